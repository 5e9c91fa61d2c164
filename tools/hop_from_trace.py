#!/usr/bin/env python3
"""Gaps between the kernels of consecutive overlapped steps, from a rocprofv3 --kernel-trace CSV:
end of sweep k -> start of heads k+1, end of heads k -> start of sweep k+1 (the two cross-stream
hand-overs), and end -> start on the same stream.  python tools/hop_from_trace.py <kernel_trace.csv>"""
import csv
import json
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
heads = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_ov2_heads" in r[name_key])
sweeps = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_ov2_sweep" in r[name_key])
# working launches only (a launch that finds nothing to do returns within a few us)
n = min(len(heads), len(sweeps))
out = {"heads": len(heads), "sweeps": len(sweeps)}
g = {"sweep_end_to_next_heads_start": [], "heads_end_to_next_sweep_start": [],
     "sweep_end_to_next_sweep_start": [], "heads_end_to_next_heads_start": [],
     "sweep_start_to_next_sweep_start": [], "heads_us": [], "sweep_us": []}
for k in range(n - 1):
    hs, he = heads[k]
    ss, se = sweeps[k]
    hs1, he1 = heads[k + 1]
    ss1, se1 = sweeps[k + 1]
    if he - hs < 100_000 or se - ss < 100_000 or he1 - hs1 < 100_000 or se1 - ss1 < 100_000:
        continue
    g["sweep_end_to_next_heads_start"].append((hs1 - se) / 1e3)
    g["heads_end_to_next_sweep_start"].append((ss1 - he) / 1e3)
    g["sweep_end_to_next_sweep_start"].append((ss1 - se) / 1e3)
    g["heads_end_to_next_heads_start"].append((hs1 - he) / 1e3)
    g["sweep_start_to_next_sweep_start"].append((ss1 - ss) / 1e3)
    g["heads_us"].append((he - hs) / 1e3)
    g["sweep_us"].append((se - ss) / 1e3)
for k, v in g.items():
    if v:
        out[k] = {"median_us": round(statistics.median(v), 2), "mean_us": round(statistics.mean(v), 2),
                  "min_us": round(min(v), 2), "max_us": round(max(v), 2), "n": len(v)}
print(json.dumps(out))
