#!/usr/bin/env python3
"""Per-kernel medians of the TCC FETCH_SIZE / WRITE_SIZE counters from two rocprofv3 --pmc passes
(separate passes: the two counters do not fit one) and the corrected HBM-side traffic of one sweep
launch, as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE
reports 1/2 of the bytes of wide (16 B/lane) streaming reads -> doubled; WRITE_SIZE exact.

    python tools/pmc_summary.py FETCH_DIR WRITE_DIR KERNEL_SUBSTRING PIVOTS_PER_LAUNCH M N > out.json
"""
import csv
import glob
import json
import os
import statistics
import sys


def per_kernel(dirname, counter):
    rows = {}
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                name = r["Kernel_Name"].split("(")[0]
                rows.setdefault(name, []).append(float(r["Counter_Value"]))
    return rows


def main():
    fdir, wdir, kern, ppl, m, n = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), \
        int(sys.argv[5]), int(sys.argv[6])
    out = {}
    med = {}
    for counter, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
        out[counter] = {}
        for name, vals in sorted(per_kernel(d, counter).items()):
            # launches that return at once on the status word move (almost) nothing: keep the
            # working launches only when a kernel has both kinds
            big = [v for v in vals if v > 0.5 * max(vals)] if max(vals) > 0 else vals
            out[counter][name] = {"dispatches": len(vals), "working_dispatches": len(big),
                                  "median_KiB": statistics.median(big), "max_KiB": max(vals)}
            if kern in name:
                med[counter] = statistics.median(big) * 1024.0
    alg = ppl * 2 * 8 * (m + 1) * (n + m + 1)
    if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
        out["k_update_traffic_per_launch"] = {
            "kernel": kern, "pivots_per_launch": ppl,
            "fetch_bytes_raw": med["FETCH_SIZE"],
            "fetch_bytes_corrected_x2": 2 * med["FETCH_SIZE"],
            "write_bytes": med["WRITE_SIZE"],
            "hbm_side_bytes": 2 * med["FETCH_SIZE"] + med["WRITE_SIZE"],
            "algorithmic_bytes": alg,
            "note": "gfx950: FETCH_SIZE reports 1/2 of wide (16 B/lane) streaming reads -> doubled "
                    "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for 16 B/lane stores; "
                    "separate --pmc passes; counters sit at the L2 fabric interface, "
                    "Infinity-Cache hits are included",
        }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
