#!/usr/bin/env python3
"""Round-2 probe on the headline tableau (m=4096, n=8192): pivots/s, sweep and step time for the
loop-head placements (confined to one XCD with L2 hand-offs / confined with memory-side hand-offs /
spread over the chip) x sweep tile shapes, and the per-phase time stamps of the lead loop-head
workgroup.  Writes JSON lines to stdout.

    python tools/r2_probe.py [--quick]
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import lpr_381_group_v22_amd as pkg  # noqa: E402

STAMPS, SPREAD, MEMSIDE = 0x10000, 0x20000, 0x40000
PHASES = ["collect_e", "col_gather", "sync", "col_chain", "ratio_reduce", "drain_col",
          "collect_r", "row_gather", "row_chain", "z_reduce_rhs", "drain_row"]


def run(eng, m, n, variant, block=0, warm=8, steps=40, label=""):
    tab = pkg.Tableau.synthetic(eng, m, n, 0)
    probe = tab.solve(max_pivots=1, variant=variant, block=block)
    B = max(1, probe.block)
    tab.solve(max_pivots=warm * B, variant=variant, block=block)
    k0, s0 = tab.kernel_stats(), tab.step_stats()
    eng.sync()
    t0 = time.perf_counter()
    res = tab.solve(max_pivots=steps * B, variant=variant, block=block, time_kernels=True)
    eng.sync()
    dt = time.perf_counter() - t0
    k1, s1 = tab.kernel_stats(), tab.step_stats()
    # the same again without events
    eng.sync()
    t0 = time.perf_counter()
    res2 = tab.solve(max_pivots=steps * B, variant=variant, block=block)
    eng.sync()
    dt2 = time.perf_counter() - t0
    out = {"label": label, "variant": hex(variant), "block": B, "pivots": int(res.pivots),
           "pivots_per_s_events": round(res.pivots / dt, 1),
           "pivots_per_s": round(res2.pivots / dt2, 1),
           "sweep_us": round(1e3 * (k1[1] - k0[1]) / max(1, k1[0] - k0[0]), 2),
           "sweeps": k1[0] - k0[0],
           "step_us": round(1e3 * (s1[1] - s0[1]) / max(1, s1[0] - s0[0]), 2),
           "steps": s1[0] - s0[0]}
    tab.destroy()
    return out


def stamps(eng, m, n, variant, label):
    tab = pkg.Tableau.synthetic(eng, m, n, 0)
    tab.solve(max_pivots=129, variant=variant)
    res = tab.solve(max_pivots=64, variant=variant | STAMPS)
    st, xcc, l2 = tab.head_stamps()
    st = st.astype(np.int64)
    d = np.diff(st, axis=1) * 0.01  # us
    per_pivot = (st[:, 11] - st[:, 0]) * 0.01
    # gap between the end of one head and the start of the next (same launch only)
    out = {"label": label, "variant": hex(variant), "xcc": xcc, "l2": l2,
           "pivots": int(res.pivots),
           "head_us_mean": round(float(per_pivot.mean()), 2),
           "head_us_median": round(float(np.median(per_pivot)), 2),
           "phases_us": {PHASES[k]: round(float(np.median(d[:, k])), 2) for k in range(11)}}
    tab.destroy()
    return out


def main():
    quick = "--quick" in sys.argv
    m, n = 4096, 8192
    eng = pkg.Engine(0)
    NOAVOID, NOHINT, DEVHAND = 0x80000, 0x100000, 0x200000
    heads = (("confined+L2, sweep leaves the XCD", 0), ("same, hand-over on the device", DEVHAND),
             ("same, live word only", NOHINT),
             ("confined+L2, sweep everywhere", NOAVOID), ("confined+memside", MEMSIDE),
             ("spread", SPREAD))
    for label, fl in heads:
        for tile in ((0x08,) if quick else (0x08, 0x24, 0x28, 0x04, 0x10)):
            r = run(eng, m, n, 0x3000 | tile | fl, label=f"ov2 {label} tile {tile:#x}")
            print(json.dumps(r), flush=True)
    for label, fl in (("confined+L2", 0), ("spread", SPREAD)):
        for tile in ((0x08,) if quick else (0x08, 0x28)):
            r = run(eng, m, n, 0x4000 | tile | fl, label=f"seq {label} tile {tile:#x}")
            print(json.dumps(r), flush=True)
    if not quick:
        r = run(eng, m, n, 0x5008, label="ov (one launch)")
        print(json.dumps(r), flush=True)
        r = run(eng, m, n, 0, block=1, warm=32, steps=128, label="one pivot per sweep")
        print(json.dumps(r), flush=True)
    for label, v in (("ov2 default", 0x3008), ("ov2 live word only", 0x3008 | NOHINT),
                     ("ov2 sweep everywhere", 0x3008 | NOAVOID),
                     ("ov2 spread", 0x3008 | SPREAD), ("seq confined+L2", 0x4008)):
        for rep in range(2):
            print(json.dumps(stamps(eng, m, n, v, label)), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
