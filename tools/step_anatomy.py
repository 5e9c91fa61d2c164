#!/usr/bin/env python3
"""Where a step of the two-stream pivot loop spends its time at the headline size (m=4096,
n=8192), from the diagnostic build's launch stamps (opts.variant bit 16, Tableau.launch_stamps):
for the last 8 steps of a 21-block call, when the loop heads entered, had their tableau buffer,
ended their last head and published; when the sweep's first workgroup entered and (0x400000) saw
the heads' completion word.  Three hand-over forms: events both ways (default), heads wait on the
device (0x200000), both wait on the device (0x600000).   python tools/step_anatomy.py"""
import sys

import numpy as np

sys.path.insert(0, ".")
import lpr_381_group_v22_amd as pkg  # noqa: E402

eng = pkg.Engine(0)
for variant, name in ((0x10000, "events both ways (default)"),
                      (0x10000 | 0x200000, "heads wait on the device (0x200000)"),
                      (0x10000 | 0x600000, "heads and sweep wait on the device (0x600000)")):
    tab = pkg.Tableau.synthetic(eng, 4096, 8192, 0)
    tab.solve(max_pivots=16 * 8, variant=variant)
    res = tab.solve(max_pivots=16 * 21, variant=variant)
    ls = tab.launch_stamps().astype(np.int64)
    print(f"{name}: {res.pivots} pivots; us")
    order = np.argsort(ls[:, 0])
    t0 = ls[order[0], 0]
    prev = None
    for p in order:
        r = ls[p]
        step = "" if prev is None else "(step %.1f)" % ((r[0] - prev) / 100.0)
        prev = r[0]
        print("  heads enter %8.1f %-13s buffer ready %+5.1f  last head done %+6.1f  published %s | "
              "sweep enters %8.1f  sees the heads' word %s" % (
                  (r[0] - t0) / 100.0, step, (r[1] - r[0]) / 100.0, (r[2] - r[0]) / 100.0,
                  "%+6.1f" % ((r[3] - r[0]) / 100.0) if r[3] else "  -   ",
                  (r[4] - t0) / 100.0, "%8.1f" % ((r[5] - t0) / 100.0) if r[5] else "-"))
    tab.destroy()
