#!/usr/bin/env python3
"""HBM-side traffic of EVERY kernel of a run from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE do not fit one pass), corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes for gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) streaming reads ->
doubled; WRITE_SIZE exact.  Medians over the working dispatches of a kernel (launches that return at
once on a status word move almost nothing and are left out).

    python tools/pmc_kernels.py FETCH_DIR WRITE_DIR > out.json
"""
import json
import sys

from pmc_summary import per_kernel
import statistics


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    out = {"note": "per dispatch: hbm_side_bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters x 1024); "
                   "gfx950 reports half of wide streaming reads; counters sit at the L2 fabric "
                   "interface, Infinity-Cache hits are included; separate --pmc passes",
           "kernels": {}}
    f = per_kernel(fdir, "FETCH_SIZE")
    w = per_kernel(wdir, "WRITE_SIZE")
    for name in sorted(set(f) | set(w)):
        def med(vals):
            if not vals:
                return 0.0, 0
            big = [v for v in vals if v > 0.5 * max(vals)] if max(vals) > 0 else vals
            return statistics.median(big) * 1024.0, len(big)
        fb, nf = med(f.get(name, []))
        wb, nw = med(w.get(name, []))
        tot = 2 * 1024.0 * sum(f.get(name, [])) + 1024.0 * sum(w.get(name, []))
        out["kernels"][name] = {"fetch_bytes_raw": fb, "fetch_bytes_corrected_x2": 2 * fb,
                                "write_bytes": wb, "hbm_side_bytes": 2 * fb + wb,
                                "working_dispatches": min(nf, nw) if nf and nw else max(nf, nw),
                                # all dispatches of the run (levels of a B&B tree differ in width)
                                "dispatches": max(len(f.get(name, [])), len(w.get(name, []))),
                                "hbm_side_bytes_all_dispatches": tot}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
