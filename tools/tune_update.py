#!/usr/bin/env python3
"""Sweep the rank-1 update kernel variants at a given LP size (GPU only).  Interleaved rounds in
one process; prints avg launch ms and algorithmic GB/s per variant."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lpr_381_group_v22_amd as pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--pivots", type=int, default=48)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", type=str, default="")
    args = ap.parse_args()
    m, n = args.m, args.n
    bytes_pp = 16 * (m + 1) * (n + m + 1)
    variants = [int(v, 0) for v in args.variants.split(",")] if args.variants else \
        list(range(1, 12)) + [0x100 + v for v in range(1, 12)]
    eng = pkg.Engine(0)
    tab = pkg.Tableau.synthetic(eng, m, n, 0)
    tab.solve(max_pivots=16)
    stats = {v: [] for v in variants}
    wall = {v: [] for v in variants}
    for r in range(args.rounds):
        for v in variants:
            k0 = tab.kernel_stats()
            t0 = time.perf_counter()
            res = tab.solve(max_pivots=args.pivots, time_kernels=True, variant=v)
            dt = time.perf_counter() - t0
            k1 = tab.kernel_stats()
            assert res.pivots == args.pivots, (res.pivots, res.status)
            stats[v].append((k1[1] - k0[1]) / (k1[0] - k0[0]))
            wall[v].append(dt / args.pivots * 1e3)
    # graph (untimed) wall per pivot
    for v in variants:
        ms = sorted(stats[v])
        med = ms[len(ms) // 2]
        print(f"variant {v:#06x}: kernel ms min {ms[0]:.4f} med {med:.4f}  "
              f"-> {bytes_pp / (ms[0] * 1e-3) / 1e9:8.1f} GB/s (best)  "
              f"wall/pivot {min(wall[v]):.4f} ms", flush=True)
    for v in variants[:2]:
        t0 = time.perf_counter()
        res = tab.solve(max_pivots=args.pivots * 4, time_kernels=False, variant=v)
        dt = time.perf_counter() - t0
        print(f"graph mode variant {v:#06x}: {dt / res.pivots * 1e3:.4f} ms/pivot "
              f"({res.pivots / dt:.1f} pivots/s)")
    tab.destroy()
    eng.close()


if __name__ == "__main__":
    main()
