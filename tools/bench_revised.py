#!/usr/bin/env python3
"""Revised-simplex timings on one MI355X (BASELINE configs[2]): per-iteration time of the
order-faithful kernels and the fp64-MFMA B^-1 * A product (GPU only)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lpr_381_group_v22_amd as pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=24)
    ap.add_argument("--gemm-reps", type=int, default=5)
    args = ap.parse_args()
    m, n = args.m, args.n
    eng = pkg.Engine(0)
    st = pkg.RevisedState.synthetic(eng, m, n, 0)
    st.solve(max_pivots=4)
    t0 = time.perf_counter()
    res = st.solve(max_pivots=args.iters)
    dt = time.perf_counter() - t0
    out = {"m": m, "n": n, "iterations": int(res.iterations), "status": int(res.status),
           "ms_per_iteration": round(dt / max(res.iterations, 1) * 1e3, 4),
           "iterations_per_s": round(res.iterations / dt, 2)}
    flops = 2.0 * m * m * n
    best = None
    for _ in range(args.gemm_reps):
        _, ms = st.binv_a(fetch=False)
        best = ms if best is None else min(best, ms)
    out["binv_a_ms"] = round(best, 4)
    out["binv_a_tflops"] = round(flops / (best * 1e-3) / 1e12, 3)
    out["binv_a_flop"] = flops
    print(json.dumps(out))
    st.destroy()
    eng.close()


if __name__ == "__main__":
    main()
