#!/usr/bin/env python3
"""Headline benchmark: full-tableau primal-simplex pivots/s on the m=4096, n=8192 dense random LP
(4097 x 12289 fp64 tableau, 402.8 MB) and the HBM-roofline fraction of the rank-1 update kernel.

    python bench.py --gpus 1 --steps 512 --warmup 64
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pivot of the hot path: k_select (entering arg-min, ratio-test arg-min, pivot-row
normalise) + k_update (rank-1 row elimination of the whole tableau).  The tableau is generated on
the device and is resident in HBM before the timed region starts; nothing crosses PCIe inside it.
With N > 1 every rank owns one GPU and solves its own LP replica (seed = rank): the path shards
by independent sub-problems with no data-path collective ("weak" scaling).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(m: int, n: int, seed: int, pivots: int):
    """The C oracle (literal restatement of PrimalSimplexSolver.cs:152-211, gcc -O2
    -ffp-contract=off, ONE thread like the reference) timed on this host on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    orc = Oracle()
    T, basis = orc.gen_dense_tableau(m, n, seed)
    t0 = time.perf_counter()
    st, piv, log = orc.primal_solve(T, basis, pivots)
    dt = time.perf_counter() - t0
    return piv / dt, piv, log


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--variant", type=int, default=0, help="rank-1 update kernel variant (0=auto)")
    ap.add_argument("--cpu-pivots", type=int, default=-1,
                    help="pivots of the CPU baseline sample (-1: sized for ~15 s, 0: skip)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="replay captured graphs instead of eager launches with HIP events")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank "
                             "per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import lpr_381_group_v22_amd as pkg

    m, n, K, W = args.m, args.n, args.steps, args.warmup
    R, C = m + 1, n + m + 1
    bytes_per_pivot = 2 * 8 * R * C  # every tableau element read once + written once
    seed = rank  # one LP replica per rank

    eng = pkg.Engine(local_rank)
    tab = pkg.Tableau.synthetic(eng, m, n, seed)
    timed = not args.no_kernel_timing

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    if W > 0:
        res = tab.solve(max_pivots=W, time_kernels=False, variant=args.variant)
        if res.pivots != W:
            raise SystemExit(f"warm-up ended after {res.pivots} pivots (status {res.status})")
    k0 = tab.kernel_stats()
    barrier()
    t0 = time.perf_counter()
    res = tab.solve(max_pivots=K, time_kernels=timed, variant=args.variant)
    barrier()
    dt = time.perf_counter() - t0
    if res.pivots != K:
        raise SystemExit(f"timed region ended after {res.pivots} of {K} pivots "
                         f"(status {res.status}); pick another seed / fewer steps")
    k1 = tab.kernel_stats()

    dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt_max = float(dt_t.item())

    launches = k1[0] - k0[0]
    kern_ms = (k1[1] - k0[1]) / launches if launches else None

    out = None
    if rank == 0:
        value = world * K / dt_max
        roof = None
        if kern_ms:
            achieved = bytes_per_pivot / (kern_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_update (rank-1 row elimination)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4),
                    "bytes_per_launch": bytes_per_pivot,
                    "avg_launch_ms": round(kern_ms, 6), "launches": launches,
                    "event_sampling": "every 4th k_update launch of the timed region",
                    "traffic": None}
            roof.update(_pmc_traffic(m, n))
        cpu = None
        if world == 1 and args.cpu_pivots != 0:
            cp = args.cpu_pivots
            if cp < 0:
                # ~16*R*C bytes per pivot at roughly 4 GB/s on one core -> aim at ~15 s
                cp = max(4, min(2000, int(15.0 / (bytes_per_pivot / 4.0e9))))
            cpu_rate, cpu_piv, cpu_log = cpu_baseline(m, n, 0, cp)
            gpu_log = tab.pivot_log(cap=cpu_piv)
            cpu = {"value": round(cpu_rate, 3), "unit": "pivots/s", "cores": 1, "kind": "port",
                   "sample": f"first {cpu_piv} pivots of the same LP (m={m}, n={n}, seed 0) by "
                             f"the C oracle of PrimalSimplexSolver.cs:152-211, 1 thread, "
                             f"snapshots off; cpu: {_cpu_model()}, {os.cpu_count()} logical",
                   "pivot_log_matches_gpu": bool(
                       (gpu_log[:min(len(gpu_log), len(cpu_log))]
                        == cpu_log[:min(len(gpu_log), len(cpu_log))]).all())}
        out = {
            "metric": "simplex pivots/sec on 4096x8192 fp64 tableau; % HBM roofline",
            "value": round(value, 2), "unit": "pivots/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": round(dt_max * 1e3 / K, 6), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"dense random LP m={m} n={n} fp64, full-tableau primal "
                                   f"simplex pivots on the {R}x{C} tableau "
                                   f"({R * C * 8 / 1e6:.1f} MB), one LP replica per GPU",
                       "m": m, "n": n, "rows": R, "cols": C, "seed": "rank",
                       "parallelism": f"replica{world}", "update_variant": args.variant,
                       "launch": "eager+events" if timed else "hipGraph"},
            "roofline": roof, "cpu_baseline": cpu,
        }
    tab.destroy()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)
    return 0


def _pmc_traffic(m: int, n: int) -> dict:
    """HBM-side bytes of one k_update launch from the rocprofv3 PMC passes (FETCH_SIZE doubled for
    gfx950 wide reads + WRITE_SIZE, separate --pmc runs).  Counters cannot be read from inside the
    process, so the newest summary committed under profiles/ for this workload is reported."""
    import glob
    best = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_summary.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
            t = d.get("k_update_traffic_per_launch", {})
            if t.get("algorithmic_bytes") == 2 * 8 * (m + 1) * (n + m + 1):
                best = {"traffic": int(t["hbm_side_bytes"]),
                        "traffic_source": os.path.relpath(path, ROOT)}
        except (OSError, ValueError, KeyError):
            continue
    return best


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    sys.exit(main())
