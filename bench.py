#!/usr/bin/env python3
"""Benchmarks of the MI355X simplex pivot engine.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 64 --warmup 8                        # headline (default)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

--workload primal (default, BASELINE.json's metric): full-tableau primal-simplex pivots/s on the
    dense random LP m=4096, n=8192 (4097 x 12289 fp64 tableau, 402.8 MB) and the HBM-roofline
    fraction of the sweep kernel.  A "step" is one pass of the hot path over the tableau: one
    sweep, which applies 16 pivots (each element read once, taken through 16 rounded
    multiply-subtract steps in registers, written once), beside the loop heads that decide the next
    16 (entering arg-min, ratio-test arg-min, pivot-row normalise); --block 1 = one pivot per step.
    The tableau is generated on the device and resident in HBM before the timed region;
    nothing crosses PCIe inside it.  With N > 1 every rank owns one GPU and solves its own LP
    replica (seed = rank): the path shards by independent sub-problems, no data-path collective.
--workload revised (BASELINE configs[2]): revised-simplex iterations/s at the same size and the
    fp64-MFMA B^-1*A product (roofline bound "mfma").
--workload sens (row f4): ChangeRHS edits re-solved by the dual simplex on the optimal tableau.
--workload bb (BASELINE configs[3]): level-synchronous Branch & Bound, sub-trees sharded over the
    ranks, ONE RCCL all-reduce(MAX) of the incumbent per level; value = sub-problem pivots/s.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
OVERLAP_BYTES = 80 << 20    # kOverlapBytes of csrc/lpr_engine.hip: above it the two-stream path runs
MFMA_F64_PEAK_TFLOPS = 78.6  # MI355X datasheet fp64 matrix (= fp64 vector) peak; the local guide
#                              has no fp64 MFMA row, tools/mfma_f64_peak.hip probes it on the device


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    return Oracle()


def _pmc_traffic(m: int, n: int, block: int = 1) -> dict:
    """HBM-side bytes of one sweep launch from the rocprofv3 PMC passes (FETCH_SIZE doubled for
    gfx950 wide reads + WRITE_SIZE, separate --pmc runs).  Counters cannot be read from inside the
    process, so the newest summary committed under profiles/ for this workload (same tableau, same
    pivots per sweep) is reported."""
    import glob
    best = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_summary.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
            t = d.get("k_update_traffic_per_launch", {})
            if t.get("algorithmic_bytes") in (block * 2 * 8 * (m + 1) * (n + m + 1),
                                              2 * 8 * (m + 1) * (n + m + 1)) and \
                    int(t.get("pivots_per_launch", 1)) == block:
                best = {"traffic": int(t["hbm_side_bytes"]),
                        "traffic_source": os.path.relpath(path, ROOT)}
        except (OSError, ValueError, KeyError):
            continue
    return best


def _pmc_kernel_traffic(workload: str) -> dict:
    """Per-kernel HBM-side bytes per dispatch from the newest profiles/r*_{workload}_pmc_summary.json
    (tools/prof_workload.sh: separate rocprofv3 --pmc passes, FETCH_SIZE doubled for gfx950)."""
    import glob
    out = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_pmc_summary.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
            short = lambda k: k.replace("void ", "").split("<")[0].split("::")[-1]  # noqa: E731
            out = {"source": os.path.relpath(path, ROOT),
                   "kernels": {short(k): int(v["hbm_side_bytes"]) for k, v in d["kernels"].items()
                               if "lpr::" in k},
                   # every dispatch of the run summed (absent in summaries of earlier rounds)
                   "totals": {short(k): int(v["hbm_side_bytes_all_dispatches"])
                              for k, v in d["kernels"].items()
                              if "lpr::" in k and "hbm_side_bytes_all_dispatches" in v}}
        except (OSError, ValueError, KeyError):
            continue
    return out


class Dist:
    """torch.distributed plumbing (backend nccl == RCCL over xGMI); a no-op at world size 1."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"WORLD_SIZE={self.world} does not match --gpus {args.gpus}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
        # Rehearsal of the N > 1 code path on a one-GPU box: every rank uses GPU 0 and the bench's
        # own collectives go over gloo (RCCL refuses two ranks on one device).  The line it prints
        # is marked "rehearsal" -- it is a test of the plumbing, never a measurement.
        self.shared_gpu = os.environ.get("LPR_BENCH_SHARED_GPU", "") == "1" and self.world > 1
        self.device_index = 0 if self.shared_gpu else self.local_rank
        self.tensor_device = "cpu" if self.shared_gpu else "cuda"
        torch.cuda.set_device(self.device_index)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.shared_gpu:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl",
                                        device_id=torch.device("cuda", self.local_rank))
            self.dist = dist

    def barrier(self, eng=None):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        if eng is not None:
            eng.sync()

    def max(self, v: float) -> float:
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.tensor_device)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, v: float) -> float:
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.tensor_device)
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------------------ primal
def _sha(a) -> str:
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_primal(args, D: Dist):
    """A step is one pass of the hot path over the whole tableau: one sweep, which applies
    `pivots_per_step` pivots (16 on the large-tableau paths, 1 with --block 1 and on the small
    paths) together with the loop heads that decide the next ones.  --steps K times exactly K of
    them in ONE lpr_primal_solve call (max_pivots = K x pivots_per_step); value = pivots/s."""
    import lpr_381_group_v22_amd as pkg
    m, n, K, W = args.m, args.n, args.steps, args.warmup
    R, C = m + 1, n + m + 1
    bytes_per_pivot = 2 * 8 * R * C  # every tableau element read once + written once
    eng = pkg.Engine(D.device_index)
    tab = pkg.Tableau.synthetic(eng, m, n, D.rank)  # one LP replica per rank (seed = rank)
    timed = not args.no_kernel_timing
    # one pivot to learn how many pivots a step of the path chosen for this tableau applies
    probe = tab.solve(max_pivots=1, variant=args.variant, block=args.block)
    B = max(1, probe.block)
    if B == 1 and args.block == 0 and args.variant == 0 and R * C * 8 <= (1 << 20):
        # the single-launch path of small (cache-resident) tableaux has no per-kernel event
        # timing (asking for it would switch to another path): whole-job figures only
        timed = False
    if W > 0:
        # warm-up in the timed region's own mode (the engine creates its HIP events on first use:
        # ~10 us each, which is not pivot time); the statistics below are differences
        res = tab.solve(max_pivots=W * B, time_kernels=(args.time_stride if timed else 0),
                        variant=args.variant, block=args.block)
        if res.pivots != W * B:
            raise SystemExit(f"warm-up ended after {res.pivots} pivots (status {res.status})")
    k0, s0 = tab.kernel_stats(), tab.step_stats()
    D.barrier(eng)
    t0 = time.perf_counter()
    res = tab.solve(max_pivots=K * B, time_kernels=(args.time_stride if timed else 0),
                    variant=args.variant, block=args.block)
    D.barrier(eng)
    dt = time.perf_counter() - t0
    if res.pivots != K * B or res.block != B:
        raise SystemExit(f"timed region ended after {res.pivots} of {K * B} pivots "
                         f"(status {res.status}); pick another seed / fewer steps")
    k1, s1 = tab.kernel_stats(), tab.step_stats()
    dt_max = D.max(dt)
    launches = k1[0] - k0[0]
    kern_ms = (k1[1] - k0[1]) / launches if launches else None
    nsteps = s1[0] - s0[0]
    step_ms = (s1[1] - s0[1]) / nsteps if nsteps else None

    out = None
    if D.rank == 0:
        value = D.world * K * B / dt_max
        two_stream = B > 1 and (args.variant & 0xff00) in (0, 0x3000) and R * C * 8 > OVERLAP_BYTES
        if B == 1 and not timed and R * C * 8 <= (1 << 20):
            kname = ("k_pivot_fused (selection + rank-1 update in one launch; the tableau is "
                     "cache-resident: effective GB/s, not an HBM fraction)")
        elif B == 1:
            kname = "k_update (rank-1 row elimination, one pivot per launch)"
        elif two_stream:
            kname = (f"k_ov2_sweep (one read + one write of the tableau, {B} pivots applied in "
                     f"registers; k_ov2_heads decides the next {B} beside it on a second stream)")
        else:
            kname = f"sweep of {B} pivots per launch (one read + one write of the tableau)"
        roof = {"bound": "hbm", "kernel": kname, "achieved": None, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": None, "traffic": None,
                "bytes_per_launch": bytes_per_pivot, "pivots_per_launch": B}
        if kern_ms:
            # PHYSICAL bytes of one launch (2*8*R*C: each element read once and written once,
            # however many pivots the launch applies) / its duration by HIP events on its stream
            achieved = bytes_per_pivot / (kern_ms * 1e-3) / 1e9
            roof.update({"achieved": round(achieved, 1),
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         # hipMemcpyDtoD of the same 403 MB on the same part: 155.6 us = 5.18 TB/s
                         # (profiles/r02_sweep_floor_probe.jsonl): what an out-of-place copy achieves
                         "frac_vs_measured_copy_5180": round(achieved / 5180.0, 4),
                         "avg_launch_ms": round(kern_ms, 6), "launches": launches,
                         "event_sampling": ((("every sweep launch" if args.time_stride <= 1 else
                                              f"one sweep launch in {args.time_stride}") +
                                             " of the timed region that applied a full block "
                                             "(start / stop events of the launch itself; "
                                             "sampling every launch costs ~1 % more)") if B > 1 else
                                            "every 4th update launch of the timed region")})
        else:
            roof["note_timing"] = "no launch of the timed region was bracketed by events"
            whole = bytes_per_pivot * value / D.world / 1e9  # one launch per pivot: bytes x rate
            if B == 1:
                roof.update({"achieved": round(whole, 1),
                             "frac": round(whole / HBM_PEAK_GBPS, 4)})
        if step_ms:
            roof["avg_step_ms"] = round(step_ms, 6)
            roof["steps_timed"] = nsteps
        if R * C * 8 <= (256 << 20) * 0.5:
            # cache-resident: an effective rate, not a fraction of the HBM roofline -- a
            # pivot here is bound by the latency of its dependent steps (two gathers, two arg-min
            # reductions), not by bytes
            eff = bytes_per_pivot * value / D.world / 1e9
            roof.update({"bound": "latency", "achieved": round(eff, 1), "frac": None,
                         "unit": "GB/s (effective: 2*8*R*C bytes per pivot x pivots/s; the "
                                 "tableau stays in L2 / Infinity Cache)",
                         "us_per_pivot": round(1e6 * D.world / value, 3)})
            if B > 1 and R <= 1024 and (C + 15) // 16 * 16 <= 2048 and args.variant == 0 \
                    and args.block == 0:
                roof["kernel"] = ("k_small_heads (the 16 loop heads of a block in one workgroup) "
                                  "+ k_small_sweep (in place)")
        roof.update(_pmc_traffic(m, n, B))
        if roof.get("traffic") and (step_ms or kern_ms):
            # what the chip's memory side sustains over a whole step (sweep + heads beside it)
            roof["hbm_side_frac"] = round(roof["traffic"] / ((step_ms or kern_ms) * 1e-3) / 1e9
                                          / HBM_PEAK_GBPS, 4)
        # SURVEY 8(d)'s per-pivot figure x pivots/s: what one-pivot-per-sweep would have to move
        # to reach this pivot rate -- NOT bytes that moved when a sweep applies several pivots
        roof["algorithmic_equivalent_gbps"] = round(bytes_per_pivot * value / D.world / 1e9, 1)
        cpu = None
        if D.world == 1 and args.cpu_pivots != 0:
            orc = _oracle()
            T, basis = orc.gen_dense_tableau(m, n, 0)
            cp = args.cpu_pivots
            c0 = time.perf_counter()
            if cp < 0:
                # a bounded sample of the same LP: about 12 s of one core, sized from the first 8
                st, piv0, log0 = orc.primal_solve(T, basis, 8)
                per = (time.perf_counter() - c0) / max(1, piv0)
                more = max(0, min(4000, int(12.0 / max(per, 1e-6))) - piv0) if st == 5 else 0
                st, piv1, log1 = orc.primal_solve(T, basis, more) if more > 0 else (st, 0, log0[:0])
                cpu_piv = piv0 + piv1
                cpu_log = np.concatenate([log0, log1]) if piv1 else log0
            else:
                st, cpu_piv, cpu_log = orc.primal_solve(T, basis, cp)
            cdt = time.perf_counter() - c0
            # the same number of pivots on a fresh device tableau (default path), then everything
            # the two sides hold is compared: pivot log, basis, the whole tableau
            chk = pkg.Tableau.synthetic(eng, m, n, 0)
            cres = chk.solve(max_pivots=cpu_piv, variant=args.variant, block=args.block)
            gpu_log = chk.pivot_log(cap=cpu_piv + 8)
            same_log = gpu_log.shape == cpu_log.shape and bool((gpu_log == cpu_log).all())
            same_basis = chk.basis().tolist() == basis.tolist()
            same_tab = _sha(chk.read()) == _sha(T)
            chk.destroy()
            cpu = {"value": round(cpu_piv / cdt, 3), "unit": "pivots/s", "cores": 1,
                   "kind": "port",
                   "sample": f"first {cpu_piv} pivots of the same LP (m={m}, n={n}, seed 0) by "
                             f"the C oracle of PrimalSimplexSolver.cs:152-211, 1 thread, "
                             f"snapshots off; cpu: {_cpu_model()}, {os.cpu_count()} logical",
                   "gpu_status_after_same_pivots": cres.status,
                   "pivot_log_matches_gpu": same_log, "basis_matches_gpu": same_basis,
                   "tableau_sha256_matches_gpu": same_tab}
        headline = (m == 4096 and n == 8192)
        out = {
            "metric": ("simplex pivots/sec on 4096x8192 fp64 tableau; % HBM roofline" if headline
                       else f"simplex pivots/sec on {m}x{n} fp64 LP ({R}x{C} tableau); % HBM "
                            f"roofline"),
            "value": round(value, 2), "unit": "pivots/s", "n_gpus": D.world, "steps": K,
            "warmup": W, "ms_per_step": round(dt_max * 1e3 / K, 6), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "pivots_per_step": B,
            "config": {"workload": f"dense random LP m={m} n={n} fp64, full-tableau primal "
                                   f"simplex pivots on the {R}x{C} tableau "
                                   f"({R * C * 8 / 1e6:.1f} MB), one LP replica per GPU; a step = "
                                   f"one sweep of the tableau = {B} pivots",
                       "m": m, "n": n, "rows": R, "cols": C, "seed": "rank",
                       "parallelism": f"replica{D.world}", "update_variant": args.variant,
                       "pivots_per_sweep": B, "timed_region": "one lpr_primal_solve call, "
                       f"max_pivots = steps x {B} (pipeline fill and drain included)",
                       "launch": ("eager+events" if timed else
                                  ("hipGraph" if B == 1 else "eager"))},
            "roofline": roof, "cpu_baseline": cpu,
        }
    tab.destroy()
    eng.close()
    return out


# ------------------------------------------------------------------------------------ revised
def run_revised(args, D: Dist):
    import lpr_381_group_v22_amd as pkg
    m, n, K, W = args.m, args.n, args.steps, args.warmup
    eng = pkg.Engine(D.device_index)
    st = pkg.RevisedState.synthetic(eng, m, n, D.rank)
    if W > 0:
        st.solve(max_pivots=W)
    D.barrier(eng)
    t0 = time.perf_counter()
    res = st.solve(max_pivots=K)
    D.barrier(eng)
    dt = time.perf_counter() - t0
    if res.iterations != K:
        raise SystemExit(f"timed region ended after {res.iterations} of {K} iterations")
    dt_max = D.max(dt)
    best = None
    for _ in range(5):  # the snapshot product B^-1 * A of CaptureSnapshot (:360), on MFMA
        _, ms = st.binv_a(fetch=False)
        best = ms if best is None else min(best, ms)
    flop = 2.0 * m * m * n
    out = None
    if D.rank == 0:
        tf = flop / (best * 1e-3) / 1e12
        pmc = _pmc_kernel_traffic("revised") if (m, n) == (4096, 8192) else {}
        cpu = None
        if D.world == 1 and args.cpu_pivots != 0:
            orc = _oracle()
            c, A, b = orc.gen_dense_lp(m, n, 0)
            cp = args.cpu_pivots if args.cpu_pivots > 0 else max(2, min(200, int(2.0e9 / (m * (m + n)) / 6)))
            c0 = time.perf_counter()
            ref = orc.revised_solve(c, A, b, False, max_iter=cp)
            cdt = time.perf_counter() - c0
            log = st.log(cap=cp)
            q = min(len(log), len(ref["log"]))
            cpu = {"value": round(ref["iterations"] / cdt, 3), "unit": "iterations/s", "cores": 1,
                   "kind": "port",
                   "sample": f"first {ref['iterations']} iterations of the same LP by the C "
                             f"oracle of RevisedPrimalSimplexSolver.cs:82-275 (snapshot GEMM "
                             f"off), 1 thread; cpu: {_cpu_model()}",
                   "pivot_log_matches_gpu": bool((log[:q] == ref["log"][:q]).all())}
        out = {
            "metric": "revised simplex iterations/sec on m=4096 n=8192 fp64; B^-1*A TFLOP/s",
            "value": round(D.world * K / dt_max, 2), "unit": "iterations/s", "n_gpus": D.world,
            "steps": K, "warmup": W, "ms_per_step": round(dt_max * 1e3 / K, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"dense random LP m={m} n={n} fp64, revised primal simplex "
                                   f"(order-faithful sums, three launches per iteration) + "
                                   f"B^-1*A on fp64 MFMA",
                       "m": m, "n": n, "parallelism": f"replica{D.world}"},
            "iteration_hbm": {
                "note": "one iteration = three launches: k_rev_rc_enter reads A once (reduced costs, "
                        "then the entering fold in its last workgroup), k_rev_xu_ratio reads B^-1 "
                        "once (x_B and u in one pass, then the ratio test), k_rev_update_y reads "
                        "and writes B^-1 once (E*B^-1 in place and the next y = c_B B^-1 in the "
                        "same pass): 24 m^2 + 8 m n bytes (round 2: 32 m^2 + 8 m n, y had a pass "
                        "of its own).  Every sum keeps the C#'s sequential order: a serial chain "
                        "of m rounded adds per output (~18 us at m = 4096) bounds each pass from "
                        "below whatever the memory side does",
                "bytes_per_iteration": int(24 * m * m + 8 * m * n),
                "achieved_gbps": round((24.0 * m * m + 8.0 * m * n) * K / dt_max / 1e9, 1),
                "frac_of_hbm_peak": round((24.0 * m * m + 8.0 * m * n) * K / dt_max / 1e9
                                          / HBM_PEAK_GBPS, 4),
                "round2_accounting_32m2_8mn_frac": round((32.0 * m * m + 8.0 * m * n) * K / dt_max
                                                         / 1e9 / HBM_PEAK_GBPS, 4),
                "pmc_traffic_per_dispatch": pmc.get("kernels"),
                "pmc_source": pmc.get("source")},
            "roofline": {"bound": "mfma", "kernel": "k_rev_gemm (B^-1 * A, mfma_f64_16x16x4)",
                         "achieved": round(tf, 2), "peak": MFMA_F64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(tf / MFMA_F64_PEAK_TFLOPS, 4),
                         "flop_per_launch": flop, "launch_ms": round(best, 4),
                         "traffic": (pmc.get("kernels") or {}).get("k_rev_gemm"),
                         "traffic_source": pmc.get("source")},
            "cpu_baseline": cpu,
        }
    st.destroy()
    eng.close()
    return out


# ------------------------------------------------------------------------------------ bb
def bb_instance(nvars: int, ncons: int, seed: int):
    """Seeded binary programme in the reference's option-3 shape: max c.x, A x <= b, plus the
    n rows x_i <= 1 that Program.cs:372-382 appends."""
    import numpy as np
    rng = np.random.RandomState(seed)
    c = rng.randint(1, 20, size=nvars).astype(float)
    A = rng.randint(1, 15, size=(ncons, nvars)).astype(float)
    b = np.floor(A.sum(axis=1) * rng.uniform(0.3, 0.6, size=ncons))
    return c, A, b


def run_bb(args, D: Dist):
    """BASELINE configs[3]: level-synchronous Branch & Bound inside the library
    (lpr_bb_solve_level_sync), sub-trees sharded over the ranks, ONE ncclAllReduce(MAX) of the
    incumbent per level issued by the library on RCCL.  torch.distributed only carries the RCCL
    unique id to the ranks and the bench's own barriers."""
    import numpy as np
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd import Constraint
    nv, nc, levels = args.bb_vars, args.bb_cons, args.bb_levels
    eng = pkg.Engine(D.device_index)
    c, A, b = bb_instance(nv, nc, 7)
    cons = [Constraint(A[i].tolist(), "<=", float(b[i])) for i in range(nc)]
    for i in range(nv):  # Program.cs:372-382
        co = [0.0] * (nv + 3)
        co[i] = 1.0
        co[nv + 1] = 1.0
        cons.append(Constraint(co, "<=", 1.0))
    primal = pkg.PrimalSimplexSolver(c.tolist(), cons, True, engine=eng, snapshots="none")
    primal.Solve()
    R0, C0 = primal.tableau.rows, primal.tableau.cols
    tree = pkg.BranchBoundTree.from_tableau(primal.tableau, nv, max_depth=levels + 2)
    comm = None
    if D.dist is not None:
        ids = [pkg.Comm.unique_id() if D.rank == 0 else None]
        D.dist.broadcast_object_list(ids, src=0)
        if D.shared_gpu:  # rehearsal: the library's collectives carried by gloo callbacks
            import torch

            def _armax(vals):
                t = torch.tensor(vals, dtype=torch.float64)
                D.dist.all_reduce(t, op=D.dist.ReduceOp.MAX)
                return t.tolist()

            def _agather(blob):
                parts = [None] * D.world
                D.dist.all_gather_object(parts, blob)
                return parts

            comm = pkg.Comm.custom(D.rank, D.world, _armax, _agather)
        else:
            comm = pkg.Comm.rccl(eng, D.rank, D.world, ids[0])
    D.barrier(eng)
    t0 = time.perf_counter()
    res = pkg.solve_level_sync_native(tree, comm, max_levels=levels)
    D.barrier(eng)
    dt = time.perf_counter() - t0
    dt_max = D.max(dt)
    out = None
    if D.rank == 0:
        # a child at depth d is (R0 + d) x (C0 + d); its pivot reads and writes it once (the C#
        # builds a fresh tableau per pivot, :257-271).  Depths are not tracked per pivot: the
        # mid-depth size stands for all of them (+-%d rows on %d)
        Rm, Cm = R0 + levels / 2.0, C0 + levels / 2.0
        bytes_per_pivot = 2 * 8 * Rm * Cm
        achieved = res["pivots"] * bytes_per_pivot / dt_max / 1e9
        pmc = _pmc_kernel_traffic("bb") if (nv, nc, levels) == (512, 64, 9) else {}
        cpu = None
        if D.world == 1 and args.cpu_pivots != 0:
            orc = _oracle()
            T = primal.tableau.read()
            c0 = time.perf_counter()
            cap = 24
            ref = orc.bb_solve(T, nv, node_cap=cap, rec_cap=1 << 12, piv_cap=1 << 20)
            cdt = time.perf_counter() - c0
            cpiv = sum(1 for t in ref["trace"] if t[1] < 2)
            cpu = {"value": round(cpiv / cdt, 2), "unit": "pivots/s", "cores": 1, "kind": "port",
                   "sample": f"the reference's DFS (BranchBoundSimplexSolver.cs:1006-1233) from "
                             f"the same root, first {ref['processed']} nodes popped ({cpiv} "
                             f"sub-problem pivots), C oracle, 1 thread; cpu: {_cpu_model()}",
                   "nodes_per_s": round(ref["processed"] / cdt, 2)}
            # the same level-synchronous search on the oracle for the first `chk` levels, against
            # a fresh tree on the device: node / pivot counts, incumbent bits, status
            from oracle_evaluator import OracleEvaluator
            chk = min(levels, args.bb_check_levels)
            if chk > 0:
                oref = pkg.solve_level_synchronous(OracleEvaluator(orc, T, nv), nv, max_levels=chk)
                tchk = pkg.BranchBoundTree.from_tableau(primal.tableau, nv, max_depth=chk + 2)
                gchk = pkg.solve_level_sync_native(tchk, max_levels=chk)
                tchk.destroy()
                same = all(gchk[k] == oref[k] for k in ("processed", "pivots", "levels", "found",
                                                        "status"))
                same = same and (np.float64(gchk["z"]).tobytes() == np.float64(oref["z"]).tobytes())
                if oref["found"]:
                    same = same and np.array(gchk["x"]).tobytes() == np.array(oref["x"]).tobytes()
                cpu["matches_oracle"] = bool(same)
                cpu["matches_oracle_over"] = (f"{chk} levels, {oref['processed']} nodes, "
                                              f"{oref['pivots']} pivots")
        out = {
            "metric": "Branch&Bound sub-problem pivots/sec (level-synchronous, sub-trees sharded)",
            "value": round(res["pivots"] / dt_max, 2), "unit": "pivots/s", "n_gpus": D.world,
            "steps": res["levels"], "warmup": 0,
            "ms_per_step": round(dt_max * 1e3 / max(res["levels"], 1), 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"binary programme {nv} vars, {nc}+{nv} rows (root tableau "
                                   f"{R0}x{C0}), {levels} levels, pruning off, cap lifted; "
                                   f"a step = one level (all children of the frontier in one "
                                   f"batch + one all-reduce)",
                       "nodes_processed": res["processed"], "pivots": res["pivots"],
                       "nodes_per_s": round(res["processed"] / dt_max, 1),
                       "incumbent_z": res["z"] if res["found"] else None,
                       "collective": "1 ncclAllReduce(MAX, 24 B)/level issued by "
                                     "lpr_bb_solve_level_sync" if comm else "none (one rank)",
                       "parallelism": f"subtree{D.world}"},
            "roofline": {"bound": "hbm", "kernel": "whole job (k_bb_select + k_bb_update: batched "
                                                   "in-place pivots of all live children; "
                                                   "k_bb_child_init, k_bb_eliminate, k_bb_finish)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         # PMC: HBM-side bytes of EVERY dispatch of the job's kernels summed (the
                         # whole-job counterpart of `achieved`'s algorithmic bytes); per kernel in
                         # pmc_traffic_job, the widest dispatches' median in pmc_traffic_per_dispatch
                         "traffic": (sum(v for k, v in pmc["totals"].items() if k.startswith("k_bb_"))
                                     if pmc.get("totals") else None),
                         "traffic_source": pmc.get("source"),
                         "algorithmic_bytes_job": int(res["pivots"] * bytes_per_pivot),
                         "pmc_traffic_job": pmc.get("totals") or None,
                         "pmc_traffic_per_dispatch": pmc.get("kernels"),
                         "bytes_per_pivot": int(bytes_per_pivot),
                         "note": "whole-job form: ALGORITHMIC bytes of all sub-problem pivots (2*8*R*C "
                                 "each, SURVEY 8d) / wall time (child set-up, selection kernels, host "
                                 "polls included).  The pivots are applied in place to the rows whose "
                                 "factor is not zero (~13 %% of them on this instance), so the bytes "
                                 "that MOVE are far fewer: see traffic.  A child tableau is ~%.1f MB, "
                                 "%d live children at the widest level"
                                 % (Rm * Cm * 8 / 1e6, 1 << (levels - 1))},
            "cpu_baseline": cpu,
        }
    if comm is not None:
        comm.destroy()
    tree.destroy()
    eng.close()
    return out


# ------------------------------------------------------------------------------------ sens
def run_sens(args, D: Dist):
    """Sensitivity re-solve (row f4): the LP is solved on the device, handed to the analyzer HBM to
    HBM, then every step is one ChangeRHS edit that tightens a constraint and is re-solved by
    DualSimplexIfNeeded + ReOptimize.  value = re-solve pivots per second over the timed edits."""
    import lpr_381_group_v22_amd as pkg
    from lpr_381_group_v22_amd.engine import SensState
    m, n, K, W = args.m, args.n, args.steps, args.warmup
    R, C = m + 1, n + m + 1
    bytes_per_pivot = 2 * 8 * R * C
    eng = pkg.Engine(D.device_index)
    tab = pkg.Tableau.synthetic(eng, m, n, D.rank)
    t0 = time.perf_counter()
    res = tab.solve()
    solve_s = time.perf_counter() - t0
    if res.status != 0:
        raise SystemExit(f"primal solve ended with status {res.status}")
    sens = SensState.from_tableau(tab, n)
    want_cpu = D.world == 1 and args.cpu_pivots != 0
    if want_cpu:  # what the CPU leg starts from: the optimal tableau itself (one 403 MB read-back)
        T_opt, basis_opt = tab.read(), tab.basis()
        x_opt, z_opt = tab.extract_solution(n)
    tab.destroy()

    def new_rhs(cur):
        return cur - 0.05 * abs(cur) - 1.0

    def edit(step, handle=None):
        h = handle or sens
        k = 1 + (step * 997) % m
        cur = float(h.read_block(k, 1, C - 1, 1)[0, 0])
        oc = h.change_rhs(k, new_rhs(cur))
        return oc, h.shape()[5]

    for w in range(W):
        edit(w)
    D.barrier(eng)
    t0 = time.perf_counter()
    piv = rolled = 0
    for k in range(K):
        oc, p = edit(W + k)
        piv += p
        rolled += 1 if oc == 8 else 0
    D.barrier(eng)
    dt = time.perf_counter() - t0
    dt_max = D.max(dt)
    piv_all = D.sum(float(piv))
    out = None
    if D.rank == 0:
        achieved = piv * bytes_per_pivot / dt / 1e9
        out = {
            "metric": "sensitivity re-solve pivots/sec (ChangeRHS edits on the optimal tableau)",
            "value": round(piv_all / dt_max, 2), "unit": "pivots/s", "n_gpus": D.world,
            "steps": K, "warmup": W, "ms_per_step": round(dt_max * 1e3 / K, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"dense random LP m={m} n={n} fp64 solved on the device "
                                   f"({res.pivots} pivots, {solve_s:.2f} s), then {K} ChangeRHS "
                                   f"edits re-solved by dual simplex on the {R}x{C} tableau",
                       "m": m, "n": n, "edits": K, "pivots": piv, "rolled_back": rolled,
                       "parallelism": f"replica{D.world}"},
            "roofline": {"bound": "hbm", "kernel": "k_sens_update (pivot with |f|<EPS row skip)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "note": "end-to-end lower bound: bytes of the pivots / wall time of the "
                                 "edits (snapshot copy, selection kernels and host polls included)",
                         "traffic": None},
            "cpu_baseline": None,
        }
        if want_cpu:
            # the C oracle of SensitivityAnalyzer.ChangeRHS (:427-470: DualSimplexIfNeeded +
            # ReOptimize, one thread) on the SAME optimal tableau, the first edits of the same
            # sequence (about 15 s), then those edits again on a fresh device handle built from the
            # same bytes: outcome and pivot count of every edit, and the final Z bits, compared
            orc = _oracle()
            o = orc.sens(T_opt, x_opt, z_opt, basis_opt)
            c0 = time.perf_counter()
            cpu_out, cpu_piv, nlog = [], [], 0
            for step in range(min(K + W, 64)):
                k = 1 + (step * 997) % m
                cur = float(o.state()["T"][k, C - 1])
                oc = o.change_rhs(k, new_rhs(cur))
                q = len(o.log())
                cpu_out.append(int(oc))
                cpu_piv.append(q - nlog)
                nlog = q
                if time.perf_counter() - c0 > 15.0:
                    break
            cdt = time.perf_counter() - c0
            chk = SensState.create(eng, T_opt, x_opt, z_opt)
            gpu_out, gpu_piv = [], []
            for step in range(len(cpu_out)):
                oc, pp = edit(step, chk)
                gpu_out.append(int(oc))
                gpu_piv.append(int(pp))
            same = gpu_out == cpu_out and gpu_piv == cpu_piv and \
                np.float64(chk.shape()[4]).tobytes() == np.float64(o.state()["z"]).tobytes()
            chk.destroy()
            out["cpu_baseline"] = {
                "value": round(sum(cpu_piv) / cdt, 3), "unit": "pivots/s", "cores": 1,
                "kind": "port",
                "sample": f"first {len(cpu_out)} ChangeRHS edits of the same sequence on the same "
                          f"optimal tableau ({sum(cpu_piv)} re-solve pivots), C oracle of "
                          f"SensitivityAnalyzer.cs:98-208,427-470, 1 thread, incl. reading the "
                          f"row's RHS; cpu: {_cpu_model()}",
                "edits_per_s": round(len(cpu_out) / cdt, 3),
                "outcomes_and_pivots_match_gpu": bool(same)}
    sens.destroy()
    eng.close()
    return out


def _self_launch(n: int) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as FRESH
    child processes (python -m torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1)
    BEFORE this process has touched the GPU -- it never does -- and relay rank 0's JSON line.  A
    process that has initialised the GPU is never re-executed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")
    if proc.returncode != 0 or not lines:
        sys.stderr.write(f"bench.py: the {n}-rank launch ended with code {proc.returncode} and "
                         f"{len(lines)} result line(s)\n")
        return proc.returncode or 1
    print(lines[-1], flush=True)
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["primal", "revised", "bb", "sens"], default="primal")
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--variant", type=int, default=0, help="rank-1 update kernel variant (0=auto)")
    ap.add_argument("--block", type=int, default=0,
                    help="pivots decided ahead and applied per sweep (0=auto, 1..8)")
    ap.add_argument("--cpu-pivots", type=int, default=-1,
                    help="pivots of the CPU baseline sample (-1: sized for ~15 s, 0: skip)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="replay captured graphs instead of eager launches with HIP events")
    ap.add_argument("--time-stride", type=int, default=0,
                    help="K-pivot paths: bracket every n-th sweep launch with HIP events (1: all; "
                         "0 = auto: every launch when steps <= 32, one in two above)")
    ap.add_argument("--bb-vars", type=int, default=512)
    ap.add_argument("--bb-cons", type=int, default=64)
    ap.add_argument("--bb-levels", type=int, default=9)
    ap.add_argument("--bb-check-levels", type=int, default=6,
                    help="levels of the same search repeated on the CPU oracle and compared")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 64 if args.workload == "primal" else 512
    if args.warmup is None:
        args.warmup = 8 if args.workload == "primal" else 64
    if args.time_stride <= 0:
        args.time_stride = 1 if args.steps <= 32 else 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _self_launch(args.gpus)
    D = Dist(args)
    out = {"primal": run_primal, "revised": run_revised, "bb": run_bb,
           "sens": run_sens}[args.workload](args, D)
    D.finish()
    if out is not None:
        if D.shared_gpu:
            out["rehearsal"] = (f"{D.world} ranks SHARE one GPU, collectives over gloo: a test of "
                                "the multi-rank plumbing, not a measurement")
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
