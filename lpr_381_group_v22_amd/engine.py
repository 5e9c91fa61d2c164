"""Handle wrappers over the C ABI (include/lpr_engine.h): Engine and device-resident Tableau."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _native as N


def _dptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _i32ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


class Engine:
    """One HIP device + one stream (lpr_engine_open / lpr_engine_close)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        N.check(N.lib.lpr_engine_open(device, C.byref(h)), "lpr_engine_open")
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            N.lib.lpr_engine_close(self._h)
            self._h = None

    def sync(self):
        N.check(N.lib.lpr_engine_sync(self._h), "lpr_engine_sync")

    @property
    def stream(self) -> int:
        return int(N.lib.lpr_engine_stream(self._h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_engine: Optional[Engine] = None


def default_engine() -> Engine:
    global _default_engine
    if _default_engine is None or _default_engine._h is None:
        _default_engine = Engine(0)
    return _default_engine


class Tableau:
    """A (rows x cols) fp64 simplex tableau resident in HBM (lpr_tableau_*)."""

    def __init__(self, engine: Engine, handle: C.c_void_p):
        self.engine = engine
        self._h = handle
        r, c, ld = C.c_int(), C.c_int(), C.c_int()
        N.check(N.lib.lpr_tableau_shape(handle, C.byref(r), C.byref(c), C.byref(ld)),
                "lpr_tableau_shape")
        self.rows, self.cols, self.ld = r.value, c.value, ld.value

    # ---- constructors -------------------------------------------------------------------
    @classmethod
    def from_lp(cls, engine: Engine, objective: Sequence[float], A: np.ndarray,
                relation: Sequence[int], rhs: Sequence[float], is_max: bool = True,
                ncoef: Optional[Sequence[int]] = None) -> "Tableau":
        obj = np.ascontiguousarray(objective, dtype=np.float64)
        n = obj.shape[0]
        rhs_a = np.ascontiguousarray(rhs, dtype=np.float64)
        m = rhs_a.shape[0]
        A = np.ascontiguousarray(A, dtype=np.float64).reshape(m, -1) if m else np.zeros((0, n))
        lda = A.shape[1] if m else n
        rel = np.ascontiguousarray(relation, dtype=np.int8)
        nc = None if ncoef is None else np.ascontiguousarray(ncoef, dtype=np.int32)
        h = C.c_void_p()
        N.check(N.lib.lpr_tableau_from_lp(
            engine._h, n, m, _dptr(obj), _dptr(A) if m and lda else None, lda,
            _i32ptr(nc), rel.ctypes.data_as(C.POINTER(C.c_int8)) if m else None,
            _dptr(rhs_a) if m else None, 1 if is_max else 0, C.byref(h)), "lpr_tableau_from_lp")
        return cls(engine, h)

    @classmethod
    def from_array(cls, engine: Engine, T: np.ndarray,
                   basis: Optional[Sequence[int]] = None) -> "Tableau":
        T = np.ascontiguousarray(T, dtype=np.float64)
        b = None if basis is None else np.ascontiguousarray(basis, dtype=np.int32)
        h = C.c_void_p()
        N.check(N.lib.lpr_tableau_create(engine._h, T.shape[0], T.shape[1], _dptr(T),
                                         _i32ptr(b), C.byref(h)), "lpr_tableau_create")
        return cls(engine, h)

    @classmethod
    def synthetic(cls, engine: Engine, m: int, n: int, seed: int) -> "Tableau":
        h = C.c_void_p()
        N.check(N.lib.lpr_tableau_synthetic(engine._h, m, n, C.c_uint64(seed), C.byref(h)),
                "lpr_tableau_synthetic")
        return cls(engine, h)

    def destroy(self):
        if self._h:
            N.lib.lpr_tableau_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # ---- the pivot loop -----------------------------------------------------------------
    def solve(self, max_pivots: int = 0, time_kernels: bool = False, batch: int = 0,
              variant: int = 0, block: int = 0) -> N.SolveResult:
        """block: pivots decided ahead and applied per sweep on large tableaux (0 auto, 1..8)."""
        opts = N.SolveOpts(max_pivots=max_pivots, time_kernels=int(time_kernels),
                           batch=batch, variant=variant, block=block)
        res = N.SolveResult()
        N.check(N.lib.lpr_primal_solve(self._h, C.byref(opts), C.byref(res)), "lpr_primal_solve")
        return res

    def select_entering(self) -> int:
        v = C.c_int32()
        N.check(N.lib.lpr_select_entering(self._h, C.byref(v)), "lpr_select_entering")
        return v.value

    def select_leaving(self, col: int) -> int:
        v = C.c_int32()
        N.check(N.lib.lpr_select_leaving(self._h, col, C.byref(v)), "lpr_select_leaving")
        return v.value

    def pivot(self, row: int, col: int) -> None:
        N.check(N.lib.lpr_pivot(self._h, row, col), "lpr_pivot")

    # ---- results ------------------------------------------------------------------------
    def extract_solution(self, n: int) -> Tuple[np.ndarray, float]:
        x = np.zeros(max(n, 1), dtype=np.float64)
        z = C.c_double()
        N.check(N.lib.lpr_extract_solution(self._h, n, _dptr(x), C.byref(z)),
                "lpr_extract_solution")
        return x[:n], z.value

    def read(self) -> np.ndarray:
        out = np.empty((self.rows, self.cols), dtype=np.float64)
        N.check(N.lib.lpr_tableau_read(self._h, _dptr(out)), "lpr_tableau_read")
        return out

    def read_block(self, row0: int, nrows: int, col0: int, ncols: int) -> np.ndarray:
        out = np.empty((nrows, ncols), dtype=np.float64)
        N.check(N.lib.lpr_tableau_read_block(self._h, row0, nrows, col0, ncols, _dptr(out)),
                "lpr_tableau_read_block")
        return out

    def basis(self) -> np.ndarray:
        out = np.zeros(max(self.rows - 1, 1), dtype=np.int32)
        N.check(N.lib.lpr_basis_read(self._h, _i32ptr(out)), "lpr_basis_read")
        return out[: self.rows - 1]

    def pivot_log(self, cap: int = 1 << 20) -> np.ndarray:
        rows = np.zeros(cap, dtype=np.int32)
        cols = np.zeros(cap, dtype=np.int32)
        cnt = C.c_int64()
        N.check(N.lib.lpr_pivot_log_read(self._h, _i32ptr(rows), _i32ptr(cols), cap,
                                         C.byref(cnt)), "lpr_pivot_log_read")
        k = cnt.value
        return np.stack([rows[:k], cols[:k]], axis=1)

    # ---- cutting-plane side path (DualSimplex.cs, PrimalSimplexSolver2.cs, CuttingPlaneSolver.cs)
    def dual_solve(self, max_iters: int = 10000, print_steps: bool = True,
                   hard_cap: int = 0) -> N.SolveResult:
        res = N.SolveResult()
        N.check(N.lib.lpr_dual_solve(self._h, max_iters, 1 if print_steps else 0, hard_cap,
                                     C.byref(res)), "lpr_dual_solve")
        return res

    def primal2_solve(self, max_iters: int = 10000, print_steps: bool = False,
                      hard_cap: int = 0) -> N.SolveResult:
        res = N.SolveResult()
        N.check(N.lib.lpr_primal2_solve(self._h, max_iters, 1 if print_steps else 0, hard_cap,
                                        C.byref(res)), "lpr_primal2_solve")
        return res

    def cutting_plane(self, max_cuts: int = 8, hard_cap: int = 0) -> Tuple[int, int]:
        """Returns (exit code, cuts); the tableau grows by one row per cut."""
        ex, cuts = C.c_int32(), C.c_int32()
        N.check(N.lib.lpr_cutting_plane(self._h, max_cuts, hard_cap, C.byref(ex), C.byref(cuts)),
                "lpr_cutting_plane")
        r, c, ld = C.c_int(), C.c_int(), C.c_int()
        N.check(N.lib.lpr_tableau_shape(self._h, C.byref(r), C.byref(c), C.byref(ld)),
                "lpr_tableau_shape")
        self.rows, self.cols, self.ld = r.value, c.value, ld.value
        return ex.value, cuts.value

    def cut_log(self, cap: int = 1 << 16):
        buf = np.zeros(cap * 3, dtype=np.int32)
        n = C.c_int64()
        N.check(N.lib.lpr_cut_log_read(self._h, _i32ptr(buf), cap, C.byref(n)),
                "lpr_cut_log_read")
        return [tuple(v) for v in buf[: 3 * n.value].reshape(-1, 3).tolist()]

    def kernel_stats(self) -> Tuple[int, float, float]:
        n, tot, avg = C.c_int64(), C.c_double(), C.c_double()
        N.check(N.lib.lpr_tableau_kernel_stats(self._h, C.byref(n), C.byref(tot), C.byref(avg)),
                "lpr_tableau_kernel_stats")
        return n.value, tot.value, avg.value

    def step_stats(self) -> Tuple[int, float]:
        """(steps timed, summed milliseconds): a step = one sweep of K pivots beside the loop
        heads of the next K (K-pivot paths with time_kernels)."""
        n, tot = C.c_int64(), C.c_double()
        N.check(N.lib.lpr_tableau_step_stats(self._h, C.byref(n), C.byref(tot)),
                "lpr_tableau_step_stats")
        return n.value, tot.value

    def head_stamps(self):
        """Diagnostic stamps of the lead loop-head workgroup (variant bit 16): (array of shape
        (64, 12) in 10 ns ticks, XCC id, 1 if the hand-offs went through the XCD's L2)."""
        buf = np.zeros(64 * 12 + 8, dtype=np.uint64)
        cnt = C.c_int64()
        N.check(N.lib.lpr_debug_head_stamps(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            buf.size, C.byref(cnt)), "lpr_debug_head_stamps")
        return buf[:64 * 12].reshape(64, 12), int(buf[64 * 12]), int(buf[64 * 12 + 1])

    def launch_stamps(self):
        """Diagnostic (variant bit 16, two-stream path): a ring over 8 steps, 10 ns ticks of
        [heads entry, tableau ready, last head done, completion published, sweep entry, sweep saw
        the heads' word, -, -]."""
        buf = np.zeros(64 * 12 + 8 + 64, dtype=np.uint64)
        cnt = C.c_int64()
        N.check(N.lib.lpr_debug_head_stamps(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            buf.size, C.byref(cnt)), "lpr_debug_head_stamps")
        return buf[64 * 12 + 8:].reshape(8, 8)


class RevisedState:
    """Device-resident state of the revised primal simplex (lpr_revised_*)."""

    def __init__(self, engine: Engine, handle: C.c_void_p, n: int, m: int):
        self.engine = engine
        self._h = handle
        self.n, self.m = n, m

    @classmethod
    def create(cls, engine: Engine, objective: Sequence[float], A: np.ndarray,
               b: Sequence[float], is_min: bool) -> "RevisedState":
        obj = np.ascontiguousarray(objective, dtype=np.float64)
        bb = np.ascontiguousarray(b, dtype=np.float64)
        n, m = obj.shape[0], bb.shape[0]
        A = np.ascontiguousarray(A, dtype=np.float64).reshape(m, n) if m and n else \
            np.zeros((max(m, 1), max(n, 1)))
        h = C.c_void_p()
        N.check(N.lib.lpr_revised_create(engine._h, n, m, _dptr(obj), _dptr(A), max(n, 1),
                                         _dptr(bb), 1 if is_min else 0, C.byref(h)),
                "lpr_revised_create")
        return cls(engine, h, n, m)

    @classmethod
    def synthetic(cls, engine: Engine, m: int, n: int, seed: int) -> "RevisedState":
        h = C.c_void_p()
        N.check(N.lib.lpr_revised_synthetic(engine._h, m, n, C.c_uint64(seed), C.byref(h)),
                "lpr_revised_synthetic")
        return cls(engine, h, n, m)

    def destroy(self):
        if self._h:
            N.lib.lpr_revised_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def solve(self, max_pivots: int = 0, batch: int = 0) -> N.RevisedResult:
        opts = N.SolveOpts(max_pivots=max_pivots, time_kernels=0, batch=batch, variant=0,
                           block=0)
        res = N.RevisedResult()
        N.check(N.lib.lpr_revised_solve(self._h, C.byref(opts), C.byref(res)),
                "lpr_revised_solve")
        return res

    def solution(self) -> Tuple[np.ndarray, float]:
        x = np.zeros(self.n, dtype=np.float64)
        z = C.c_double()
        N.check(N.lib.lpr_revised_solution(self._h, _dptr(x), C.byref(z)), "lpr_revised_solution")
        return x, z.value

    def basis(self) -> np.ndarray:
        out = np.zeros(self.m, dtype=np.int32)
        N.check(N.lib.lpr_revised_basis_read(self._h, _i32ptr(out)), "lpr_revised_basis_read")
        return out

    def log(self, cap: int = 1 << 20) -> np.ndarray:
        r = np.zeros(cap, dtype=np.int32)
        e = np.zeros(cap, dtype=np.int32)
        lv = np.zeros(cap, dtype=np.int32)
        cnt = C.c_int64()
        N.check(N.lib.lpr_revised_log_read(self._h, _i32ptr(r), _i32ptr(e), _i32ptr(lv), cap,
                                           C.byref(cnt)), "lpr_revised_log_read")
        k = cnt.value
        return np.stack([r[:k], e[:k], lv[:k]], axis=1)

    def binv(self) -> np.ndarray:
        out = np.empty((self.m, self.m), dtype=np.float64)
        N.check(N.lib.lpr_revised_binv_read(self._h, _dptr(out)), "lpr_revised_binv_read")
        return out

    def xb(self) -> np.ndarray:
        out = np.empty(self.m, dtype=np.float64)
        N.check(N.lib.lpr_revised_xb_read(self._h, _dptr(out)), "lpr_revised_xb_read")
        return out

    def binv_a(self, fetch: bool = True) -> Tuple[Optional[np.ndarray], float]:
        """B^-1 * A on the fp64 matrix cores; returns (product or None, kernel milliseconds)."""
        out = np.empty((self.m, self.n), dtype=np.float64) if fetch else None
        ms = C.c_double()
        N.check(N.lib.lpr_revised_binv_a(self._h, _dptr(out), C.byref(ms)), "lpr_revised_binv_a")
        return out, ms.value


    # ---- IterationSnapshots (RevisedPrimalSimplexSolver.cs:294-387) -----------------------
    def step(self) -> N.RevisedSnapshotInfo:
        """One pass of Solve()'s loop incl. the post-pivot quantities (lpr_revised_step)."""
        info = N.RevisedSnapshotInfo()
        N.check(N.lib.lpr_revised_step(self._h, C.byref(info)), "lpr_revised_step")
        return info

    def snapshot(self):
        """(y, rc[n+m], u_pre, ratios_pre, basis_pre, xB) as left by the last step()."""
        y = np.zeros(self.m)
        rc = np.zeros(self.n + self.m)
        u = np.zeros(self.m)
        ratios = np.zeros(self.m)
        bpre = np.zeros(self.m, dtype=np.int32)
        xb = np.zeros(self.m)
        N.check(N.lib.lpr_revised_snapshot_read(self._h, _dptr(y), _dptr(rc), _dptr(u),
                                                _dptr(ratios), _i32ptr(bpre), _dptr(xb)),
                "lpr_revised_snapshot_read")
        return y, rc, u, ratios, bpre, xb

    def binv_a_exact(self) -> np.ndarray:
        out = np.empty((self.m, self.n), dtype=np.float64)
        N.check(N.lib.lpr_revised_binv_a_exact(self._h, _dptr(out)), "lpr_revised_binv_a_exact")
        return out


class SensState:
    """Device-resident tableau of a SensitivityAnalyzer (lpr_sens_*).  Edits return the
    lpr_sens_outcome of the re-solve."""

    def __init__(self, engine: Engine, handle: C.c_void_p):
        self.engine = engine
        self._h = handle

    @classmethod
    def create(cls, engine: Engine, final_tableau: np.ndarray, solution: Sequence[float],
               z: float) -> "SensState":
        T = np.ascontiguousarray(final_tableau, dtype=np.float64)
        sol = np.ascontiguousarray(solution, dtype=np.float64)
        h = C.c_void_p()
        N.check(N.lib.lpr_sens_create(engine._h, _dptr(T), T.shape[0], T.shape[1],
                                      _dptr(sol) if sol.size else None, sol.shape[0], float(z),
                                      C.byref(h)), "lpr_sens_create")
        return cls(engine, h)

    @classmethod
    def from_tableau(cls, tableau: Tableau, n_decision: int) -> "SensState":
        h = C.c_void_p()
        N.check(N.lib.lpr_sens_create_from_tableau(tableau._h, n_decision, C.byref(h)),
                "lpr_sens_create_from_tableau")
        return cls(tableau.engine, h)

    def destroy(self):
        if self._h:
            N.lib.lpr_sens_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def shape(self):
        """(rows, cols, nsol, nbasic, z, pivots of the last edit)"""
        r, c, ns, nb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        z, p = C.c_double(), C.c_int64()
        N.check(N.lib.lpr_sens_shape(self._h, C.byref(r), C.byref(c), C.byref(ns), C.byref(nb),
                                     C.byref(z), C.byref(p)), "lpr_sens_shape")
        return r.value, c.value, ns.value, nb.value, z.value, p.value

    def read(self, tableau: bool = True):
        """(tableau or None, basicVars, solutionVector)"""
        r, c, ns, nb, _, _ = self.shape()
        T = np.empty((r, c), dtype=np.float64) if tableau else None
        basic = np.zeros(max(nb, 1), dtype=np.int32)
        sol = np.zeros(max(ns, 1), dtype=np.float64)
        N.check(N.lib.lpr_sens_read(self._h, _dptr(T), _i32ptr(basic), _dptr(sol)),
                "lpr_sens_read")
        return T, basic[:nb], sol[:ns]

    def read_block(self, row0: int, nrows: int, col0: int, ncols: int) -> np.ndarray:
        out = np.empty((nrows, ncols), dtype=np.float64)
        N.check(N.lib.lpr_sens_read_block(self._h, row0, nrows, col0, ncols, _dptr(out)),
                "lpr_sens_read_block")
        return out

    def basic_row(self, col: int) -> int:
        r = C.c_int32()
        N.check(N.lib.lpr_sens_basic_row(self._h, col, C.byref(r)), "lpr_sens_basic_row")
        return r.value

    def log(self):
        cnt = C.c_int64()
        N.check(N.lib.lpr_sens_log_read(self._h, None, 0, C.byref(cnt)), "lpr_sens_log_read")
        n = cnt.value
        buf = np.zeros(max(3 * n, 3), dtype=np.int32)
        N.check(N.lib.lpr_sens_log_read(self._h, _i32ptr(buf), n, C.byref(cnt)),
                "lpr_sens_log_read")
        return [tuple(v) for v in buf[:3 * n].reshape(-1, 3).tolist()]

    def column_fold(self, w: Sequence[float], init: Optional[Sequence[float]],
                    ncols: int) -> np.ndarray:
        ww = np.ascontiguousarray(w, dtype=np.float64)
        ii = None if init is None else np.ascontiguousarray(init, dtype=np.float64)
        out = np.zeros(max(ncols, 1), dtype=np.float64)
        N.check(N.lib.lpr_sens_column_fold(self._h, _dptr(ww), ww.shape[0], _dptr(ii), ncols,
                                           _dptr(out)), "lpr_sens_column_fold")
        return out[:ncols]

    def _edit(self, name: str, *args) -> int:
        oc = C.c_int32()
        N.check(getattr(N.lib, name)(self._h, *args, C.byref(oc)), name)
        return oc.value

    def resolve_all(self) -> int:
        return self._edit("lpr_sens_resolve_all")

    def change_nonbasic_cbar(self, index: int, new_cbar: float) -> int:
        return self._edit("lpr_sens_change_nonbasic_cbar", index, float(new_cbar))

    def change_basic(self, col: int, delta: float) -> int:
        return self._edit("lpr_sens_change_basic", col, float(delta))

    def change_rhs(self, k: int, new_b: float) -> int:
        return self._edit("lpr_sens_change_rhs", k, float(new_b))

    def change_nonbasic_column(self, row: int, col: int, new_val: float) -> int:
        return self._edit("lpr_sens_change_nonbasic_column", row, col, float(new_val))

    def add_activity(self, c_new: float, a_new: Sequence[float]) -> int:
        a = np.ascontiguousarray(a_new, dtype=np.float64)
        return self._edit("lpr_sens_add_activity", float(c_new), _dptr(a), a.shape[0])

    def add_constraint(self, tech: Sequence[float], rhs: float) -> int:
        t = np.ascontiguousarray(tech, dtype=np.float64)
        return self._edit("lpr_sens_add_constraint", _dptr(t), t.shape[0], float(rhs))
