"""ctypes binding of oracle/_build/liblpr_oracle.so (TEST ONLY -- the checker, never the product)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "liblpr_oracle.so")

_D = C.POINTER(C.c_double)
_I32 = C.POINTER(C.c_int32)
_I8 = C.POINTER(C.c_int8)
_I64 = C.POINTER(C.c_int64)


def _newest_src() -> float:
    return max(os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in os.listdir(ORACLE_DIR)
               if f.endswith((".c", ".h")) or f == "Makefile")


def build_oracle() -> str:
    # LPR_ORACLE_SANITIZE=1 (tests/test_sanitizers.py): the -fsanitize=address,undefined build
    if os.environ.get("LPR_ORACLE_SANITIZE") == "1":
        so = os.path.join(ORACLE_DIR, "_build", "liblpr_oracle_asan.so")
        if not os.path.exists(so) or os.path.getmtime(so) < _newest_src():
            subprocess.run(["make", "-C", ORACLE_DIR, "asan"], check=True, capture_output=True)
        return so
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < _newest_src():
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return ORACLE_SO


def _dp(a):
    return None if a is None else a.ctypes.data_as(_D)


def _ip(a):
    return None if a is None else a.ctypes.data_as(_I32)


class Oracle:
    def __init__(self):
        self.lib = lib = C.CDLL(build_oracle())
        lib.orc_splitmix64.restype = C.c_uint64
        lib.orc_splitmix64.argtypes = [C.c_uint64]
        lib.orc_u01.restype = C.c_double
        lib.orc_u01.argtypes = [C.c_uint64] * 4
        lib.orc_gen_dense_lp.restype = None
        lib.orc_gen_dense_lp.argtypes = [C.c_int, C.c_int, C.c_uint64, _D, _D, _D]
        lib.orc_gen_dense_tableau.restype = None
        lib.orc_gen_dense_tableau.argtypes = [C.c_int, C.c_int, C.c_uint64, _D, _I32]
        lib.orc_primal_build.restype = C.c_int
        lib.orc_primal_build.argtypes = [C.c_int, C.c_int, _D, _D, C.c_int, _I32, _I8, _D,
                                         C.c_int, _D, _I32]
        lib.orc_find_entering.restype = C.c_int
        lib.orc_find_entering.argtypes = [_D, C.c_int, C.c_int]
        lib.orc_find_leaving.restype = C.c_int
        lib.orc_find_leaving.argtypes = [_D, C.c_int, C.c_int, C.c_int]
        lib.orc_pivot.restype = None
        lib.orc_pivot.argtypes = [_D, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.orc_primal_solve.restype = C.c_int
        lib.orc_primal_solve.argtypes = [_D, C.c_int, C.c_int, _I32, C.c_int64, _I32, _I32,
                                         C.c_int64, _I64]
        lib.orc_extract_solution.restype = None
        lib.orc_extract_solution.argtypes = [_D, C.c_int, C.c_int, C.c_int, _D, _D]

        lib.orc_matmul_skip.restype = None
        lib.orc_matmul_skip.argtypes = [_D, C.c_int, C.c_int, _D, C.c_int, _D]
        lib.orc_update_binverse.restype = C.c_int
        lib.orc_update_binverse.argtypes = [_D, C.c_int, C.c_int, _D, _D]
        lib.orc_revised_solve.restype = C.c_int
        lib.orc_revised_solve.argtypes = [C.c_int, C.c_int, _D, _D, _D, C.c_int, C.c_int64, _D,
                                          _D, _I32, _D, _D, _I32, _I32, _I32, C.c_int64, _I64]

        lib.orc_round4.restype = C.c_double
        lib.orc_round4.argtypes = [C.c_double]
        lib.orc_round_int.restype = C.c_double
        lib.orc_round_int.argtypes = [C.c_double]
        _IP = C.POINTER(C.c_int)
        lib.orc_bb_solve.restype = C.c_int
        lib.orc_bb_solve.argtypes = [_D, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _D, _D, _IP,
                                     _IP, _IP, _I32, _I32, _I32, _I32, _D, _I32, _D, C.c_int, _IP,
                                     _I32, _I32, C.c_int64, _I64]
        lib.orc_bb_add_constraint.restype = C.c_int
        lib.orc_bb_add_constraint.argtypes = [_D, C.c_int, C.c_int, _D, C.c_int, _D]
        lib.orc_bb_dual_simplex.restype = C.c_int
        lib.orc_bb_dual_simplex.argtypes = [_D, C.c_int, C.c_int, _D, _IP, _I32, C.c_int64, _I64]

        for fn in (lib.orc_dual_solve, lib.orc_primal2_solve):
            fn.restype = C.c_int
            fn.argtypes = [_D, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _I32, C.c_int64,
                           _I64, _I64]
        lib.orc_cutting_plane.restype = C.c_int
        lib.orc_cutting_plane.argtypes = [_D, _IP, C.c_int, C.c_int, C.c_int, C.c_int64, _I32,
                                          C.c_int64, _I64, _IP]

        lib.orc_sens_create.restype = C.c_void_p
        lib.orc_sens_create.argtypes = [_D, C.c_int, C.c_int, _D, C.c_int, C.c_double, _I32,
                                        C.c_int]
        lib.orc_sens_destroy.restype = None
        lib.orc_sens_destroy.argtypes = [C.c_void_p]
        lib.orc_sens_shape.restype = None
        lib.orc_sens_shape.argtypes = [C.c_void_p, _IP, _IP, _IP, _IP, _D]
        lib.orc_sens_read.restype = None
        lib.orc_sens_read.argtypes = [C.c_void_p, _D, _I32, _D]
        lib.orc_sens_log_read.restype = C.c_int64
        lib.orc_sens_log_read.argtypes = [C.c_void_p, _I32, C.c_int64]
        lib.orc_sens_resolve_all.restype = C.c_int
        lib.orc_sens_resolve_all.argtypes = [C.c_void_p]
        lib.orc_sens_change_nonbasic_cbar.restype = C.c_int
        lib.orc_sens_change_nonbasic_cbar.argtypes = [C.c_void_p, C.c_int, C.c_double]
        lib.orc_sens_change_basic.restype = C.c_int
        lib.orc_sens_change_basic.argtypes = [C.c_void_p, C.c_int, C.c_double]
        lib.orc_sens_change_rhs.restype = C.c_int
        lib.orc_sens_change_rhs.argtypes = [C.c_void_p, C.c_int, C.c_double]
        lib.orc_sens_change_nonbasic_column.restype = C.c_int
        lib.orc_sens_change_nonbasic_column.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
        lib.orc_sens_add_activity.restype = C.c_int
        lib.orc_sens_add_activity.argtypes = [C.c_void_p, C.c_double, _D]
        lib.orc_sens_add_constraint.restype = C.c_int
        lib.orc_sens_add_constraint.argtypes = [C.c_void_p, _D, C.c_int, C.c_double]

    def sens(self, final_tableau, solution, z, basic):
        return OracleSens(self.lib, final_tableau, solution, z, basic)

    # ---- cutting-plane side path (T: row 0 = objective row; in place) ----
    def _cut_solver(self, fn, T, max_iters, print_steps, hard_cap, log_cap):
        assert T.flags["C_CONTIGUOUS"] and T.dtype == np.float64
        log = np.zeros(log_cap * 3, dtype=np.int32)
        nlog, piv = C.c_int64(), C.c_int64()
        rc = fn(_dp(T), T.shape[0], T.shape[1], max_iters, 1 if print_steps else 0, hard_cap,
                _ip(log), log_cap, C.byref(nlog), C.byref(piv))
        q = min(nlog.value, log_cap)
        return rc, piv.value, [tuple(v) for v in log[:3 * q].reshape(-1, 3).tolist()]

    def dual_solve(self, T, max_iters=10000, print_steps=True, hard_cap=0, log_cap=1 << 14):
        return self._cut_solver(self.lib.orc_dual_solve, T, max_iters, print_steps, hard_cap,
                                log_cap)

    def primal2_solve(self, T, max_iters=10000, print_steps=False, hard_cap=0, log_cap=1 << 14):
        return self._cut_solver(self.lib.orc_primal2_solve, T, max_iters, print_steps, hard_cap,
                                log_cap)

    def cutting_plane(self, T, max_cuts=8, hard_cap=0, log_cap=1 << 14):
        """Returns (exit code, cuts, final tableau (rows grown), log)."""
        R, Cc = T.shape
        buf = np.zeros((R + max_cuts, Cc))
        buf[:R] = T
        r_io = C.c_int(R)
        log = np.zeros(log_cap * 3, dtype=np.int32)
        nlog = C.c_int64()
        cuts = C.c_int()
        rc = self.lib.orc_cutting_plane(_dp(buf), C.byref(r_io), R + max_cuts, Cc, max_cuts,
                                        hard_cap, _ip(log), log_cap, C.byref(nlog),
                                        C.byref(cuts))
        q = min(nlog.value, log_cap)
        return rc, cuts.value, buf[:r_io.value].copy(), \
            [tuple(v) for v in log[:3 * q].reshape(-1, 3).tolist()]

    # ---- branch & bound ----
    def round4(self, x):
        return self.lib.orc_round4(x)

    def round_int(self, x):
        return self.lib.orc_round_int(x)

    def bb_solve(self, final_tableau, nvars, enable_pruning=False, node_cap=20, rec_cap=4096,
                 piv_cap=1 << 16):
        T = np.ascontiguousarray(final_tableau, dtype=np.float64)
        x = np.zeros(max(nvars, 1))
        z = C.c_double()
        found, best, processed, nrec = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rp = np.zeros(rec_cap, dtype=np.int32)
        rk = np.zeros(rec_cap, dtype=np.int32)
        rd = np.zeros(rec_cap, dtype=np.int32)
        rv = np.zeros(rec_cap, dtype=np.int32)
        rb = np.zeros(rec_cap)
        rs = np.zeros(rec_cap, dtype=np.int32)
        rz = np.zeros(rec_cap)
        pop = np.zeros(rec_cap, dtype=np.int32)
        piv = np.zeros(piv_cap * 4, dtype=np.int32)
        npiv = C.c_int64()
        st = self.lib.orc_bb_solve(_dp(T), T.shape[0], T.shape[1], nvars,
                                   1 if enable_pruning else 0, node_cap, _dp(x), C.byref(z),
                                   C.byref(found), C.byref(best), C.byref(processed), _ip(rp),
                                   _ip(rk), _ip(rd), _ip(rv), _dp(rb), _ip(rs), _dp(rz), rec_cap,
                                   C.byref(nrec), _ip(pop), _ip(piv), piv_cap, C.byref(npiv))
        k = min(nrec.value, rec_cap)
        recs = [dict(parent=int(rp[i]), kind=int(rk[i]), depth=int(rd[i]), var=int(rv[i]),
                     bound=float(rb[i]), status=int(rs[i]), z=float(rz[i])) for i in range(k)]
        q = min(npiv.value, piv_cap)
        return dict(status=st, x=x[:nvars] if found.value else None, z=z.value,
                    found=bool(found.value), best_node=best.value, processed=processed.value,
                    records=recs, pop_order=pop[:min(processed.value, rec_cap)].tolist(),
                    trace=[tuple(v) for v in piv[:4 * q].reshape(-1, 4).tolist()])

    def bb_add_constraint(self, base, con):
        base = np.ascontiguousarray(base, dtype=np.float64)
        con = np.ascontiguousarray(con, dtype=np.float64)
        out = np.zeros((base.shape[0] + 1, base.shape[1] + 1))
        self.lib.orc_bb_add_constraint(_dp(base), base.shape[0], base.shape[1], _dp(con),
                                       con.shape[0], _dp(out))
        return out

    def bb_round_tableau(self, T):
        """RoundTableau :552-567 on a copy."""
        T = np.array(T, dtype=np.float64, order="C", copy=True)
        self.lib.orc_bb_round_tableau.restype = None
        self.lib.orc_bb_round_tableau(_dp(T), T.shape[0], T.shape[1])
        return T

    def bb_node_info(self, T, nvars):
        """A popped node: (rounded tableau :1047, z :892-897, decision values :805-857)."""
        T = np.array(T, dtype=np.float64, order="C", copy=True)
        z = C.c_double()
        vals = np.zeros(max(nvars, 1))
        self.lib.orc_bb_node_info.restype = None
        self.lib.orc_bb_node_info(_dp(T), T.shape[0], T.shape[1], nvars, C.byref(z), _dp(vals))
        return T, z.value, vals[:nvars]

    def bb_dual_simplex(self, start, piv_cap=1 << 14):
        start = np.ascontiguousarray(start, dtype=np.float64)
        out = np.zeros_like(start)
        npv = C.c_int()
        piv = np.zeros(piv_cap * 4, dtype=np.int32)
        n = C.c_int64()
        rc = self.lib.orc_bb_dual_simplex(_dp(start), start.shape[0], start.shape[1], _dp(out),
                                          C.byref(npv), _ip(piv), piv_cap, C.byref(n))
        q = min(n.value, piv_cap)
        return rc, out, npv.value, [tuple(v[1:]) for v in piv[:4 * q].reshape(-1, 4).tolist()]

    # ---- revised ----
    def matmul_skip(self, A, B):
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        R = np.zeros((A.shape[0], B.shape[1]))
        self.lib.orc_matmul_skip(_dp(A), A.shape[0], A.shape[1], _dp(B), B.shape[1], _dp(R))
        return R

    def update_binverse(self, Binv, pivot_row, u):
        """In place on Binv; returns 0 or ORC_PIVOT_TOO_SMALL."""
        m = Binv.shape[0]
        scratch = np.zeros((m, m))
        u = np.ascontiguousarray(u, dtype=np.float64)
        return self.lib.orc_update_binverse(_dp(Binv), m, pivot_row, _dp(u), _dp(scratch))

    def revised_solve(self, objective, A, b, is_min=False, max_iter=0, log_cap=1 << 16):
        obj = np.ascontiguousarray(objective, dtype=np.float64)
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        m, n = A.shape
        x = np.zeros(n)
        z = C.c_double()
        basis = np.zeros(m, dtype=np.int32)
        Binv = np.zeros((m, m))
        xB = np.zeros(m)
        lr = np.zeros(log_cap, dtype=np.int32)
        le = np.zeros(log_cap, dtype=np.int32)
        ll = np.zeros(log_cap, dtype=np.int32)
        it = C.c_int64()
        st = self.lib.orc_revised_solve(n, m, _dp(obj), _dp(A), _dp(b), 1 if is_min else 0,
                                        max_iter, _dp(x), C.byref(z), _ip(basis), _dp(Binv),
                                        _dp(xB), _ip(lr), _ip(le), _ip(ll), log_cap, C.byref(it))
        k = min(it.value, log_cap)
        return dict(status=st, iterations=it.value, x=x, z=z.value, basis=basis, Binv=Binv,
                    xB=xB, log=np.stack([lr[:k], le[:k], ll[:k]], axis=1))

    def revised_iterate_from(self, objective, A, b, Binv, basis, is_min=False):
        """One pass of Solve()'s loop (:89-215) from a given state.  Binv / basis are copied; the
        updated ones are returned with the pre-pivot vectors."""
        obj = np.ascontiguousarray(objective, dtype=np.float64)
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        m, n = A.shape
        Binv = np.array(Binv, dtype=np.float64, order="C", copy=True)
        basis = np.array(basis, dtype=np.int32, copy=True)
        xB, y, rcX, rcS, u, ratios = (np.zeros(m), np.zeros(m), np.zeros(n), np.zeros(m),
                                      np.zeros(m), np.zeros(m))
        ent, lrow = C.c_int32(), C.c_int32()
        st = self.lib.orc_revised_iterate_from(n, m, _dp(obj), _dp(A), _dp(b), 1 if is_min else 0,
                                               _dp(Binv), _ip(basis), _dp(xB), _dp(y), _dp(rcX),
                                               _dp(rcS), _dp(u), _dp(ratios), C.byref(ent),
                                               C.byref(lrow))
        return dict(status=st, entering=ent.value, leaving_row=lrow.value, xB=xB, y=y, rcX=rcX,
                    rcS=rcS, u=u, ratios=ratios, Binv=Binv, basis=basis)

    def revised_trace(self, objective, A, b, is_min=False, max_iter=0, cap=64):
        """Every CaptureSnapshot (:294-387) as a dict of numbers (layout: oracle_revised.c)."""
        obj = np.ascontiguousarray(objective, dtype=np.float64)
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        m, n = A.shape
        self.lib.orc_revised_trace_stride.restype = C.c_int64
        S = self.lib.orc_revised_trace_stride(n, m)
        buf = np.zeros(cap * S)
        x = np.zeros(n)
        z = C.c_double()
        basis = np.zeros(m, dtype=np.int32)
        ns, it = C.c_int64(), C.c_int64()
        st = self.lib.orc_revised_solve_trace(n, m, _dp(obj), _dp(A), _dp(b), 1 if is_min else 0,
                                              C.c_int64(max_iter), _dp(x), C.byref(z), _ip(basis),
                                              _dp(buf), C.c_int64(cap), C.byref(ns), C.byref(it))
        snaps = []
        for k in range(min(ns.value, cap)):
            r = buf[k * S:(k + 1) * S]
            o = 6
            d = dict(entering=int(r[0]), leaving_row=int(r[1]), leaving_var=int(r[2]),
                     rc_pre=r[3], z_working=r[4], z_original=r[5])
            for name, ln in (("y", m), ("rcX", n), ("rcS", m), ("u_pre", m), ("ratios_pre", m),
                             ("basis_pre", m), ("basis_post", m), ("xB", m)):
                d[name] = r[o:o + ln].copy()
                o += ln
            d["BInvA"] = r[o:o + m * n].reshape(m, n).copy()
            o += m * n
            d["BInv"] = r[o:o + m * m].reshape(m, m).copy()
            d["basis_pre"] = d["basis_pre"].astype(np.int32)
            d["basis_post"] = d["basis_post"].astype(np.int32)
            snaps.append(d)
        return dict(status=st, iterations=it.value, snapshots=snaps, count=ns.value, x=x,
                    z=z.value, basis=basis)

    # ---- generator ----
    def u01(self, seed, stream, i, j) -> float:
        return self.lib.orc_u01(seed, stream, i, j)

    def gen_dense_lp(self, m: int, n: int, seed: int):
        c = np.zeros(n)
        A = np.zeros((m, n))
        b = np.zeros(m)
        self.lib.orc_gen_dense_lp(m, n, seed, _dp(c), _dp(A), _dp(b))
        return c, A, b

    def gen_dense_tableau(self, m: int, n: int, seed: int):
        T = np.zeros((m + 1, n + m + 1))
        basis = np.zeros(m, dtype=np.int32)
        self.lib.orc_gen_dense_tableau(m, n, seed, _dp(T), _ip(basis))
        return T, basis

    # ---- primal ----
    def primal_build(self, objective, A, relation, rhs, is_max=True, ncoef=None):
        obj = np.ascontiguousarray(objective, dtype=np.float64)
        n = obj.shape[0]
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        m = rhs.shape[0]
        A = np.ascontiguousarray(A, dtype=np.float64).reshape(m, -1) if m else np.zeros((0, n))
        lda = A.shape[1] if m else n
        rel = np.ascontiguousarray(relation, dtype=np.int8)
        nc = None if ncoef is None else np.ascontiguousarray(ncoef, dtype=np.int32)
        T = np.zeros((m + 1, n + m + 1))
        basis = np.zeros(max(m, 1), dtype=np.int32)
        rc = self.lib.orc_primal_build(n, m, _dp(obj), _dp(A), lda, _ip(nc),
                                       rel.ctypes.data_as(_I8), _dp(rhs), 1 if is_max else 0,
                                       _dp(T), _ip(basis))
        assert rc == 0
        return T, basis[:m]

    def find_entering(self, T) -> int:
        return self.lib.orc_find_entering(_dp(T), T.shape[0], T.shape[1])

    def find_leaving(self, T, e) -> int:
        return self.lib.orc_find_leaving(_dp(T), T.shape[0], T.shape[1], e)

    def pivot(self, T, r, e) -> None:
        self.lib.orc_pivot(_dp(T), T.shape[0], T.shape[1], r, e)

    def primal_solve(self, T, basis: Optional[np.ndarray] = None, max_pivots: int = 0,
                     log_cap: int = 1 << 16) -> Tuple[int, int, np.ndarray]:
        """In place on T (and basis).  Returns (status, pivots, log[(row, col)])."""
        assert T.flags["C_CONTIGUOUS"] and T.dtype == np.float64
        lr = np.zeros(log_cap, dtype=np.int32)
        lc = np.zeros(log_cap, dtype=np.int32)
        piv = C.c_int64()
        st = self.lib.orc_primal_solve(_dp(T), T.shape[0], T.shape[1], _ip(basis), max_pivots,
                                       _ip(lr), _ip(lc), log_cap, C.byref(piv))
        k = min(piv.value, log_cap)
        return st, piv.value, np.stack([lr[:k], lc[:k]], axis=1)

    def extract_solution(self, T, n):
        x = np.zeros(max(n, 1))
        z = C.c_double()
        self.lib.orc_extract_solution(_dp(T), T.shape[0], T.shape[1], n, _dp(x), C.byref(z))
        return x[:n], z.value


class OracleSens:
    """Handle on an orc_sens (oracle/oracle_sens.c): the re-solve half of SensitivityAnalyzer."""

    def __init__(self, lib, final_tableau, solution, z, basic):
        self.lib = lib
        T = np.ascontiguousarray(final_tableau, dtype=np.float64)
        sol = np.ascontiguousarray(solution, dtype=np.float64)
        b = np.ascontiguousarray(basic, dtype=np.int32)
        self.h = lib.orc_sens_create(_dp(T), T.shape[0], T.shape[1], _dp(sol), sol.shape[0],
                                     float(z), _ip(b), b.shape[0])

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_sens_destroy(self.h)
            self.h = None

    def state(self):
        R, Cc, ns, nb = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        z = C.c_double()
        self.lib.orc_sens_shape(self.h, C.byref(R), C.byref(Cc), C.byref(ns), C.byref(nb),
                                C.byref(z))
        T = np.zeros((R.value, Cc.value))
        basic = np.zeros(max(nb.value, 1), dtype=np.int32)
        sol = np.zeros(max(ns.value, 1))
        self.lib.orc_sens_read(self.h, _dp(T), _ip(basic), _dp(sol))
        return dict(T=T, basic=basic[:nb.value].tolist(), sol=sol[:ns.value], z=z.value)

    def log(self):
        n = self.lib.orc_sens_log_read(self.h, None, 0)
        buf = np.zeros(max(3 * n, 3), dtype=np.int32)
        self.lib.orc_sens_log_read(self.h, _ip(buf), n)
        return [tuple(v) for v in buf[:3 * n].reshape(-1, 3).tolist()]

    def resolve_all(self):
        return self.lib.orc_sens_resolve_all(self.h)

    def change_nonbasic_cbar(self, index, new):
        return self.lib.orc_sens_change_nonbasic_cbar(self.h, index, float(new))

    def change_basic(self, col, delta):
        return self.lib.orc_sens_change_basic(self.h, col, float(delta))

    def change_rhs(self, k, new_b):
        return self.lib.orc_sens_change_rhs(self.h, k, float(new_b))

    def change_nonbasic_column(self, row, col, new):
        return self.lib.orc_sens_change_nonbasic_column(self.h, row, col, float(new))

    def add_activity(self, c_new, a_new):
        a = np.ascontiguousarray(a_new, dtype=np.float64)
        return self.lib.orc_sens_add_activity(self.h, float(c_new), _dp(a))

    def add_constraint(self, tech, rhs):
        t = np.ascontiguousarray(tech, dtype=np.float64)
        return self.lib.orc_sens_add_constraint(self.h, _dp(t), t.shape[0], float(rhs))
