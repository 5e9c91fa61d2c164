"""ctypes binding of the C ABI declared in include/lpr_engine.h.

This is plumbing only: every solver entry point of the package goes through liblpr_engine.so
(hand-written HIP for gfx950).  There is no CPU fallback -- if the library is missing the import
fails loudly, and if no gfx950 device is present ``lpr_engine_open`` returns LPR_DEVICE_ERROR.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "liblpr_engine.so")

# lpr_status (include/lpr_engine.h)
LPR_OK_OPTIMAL = 0
LPR_UNBOUNDED = 1
LPR_INFEASIBLE_BASIS = 2
LPR_PIVOT_TOO_SMALL = 3
LPR_ENTERING_ALREADY_BASIC = 4
LPR_PIVOT_LIMIT = 5
LPR_BB_NODE_CAP = 6
LPR_BB_DEPTH_CAP = 7
LPR_BAD_ARGUMENT = -1
LPR_DEVICE_ERROR = -2
LPR_OUT_OF_MEMORY = -3

STATUS_NAMES = {
    0: "OPTIMAL", 1: "UNBOUNDED", 2: "INFEASIBLE_BASIS", 3: "PIVOT_TOO_SMALL",
    4: "ENTERING_ALREADY_BASIC", 5: "PIVOT_LIMIT", 6: "BB_NODE_CAP", 7: "BB_DEPTH_CAP",
    -1: "BAD_ARGUMENT", -2: "DEVICE_ERROR", -3: "OUT_OF_MEMORY",
}

LPR_REL_LE, LPR_REL_GE, LPR_REL_EQ = 0, 1, 2

# lpr_sens_outcome
LPR_SENS_OK = 0
LPR_SENS_UNBOUNDED = 1
LPR_SENS_INFEASIBLE = 2
LPR_SENS_ZERO_PIVOT = 3
LPR_SENS_ITER_LIMIT = 5
LPR_SENS_ROLLED_BACK = 8
LPR_SENS_INDEX_OUT_OF_RANGE = 9
LPR_SENS_INVALID_INDEX = -1


class SolveOpts(C.Structure):
    _fields_ = [
        ("max_pivots", C.c_int64),
        ("time_kernels", C.c_int32),
        ("batch", C.c_int32),
        ("variant", C.c_int32),
        ("block", C.c_int32),
    ]


class SolveResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("block", C.c_int32),
        ("pivots", C.c_int64),
        ("total_pivots", C.c_int64),
        ("z", C.c_double),
    ]


class RevisedResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("reserved", C.c_int32),
        ("iterations", C.c_int64),
        ("total_iterations", C.c_int64),
        ("z", C.c_double),
    ]


class RevisedSnapshotInfo(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("entering", C.c_int32),
        ("leaving_row", C.c_int32),
        ("leaving_var", C.c_int32),
        ("entering_rc_pre", C.c_double),
        ("z_working", C.c_double),
        ("z_original", C.c_double),
    ]


class BBOpts(C.Structure):
    _fields_ = [
        ("enable_pruning", C.c_int32),
        ("node_cap", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


class BBSyncOpts(C.Structure):
    _fields_ = [
        ("enable_pruning", C.c_int32),
        ("max_levels", C.c_int32),
        ("max_nodes", C.c_int64),
    ]


class BBSyncResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("found", C.c_int32),
        ("processed", C.c_int64),
        ("pivots", C.c_int64),
        ("levels", C.c_int32),
        ("path_len", C.c_int32),
        ("path_bits", C.c_uint64),
        ("z", C.c_double),
    ]


ALLREDUCE_MAX_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)
LPR_COMM_ID_BYTES = 128


class BBResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("found", C.c_int32),
        ("processed", C.c_int64),
        ("best_node", C.c_int32),
        ("reserved", C.c_int32),
        ("z", C.c_double),
        ("pivots", C.c_int64),
        ("nodes_created", C.c_int64),
    ]


_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_D = C.POINTER(C.c_double)
_I32 = C.POINTER(C.c_int32)
_I8 = C.POINTER(C.c_int8)
_I64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes); the symbol list is also what tests/test_abi.py checks against the
# declarations in include/lpr_engine.h.
SIGNATURES = {
    "lpr_abi_version": (C.c_int, []),
    "lpr_last_error": (C.c_char_p, []),
    "lpr_engine_open": (C.c_int, [C.c_int, _PP]),
    "lpr_engine_close": (C.c_int, [_P]),
    "lpr_engine_sync": (C.c_int, [_P]),
    "lpr_engine_stream": (C.c_uint64, [_P]),
    "lpr_tableau_from_lp": (C.c_int, [_P, C.c_int, C.c_int, _D, _D, C.c_int, _I32, _I8, _D,
                                      C.c_int, _PP]),
    "lpr_tableau_create": (C.c_int, [_P, C.c_int, C.c_int, _D, _I32, _PP]),
    "lpr_tableau_synthetic": (C.c_int, [_P, C.c_int, C.c_int, C.c_uint64, _PP]),
    "lpr_tableau_destroy": (C.c_int, [_P]),
    "lpr_tableau_shape": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int)]),
    "lpr_primal_solve": (C.c_int, [_P, C.POINTER(SolveOpts), C.POINTER(SolveResult)]),
    "lpr_select_entering": (C.c_int, [_P, _I32]),
    "lpr_select_leaving": (C.c_int, [_P, C.c_int32, _I32]),
    "lpr_pivot": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "lpr_extract_solution": (C.c_int, [_P, C.c_int, _D, _D]),
    "lpr_tableau_read": (C.c_int, [_P, _D]),
    "lpr_tableau_read_block": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _D]),
    "lpr_basis_read": (C.c_int, [_P, _I32]),
    "lpr_pivot_log_read": (C.c_int, [_P, _I32, _I32, C.c_int64, _I64]),
    "lpr_tableau_kernel_stats": (C.c_int, [_P, _I64, _D, _D]),
    "lpr_tableau_step_stats": (C.c_int, [_P, _I64, _D]),
    "lpr_debug_head_stamps": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int64, _I64]),
    "lpr_dual_solve": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, C.POINTER(SolveResult)]),
    "lpr_primal2_solve": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, C.POINTER(SolveResult)]),
    "lpr_cutting_plane": (C.c_int, [_P, C.c_int, C.c_int64, _I32, _I32]),
    "lpr_cut_log_read": (C.c_int, [_P, _I32, C.c_int64, _I64]),
    "lpr_revised_create": (C.c_int, [_P, C.c_int, C.c_int, _D, _D, C.c_int, _D, C.c_int, _PP]),
    "lpr_revised_synthetic": (C.c_int, [_P, C.c_int, C.c_int, C.c_uint64, _PP]),
    "lpr_revised_destroy": (C.c_int, [_P]),
    "lpr_revised_solve": (C.c_int, [_P, C.POINTER(SolveOpts), C.POINTER(RevisedResult)]),
    "lpr_revised_solution": (C.c_int, [_P, _D, _D]),
    "lpr_revised_basis_read": (C.c_int, [_P, _I32]),
    "lpr_revised_log_read": (C.c_int, [_P, _I32, _I32, _I32, C.c_int64, _I64]),
    "lpr_revised_binv_read": (C.c_int, [_P, _D]),
    "lpr_revised_xb_read": (C.c_int, [_P, _D]),
    "lpr_revised_binv_a": (C.c_int, [_P, _D, _D]),
    "lpr_revised_step": (C.c_int, [_P, C.POINTER(RevisedSnapshotInfo)]),
    "lpr_revised_snapshot_read": (C.c_int, [_P, _D, _D, _D, _D, _I32, _D]),
    "lpr_revised_binv_a_exact": (C.c_int, [_P, _D]),
    "lpr_bb_create": (C.c_int, [_P, _D, C.c_int, C.c_int, C.c_int, C.c_int, _PP]),
    "lpr_bb_create_from_tableau": (C.c_int, [_P, C.c_int, C.c_int, _PP]),
    "lpr_bb_destroy": (C.c_int, [_P]),
    "lpr_bb_run": (C.c_int, [_P, C.POINTER(BBOpts), _D, C.POINTER(BBResult)]),
    "lpr_bb_records_read": (C.c_int, [_P, _I32, _I32, _I32, _I32, _D, _I32, _D, C.c_int64, _I64]),
    "lpr_bb_pop_order_read": (C.c_int, [_P, _I32, C.c_int64, _I64]),
    "lpr_bb_trace_read": (C.c_int, [_P, _I32, C.c_int64, _I64]),
    "lpr_bb_node_info": (C.c_int, [_P, _I32, C.c_int, _D, _D]),
    "lpr_bb_expand": (C.c_int, [_P, C.c_int, _I32, _I32, _D, _I32, _I32, _I32, _I32]),
    "lpr_bb_expand_traced": (C.c_int, [_P, C.c_int, _I32, _I32, _D, _I32, _I32, _I32, _I32, _I32,
                                       C.c_int64, C.POINTER(C.c_int64), _D, C.c_int64,
                                       C.POINTER(C.c_int64), _I32]),
    "lpr_bb_release": (C.c_int, [_P, _I32, C.c_int]),
    "lpr_bb_node_read": (C.c_int, [_P, C.c_int32, _D, _I32, _I32]),
    "lpr_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "lpr_comm_init": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_uint8), _PP]),
    "lpr_comm_init_custom": (C.c_int, [C.c_int, C.c_int, ALLREDUCE_MAX_FN, ALLGATHER_FN,
                                       C.c_void_p, _PP]),
    "lpr_comm_destroy": (C.c_int, [_P]),
    "lpr_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), _I64, _I64]),
    "lpr_comm_all_reduce_max": (C.c_int, [_P, _D, C.c_int]),
    "lpr_bb_solve_level_sync": (C.c_int, [_P, _P, C.POINTER(BBSyncOpts), _D,
                                          C.POINTER(BBSyncResult)]),
    "lpr_sens_create": (C.c_int, [_P, _D, C.c_int32, C.c_int32, _D, C.c_int32, C.c_double, _PP]),
    "lpr_sens_create_from_tableau": (C.c_int, [_P, C.c_int32, _PP]),
    "lpr_sens_destroy": (C.c_int, [_P]),
    "lpr_sens_shape": (C.c_int, [_P, _I32, _I32, _I32, _I32, _D, _I64]),
    "lpr_sens_read": (C.c_int, [_P, _D, _I32, _D]),
    "lpr_sens_read_block": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _D]),
    "lpr_sens_basic_row": (C.c_int, [_P, C.c_int32, _I32]),
    "lpr_sens_log_read": (C.c_int, [_P, _I32, C.c_int64, _I64]),
    "lpr_sens_column_fold": (C.c_int, [_P, _D, C.c_int32, _D, C.c_int32, _D]),
    "lpr_sens_resolve_all": (C.c_int, [_P, _I32]),
    "lpr_sens_change_nonbasic_cbar": (C.c_int, [_P, C.c_int32, C.c_double, _I32]),
    "lpr_sens_change_basic": (C.c_int, [_P, C.c_int32, C.c_double, _I32]),
    "lpr_sens_change_rhs": (C.c_int, [_P, C.c_int32, C.c_double, _I32]),
    "lpr_sens_change_nonbasic_column": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_double, _I32]),
    "lpr_sens_add_activity": (C.c_int, [_P, C.c_double, _D, C.c_int32, _I32]),
    "lpr_sens_add_constraint": (C.c_int, [_P, _D, C.c_int32, C.c_double, _I32]),
}


def load_library(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP engine has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            "lpr_381_group_v22_amd/csrc`). This package has no CPU fallback."
        )
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load_library()


class EngineError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        msg = lib.lpr_last_error()
        super().__init__(
            f"{where}: {STATUS_NAMES.get(status, status)}"
            + (f" -- {msg.decode(errors='replace')}" if msg else "")
        )


def check(status: int, where: str) -> int:
    """Raise on ABI errors (< 0); solver outcomes (>= 0) are returned to the caller."""
    if status < 0:
        raise EngineError(status, where)
    return status
