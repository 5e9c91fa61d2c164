// bb_common.hpp -- internal state of the Branch & Bound engine (IntegerProgramming/
// BranchBoundSimplexSolver.cs on the device).  Not part of the ABI.
#pragma once

#include "engine_common.hpp"

namespace lpr {

// k_bb_eliminate keeps one int per column of a child in dynamic LDS (above its 8 KB of static LDS):
// 144 KB of the CU's 160 KB = 36 864 columns.  lpr_bb_create rejects wider trees.
constexpr size_t kBBEliminateLdsMax = (size_t)144 << 10;

// States of one child LP while DoDualSimplex (:289-468) runs on it.
enum : int32_t {
    kBBDual = 0,        // dual phase (:305-343)
    kBBPrimal = 1,      // primal phase (:352-390)
    kBBSolved = 2,      // optimalValue != null
    kBBInfeasible = 3,  // PerformDualPivot gave up -> optimalValue == null (:324-331)
    kBBFailed = 4       // an exception escaped DoDualSimplex (RemoveAt on an empty list, :396-399)
};

// One child sub-problem being evaluated (device memory, one per slot of the current batch).
struct BBSlot {
    double* cur;       // current tableau ("tableaux.Last()"), rows x ld: pivots are applied IN PLACE
    double* nxt;       // backup buffer: the old contents of the rows the last pivot changed, kept
                       // only when that pivot may still be dropped (:395-400)
    int32_t rows, cols;
    int32_t state;
    int32_t pivots;    // tableaux.Count - 1
    int32_t pr, pc;    // pivot chosen by k_bb_select for the pending update
    int32_t do_update; // 1: k_bb_update must apply the pivot (pr, pc) to the rows of the row list
    int32_t backup;    // 1: ... and first save those rows into nxt (the pivot may be dropped)
    int32_t restore;   // 1: the last pivot IS dropped: k_bb_update copies the saved rows back
    int32_t nlist;     // rows in this slot's row list (rows with a non-zero factor + the pivot row)
    int32_t reverse;   // constraint type of the branching row (1: ">=", AddConstraint :774-775)
    int32_t var;       // branching variable
    int32_t crow;      // index of the appended constraint row
    int32_t trace_n;   // entries written to this slot's pivot trace
    int32_t big;       // set by k_bb_round: a rounded entry is >= 1e11 or not finite
    double bound;
    const double* parent;  // parent node's (rounded) tableau, (rows-1) x (cols-1)
    int32_t inplace;   // 1: this child takes its parent's buffer over (cur == parent): k_bb_child_init
                       // leaves it alone, k_bb_child_inplace turns the parent into the child
    int32_t pscan;     // row of bflag / bkey holding this child's PARENT's basic-column scan
    int32_t rep;       // slots[u].rep: a child slot of the u-th distinct parent of the batch
};

}  // namespace lpr

struct lpr_bb {
    lpr_engine* eng = nullptr;
    int rows0 = 0, cols0 = 0;   // root tableau shape
    int nvars = 0;
    int max_depth = 0;
    int rows_cap = 0, ld = 0;   // every node buffer is (rows_cap + 2) x ld doubles: the tableau, then
                                // the two rows k_bb_finish leaves behind it
    size_t buf_elems = 0;
    // node pool
    struct Node {
        double* T = nullptr;
        int rows = 0, cols = 0, depth = 0;
        bool live = false;
        bool big = true;   // rounding the stored tableau again could change it (unknown: yes)
        bool side = false; // rows rows_cap, rows_cap + 1 of the buffer hold its scan / scores (k_bb_finish)
    };
    std::vector<Node> nodes;          // node id -> buffer
    std::vector<double*> free_bufs;   // recycled device buffers
    std::vector<double*> all_bufs;    // everything ever allocated (freed at destroy)
    size_t pool_bytes = 0;            // ... and how much that is
    // batch scratch (grown on demand)
    int slot_cap = 0;
    lpr::BBSlot* d_slots = nullptr;
    lpr::BBSlot* h_slots = nullptr;   // pinned
    double* rowbuf = nullptr;         // slot_cap x ld
    double* colbuf = nullptr;         // slot_cap x rows_cap
    int32_t* bflag = nullptr;         // slot_cap x ld   IdentifyBasicVariables: column is "basic"
    int32_t* bkey = nullptr;          // slot_cap x ld   row of its first 1.0 (or rows)
    int32_t* blist = nullptr;         // slot_cap x ld   sorted basic columns
    int32_t* bcount = nullptr;        // slot_cap
    int32_t* rowlist = nullptr;       // slot_cap x rows_cap: rows the pending pivot changes
    uint8_t* touched = nullptr;       // slot_cap x align_up(rows_cap, 16): 1 = some pivot of this
                                      // child's DoDualSimplex (or AddConstraint's elimination) has
                                      // written the row since k_bb_child_init rounded it
    int32_t* trace = nullptr;         // slot_cap x trace_cap x 3 (phase, row, col)
    int trace_cap = 0;
    double* info = nullptr;           // slot_cap x (nvars + 1): z, decision values
    double* h_info = nullptr;         // pinned
    int32_t* d_running = nullptr;     // number of slots still in kBBDual / kBBPrimal
    int32_t* h_running = nullptr;     // pinned ([1]: last pivot step of the batch that found work)
    int last_steps = 0;               // ... of the previous batch: sizes the next first batch
    // host-side results of lpr_bb_run
    struct Rec {
        int32_t parent, kind, depth, var, status;
        double bound, z;
    };
    std::vector<Rec> records;
    std::vector<int32_t> pop_order;
    std::vector<int32_t> piv_trace;   // quads (record id, phase, row, col)
    std::vector<double> best_x;
    double best_z = 0.0;
    int best_node = -1;
    bool found = false;
    int64_t total_pivots = 0;
    // host-side wall time of the level-synchronous driver by part (seconds; LPR_BB_TIMING=1 prints
    // them to stderr when lpr_bb_solve_level_sync returns)
    struct Prof {
        double alloc = 0, slots = 0, info = 0, expand = 0, poll = 0, comm = 0;
        int mallocs = 0, polls = 0, steps = 0;
    } prof;
};
