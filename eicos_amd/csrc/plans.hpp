// Sliced-ELL plans of the level-scheduled triangular sweeps (see SliceMeta in device_types.hpp).
#pragma once
#include <vector>

#include "device_types.hpp"
#include "symbolic.hpp"

namespace eicos {

struct TriPlan {
    std::vector<SliceMeta> sl; // slices in sweep order (forward: levels up, backward: levels down)
    // The narrow top of the elimination tree (levels that fit one wavefront in <= 2 slices) is laid out for 64
    // lanes and run by a single wavefront without workgroup barriers: forward sl = [wide | solo], backward
    // sl = [solo | wide]; each part padded to a multiple of TRI_DEPTH slices.
    int n_wide = 0, n_solo = 0;
    // hybrid (Symbolic::tile == 2), forward plan only: one more level after the solo part, laid out for the whole workgroup --
    // the rows of the top block with their entries in columns < n0 (the block's own columns belong to the tile sweep)
    int n_ext = 0;
    std::vector<int> idx;      // per slot: index of the gathered solve-vector entry (padding -> N)
    std::vector<int> pos;      // per CSC entry of L: its slot
    int slots = 0;
    // dense apex (Symbolic::apex0 >= 0): the sweeps stop below the apex like the hybrid's (forward: its rows against the columns below as the
    // `ext` level); the entries of L INSIDE the apex have their slots in a dense image behind the dummy slot, from apex_base on (a multiple
    // of 64): the FOLDED image of device_types.hpp, apex_img_at(i - apex0, k - apex0), APEX_IMG doubles, in both plans.  Forward = the
    // unit-lower L the apex sweeps read -- the order the LDS copy has, so that copy is a straight one.  Backward = U = L.*D, the operands of
    // the factor program (read by the sweeps only when no LDS vector exists).  Image positions no entry names stay zero.
    int apex_base = 0;
    int ulen = 1;              // length of the plan's value array: slots + 1 (the dummy), or apex_base + the image
    // dense apex, forward plan: ONE row of the apex that is much longer than the others (the root of an MPC tree: 935 entries in the columns
    // below the block against 18..26) is cut into split_n parts that sit side by side in ONE slice of the `ext` level -- part p is the
    // pseudo-row split_slot0 + p of the sweep vector (spare slots behind the zero slot N: zero when the sweep starts, so a part leaves
    // -(its partial sum) there) -- instead of split_n sub-slices of a 64-lane slice that run one after the other; apex_solve adds the
    // parts to the row and zeroes them again.  split_row < 0: none.
    int split_row = -1, split_slot0 = 0, split_n = 0;
};
// length of the KKT-space vectors of the scalar path on the device (>= N + 1: slot N is the always-zero target of ELL padding)
inline int scalar_npad(int N) { return (N + 1 + 15) & ~15; }

// T = workgroup size the plan is laid out for.
TriPlan build_tri_plan(const Symbolic &S, int T, bool forward, bool allow_solo = true);

// Sliced-ELL plan of a plain row-wise sparse product (no levels): rows [0,nrows) of a CSR-like
// pattern `ptr`, consecutive rows per slice.  src[slot] = CSR entry stored in that slot, -1 = padding.
struct EllPlan {
    std::vector<SliceMeta> sl;
    std::vector<int> src;
    int slots = 0;
};
EllPlan build_ell_plan(const std::vector<int> &ptr, int nrows, int T);

// Sliced-ELL form of the numeric factorisation program.  Per elimination-tree level the targets
// (diagonal j or strictly-lower entry e of L, Symbolic::ftask order = decreasing pair count) are cut
// into slices; lane t of a slice owns pairs q, q+g, ... of target row0 + t/g.  Padding pairs read the
// dummy value slot (value 0).
struct FactorPlan {
    std::vector<SliceMeta> sl;       // row0 = index of the slice's first target in the per-target arrays
    std::vector<int> pa, pb;         // per slot: U[i,k] in the backward (column) slot order, L[j,k] in the forward (row) slot order
    std::vector<int> pbU, pk;        // the same pair as U[j,k] (backward slot) and its pivot column k: term = U[i,k] * (U[j,k] / D[k])
    std::vector<int> target;         // per target: Symbolic target id (j < N: diagonal, N + e: entry e)
    int slots = 0;
};
FactorPlan build_factor_plan(const Symbolic &S, int T, const std::vector<int> &posB, int dummyB,
                             const std::vector<int> &posF, int dummyF);

} // namespace eicos
