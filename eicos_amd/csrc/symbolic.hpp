// Host-side symbolic analysis for the batched KKT LDL' (one pattern, many numeric instances).
//
// Replaces, for the product path, what the reference obtains from
// Eigen::SimplicialLDLT::analyzePattern (reference src/eicos.cpp:897) and from
// setupKKT/cacheIndices (src/eicos.cpp:1734-1988): the KKT pattern, a fill-reducing
// ordering, the elimination tree, the pattern of L -- plus what a GPU needs on top:
// a level schedule (nodes renumbered so each etree level is a contiguous index range),
// CSR and CSC views of L, and an explicit update program for the numeric factorisation.
#pragma once
#include <cstdint>
#include <vector>

namespace eicos {

struct ProblemPattern {
    int n = 0, m = 0, p = 0, l = 0, nc = 0;
    std::vector<int> q;
    std::vector<int> Gjc, Gir, Ajc, Air; // CSC patterns (G: m x n, A: p x n)
    int nnzG() const { return (int)Gir.size(); }
    int nnzA() const { return (int)Air.size(); }
};

// Where one KKT entry's numeric value comes from, per instance.
enum SrcKind : int { SRC_ZERO = 0, SRC_A = 1, SRC_G = 2, SRC_V = 3, SRC_POSDELTA = 4, SRC_NEGDELTA = 5 };

struct Symbolic {
    // dimensions (reference src/eicos.cpp:152-165)
    int n = 0, p = 0, m = 0, l = 0, nc = 0, N = 0, mt = 0, nV = 0, nnzA = 0, nnzG = 0, nnzK = 0;
    std::vector<int> q, cone_off;   // cone_off[c] = first row of cone c in z/s (un-expanded)
    // transposed (row-major) views of A and G: ptr over rows, column index, position in CSC values
    std::vector<int> At_ptr, At_col, At_pos, Gt_ptr, Gt_col, Gt_pos;
    // KKT upper triangle in the reference's column layout (Appendix B of SURVEY.md)
    std::vector<int> K_row, K_col, K_kind, K_src; // per entry: coordinates, SrcKind, index into that source
    // ordering: perm[new] = old KKT index ; iperm[old] = new
    std::vector<int> perm, iperm;
    // L pattern (strictly lower), nodes in level order
    int nnzL = 0, nlev = 0;
    std::vector<int> lev_ptr;           // nlev+1: node range of each level
    std::vector<int> Lp, Li;            // CSC: column j holds rows i>j   (used by backward solve)
    std::vector<int> Rp, Rj, Rpos;      // CSR: row i holds cols k<i ; Rpos = position in CSC arrays
    std::vector<int> Cpos;              // CSC entry -> position in CSR arrays
    std::vector<int> parent;
    // numeric sources of the permuted lower triangle: per CSC entry / per diagonal: (kind, src) or SRC_ZERO (fill)
    std::vector<int> Lkind, Lsrc, Dkind, Dsrc;
    // factor program: target t in [0,N) = diagonal t ; t in [N, N+nnzL) = CSC entry t-N.
    // pairs of target t: [tp[t], tp[t+1]) ; value -= U[pa]*U[pb]*invD[pk]   (positions in CSC order)
    std::vector<int64_t> tp;
    std::vector<int> pa, pb, pk;
    int64_t npairs = 0;
    // per-level task lists (targets sorted by decreasing pair count)
    std::vector<int> ftask_ptr, ftask;  // nlev+1 ; target ids
    double flops_factor = 0;            // 2*npairs + divisions
    int max_row_len = 0, max_col_len = 0;
    int order_mode = 0;                 // slack+1 actually used
    int cone_order = 0;                 // 1: ordered with the constraint "cone rows -> v -> u" for every second-order cone (analyze_mode)
    // ---- tile mode (patterns with dense fronts; tiles.hpp): the elimination order is an etree postorder cut into
    // blocks of <= 16 consecutive nodes aligned with the supernodes, blocks renumbered by block level.  L is then a
    // block-sparse matrix of dense 16 x 16 tiles; the scalar factor program (tp/pa/pb/pk, ftask) is not built.
    // tile = 2: HYBRID.  The scalar (level-ordered) elimination order is kept; its narrow top -- the levels lev_cut.. of the
    // schedule, nodes n0..N-1, typically one dense supernode whose scalar schedule is a chain of single-node levels -- is cut
    // into plain 16-node blocks and handled by the tile path (one block level per block), everything below stays on the
    // scalar sliced-ELL programs.  The scalar factor program then covers: every target in a column < n0, and, as one extra
    // level ftask_ptr[lev_cut..lev_cut+1], every target of the top block with its pairs from columns < n0 only (the Schur
    // complement the tile factorisation starts from).
    int tile = 0;
    int lev_cut = 0, n0 = 0;
    int nblk = 0, nblev = 0;
    std::vector<int> blk_ptr;           // nblk+1: node range of each block
    // ---- dense apex (scalar path, tile == 0).  The last levels of the schedule -- the top of the elimination tree: levels apex_lev..nlev-1,
    // nodes apex0..N-1, at most APEX_MAX of them -- are narrow (1..16 rows) and each costs both sweeps a dependent slice step on one
    // wavefront.  They are swept instead as ONE dense triangular system by one wavefront, lane = node, the iterate in a register, a column
    // (forward) / row (backward) of the block per step: na dependent multiply-subtract steps instead of (nlev - apex_lev) slice steps.
    // The factor program is unchanged; the entries of L inside the block get slots in a dense (folded, APEX_IMG doubles) image behind each sweep plan's
    // value array (plans.hpp: TriPlan::apex_base), everything else of the rows of the block (their entries in columns < apex0) becomes one
    // workgroup-wide level of the forward plan (as in the hybrid).  apex0 < 0: none.
    int apex0 = -1, apex_lev = 0;
    std::vector<int> blev_ptr;          // nblev+1: block range of each block level
};

// order_mode: 0 = minimum degree, ties by index (sequential; deep trees)
//             k>=1 = multiple independent elimination of nodes within k-1 of the minimum degree
//             <0 = try several k, keep the best under a fill + tree-height cost model (default)
// tile: 0 = scalar (sliced-ELL) path, 1 = tile path, 2 = scalar with the top of the tree on tiles (hybrid) when it pays,
// < 0 = choose: tiles when L is dense (nnz(L) >= 16 dim_K), otherwise hybrid when the scalar schedule ends in a long chain
Symbolic analyze(const ProblemPattern &P, int order_mode = -1, int tile = -1);
constexpr int APEX_MAX = 64;        // nodes of the dense apex: one per lane of a wavefront
constexpr int APEX_MIN_LEVELS = 5;  // worth it from this many levels on (it costs about three slice steps per solve itself)
constexpr int APEX_MIN_N = 96;      // (128-thread workgroups carry an apex only in the LDS-resident build: api.cpp decides)
constexpr int APEX_OVER_HYBRID_BELOW = 512; // below this size the apex replaces the hybrid's tile block when both apply (api.cpp; measured on lp_adlittle, lp_blend: +17 %)

} // namespace eicos
