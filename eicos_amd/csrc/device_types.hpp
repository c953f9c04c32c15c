// Shared host/device plain-data types of the batched solver.
#pragma once
#include <cstddef>

namespace eicos {

// On the device every slab / pattern pointer is tagged with the global address space: pointers
// that reach a kernel through a struct in memory (or through a non-inlined function) are
// otherwise "generic", hipcc emits flat_load/flat_store for them, and flat accesses count on
// BOTH vmcnt and lgkmcnt -- which couples HBM latency into every LDS wait.
#if defined(__HIP_DEVICE_COMPILE__)
#define EICOS_GLOBAL __attribute__((address_space(1)))
#else
#define EICOS_GLOBAL
#endif
typedef const int EICOS_GLOBAL *gint_p;
typedef double EICOS_GLOBAL *gdbl_p;
typedef const double EICOS_GLOBAL *gcdbl_p;

// Per-instance persistent scalar state: struct Work's scalars + struct Information
// (reference include/eicos.hpp:49-73,97-114) with std::optional flattened.
struct DevInfo {
    double pcost, dcost, pres, dres, gap, relgap, sigma, mu, step, step_aff, kapovert;
    double pinfres, dinfres, tau, kap, cx, by, hz;
    int has_relgap, has_pinfres, has_dinfres, pinf, dinf;
    int iter, nitref1, nitref2, nitref3, exitcode, n_factor, n_ldlsolve, equilibrated;
    int n_sweep;     // passes over the factor L in the last solve: = n_ldlsolve, except that a dual right-hand-side solve counts once for both
    double solve_us; // wall time this instance's workgroup spent on the last solve (100 MHz counter), for the launch's tail statistics
};
static_assert(sizeof(DevInfo) == 208, "DevInfo layout");
constexpr int DEVINFO_DOUBLES = 32;

// One slice of a sliced-ELL triangular-solve plan: `cnt` consecutive rows (or columns) of L
// starting at node row0, each handled by g = 1<<lg adjacent lanes; lane t of the workgroup owns
// entries q, q+g, ... of row row0 + t/g (q = t%g); entry kk of lane t sits at slot
// off + kk*(cnt*g) + t, so every load is unit-stride across the workgroup and needs no row
// pointer.  K = entries per lane (padding slots hold value 0 / index N).  newlev = 1 on the first
// slice of an elimination-tree level: a workgroup barrier separates it from the previous level.
// One slice of a sliced-ELL program: rows/targets [row0, row0+cnt), 2^lg lanes per row, K entries per lane at
// slot off + k*lanes + lane.  Rows longer than ELL_KMAX << lg are cut into consecutive sub-slices of K <= ELL_KMAX
// over the same rows: `more` = another sub-slice of these rows follows, `cont` = this one continues the previous.
struct SliceMeta { int row0, cnt, lg, K, off, newlev, more, cont; }; // host form (plan building, host emulation)
constexpr int ELL_KMAX = 4;  // entries per lane that are software-prefetched
// Destination codes of the factor program (DevPat::fac_dst), checked against each other in eicos_batch_create:
constexpr int DIAG_POS = 1 << 30; // diagonal targets: dst = -(j + 1) - (quasi-definite sign of pivot j is + ? DIAG_POS : 0)
constexpr int IMG_BASE = 1 << 28; // hybrid: dst >= IMG_BASE = entry (dst - IMG_BASE) of the top block's tile image; UB slots stay below IMG_BASE

// Device form of a slice: 16 bytes = one ds_read_b128 / s_load_dwordx4.  off16 = index of the slice's first lane
// in the plan's packed 16-bit gather-index array (one 8-byte entry = ELL_KMAX indices per lane), see api.cpp.
// bits: cnt [0,11) | lg [11,14) | K [14,17) | newlev 17 | last slice of its level 18 (factor plan) | more 19 | cont 20
struct PackedSlice { int row0, off, off16, bits; };
constexpr int PS_LG = 11, PS_K = 14, PS_NEWLEV = 17, PS_LAST = 18, PS_MORE = 19, PS_CONT = 20;
inline PackedSlice pack_slice(const SliceMeta &m, int off16) {
    return PackedSlice{m.row0, m.off, off16,
                       m.cnt | (m.lg << PS_LG) | (m.K << PS_K) | ((m.newlev & 1) << PS_NEWLEV) | (((m.newlev >> 1) & 1) << PS_LAST) |
                           ((m.more & 1) << PS_MORE) | ((m.cont & 1) << PS_CONT)};
}
static_assert(ELL_KMAX == 4, "the packed index entries hold four 16-bit indices per lane");
#ifndef EICOS_TRI_DEPTH
#define EICOS_TRI_DEPTH 3
#endif
#ifndef EICOS_ELL_DEPTH
#define EICOS_ELL_DEPTH 2
#endif
#ifndef EICOS_FAC_DEPTH
#define EICOS_FAC_DEPTH 2
#endif
#ifndef EICOS_TRI_DEPTH_SOLO
#define EICOS_TRI_DEPTH_SOLO 3
#endif
constexpr int TRI_DEPTH_SOLO = EICOS_TRI_DEPTH_SOLO; // queue depth (= slices per trip) of the single-wavefront part of the sweeps
constexpr int TRI_DEPTH = EICOS_TRI_DEPTH; // ... this many slices ahead of their use (plans are padded to a multiple)
#ifndef EICOS_TRI_TRIP
#define EICOS_TRI_TRIP 6
#endif
constexpr int TRI_TRIP = EICOS_TRI_TRIP;   // slices per trip of the (unrolled) sweep loops, see ELL_TRIP
static_assert(TRI_TRIP % TRI_DEPTH == 0, "the register queue rotates inside a trip");
constexpr int ELL_DEPTH = EICOS_ELL_DEPTH; // same for the matrix-vector products
// Slices per trip of the (unrolled) product loops (the remainder of a plan runs in trips of ELL_DEPTH).  The compiler's s_waitcnt insertion is
// exact inside a trip but drains the whole load queue at the loop head (s_waitcnt vmcnt(0)), so a trip of ELL_DEPTH slices
// exposes a full memory round trip every ELL_DEPTH slices.
#ifndef EICOS_ELL_TRIP
#define EICOS_ELL_TRIP 6
#endif
constexpr int ELL_TRIP = EICOS_ELL_TRIP;
static_assert(ELL_TRIP % ELL_DEPTH == 0, "the register queue rotates inside a trip");
constexpr int FAC_DEPTH = EICOS_FAC_DEPTH; // and for the static part of the factor program (no padding needed)

// Everything the kernels need to know about the (shared) pattern.  All pointers are device
// pointers into one int32 pattern buffer; all i_* / w_* members are offsets in doubles into
// the per-instance slab / the per-resident-workgroup workspace slab.
// Dense apex: where entry (i, k), k < i < 64, of the block's strictly lower triangle sits in its LDS image: 32 rows of 65 doubles, row i >= 32 in
// image row 63 - i at columns 0 .. i - 1, row i < 32 in image row i from column 63 down (kernels.hip: apex_solve_lds)
constexpr int APEX_IMG = 32 * 65;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int apex_img_at(int i, int k) { return i >= 32 ? (63 - i) * 65 + k : i * 65 + 63 - k; }

struct DevPat {
    int n, p, m, l, nc, N, mt, nV, nnzA, nnzG, nnzL, nlev;
    int Npad; // N rounded up to 16 doubles: stride of the LDS-resident KKT-space vectors
    // A, G in CSC (column) form and row pointers + CSC positions of the transposed form (updateData)
    gint_p Ajc, Air, At_ptr, At_pos;
    gint_p Gjc, Gir, Gt_ptr, Gt_pos;
    gint_p Acol, Gcol; // column of every CSC entry (entry-parallel updateData)
    // sliced-ELL plans of the matrix-vector products (plans.hpp: EllPlan): stacked columns of [A;G]
    // (cag: x-space results), rows of A (rA), rows of G (rG).  *_src: slot -> offset of the CSC value
    // relative to Av (-1 = padding); cag has two gather-index sets: KKT indices (refinement) and
    // offsets into the contiguous (y, z) block of the instance slab (residuals).
    const PackedSlice EICOS_GLOBAL *cag_sl; const PackedSlice EICOS_GLOBAL *rA_sl; const PackedSlice EICOS_GLOBAL *rG_sl;
    int cag_ns, rA_ns, rG_ns, cag_slots, rA_slots, rG_slots;
    gint_p cag_idx_k, cag_idx_yz, cag_src, rA_idx, rA_idx_k, rA_src, rG_idx, rG_idx_k, rG_src;
    // KKT-space vectors live in the (level-ordered) elimination order on the device: position of
    // variable j / equality row r / cone-block row i / the v- and u-expansion slot of cone c
    gint_p ipx, ipy, ipz, ipv, ipu; gint_p ipk; // [ipx | ipy | ipz] in one array
    gint_p zpos; int nzpos; // the slots of the sweep vector [0, Npad) that are not an x / y / z position (cone expansions, padding): kept zero
    // cones
    gint_p cq, cone_off, cone_vbase, cone_small, cone_big;
    int n_small, n_big;
    // "tiny" cones (dimension <= TINY_D): the hot per-cone loops keep a tiny cone's rows in registers, one thread per cone, all loads of
    // a thread's cones issued up front (kernels.hip: for_tiny).  cone_tiny = TINY_INTS ints per tiny cone: first row o, dimension d, cone
    // id c, elimination slot of its v and u expansion rows, elimination slots of its rows, first slot of its scaling block (cone_vbase);
    // cone_mid = the other cones below CONE_BIG
    gint_p cone_tiny, cone_mid;
    int n_tiny, n_mid;
    // "wave" cones (CONE_BIG <= dimension <= 64): one wavefront per cone with LANE-PER-ROW registers in the hot loops (kernels.hip: for_wave);
    // cone_huge = the cones above 64 rows (generic wavefront loops)
    gint_p cone_wave, cone_huge;
    int n_wave, n_huge;
    gint_p zdsign;  // [m] +1 / -1: sign of the static-regularisation term in refinement
    // LDL' pattern, level ordered
    // triangular solves: sliced-ELL plans (see SliceMeta).  UF = unit-lower L in the forward (row) slot order,
    // UB = U = L.*D (column-scaled) in the backward (column) slot order (the CSC-entry -> slot maps stay on the host)
    const PackedSlice EICOS_GLOBAL *fsl; const PackedSlice EICOS_GLOBAL *bsl;
    int nfs, nbs, nUF, nUB;   // nfs / nbs = slices of the workgroup-wide part of the forward / backward plan
    int nfs_solo, nbs_solo;   // single-wavefront part (top of the tree): fsl = [wide | solo], bsl = [solo | wide]
    // the REAL slices of every section (the counts above include the empty slices the host pads a section with: never stepped through)
    int nfs_r, nbs_r, nfs_solo_r, nbs_solo_r, nfs_ext_r, cag_ns_r, rA_ns_r, rG_ns_r;
    int meta_lds; // 1: the slice tables are staged in LDS behind the NLDS vectors, at these slice offsets:
    int lm_f, lm_b, lm_cag, lm_rA, lm_rG, lm_fac, lm_total; // lm_fac < 0: factor table stays in global memory
    gint_p f_idx, b_idx;
    // 16-bit gather indices (idx16 = 1: every index of every plan fits 16 bits): per plan one 8-byte entry per lane and
    // slice holding its ELL_KMAX indices, PackedSlice::off16 + lane; *_d16 = the all-padding entry inactive lanes read
    int idx16, f_d16, b_d16, cag_d16, rA_d16, rG_d16;
    gint_p f_idx16, b_idx16, cag_k16, cag_yz16, rA_16, rA_k16, rG_16, rG_k16;
    // numeric factorisation: sliced-ELL program (plans.hpp: FactorPlan); per target: source offset of its
    // K value in the instance slab, destination (>= 0: UB slot, < 0: -(diagonal index)-1) and UF slot
    const PackedSlice EICOS_GLOBAL *fac_sl;
    int fac_ns, fac_slots, fac_nt; // slices, pair slots, targets
    int fac_s1, fac_nd0, fac_nt0;  // level 0 (no pairs): its slices are [0, fac_s1), its targets [0, fac_nt0), the first fac_nd0 of them diagonals
    int w_Kt;                      // [fac_nt] KKT entry of every target, in target order (workspace slab)
    gint_p v2t;                    // [nV] scaling-block entry -> its target
    gint_p fac_pa, fac_pb, fac_src, fac_dst, fac_dstF, fac_col;
    gint_p fac_p16; int fac_d16; // idx16: per lane and slice the four (pa, pb) pairs as eight 16-bit slot numbers (16 bytes)
    // fac_defer = 1 (needs an LDS vector: NLDS >= 1): pb names U[j,k] (a UB slot) and fac_pk / fac_k16 the pivot column k; the
    // factorisation keeps a mirror of 1/D in the (then idle) LDS solve vector and forms L[j,k] = U[j,k] * (1/D[k]) on the fly --
    // bit-identical to the stored L -- so no level needs a second phase; L goes to its forward slots in one pass at the end
    // NLDS = 1 without dual right-hand sides: elimination positions [0, e_lds) of the refinement residual E are kept in LDS at g_dyn + e_off
    // (doubles) -- the part of the CU's LDS that the chosen number of resident workgroups leaves unused
    int e_lds, e_off;
    // U-in-LDS builds (kernels_ubl*.hip; one workgroup per CU): the factor operand array U = L.*D (w_UB, ub_len doubles incl. padding and dummy slots)
    // lives at g_dyn + ub_lds (doubles) instead of in the workspace slab; -1 = not this build
    int ub_lds, ub_len;
    int fac_defer, fac_kpad; gint_p fac_pk, fac_k16; // fac_kpad: the pivot-column index of padding pairs (its mirror slot holds 0)
    // ---- tile mode (dense fronts, tiles.hpp): L = block-sparse matrix of dense 16 x 16 tiles; D.N is then 16 * nb ----
    int tile, nb, nt, nblev;       // 1 = tile path, 2 = hybrid (top block of the tree on tiles); blocks, off-diagonal tiles, block levels
    int tl_base;                   // slot of block 0 in the KKT-space vectors (hybrid: the scalar part comes first)
    int w_Kimg;                    // workspace: dense tile image of K the tile factorisation starts from (= w_Kt in pure tile mode)
    int nfs_ext;                   // hybrid / dense apex: slices of the forward plan's extra level (rows of the top block, columns below it)
    // dense apex (symbolic.hpp): nodes apex_n0 .. apex_n0 + apex_na - 1 (apex_na = 0: none; <= 64), swept by wavefront 0 alone from the
    // folded images (apex_img_at below) UF + apex_f (the block's unit-lower L) and UB + apex_b (U = L.*D)
    int apex_na, apex_n0, apex_f, apex_b;
    // apex_lds >= 0 (NLDS >= 1 and room in LDS): offset (doubles, in the dynamic LDS) of the FOLDED strictly-lower image of the block's unit-lower L
    // (apex_img_at below, APEX_IMG doubles) -- copied from the forward image after every factorisation; both sweeps then read LDS (a global
    // load per step is a memory round trip per APEX_QD steps: measured, the sweeps got SLOWER with the apex on global images)
    int apex_lds;
    int apex_inplace; // 1 (LDS-resident build): apex_lds points at the forward image inside the LDS copy of the workspace slab -- nothing to copy after a factorisation
    // one long row of the apex cut into apex_split_n parts by the forward plan's `ext` level (plans.hpp: TriPlan::split_row): the parts' sums sit, negated, in
    // the sweep-vector slots apex_split_slot .. + apex_split_n - 1 when the apex sweep starts; it adds them to lane apex_split_lane and zeroes them
    int apex_split_lane, apex_split_slot, apex_split_n;
    int tl_nimg, tl_scratch;       // entries of the K image scatter; offset (doubles) of the per-wave LDS scratch
    gint_p tl_blev, tl_tgt_lev, tl_tgt, tl_tp, tl_pa, tl_pb, tl_pk, tl_fin_lev, tl_fin; // levels, factor targets / pairs, finalise lists
    gint_p tl_trow, tl_tcol, tl_tc_ptr, tl_tr_ptr, tl_tr_tile; // tiles: block row / column; CSC pointer; CSR view
    gint_p tl_fops, tl_bops, tl_fptr, tl_bptr; // per-wavefront flat schedules of the two sweeps (tiles.hpp: TileSweeps), int4 per op; ptr: [(level * 2 + phase) * NW + wave]
    gint_p tl_fsplit, tl_bsplit;               // per level (sweep order): 1 = the level has split blocks, i.e. a second phase behind a barrier
    gint_p tl_fend, tl_bend;                   // [(level * 2 + phase) * NW + wave]: end of the range's REAL operations (the rest of the range is padding: never executed)
    int tl_part;                               // offset (doubles) of the TILE_PARTS partial-sum slots (16 x KI_MAX doubles each) in the dynamic LDS
    gint_p tl_facops, tl_facptr;               // per-wavefront flat schedule of the factorisation's accumulation phase (TileFactorOps)
    gint_p tl_ident;               // per block: 1 = the diagonal tile of L is the identity (skipped by the sweeps)
    gint_p tl_img_dst, tl_img_src, tl_psign; // K image scatter (slab offset -> image index); quasi-definite pivot sign per slot
    int w_LC, w_LR, w_DL;                // workspace: L tiles column- / row-major, strictly lower part of the diagonal tiles (row-major)
    // instance slab offsets
    int i_Av, i_Gv, i_cag, i_rA, i_rG, i_c, i_h, i_b, i_xe, i_ae, i_ge, i_Vv, i_cst, i_x, i_y, i_z, i_s, i_info;
    // workspace slab offsets
    int w_lam, w_bx, w_by, w_bz, w_bs, w_rz, w_rhs1k, w_rhs2k;
    int w_dx1, w_dy1, w_dz1, w_dx2, w_dy2, w_dz2, w_dsw, w_wdz, w_dsa, w_t1, w_t2;
    int w_lpw, w_lpv, w_csc, w_qv, w_xk, w_ek, w_dxr, w_UF, w_UB, w_D, w_invD, w_trace;
    int lds_tab;                     // dynamic LDS: offset (doubles) of the slice tables behind the KKT-space vector(s)
    int dual, w_dual_xk, w_dual_ek;  // dual right-hand-side solves: flag + the two 2-interleaved vectors in the workspace
    int lr_inst, lr_work;            // LDS-resident variant: offsets (doubles) of the instance slab and the workspace slab in the dynamic LDS
    // G in dense 16 x 16 tiles (api.cpp): gt_nrb row blocks of 16 rows, tiles [gt_rbptr[rb], gt_rbptr[rb+1]) of row block rb,
    // 16 columns per tile (gt_col: variable index or -1, gt_colk: elimination-order slot), gt_zslot: slot of z_i per row,
    // gt_cidx: per column gt_W indices into the partial sums, gt_src: value source of every tile element (updateData)
    int gt_on, gt_nrb, gt_nt, gt_W, i_Gt, w_gpart, w_gx, w_gz;
    gint_p gt_rbptr, gt_col, gt_colk, gt_zslot, gt_cidx, gt_src;
    size_t inst_stride, work_stride; // in doubles
};

// Tile-internal element order (tile mode): a 16 x 16 tile is stored so that lane l of a wavefront owns the four
// consecutive doubles 4 l .. 4 l + 3 (two 16-byte loads per lane, 2 KB contiguous per wavefront):
//   operand order (LC): element (row r, column k) at tile_op(r, k)  -- lane (k&3)*16 + r holds K-step k>>2: exactly what
//     lane l of v_mfma_f64_16x16x4_f64 needs as A[r = l&15][4 s + (l>>4)] (and, for the transposed factor, as B);
//   result order (LR, the K image): element (r, c) at tile_res(r, c) = tile_op(c, r) -- lane (r&3)*16 + c, register r>>2:
//     exactly the MFMA result layout C[(l>>4) + 4 reg][l&15], so an accumulator tile is stored with one 32-byte store per lane.
constexpr int tile_op(int r, int k) { return (((k & 3) * 16 + r) << 2) + (k >> 2); }
constexpr int tile_res(int r, int c) { return (((r & 3) * 16 + c) << 2) + (r >> 2); }
constexpr int TOP_DIAG = 1, TOP_IDENT = 2; // tile sweep op flags: closes its block (diagonal tile) / that diagonal tile is the identity (no load)
// A block whose row (forward) / column (backward) of tiles is much longer than a wavefront's share of its level is SPLIT over several
// wavefronts (tiles.hpp: build_tile_sweeps): every part ends with a TOP_PART operation that writes the wavefront's partial sum to an LDS
// slot (op.z), and the block is closed in a second phase of the level, after a barrier, by a TOP_DIAG operation that adds its
// TOP_NPART (flags >> TOP_NPART_SHIFT) partial sums, slots op.z ..., in slot order.  TILE_PARTS = slots per level.
constexpr int TOP_PART = 4, TOP_NPART_SHIFT = 8, TILE_PARTS = 32;
constexpr int TILE_PF = 6;          // tile loads in flight per wavefront in the tile sweeps (op lists are padded to a multiple)
#ifndef EICOS_TILE_FPF
#define EICOS_TILE_FPF 3
#endif
constexpr int TILE_FPF = EICOS_TILE_FPF;         // operations (two tiles + a D block each) in flight per wavefront in the tile factorisation
constexpr int TILE_FTRIP = TILE_FPF;             // operations per trip of the unrolled loops (factor / sweeps): see ELL_TRIP; op lists are padded to a multiple
constexpr int TILE_STRIP = 6;                    // (longer trips measured in round 3: no change on the bandwidth-bound tile path)
static_assert(TILE_FTRIP % TILE_FPF == 0 && TILE_STRIP % TILE_PF == 0, "the register queues rotate inside a trip");
constexpr int FOP_INIT = 1, FOP_END = 2, FOP_PAD = 4, FOP_ZERO = 8, FOP_SHIFT = 4; // tile factor op flags: start a target from its K tile / finish it / padding / (with INIT) the K tile is structurally zero: start from 0, no 2 KB load; target id above
constexpr int TILE_SCR = 16 * 17;   // doubles of LDS scratch per wavefront in tile mode (one padded 16 x 16 tile)
constexpr int TRACE_COLS = 12, TRACE_ROWS = 102; // per-iteration history rows (iter 0..100)
constexpr int CONE_BIG = 32;       // cones of at least this dimension get a wavefront each
constexpr int TINY_D = 4, TINY_INTS = 12; // tiny cones: dimension <= TINY_D (DevPat::cone_tiny)
constexpr int CSC_STRIDE = 24;     // doubles of scaling state per cone: three 64-byte sectors, the committed scalars CS_* (what the per-solve cone loops read) fill the first
// per-cone scaling scalars (reference struct SOCone, include/eicos.hpp:81-95): CS_* committed,
// CN_* candidates of the current updateScalings pass (committed only if no earlier cone failed)
enum { CS_A = 0, CS_D1, CS_W, CS_ETA, CS_ETA2, CS_U0, CS_U1, CS_V1,
       CN_A, CN_D1, CN_W, CN_ETA2, CN_U0, CN_U1, CN_V1, CN_SN, CN_ZN, CN_GAM,
       CN_MODE /* 0 = candidate complete, 2 = failed at the c2byu02 - d test (ref :460-463: eta, eta^2 and q are already new) */ };
static_assert(CN_MODE < CSC_STRIDE, "per-cone scaling state");

} // namespace eicos
