// Shared host/device plain-data types of the batched solver.
#pragma once
#include <cstddef>

namespace eicos {

// Per-instance persistent scalar state: struct Work's scalars + struct Information
// (reference include/eicos.hpp:49-73,97-114) with std::optional flattened.
struct DevInfo {
    double pcost, dcost, pres, dres, gap, relgap, sigma, mu, step, step_aff, kapovert;
    double pinfres, dinfres, tau, kap, cx, by, hz;
    int has_relgap, has_pinfres, has_dinfres, pinf, dinf;
    int iter, nitref1, nitref2, nitref3, exitcode, n_factor, n_ldlsolve, equilibrated, pad_;
};
static_assert(sizeof(DevInfo) == 200, "DevInfo layout");
constexpr int DEVINFO_DOUBLES = 32;

// Everything the kernels need to know about the (shared) pattern.  All pointers are device
// pointers into one int32 pattern buffer; all i_* / w_* members are offsets in doubles into
// the per-instance slab / the per-resident-workgroup workspace slab.
struct DevPat {
    int n, p, m, l, nc, N, mt, nV, nnzA, nnzG, nnzL, nlev;
    // A, G in CSC (column) and transposed (row) form; *_k = row index as KKT index
    const int *Ajc, *Air, *Air_k, *At_ptr, *At_col, *At_pos;
    const int *Gjc, *Gir, *Gir_k, *Gt_ptr, *Gt_col, *Gt_pos;
    const int *A_long, *At_long, *G_long, *Gt_long; // columns / rows longer than LONG_SEG
    int nA_long, nAt_long, nG_long, nGt_long;
    // cones
    const int *cq, *cone_off, *cone_vbase, *cone_small, *cone_big;
    int n_small, n_big;
    const int *zexp;    // [m] expanded (rhs / KKT cone block) position of z row i
    const int *zdsign;  // [m] +1 / -1: sign of the static-regularisation term in refinement
    // LDL' pattern, level ordered
    const int *perm, *lev_ptr, *Rp, *Rj, *Lp, *Li, *Cpos;
    const int *fwd_long_ptr, *fwd_long, *bwd_long_ptr, *bwd_long;
    const int *ftask_ptr, *ftask, *ftask_nlong, *tp, *pa, *pb, *pk, *Lsrc, *Dsrc;
    // instance slab offsets
    int i_Av, i_Gv, i_Atv, i_Gtv, i_c, i_h, i_b, i_xe, i_ae, i_ge, i_Vv, i_cst, i_x, i_y, i_z, i_s, i_info;
    // workspace slab offsets
    int w_lam, w_bx, w_by, w_bz, w_bs, w_blam, w_rx, w_ry, w_rz, w_rhs1, w_rhs2;
    int w_dx1, w_dy1, w_dz1, w_dx2, w_dy2, w_dz2, w_dsw, w_wdz, w_dsa, w_t1, w_t2;
    int w_lpw, w_lpv, w_csc, w_qv, w_xk, w_ek, w_dxr, w_ws, w_U, w_Ur, w_D, w_invD, w_trace;
    size_t inst_stride, work_stride; // in doubles
};

constexpr int TRACE_COLS = 12, TRACE_ROWS = 102; // per-iteration history rows (iter 0..100)
constexpr int LONG_SEG = 48;       // segments longer than this are reduced by a whole wavefront
constexpr int CONE_BIG = 32;       // cones of at least this dimension get a wavefront each
constexpr int CSC_STRIDE = 20;     // doubles of scaling state per cone
// per-cone scaling scalars (reference struct SOCone, include/eicos.hpp:81-95): CS_* committed,
// CN_* candidates of the current updateScalings pass (committed only if no earlier cone failed)
enum { CS_A = 0, CS_D1, CS_W, CS_ETA, CS_ETA2, CS_U0, CS_U1, CS_V1,
       CN_A, CN_D1, CN_W, CN_ETA2, CN_U0, CN_U1, CN_V1, CN_SN, CN_ZN, CN_GAM };

} // namespace eicos
