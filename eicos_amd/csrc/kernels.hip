// HIP kernels (gfx950 / CDNA4) of the batched SOCP interior-point solver.
//
// Execution model: ONE WORKGROUP PER PROBLEM INSTANCE, persistent over the whole
// interior-point solve.  The grid is sized to what is co-resident on the chip
// (256 CUs x blocks/CU); each workgroup walks instances blockIdx.x, +gridDim.x, ...
// Every step of the reference's solve() (reference src/eicos.cpp:848-1262) is executed by
// all threads of the workgroup with wave-uniform control flow; scalars are computed
// redundantly from workgroup-wide reductions (64-lane shuffle trees + one LDS hop), so the
// per-instance state machine (exit tests, refinement counts, safeguards) needs no
// inter-workgroup communication and no active-mask compaction.
//
// Data: per-instance values live in a slab in HBM (coalesced by construction: consecutive
// lanes walk consecutive entries of one instance's CSC/CSR value arrays); the index arrays
// are shared by all instances and stay L2-resident.  The sparse LDL' is level scheduled:
// nodes are renumbered on the host so each elimination-tree level is a contiguous range;
// factorisation is a left-looking "one thread (or one wavefront) per target entry" program,
// triangular solves are gather-form segmented dot products per level.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <type_traits>

#include "device_types.hpp"
#include "launch.hpp"

namespace eicos {

// ---- constants of struct Settings (reference include/eicos.hpp:23-47) ----
__device__ constexpr double GAMMA = 0.99, DELTASTAT = 7e-8;
__device__ constexpr double FEASTOL = 1e-8, ABSTOL = 1e-8, RELTOL = 1e-8;
__device__ constexpr double FEASTOL_INACC = 1e-4, ABSTOL_INACC = 5e-5, RELTOL_INACC = 5e-5;
__device__ constexpr int NITREF = 9, EQUIL_ITERS = 3, ITER_MAX = 100;
__device__ constexpr double LINSYSACC = 1e-14, IRERRFACT = 6., STEPMIN = 1e-6, STEPMAX = 0.999;
__device__ constexpr double SIGMAMIN = 1e-4, SIGMAMAX = 1.0, SAFEGUARD = 500.;
constexpr int EX_NOT_CONVERGED = -87;

constexpr int RED_SLOTS = 16 * 8; // up to 16 wavefronts x 8 values per reduction

template <int T>
struct Blk {
    double *red; // LDS, 2*RED_SLOTS doubles (ping-pong)
    int *flag;   // LDS ints
    int phase;
    int tid, lane, wave;
    static constexpr int NW = T / 64;
};

struct OpSum { __device__ static double f(double a, double b) { return a + b; } };
struct OpMax { __device__ static double f(double a, double b) { return fmax(a, b); } };
struct OpMin { __device__ static double f(double a, double b) { return fmin(a, b); } };

template <class Op>
__device__ __forceinline__ double wave_reduce(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = Op::f(v, __shfl_xor(v, o, 64));
    return v;
}

// Workgroup-wide reduction of NV values; every thread gets the result.  One barrier:
// the LDS scratch is ping-ponged between consecutive reductions.
template <class Op, int T, int NV>
__device__ __forceinline__ void blk_reduce(Blk<T> &b, double (&v)[NV]) {
    static_assert(NV <= 8, "too many values");
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = wave_reduce<Op>(v[i]);
    double *buf = b.red + b.phase * RED_SLOTS;
    b.phase ^= 1;
    if (b.lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) buf[b.wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) {
        double r = buf[i];
        for (int w = 1; w < Blk<T>::NW; w++) r = Op::f(r, buf[w * NV + i]);
        v[i] = r;
    }
}
template <class Op, int T>
__device__ __forceinline__ double blk_reduce1(Blk<T> &b, double x) {
    double v[1] = {x};
    blk_reduce<Op, T, 1>(b, v);
    return v[0];
}

// Segmented sparse dot products over rows/columns [r0,r1): short segments one thread each,
// long ones (listed in longlist) one wavefront each.  epi(r, sum) runs on exactly one thread.
template <int T, class Epi>
__device__ __forceinline__ void seg_dots(const Blk<T> &b, int r0, int r1, const int *__restrict__ ptr,
                                         const int *__restrict__ idx, const double *__restrict__ val,
                                         const double *__restrict__ x, const int *__restrict__ longlist, int nlong,
                                         Epi &&epi) {
    for (int r = r0 + b.tid; r < r1; r += T) {
        const int k0 = ptr[r], k1 = ptr[r + 1];
        if (k1 - k0 > LONG_SEG) continue;
        double s = 0.;
        for (int k = k0; k < k1; k++) s += val[k] * x[idx[k]];
        epi(r, s);
    }
    for (int q = b.wave; q < nlong; q += Blk<T>::NW) {
        const int r = longlist[q];
        const int k0 = ptr[r], k1 = ptr[r + 1];
        double s = 0.;
        for (int k = k0 + b.lane; k < k1; k += 64) s += val[k] * x[idx[k]];
        s = wave_reduce<OpSum>(s);
        if (b.lane == 0) epi(r, s);
    }
}

// Column products with the stacked matrix [A; G]: s_j = sum_k A[k,j] xa[ia[k]] + sum_k G[k,j] xg[ig[k]]
template <int T, class Epi>
__device__ __forceinline__ void col_dots_AG(const Blk<T> &b, const DevPat &P, const double *__restrict__ Av,
                                            const double *__restrict__ Gv, const int *__restrict__ ia,
                                            const int *__restrict__ ig, const double *__restrict__ xa,
                                            const double *__restrict__ xg, Epi &&epi) {
    for (int j = b.tid; j < P.n; j += T) {
        const int a0 = P.Ajc[j], a1 = P.Ajc[j + 1], g0 = P.Gjc[j], g1 = P.Gjc[j + 1];
        if ((a1 - a0) + (g1 - g0) > LONG_SEG) continue;
        double s = 0.;
        for (int k = a0; k < a1; k++) s += Av[k] * xa[ia[k]];
        for (int k = g0; k < g1; k++) s += Gv[k] * xg[ig[k]];
        epi(j, s);
    }
    for (int q = b.wave; q < P.nA_long; q += Blk<T>::NW) { // A_long lists the long columns of [A;G]
        const int j = P.A_long[q];
        const int a0 = P.Ajc[j], a1 = P.Ajc[j + 1], g0 = P.Gjc[j], g1 = P.Gjc[j + 1];
        double s = 0.;
        for (int k = a0 + b.lane; k < a1; k += 64) s += Av[k] * xa[ia[k]];
        for (int k = g0 + b.lane; k < g1; k += 64) s += Gv[k] * xg[ig[k]];
        s = wave_reduce<OpSum>(s);
        if (b.lane == 0) epi(j, s);
    }
}

template <int G> __device__ __forceinline__ double grp_sum(double v) {
    if constexpr (G == 64) return wave_reduce<OpSum>(v);
    else return v;
}

// Run body(c, integral_constant<G>, lane) once per cone: small cones one thread each (G=1),
// big cones one wavefront each (G=64).
template <int T, class Body>
__device__ __forceinline__ void for_cones(const Blk<T> &b, const DevPat &P, Body &&body) {
    for (int q = b.tid; q < P.n_small; q += T) body(P.cone_small[q], std::integral_constant<int, 1>{}, 0);
    for (int q = b.wave; q < P.n_big; q += Blk<T>::NW) body(P.cone_big[q], std::integral_constant<int, 64>{}, b.lane);
}

#define FOR_T(i, cnt) for (int i = b.tid; i < (cnt); i += T)

// ============================================================================================
// One instance, whole solve.  Follows reference Solver::solve (src/eicos.cpp:848-1262).
// ============================================================================================
template <int T>
__device__ void solve_instance(const DevPat &P, double *__restrict__ I, double *__restrict__ W, Blk<T> &b) {
    const int n = P.n, p = P.p, m = P.m, l = P.l, N = P.N, mt = P.mt, np = P.n + P.p;
    double *Av = I + P.i_Av, *Gv = I + P.i_Gv, *Atv = I + P.i_Atv, *Gtv = I + P.i_Gtv;
    double *cv = I + P.i_c, *hv = I + P.i_h, *bv = I + P.i_b;
    double *xe = I + P.i_xe, *ae = I + P.i_ae, *ge = I + P.i_ge, *Vv = I + P.i_Vv;
    double *wx = I + P.i_x, *wy = I + P.i_y, *wz = I + P.i_z, *wsl = I + P.i_s;
    DevInfo *ginfo = reinterpret_cast<DevInfo *>(I + P.i_info);
    double *lam = W + P.w_lam, *bx_ = W + P.w_bx, *by_ = W + P.w_by, *bz_ = W + P.w_bz, *bs_ = W + P.w_bs, *blam = W + P.w_blam;
    double *rx = W + P.w_rx, *ry = W + P.w_ry, *rz = W + P.w_rz, *rhs1 = W + P.w_rhs1, *rhs2 = W + P.w_rhs2;
    double *dx1 = W + P.w_dx1, *dy1 = W + P.w_dy1, *dz1 = W + P.w_dz1, *dx2 = W + P.w_dx2, *dy2 = W + P.w_dy2, *dz2 = W + P.w_dz2;
    double *dsw = W + P.w_dsw, *wdz = W + P.w_wdz, *dsa = W + P.w_dsa, *t1 = W + P.w_t1, *t2 = W + P.w_t2;
    double *lpw = W + P.w_lpw, *lpv = W + P.w_lpv, *csc = W + P.w_csc, *qv = W + P.w_qv;
    double *xk = W + P.w_xk, *ek = W + P.w_ek, *dxr = W + P.w_dxr, *ws = W + P.w_ws;
    double *U = W + P.w_U, *Ur = W + P.w_Ur, *D = W + P.w_D, *invD = W + P.w_invD;

    DevInfo wi = *ginfo; // sticky across solve() calls like the reference's w.i
    DevInfo bi = wi;
    wi.n_factor = 0; wi.n_ldlsolve = 0;

    // ---------------- numeric LDL' (replaces ldlt.factorize, ref :900,1164) ----------------
    auto factor = [&]() -> bool {
        if (b.tid == 0) b.flag[0] = 0;
        __syncthreads();
        for (int v = 0; v < P.nlev; v++) {
            const int q0 = P.ftask_ptr[v], q1 = P.ftask_ptr[v + 1], nl = P.ftask_nlong[v];
            auto store = [&](int tgt, double s) {
                if (tgt < N) {
                    const double d = I[P.Dsrc[tgt]] - s;
                    D[tgt] = d; invD[tgt] = 1. / d;
                    if (d == 0.) b.flag[0] = 1; // zero pivot -> fatal (Eigen NumericalIssue)
                } else {
                    const int e = tgt - N;
                    const double u = I[P.Lsrc[e]] - s;
                    U[e] = u; Ur[P.Cpos[e]] = u;
                }
            };
            for (int q = q0 + b.wave; q < q0 + nl; q += Blk<T>::NW) {
                const int tgt = P.ftask[q];
                const int k0 = P.tp[tgt], k1 = P.tp[tgt + 1];
                double s = 0.;
                for (int k = k0 + b.lane; k < k1; k += 64) s += U[P.pa[k]] * U[P.pb[k]] * invD[P.pk[k]];
                s = wave_reduce<OpSum>(s);
                if (b.lane == 0) store(tgt, s);
            }
            for (int q = q0 + nl + b.tid; q < q1; q += T) {
                const int tgt = P.ftask[q];
                const int k0 = P.tp[tgt], k1 = P.tp[tgt + 1];
                double s = 0.;
                for (int k = k0; k < k1; k++) s += U[P.pa[k]] * U[P.pb[k]] * invD[P.pk[k]];
                store(tgt, s);
            }
            __syncthreads();
        }
        wi.n_factor++;
        return b.flag[0] == 0;
    };

    // ---------------- x = P' L^-T D^-1 L^-1 P rhs (replaces ldlt.solve, ref :1477,1599) ------
    auto ldl_solve = [&](const double *__restrict__ rhs, double *__restrict__ out) {
        for (int v = 0; v < P.nlev; v++) {
            const int f0 = P.fwd_long_ptr[v];
            seg_dots(b, P.lev_ptr[v], P.lev_ptr[v + 1], P.Rp, P.Rj, Ur, ws, P.fwd_long + f0, P.fwd_long_ptr[v + 1] - f0,
                     [&](int i, double s) { ws[i] = (rhs[P.perm[i]] - s) * invD[i]; });
            __syncthreads();
        }
        for (int v = P.nlev - 1; v >= 0; v--) {
            const int f0 = P.bwd_long_ptr[v];
            seg_dots(b, P.lev_ptr[v], P.lev_ptr[v + 1], P.Lp, P.Li, U, ws, P.bwd_long + f0, P.bwd_long_ptr[v + 1] - f0,
                     [&](int j, double s) { const double xj = ws[j] - invD[j] * s; ws[j] = xj; out[P.perm[j]] = xj; });
            __syncthreads();
        }
        wi.n_ldlsolve++;
    };

    // ---------------- lambda = W z (ref scale :485-507) ----------------
    auto scale = [&](const double *__restrict__ zz, double *__restrict__ out) {
        FOR_T(i, l) out[i] = lpw[i] * zz[i];
        for_cones(b, P, [&](int c, auto G, int lane) {
            constexpr int g = decltype(G)::value;
            const int o = P.cone_off[c], d = P.cq[c];
            const double *cs = csc + c * CSC_STRIDE;
            double zeta = 0.;
            for (int k = 1 + lane; k < d; k += g) zeta += qv[o + k] * zz[o + k];
            zeta = grp_sum<g>(zeta);
            const double z0 = zz[o];
            const double factor = z0 + zeta / (1. + cs[CS_A]);
            const double eta = cs[CS_ETA];
            for (int k = 1 + lane; k < d; k += g) out[o + k] = eta * (zz[o + k] + factor * qv[o + k]);
            if (lane == 0) out[o] = eta * (cs[CS_A] * z0 + zeta);
        });
        __syncthreads();
    };

    // ---------------- solveKKT (ref :1471-1620) ----------------
    auto solve_kkt = [&](const double *__restrict__ rhs, double *__restrict__ dx, double *__restrict__ dy,
                         double *__restrict__ dz, bool init) -> int {
        ldl_solve(rhs, xk);
        double nr = 0.;
        FOR_T(i, N) nr = fmax(nr, fabs(rhs[i]));
        nr = blk_reduce1<OpMax>(b, nr);
        const double thr = (1. + nr) * LINSYSACC;
        double nerr_prev = DBL_MAX;
        const double *bx = rhs, *by = rhs + n, *bz = rhs + np;
        double *ex = ek, *ey = ek + n, *ez = ek + np;
        const double *xz = xk + np; // expanded dz ("dz_true")
        int k;
        for (k = 0; k <= NITREF; k++) {
            // ex = bx - G'dz - A'dy - delta dx   (ref :1515-1521); dz, dy read straight from xk
            double nex = 0., ney = 0., nez = 0.;
            col_dots_AG(b, P, Av, Gv, P.Air_k, P.Gir_k, xk, xk, [&](int j, double s) {
                const double e = bx[j] - s - DELTASTAT * xk[j];
                ex[j] = e; nex = fmax(nex, fabs(e));
            });
            // ey = by - A dx + delta dy   (ref :1525-1531)
            seg_dots(b, 0, p, P.At_ptr, P.At_col, Atv, xk, P.At_long, P.nAt_long, [&](int r, double s) {
                const double e = by[r] - s + DELTASTAT * xk[n + r];
                ey[r] = e; ney = fmax(ney, fabs(e));
            });
            // ez (rows of G) = bz - G dx +/- delta dz  (ref :1535-1555), then + V dz_true
            seg_dots(b, 0, m, P.Gt_ptr, P.Gt_col, Gtv, xk, P.Gt_long, P.nGt_long, [&](int i, double s) {
                const int e = P.zexp[i];
                double v = bz[e] - s + (double)P.zdsign[i] * DELTASTAT * xz[e];
                if (i < l) { v += init ? xz[e] : lpv[i] * xz[e]; nez = fmax(nez, fabs(v)); }
                ez[e] = v;
            });
            if (P.nc > 0) {
                __syncthreads();
                // cone blocks (expanded): ez += dz_true (init) or scale2add (ref :1629-1662)
                for_cones(b, P, [&](int c, auto G, int lane) {
                    constexpr int g = decltype(G)::value;
                    const int d = P.cq[c], o = P.cone_off[c], i1 = o + 2 * c, i3 = i1 + d, i4 = i3 + 1;
                    double mx = 0.;
                    if (init) {
                        for (int q = lane; q < d; q += g) { const double v = ez[i1 + q] + xz[i1 + q]; ez[i1 + q] = v; mx = fmax(mx, fabs(v)); }
                        if (lane == 0) { ez[i3] = xz[i3]; ez[i4] = xz[i4]; mx = fmax(mx, fmax(fabs(xz[i3]), fabs(xz[i4]))); }
                    } else {
                        const double *cs = csc + c * CSC_STRIDE;
                        const double eta2 = cs[CS_ETA2], x1 = xz[i1], x3 = xz[i3], x4 = xz[i4];
                        const double tt = cs[CS_V1] * x3 + cs[CS_U1] * x4;
                        double qtx = 0.;
                        for (int q = 1 + lane; q < d; q += g) {
                            const double qq = qv[o + q], xq = xz[i1 + q];
                            const double v = ez[i1 + q] + eta2 * (xq + tt * qq);
                            ez[i1 + q] = v; mx = fmax(mx, fabs(v));
                            qtx += qq * xq;
                        }
                        qtx = grp_sum<g>(qtx);
                        if (lane == 0) {
                            const double v1 = ez[i1] + eta2 * (cs[CS_D1] * x1 + cs[CS_U0] * x4);
                            const double v3 = eta2 * (cs[CS_V1] * qtx + x3);
                            const double v4 = eta2 * (cs[CS_U0] * x1 + cs[CS_U1] * qtx - x4);
                            ez[i1] = v1; ez[i3] = v3; ez[i4] = v4;
                            mx = fmax(mx, fmax(fabs(v1), fmax(fabs(v3), fabs(v4))));
                        }
                    }
                    nez = fmax(nez, mx);
                });
            }
            double nv[3] = {nex, ney, nez};
            blk_reduce<OpMax, T, 3>(b, nv);
            double nerr = fmax(nv[0], nv[2]);
            if (p > 0) nerr = fmax(nerr, nv[1]);
            if (k > 0 && nerr > nerr_prev) { // got worse: undo and quit (ref :1579-1585)
                FOR_T(i, N) xk[i] -= dxr[i];
                k--;
                break;
            }
            if (k == NITREF || nerr < thr || (k > 0 && nerr_prev < IRERRFACT * nerr)) break;
            nerr_prev = nerr;
            ldl_solve(ek, dxr);
            FOR_T(i, N) xk[i] += dxr[i];
            __syncthreads();
        }
        __syncthreads();
        FOR_T(j, n) dx[j] = xk[j];
        FOR_T(j, p) dy[j] = xk[n + j];
        FOR_T(i, m) dz[i] = xz[P.zexp[i]];
        __syncthreads();
        return k;
    };

    // ---------------- bringToCone (ref :761-805): s = sgn*r shifted into the cone ----------------
    auto bring_to_cone = [&](const double *__restrict__ r, double sgn, double *__restrict__ s) {
        double a = -GAMMA;
        FOR_T(i, l) { const double ri = sgn * r[i]; if (ri <= 0. && -ri > a) a = -ri; }
        for_cones(b, P, [&](int c, auto G, int lane) {
            constexpr int g = decltype(G)::value;
            const int o = P.cone_off[c], d = P.cq[c];
            double t = 0.;
            for (int k = 1 + lane; k < d; k += g) t += r[o + k] * r[o + k];
            t = grp_sum<g>(t);
            const double cres = sgn * r[o] - sqrt(t);
            if (cres <= 0. && -cres > a) a = -cres;
        });
        a = blk_reduce1<OpMax>(b, a) + 1.;
        FOR_T(i, m) s[i] = sgn * r[i];
        __syncthreads();
        FOR_T(i, l) s[i] += a;
        FOR_T(c, P.nc) s[P.cone_off[c]] += a;
        __syncthreads();
    };

    // ---------------- checkExitConditions (ref :526-641) ----------------
    auto check_exit = [&](bool reduced) -> int {
        const double feastol = reduced ? FEASTOL_INACC : FEASTOL;
        const double abstol = reduced ? ABSTOL_INACC : ABSTOL;
        const double reltol = reduced ? RELTOL_INACC : RELTOL;
        const bool relgap_lt = !wi.has_relgap || wi.relgap < reltol;    // optional<double> < x: true if empty
        const bool pinfres_lt = !wi.has_pinfres || wi.pinfres < feastol;
        if ((-wi.cx > 0. || -wi.by - wi.hz >= -abstol) && (wi.pres < feastol && wi.dres < feastol) &&
            (wi.gap < abstol || relgap_lt)) {
            wi.pinf = 0; wi.dinf = 0;
            return 0 + (reduced ? 10 : 0);
        }
        if (wi.has_dinfres && wi.dinfres < feastol && wi.tau < wi.kap) {
            wi.pinf = 0; wi.dinf = 1;
            return 2 + (reduced ? 10 : 0);
        }
        if ((wi.has_pinfres && wi.pinfres < feastol && wi.tau < wi.kap) ||
            (wi.tau < feastol && wi.kap < feastol && pinfres_lt)) {
            wi.pinf = 1; wi.dinf = 0;
            return 1 + (reduced ? 10 : 0);
        }
        return EX_NOT_CONVERGED;
    };
    // Information::isBetterThan (ref :23-68)
    auto better_than = [&](const DevInfo &a, const DevInfo &o) -> bool {
        const bool gap_ok = a.gap > 0. && o.gap > 0. && a.gap < o.gap;
        const bool mu_ok = a.mu > 0. && a.mu < o.mu;
        if (a.has_pinfres && a.kapovert > 1.) {
            if (o.has_pinfres) return gap_ok && (a.pinfres > 0. && a.pinfres < o.pres) && mu_ok;
            return gap_ok && mu_ok;
        }
        return gap_ok && (a.pres > 0. && a.pres < o.pres) && (a.dres > 0. && a.dres < o.dres) &&
               (a.kapovert > 0. && a.kapovert < o.kapovert) && mu_ok;
    };
    auto save_best = [&]() { // w_best = w (ref :1153,1157)
        FOR_T(j, n) bx_[j] = wx[j];
        FOR_T(j, p) by_[j] = wy[j];
        FOR_T(i, m) { bz_[i] = wz[i]; bs_[i] = wsl[i]; blam[i] = lam[i]; }
        bi = wi;
    };
    auto restore_best = [&]() { // w = w_best
        FOR_T(j, n) wx[j] = bx_[j];
        FOR_T(j, p) wy[j] = by_[j];
        FOR_T(i, m) { wz[i] = bz_[i]; wsl[i] = bs_[i]; lam[i] = blam[i]; }
        const int nf = wi.n_factor, ns = wi.n_ldlsolve;
        wi = bi; wi.n_factor = nf; wi.n_ldlsolve = ns;
        __syncthreads();
    };

    // ---------------- lineSearch (ref :1380-1469) ----------------
    auto line_search = [&](const double *__restrict__ ds, const double *__restrict__ dz, double tau, double dtau,
                           double kap, double dkap) -> double {
        double rmin = DBL_MAX, smin = DBL_MAX, cstep = 0., bad = 0.;
        FOR_T(i, l) { const double li = lam[i]; rmin = fmin(rmin, ds[i] / li); smin = fmin(smin, dz[i] / li); }
        auto cone_step = [&](int o, int d, auto G, int lane, bool &skipped) -> double {
            constexpr int g = decltype(G)::value;
            double l1 = 0.;
            for (int k = 1 + lane; k < d; k += g) l1 += lam[o + k] * lam[o + k];
            l1 = grp_sum<g>(l1);
            const double lknorm2 = lam[o] * lam[o] - l1;
            if (lknorm2 <= 0.) { skipped = true; return 0.; }
            skipped = false;
            const double lknorm = sqrt(lknorm2), inv = 1. / lknorm, lk0 = lam[o] / lknorm;
            double ld = 0., lz = 0.;
            for (int k = 1 + lane; k < d; k += g) { const double lb = lam[o + k] / lknorm; ld += lb * ds[o + k]; lz += lb * dz[o + k]; }
            ld = grp_sum<g>(ld); lz = grp_sum<g>(lz);
            const double lds = lk0 * ds[o] - ld, ldz = lk0 * dz[o] - lz;
            const double rho0 = inv * lds, fr = (lds + ds[o]) / (lk0 + 1.);
            const double sig0 = inv * ldz, fs = (ldz + dz[o]) / (lk0 + 1.);
            double rn = 0., sn = 0.;
            for (int k = 1 + lane; k < d; k += g) {
                const double lb = lam[o + k] / lknorm;
                const double r = inv * (ds[o + k] - fr * lb), s = inv * (dz[o + k] - fs * lb);
                rn += r * r; sn += s * s;
            }
            rn = grp_sum<g>(rn); sn = grp_sum<g>(sn);
            return fmax(0., fmax(sqrt(sn) - sig0, sqrt(rn) - rho0));
        };
        for_cones(b, P, [&](int c, auto G, int lane) {
            bool sk;
            const double st = cone_step(P.cone_off[c], P.cq[c], G, lane, sk);
            if (sk) bad = 1.; else cstep = fmax(cstep, st);
        });
        double v4[4] = {-rmin, -smin, cstep, bad};
        blk_reduce<OpMax, T, 4>(b, v4);
        rmin = -v4[0]; smin = -v4[1]; cstep = v4[2];
        double alpha;
        if (l > 0) {
            const double eps = 1e-13;
            if (-smin > -rmin) alpha = smin < 0. ? 1. / (-smin) : 1. / eps;
            else alpha = rmin < 0. ? 1. / (-rmin) : 1. / eps;
        } else alpha = 10.;
        const double mtd = -tau / dtau, mkd = -kap / dkap;
        if (mtd > 0. && mtd < alpha) alpha = mtd;
        if (mkd > 0. && mkd < alpha) alpha = mkd;
        if (v4[3] == 0.) {
            if (cstep != 0.) alpha = fmin(1. / cstep, alpha);
        } else {
            // Rare path: some cone has lknorm2 <= 0.  The reference `continue`s WITHOUT advancing
            // cone_start (ref :1423-1424), so later cones read shifted segments; emulate that
            // sequentially (every thread redundantly, G=1).
            int o = l;
            for (int c = 0; c < P.nc; c++) {
                bool sk;
                const double st = cone_step(o, P.cq[c], std::integral_constant<int, 1>{}, 0, sk);
                if (sk) continue;
                if (st != 0.) alpha = fmin(1. / st, alpha);
                o += P.cq[c];
            }
        }
        return fmin(fmax(alpha, STEPMIN), STEPMAX);
    };

    // ======================= solve() body =======================
    int code = -7;
    // resetKKTScalings (ref :807-846)
    FOR_T(i, l) Vv[i] = -1.;
    for_cones(b, P, [&](int c, auto G, int lane) {
        constexpr int g = decltype(G)::value;
        const int d = P.cq[c];
        double *v = Vv + P.cone_vbase[c];
        for (int k = lane; k < d; k += g) { v[k] = -1.; v[2 * d + 1 + k] = 0.; if (k >= 1) v[d + k] = 0.; }
        if (lane == 0) { v[d] = -1.; v[2 * d] = 1.; }
    });
    // rhs1 = [0; b; h expanded], rhs2 = [-c; 0; 0]   (ref :865-886)
    FOR_T(i, N) { rhs1[i] = 0.; rhs2[i] = 0.; }
    __syncthreads();
    double nr3[3] = {0., 0., 0.};
    FOR_T(j, n) { const double c_ = cv[j]; rhs2[j] = -c_; nr3[0] += c_ * c_; }
    FOR_T(r, p) { const double b_ = bv[r]; rhs1[n + r] = b_; nr3[1] += b_ * b_; }
    FOR_T(i, m) { const double h_ = hv[i]; rhs1[np + P.zexp[i]] = h_; nr3[2] += h_ * h_; }
    blk_reduce<OpSum, T, 3>(b, nr3);
    const double resx0 = fmax(1., sqrt(nr3[0])), resy0 = fmax(1., sqrt(nr3[1])), resz0 = fmax(1., sqrt(nr3[2]));

    bool fatal = !factor(); // ref :900-905
    if (!fatal) {
        wi.nitref1 = solve_kkt(rhs1, dx1, dy1, dz1, true);
        FOR_T(j, n) wx[j] = dx1[j];
        bring_to_cone(dz1, -1., wsl);
        wi.nitref2 = solve_kkt(rhs2, dx2, dy2, dz2, true);
        FOR_T(j, p) wy[j] = dy2[j];
        bring_to_cone(dz2, 1., wz);
        FOR_T(j, n) rhs1[j] = -cv[j];
        wi.kap = 1.; wi.tau = 1.;
        wi.step = 0.; wi.step_aff = 0.; wi.pinf = 0; wi.dinf = 0;
        double pres_prev = DBL_MAX;
        __syncthreads();

        for (wi.iter = 0; wi.iter <= ITER_MAX; wi.iter++) {
            // ---- computeResiduals (ref :643-689) + updateStatistics (ref :691-754) ----
            double r8[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // hresx2 rx2 cx nx2 | hresy2 ry2 by ny2
            col_dots_AG(b, P, Av, Gv, P.Air, P.Gir, wy, wz, [&](int j, double s) {
                const double hr = -s, c_ = cv[j], xj = wx[j];
                const double r = hr - wi.tau * c_;
                rx[j] = r;
                r8[0] += hr * hr; r8[1] += r * r; r8[2] += c_ * xj; r8[3] += xj * xj;
            });
            seg_dots(b, 0, p, P.At_ptr, P.At_col, Atv, wx, P.At_long, P.nAt_long, [&](int r, double s) {
                const double b_ = bv[r], yr = wy[r];
                const double rr = s - wi.tau * b_;
                ry[r] = rr;
                r8[4] += s * s; r8[5] += rr * rr; r8[6] += b_ * yr; r8[7] += yr * yr;
            });
            double q6[6] = {0, 0, 0, 0, 0, 0}; // hresz2 rz2 hz nz2 ns2 gap
            seg_dots(b, 0, m, P.Gt_ptr, P.Gt_col, Gtv, wx, P.Gt_long, P.nGt_long, [&](int i, double s) {
                const double si = wsl[i], zi = wz[i], h_ = hv[i];
                const double hr = si + s, r = hr - wi.tau * h_;
                rz[i] = r;
                q6[0] += hr * hr; q6[1] += r * r; q6[2] += h_ * zi; q6[3] += zi * zi; q6[4] += si * si; q6[5] += si * zi;
            });
            blk_reduce<OpSum, T, 8>(b, r8);
            blk_reduce<OpSum, T, 6>(b, q6);
            const double hresx = sqrt(r8[0]), nrx = sqrt(r8[1]), nx = sqrt(r8[3]);
            const double hresy = p > 0 ? sqrt(r8[4]) : 0., nry2 = sqrt(r8[5]), ny = sqrt(r8[7]);
            const double hresz = sqrt(q6[0]), nrz2 = sqrt(q6[1]), nz = sqrt(q6[3]), ns = sqrt(q6[4]);
            wi.cx = r8[2]; wi.by = p > 0 ? r8[6] : 0.; wi.hz = q6[2];
            const double rt = wi.kap + wi.cx + wi.by + wi.hz;
            wi.gap = q6[5];
            wi.mu = (wi.gap + wi.kap * wi.tau) / (double)((l + P.nc) + 1);
            wi.kapovert = wi.kap / wi.tau;
            wi.pcost = wi.cx / wi.tau;
            wi.dcost = -(wi.hz + wi.by) / wi.tau;
            if (wi.pcost < 0.) { wi.relgap = wi.gap / (-wi.pcost); wi.has_relgap = 1; }
            else if (wi.dcost > 0.) { wi.relgap = wi.gap / wi.dcost; wi.has_relgap = 1; }
            else wi.has_relgap = 0;
            {
                const double nry = p > 0 ? nry2 / fmax(resy0 + nx, 1.) : 0.;
                const double nrz = nrz2 / fmax(resz0 + nx + ns, 1.);
                wi.pres = fmax(nry, nrz) / wi.tau;
                wi.dres = nrx / fmax(resx0 + ny + nz, 1.) / wi.tau;
                if ((wi.hz + wi.by) / fmax(ny + nz, 1.) < -RELTOL) { wi.pinfres = hresx / fmax(ny + nz, 1.); wi.has_pinfres = 1; }
                if (wi.cx / fmax(nx, 1.) < -RELTOL) {
                    wi.dinfres = fmax(hresy / fmax(nx, 1.), hresz / fmax(nx + ns, 1.)); wi.has_dinfres = 1;
                }
            }
            // ---- safeguard / exit logic (ref :1010-1158) ----
            if (wi.iter > 0 && (wi.pres > SAFEGUARD * pres_prev || wi.gap < 0.)) {
                restore_best();
                code = check_exit(true);
                if (code == EX_NOT_CONVERGED) code = -2;
                break;
            }
            pres_prev = wi.pres;
            code = check_exit(false);
            if (code == EX_NOT_CONVERGED) {
                if (wi.iter > 0 && wi.step == STEPMIN * GAMMA) {
                    restore_best();
                    code = check_exit(true);
                    if (code == EX_NOT_CONVERGED) code = -2;
                    break;
                } else if (wi.iter == ITER_MAX) {
                    if (!better_than(wi, bi)) restore_best();
                    code = check_exit(true);
                    if (code == EX_NOT_CONVERGED) code = -1;
                    break;
                } else if (isnan(wi.pcost)) {
                    if (!(wi.iter == 0 || better_than(wi, bi))) {
                        restore_best();
                        code = check_exit(true);
                        if (code == EX_NOT_CONVERGED) code = -2;
                    }
                    break;
                }
            } else break;
            if (wi.iter == 0 || better_than(wi, bi)) save_best();

            // ---- updateScalings (ref :411-479) + updateKKTScalings (ref :1691-1732) ----
            FOR_T(i, l) { const double v = wsl[i] / wz[i]; lpv[i] = v; lpw[i] = sqrt(v); Vv[i] = -v - DELTASTAT; }
            double firstfail = 1e300;
            for_cones(b, P, [&](int c, auto G, int lane) { // phase 1: candidate scalings per cone
                constexpr int g = decltype(G)::value;
                const int o = P.cone_off[c], d = P.cq[c];
                double *cs = csc + c * CSC_STRIDE;
                double s1 = 0., z1 = 0.;
                for (int k = 1 + lane; k < d; k += g) { s1 += wsl[o + k] * wsl[o + k]; z1 += wz[o + k] * wz[o + k]; }
                s1 = grp_sum<g>(s1); z1 = grp_sum<g>(z1);
                const double s0 = wsl[o], z0 = wz[o];
                const double sres = s0 * s0 - s1, zres = z0 * z0 - z1;
                bool fail = (sres <= 0. || zres <= 0.); // uniform across the cone's lanes
                if (!fail) {
                    const double snorm = sqrt(sres), znorm = sqrt(zres);
                    double sz = 0., ww = 0.;
                    for (int k = lane; k < d; k += g) sz += (wsl[o + k] / snorm) * (wz[o + k] / znorm);
                    sz = grp_sum<g>(sz);
                    const double gam = sqrt(0.5 * (1. + sz));
                    const double a = (0.5 / gam) * (s0 / snorm + z0 / znorm);
                    for (int k = 1 + lane; k < d; k += g) {
                        const double qk = (0.5 / gam) * (wsl[o + k] / snorm - wz[o + k] / znorm);
                        ww += qk * qk;
                    }
                    ww = grp_sum<g>(ww);
                    const double cc = (1. + a) + ww / (1. + a);
                    const double dd = 1. + 2. / (1. + a) + ww / ((1. + a) * (1. + a));
                    const double d1 = fmax(0., 0.5 * (a * a + ww * (1. - cc * cc / (1. + ww * dd))));
                    const double u0sq = a * a + ww - d1;
                    const double c2 = cc * cc / u0sq;
                    if (c2 - dd <= 0.) fail = true;
                    else if (lane == 0) {
                        cs[CN_A] = a; cs[CN_D1] = d1; cs[CN_W] = ww; cs[CN_ETA2] = snorm / znorm;
                        cs[CN_U0] = sqrt(u0sq); cs[CN_U1] = sqrt(c2); cs[CN_V1] = sqrt(c2 - dd);
                        cs[CN_SN] = snorm; cs[CN_ZN] = znorm; cs[CN_GAM] = gam;
                    }
                }
                if (fail) firstfail = fmin(firstfail, (double)c);
            });
            // (barrier inside) index of the first cone that left the cone; 1e300 if none.  The reference
            // returns at that cone (ref :428-431,460-463): earlier cones keep their new scalings, later
            // ones their old ones, and lambda is not refreshed.
            firstfail = blk_reduce1<OpMin>(b, firstfail);
            for_cones(b, P, [&](int c, auto G, int lane) { // phase 2: commit + updateKKTScalings
                constexpr int g = decltype(G)::value;
                if ((double)c >= firstfail) return;
                const int o = P.cone_off[c], d = P.cq[c];
                double *cs = csc + c * CSC_STRIDE;
                double *v = Vv + P.cone_vbase[c];
                const double a = cs[CN_A], d1 = cs[CN_D1], eta2 = cs[CN_ETA2], u0 = cs[CN_U0], u1 = cs[CN_U1], v1 = cs[CN_V1];
                const double snorm = cs[CN_SN], znorm = cs[CN_ZN], gam = cs[CN_GAM];
                if (lane == 0) {
                    cs[CS_A] = a; cs[CS_D1] = d1; cs[CS_W] = cs[CN_W]; cs[CS_ETA2] = eta2; cs[CS_ETA] = sqrt(eta2);
                    cs[CS_U0] = u0; cs[CS_U1] = u1; cs[CS_V1] = v1;
                }
                // KKT scaling block, slot order of ref cacheIndices :1955-1986: D[d], vdiag, v[d-1], udiag, u[d]
                for (int k = lane; k < d; k += g) {
                    const double qk = (k >= 1) ? (0.5 / gam) * (wsl[o + k] / snorm - wz[o + k] / znorm) : 0.;
                    if (k >= 1) { qv[o + k] = qk; v[d + k] = -eta2 * v1 * qk; }
                    v[k] = (k == 0) ? -eta2 * d1 - DELTASTAT : -eta2 - DELTASTAT;
                    v[2 * d + 1 + k] = (k == 0) ? -eta2 * u0 : -eta2 * u1 * qk;
                }
                if (lane == 0) { v[d] = -eta2; v[2 * d] = eta2 + DELTASTAT; }
            });
            __syncthreads();
            if (firstfail >= 1e299) scale(wz, lam); // lambda = W z only when every cone succeeded (ref :476)
            // (when a cone failed, cones >= firstfail also keep their previous KKT block: the
            //  reference's updateKKTScalings rewrites them from the stale scalars = same values)

            fatal = !factor(); // ref :1164-1170
            if (fatal) break;

            solve_kkt(rhs1, dx1, dy1, dz1, false);
            // RHSaffine (ref :1670-1689)
            FOR_T(j, n) rhs2[j] = rx[j];
            FOR_T(r, p) rhs2[n + r] = -ry[r];
            FOR_T(i, m) rhs2[np + P.zexp[i]] = wsl[i] - rz[i];
            __syncthreads();
            solve_kkt(rhs2, dx2, dy2, dz2, false);
            double d6[6] = {0, 0, 0, 0, 0, 0}; // c.dx1 b.dy1 h.dz1 c.dx2 b.dy2 h.dz2
            FOR_T(j, n) { const double c_ = cv[j]; d6[0] += c_ * dx1[j]; d6[3] += c_ * dx2[j]; }
            FOR_T(r, p) { const double b_ = bv[r]; d6[1] += b_ * dy1[r]; d6[4] += b_ * dy2[r]; }
            FOR_T(i, m) { const double h_ = hv[i]; d6[2] += h_ * dz1[i]; d6[5] += h_ * dz2[i]; }
            blk_reduce<OpSum, T, 6>(b, d6);
            const double dtau_denom = wi.kap / wi.tau - d6[0] - d6[1] - d6[2];
            const double dtauaff = (rt - wi.kap + d6[3] + d6[4] + d6[5]) / dtau_denom;
            FOR_T(i, m) dz2[i] += dtauaff * dz1[i];
            __syncthreads();
            scale(dz2, wdz);
            FOR_T(i, m) dsw[i] = -wdz[i] - lam[i];
            __syncthreads();
            const double dkapaff = -wi.kap - wi.kap / wi.tau * dtauaff;
            wi.step_aff = line_search(dsw, wdz, wi.tau, dtauaff, wi.kap, dkapaff);
            const double oms_ = 1. - wi.step_aff;
            const double sigma = fmin(fmax(oms_ * oms_ * oms_, SIGMAMIN), SIGMAMAX);
            wi.sigma = sigma;
            // ---- RHScombined (ref :1282-1325) ----
            {
                const double sigmamu = sigma * wi.mu, oms = 1. - sigma;
                // LP part: ds1 = lam*lam + dsw*wdz - sigmamu ; dsw = ds1/lam ; ds1' = w*dsw
                FOR_T(i, l) {
                    const double li = lam[i];
                    const double d1_ = li * li + dsw[i] * wdz[i] - sigmamu;
                    const double q_ = d1_ / li;
                    dsw[i] = q_;
                    t1[i] = lpw[i] * q_;
                }
                for_cones(b, P, [&](int c, auto G, int lane) {
                    constexpr int g = decltype(G)::value;
                    const int o = P.cone_off[c], d = P.cq[c];
                    // conic products (ref :1357-1378): ds1 = lam o lam + dsw o wdz - sigmamu e
                    double ll = 0., dw = 0., u1sq = 0.;
                    for (int k = lane; k < d; k += g) { ll += lam[o + k] * lam[o + k]; dw += dsw[o + k] * wdz[o + k]; }
                    ll = grp_sum<g>(ll); dw = grp_sum<g>(dw);
                    const double l0 = lam[o], a0 = dsw[o], w0_ = wdz[o];
                    const double p0 = ll - sigmamu + dw; // ds1(k) -= sigmamu, then += ds2
                    // conic division v = lam \ ds1 (ref :1330-1351)
                    double zeta = 0.;
                    for (int k = 1 + lane; k < d; k += g) {
                        const double lk = lam[o + k];
                        const double pk_ = (l0 * lk + l0 * lk) + (a0 * wdz[o + k] + w0_ * dsw[o + k]);
                        t2[o + k] = pk_; // ds1 tail
                        u1sq += lk * lk; zeta += lk * pk_;
                    }
                    u1sq = grp_sum<g>(u1sq); zeta = grp_sum<g>(zeta);
                    const double rho = l0 * l0 - u1sq;
                    const double factor = (zeta / l0 - p0) / rho;
                    if constexpr (g == 64) __builtin_amdgcn_wave_barrier();
                    for (int k = 1 + lane; k < d; k += g) dsw[o + k] = factor * lam[o + k] + t2[o + k] / l0;
                    if (lane == 0) dsw[o] = (l0 * p0 - zeta) / rho;
                });
                __syncthreads();
                if (P.nc > 0) { // ds1 = W * (lam \ ds) on the cone part (LP part done above)
                    for_cones(b, P, [&](int c, auto G, int lane) {
                        constexpr int g = decltype(G)::value;
                        const int o = P.cone_off[c], d = P.cq[c];
                        const double *cs = csc + c * CSC_STRIDE;
                        double zeta = 0.;
                        for (int k = 1 + lane; k < d; k += g) zeta += qv[o + k] * dsw[o + k];
                        zeta = grp_sum<g>(zeta);
                        const double z0 = dsw[o], factor = z0 + zeta / (1. + cs[CS_A]), eta = cs[CS_ETA];
                        for (int k = 1 + lane; k < d; k += g) t1[o + k] = eta * (dsw[o + k] + factor * qv[o + k]);
                        if (lane == 0) t1[o] = eta * (cs[CS_A] * z0 + zeta);
                    });
                    __syncthreads();
                }
                FOR_T(j, np) rhs2[j] *= oms;
                FOR_T(i, m) rhs2[np + P.zexp[i]] = -oms * rz[i] + t1[i];
                __syncthreads();
            }
            wi.nitref3 = solve_kkt(rhs2, dx2, dy2, dz2, false);
            double e3[3] = {0, 0, 0};
            FOR_T(j, n) e3[0] += cv[j] * dx2[j];
            FOR_T(r, p) e3[1] += bv[r] * dy2[r];
            FOR_T(i, m) e3[2] += hv[i] * dz2[i];
            blk_reduce<OpSum, T, 3>(b, e3);
            const double bkap = wi.kap * wi.tau + dkapaff * dtauaff - sigma * wi.mu;
            const double dtau = ((1. - sigma) * rt - bkap / wi.tau + e3[0] + e3[1] + e3[2]) / dtau_denom;
            FOR_T(j, n) dx2[j] += dtau * dx1[j];
            FOR_T(r, p) dy2[r] += dtau * dy1[r];
            FOR_T(i, m) dz2[i] += dtau * dz1[i];
            __syncthreads();
            scale(dz2, wdz);
            FOR_T(i, m) dsw[i] = -(dsw[i] + wdz[i]);
            __syncthreads();
            const double dkap = -(bkap + wi.kap * dtau) / wi.tau;
            wi.step = GAMMA * line_search(dsw, wdz, wi.tau, dtau, wi.kap, dkap);
            scale(dsw, dsa);
            const double st = wi.step;
            FOR_T(j, n) wx[j] += st * dx2[j];
            FOR_T(r, p) wy[r] += st * dy2[r];
            FOR_T(i, m) { wz[i] += st * dz2[i]; wsl[i] += st * dsa[i]; }
            wi.kap += st * dkap;
            wi.tau += st * dtau;
            __syncthreads();
        }
    }
    if (fatal) code = -7; // no backscale on fatal (ref :904,1169)
    else {
        // backscale (ref :1271-1277)
        __syncthreads();
        FOR_T(j, n) wx[j] = wx[j] / (xe[j] * wi.tau);
        FOR_T(r, p) wy[r] = wy[r] / (ae[r] * wi.tau);
        FOR_T(i, m) { wz[i] = wz[i] / (ge[i] * wi.tau); wsl[i] = wsl[i] * (ge[i] / wi.tau); }
    }
    wi.exitcode = code;
    __syncthreads();
    if (b.tid == 0) *ginfo = wi;
}

template <int T>
__global__ __launch_bounds__(T) void k_solve(DevPat P, double *inst, double *work, int B) {
    __shared__ double red[2 * RED_SLOTS];
    __shared__ int flag[4];
    Blk<T> b;
    b.red = red; b.flag = flag; b.phase = 0;
    b.tid = threadIdx.x; b.lane = threadIdx.x & 63; b.wave = threadIdx.x >> 6;
    double *W = work + (size_t)blockIdx.x * P.work_stride;
    for (int i = blockIdx.x; i < B; i += gridDim.x) {
        solve_instance<T>(P, inst + (size_t)i * P.inst_stride, W, b);
        __syncthreads();
    }
}

// ============================================================================================
// updateData for a range of instances: reference updateData(double*...) src/eicos.cpp:2053-2082
// = unsetEquilibration (:389-404) -> copy-in -> setEquilibration (:302-374) -> transposes.
// (The KKT "AG" scatter of updateKKTAG :1990-2030 is implicit: the factor kernel reads the
//  equilibrated A/G values in place through DevPat::Lsrc.)
// ============================================================================================
template <int T>
__global__ __launch_bounds__(T) void k_update(DevPat P, double *inst, int first, int count, const double *Gpr,
                                              const double *Apr, const double *cin, const double *hin,
                                              const double *bin, double *scratch) {
    __shared__ double red[2 * RED_SLOTS];
    Blk<T> b;
    b.red = red; b.flag = nullptr; b.phase = 0;
    b.tid = threadIdx.x; b.lane = threadIdx.x & 63; b.wave = threadIdx.x >> 6;
    const int n = P.n, p = P.p, m = P.m, l = P.l;
    double *xt = scratch + (size_t)blockIdx.x * (size_t)(n + p + m), *at = xt + n, *gt = at + p;
    for (int q = blockIdx.x; q < count; q += gridDim.x) {
        double *I = inst + (size_t)(first + q) * P.inst_stride;
        double *Av = I + P.i_Av, *Gv = I + P.i_Gv, *Atv = I + P.i_Atv, *Gtv = I + P.i_Gtv;
        double *cv = I + P.i_c, *hv = I + P.i_h, *bv = I + P.i_b, *xe = I + P.i_xe, *ae = I + P.i_ae, *ge = I + P.i_ge;
        DevInfo *ginfo = reinterpret_cast<DevInfo *>(I + P.i_info);
        const bool was_eq = ginfo->equilibrated != 0;
        __syncthreads();
        // un-equilibrate what is kept, overwrite what is given
        FOR_T(j, n) {
            const double xj = was_eq ? xe[j] : 1.;
            for (int k = P.Ajc[j]; k < P.Ajc[j + 1]; k++)
                Av[k] = Apr ? Apr[(size_t)q * P.nnzA + k] : (was_eq ? Av[k] * (ae[P.Air[k]] * xj) : Av[k]);
            for (int k = P.Gjc[j]; k < P.Gjc[j + 1]; k++)
                Gv[k] = Gpr ? Gpr[(size_t)q * P.nnzG + k] : (was_eq ? Gv[k] * (ge[P.Gir[k]] * xj) : Gv[k]);
            cv[j] = cin ? cin[(size_t)q * n + j] : (was_eq ? cv[j] * xj : cv[j]);
        }
        FOR_T(r, p) bv[r] = Apr ? bin[(size_t)q * p + r] : (was_eq ? bv[r] * ae[r] : bv[r]);
        FOR_T(i, m) hv[i] = Gpr ? hin[(size_t)q * m + i] : (was_eq ? hv[i] * ge[i] : hv[i]);
        __syncthreads();
        FOR_T(j, n) xe[j] = 1.;
        FOR_T(r, p) ae[r] = 1.;
        FOR_T(i, m) ge[i] = 1.;
        auto sq = [](double a) { return fabs(a) < 1e-6 ? 1. : sqrt(a); };
        for (int it = 0; it < EQUIL_ITERS; it++) {
            // column maxima over A and G; row maxima via the transposed position maps
            FOR_T(j, n) {
                double mx = 0.;
                for (int k = P.Ajc[j]; k < P.Ajc[j + 1]; k++) mx = fmax(fabs(Av[k]), mx);
                for (int k = P.Gjc[j]; k < P.Gjc[j + 1]; k++) mx = fmax(fabs(Gv[k]), mx);
                xt[j] = sq(mx);
            }
            FOR_T(r, p) {
                double mx = 0.;
                for (int k = P.At_ptr[r]; k < P.At_ptr[r + 1]; k++) mx = fmax(fabs(Av[P.At_pos[k]]), mx);
                at[r] = sq(mx);
            }
            FOR_T(i, m) {
                double mx = 0.;
                for (int k = P.Gt_ptr[i]; k < P.Gt_ptr[i + 1]; k++) mx = fmax(fabs(Gv[P.Gt_pos[k]]), mx);
                gt[i] = (i < l) ? sq(mx) : mx; // cone rows: sqrt after the per-cone sum (ref :338-350)
            }
            __syncthreads();
            FOR_T(c, P.nc) { // cone rows share the SUM of their row maxima
                const int o = P.cone_off[c], d = P.cq[c];
                double tot = 0.;
                for (int k = 0; k < d; k++) tot += gt[o + k];
                tot = sq(tot);
                for (int k = 0; k < d; k++) gt[o + k] = tot;
            }
            __syncthreads();
            FOR_T(j, n) { // rows first, then columns -- same division order as the reference (:353-356)
                const double xj = xt[j];
                for (int k = P.Ajc[j]; k < P.Ajc[j + 1]; k++) Av[k] = (Av[k] / at[P.Air[k]]) / xj;
                for (int k = P.Gjc[j]; k < P.Gjc[j + 1]; k++) Gv[k] = (Gv[k] / gt[P.Gir[k]]) / xj;
                xe[j] *= xj;
            }
            FOR_T(r, p) ae[r] *= at[r];
            FOR_T(i, m) ge[i] *= gt[i];
            __syncthreads();
        }
        FOR_T(j, n) cv[j] /= xe[j];
        FOR_T(r, p) bv[r] /= ae[r];
        FOR_T(i, m) hv[i] /= ge[i];
        // transposed value copies (reference Gt/At, :2078-2079)
        FOR_T(k, P.nnzA) Atv[k] = Av[P.At_pos[k]];
        FOR_T(k, P.nnzG) Gtv[k] = Gv[P.Gt_pos[k]];
        // static-regularisation constants read by the factor program
        if (b.tid == 0) {
            double *cst = I + P.i_cst;
            cst[0] = DELTASTAT; cst[1] = -DELTASTAT; cst[2] = 0.; cst[3] = 0.;
            ginfo->equilibrated = 1;
        }
        __syncthreads();
    }
}

// Debug: factorise instance `i` with the KKT scaling block as it stands in memory.
template <int T>
__global__ __launch_bounds__(T) void k_debug_factor(DevPat P, double *inst, double *work, int i) {
    __shared__ int flag[4];
    const int tid = threadIdx.x;
    double *I = inst + (size_t)i * P.inst_stride, *W = work;
    double *U = W + P.w_U, *Ur = W + P.w_Ur, *D = W + P.w_D, *invD = W + P.w_invD;
    (void)flag;
    for (int v = 0; v < P.nlev; v++) {
        for (int q = P.ftask_ptr[v] + tid; q < P.ftask_ptr[v + 1]; q += T) {
            const int tgt = P.ftask[q];
            double s = 0.;
            for (int k = P.tp[tgt]; k < P.tp[tgt + 1]; k++) s += U[P.pa[k]] * U[P.pb[k]] * invD[P.pk[k]];
            if (tgt < P.N) { const double d = I[P.Dsrc[tgt]] - s; D[tgt] = d; invD[tgt] = 1. / d; }
            else { const int e = tgt - P.N; const double u = I[P.Lsrc[e]] - s; U[e] = u; Ur[P.Cpos[e]] = u; }
        }
        __syncthreads();
    }
}

// ---- launchers (called from api.cpp) ----
hipError_t launch_solve(const DevPat &P, double *inst, double *work, int B, int grid, int threads, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    if (threads == 512) hipLaunchKernelGGL(k_solve<512>, dim3(grid), dim3(512), 0, st, P, inst, work, B);
    else hipLaunchKernelGGL(k_solve<256>, dim3(grid), dim3(256), 0, st, P, inst, work, B);
    return hipGetLastError();
}
hipError_t launch_update(const DevPat &P, double *inst, int first, int count, const double *Gpr, const double *Apr,
                         const double *c, const double *h, const double *b, double *scratch, int grid, hipStream_t st) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_update<256>, dim3(grid), dim3(256), 0, st, P, inst, first, count, Gpr, Apr, c, h, b, scratch);
    return hipGetLastError();
}
hipError_t launch_debug_factor(const DevPat &P, double *inst, double *work, int i, hipStream_t st) {
    hipLaunchKernelGGL(k_debug_factor<256>, dim3(1), dim3(256), 0, st, P, inst, work, i);
    return hipGetLastError();
}
hipError_t solve_occupancy(int threads, int *blocks_per_cu) {
    if (threads == 512) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_solve<512>, 512, 0);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_solve<256>, 256, 0);
}

} // namespace eicos
