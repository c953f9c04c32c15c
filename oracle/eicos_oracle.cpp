/*
 * oracle/eicos_oracle.cpp -- CPU ORACLE for the EiCOS hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see eicos_oracle.h).  Nothing under eicos_amd/ or
 * include/ may use this file.
 *
 * What it is: a from-scratch CPU restatement (C++17, std::vector only, no Eigen) of the
 * interior-point algorithm of the reference /root/reference/src/eicos.cpp.  "ref:" comments
 * give the reference lines each routine follows.  The sparse LDL' the reference delegates to
 * Eigen::SimplicialLDLT<SparseMatrix<double>,Upper> (include/eicos.hpp:221-222; Eigen >= 3.3
 * per CMakeLists.txt:6, un-vendored, absent from this image) is restated here from its
 * published algorithm: fill-reducing symmetric ordering (here: plain minimum degree instead
 * of AMD), elimination tree + column counts, up-looking numeric LDL' without pivoting that
 * accepts negative pivots and fails only on an exactly-zero pivot, solve = P' L^-T D^-1 L^-1 P.
 *
 * PARITY PINS (tests/test_oracle_golden.py):
 *   - the exit code of each of the 18 reference tests available in the mount
 *     (test/ecostester.cpp:54-72 + each header's mu_assert), incl. DINF/PINF/empty/SOC cases;
 *   - udd_optval1/2 of test/updateData/update_data.h:1654-1655 (6 digits);
 *   - independent HiGHS optima for every LP-only fixture (tests/golden/expected.json).
 *   The reference binary itself cannot be built here (Eigen absent; MPC01 blobs missing), so
 *   the LDL' boundary itself is "parity unpinned": results are pinned end-to-end only.
 */
#include "eicos_oracle.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>
#include <vector>

namespace {

using vec = std::vector<double>;
using ivec = std::vector<int>;

/* ---- settings: constants of struct Settings, ref include/eicos.hpp:23-47 ---- */
constexpr double GAMMA = 0.99, DELTASTAT = 7e-8;
constexpr double FEASTOL = 1e-8, ABSTOL = 1e-8, RELTOL = 1e-8;
constexpr double FEASTOL_INACC = 1e-4, ABSTOL_INACC = 5e-5, RELTOL_INACC = 5e-5;
constexpr int NITREF = 9, EQUIL_ITERS = 3, ITER_MAX = 100;
constexpr double LINSYSACC = 1e-14, IRERRFACT = 6, STEPMIN = 1e-6, STEPMAX = 0.999;
constexpr double SIGMAMIN = 1e-4, SIGMAMAX = 1.0, SAFEGUARD = 500;

enum { EX_OPTIMAL = 0, EX_PINF = 1, EX_DINF = 2, EX_MAXIT = -1, EX_NUMERICS = -2,
       EX_FATAL = -7, EX_INACC = 10, EX_NOT_CONVERGED = -87 };

struct Csc { /* compressed sparse column, rows x cols */
    int rows = 0, cols = 0;
    ivec ptr, idx;
    vec val;
    int nnz() const { return (int)idx.size(); }
};

/* transpose with a position map: tpos[k] = position in T of entry k of M */
Csc transpose(const Csc &M, ivec *tpos = nullptr) {
    Csc T; T.rows = M.cols; T.cols = M.rows;
    T.ptr.assign(M.rows + 1, 0); T.idx.resize(M.nnz()); T.val.resize(M.nnz());
    for (int k = 0; k < M.nnz(); k++) T.ptr[M.idx[k] + 1]++;
    for (int r = 0; r < M.rows; r++) T.ptr[r + 1] += T.ptr[r];
    ivec next(T.ptr.begin(), T.ptr.end() - 1);
    if (tpos) tpos->resize(M.nnz());
    for (int j = 0; j < M.cols; j++)
        for (int k = M.ptr[j]; k < M.ptr[j + 1]; k++) {
            int d = next[M.idx[k]]++;
            T.idx[d] = j; T.val[d] = M.val[k];
            if (tpos) (*tpos)[k] = d;
        }
    return T;
}

/* y (+)= sign * M * x */
void spmv(const Csc &M, const double *x, double *y, double sign, bool accumulate) {
    if (!accumulate) std::fill(y, y + M.rows, 0.0);
    for (int j = 0; j < M.cols; j++) {
        const double xj = sign * x[j];
        for (int k = M.ptr[j]; k < M.ptr[j + 1]; k++) y[M.idx[k]] += M.val[k] * xj;
    }
}

double norm2(const vec &v) { double s = 0; for (double a : v) s += a * a; return std::sqrt(s); }
double norminf(const double *v, int n) { double s = 0; for (int i = 0; i < n; i++) s = std::max(s, std::fabs(v[i])); return s; }
double dot(const double *a, const double *b, int n) { double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s; }

/* ---- sparse LDL' (stands in for Eigen::SimplicialLDLT<...,Upper>) ---- */
struct Ldl {
    int N = 0;
    ivec perm, iperm;      /* perm[new] = old */
    Csc C;                 /* P K P' upper triangle, CSC */
    ivec kmap;             /* K value position -> C value position */
    ivec parent, Lp, Li, Lnz, flag, pattern;
    vec Lx, D, Y;
    bool ok = false;
    /* N4 of SURVEY.md 8f (NOT in the reference, whose `delta`/`eps` settings are dead, include/eicos.hpp:26,28; off
     * by default): ECOS-style dynamic regularisation -- a pivot whose sign disagrees with the quasi-definite sign
     * pattern, or is smaller than dyn_eps in magnitude, is replaced by sign * dyn_delta.  sign[] is by ORIGINAL index. */
    double dyn_delta = 0., dyn_eps = 0.;
    ivec sign;

    /* plain minimum-degree ordering on the elimination graph (ties -> lowest index);
     * replaces Eigen's AMD (analyzePattern, ref src/eicos.cpp:897). */
    static ivec min_degree(const Csc &K) {
        const int N = K.cols;
        std::vector<ivec> adj(N);
        for (int j = 0; j < N; j++)
            for (int k = K.ptr[j]; k < K.ptr[j + 1]; k++) {
                int i = K.idx[k];
                if (i != j) { adj[i].push_back(j); adj[j].push_back(i); }
            }
        for (auto &a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
        std::vector<char> dead(N, 0);
        ivec order; order.reserve(N);
        ivec merged;
        for (int step = 0; step < N; step++) {
            int best = -1; size_t bd = 0;
            for (int v = 0; v < N; v++)
                if (!dead[v] && (best < 0 || adj[v].size() < bd)) { best = v; bd = adj[v].size(); }
            const int v = best;
            dead[v] = 1; order.push_back(v);
            const ivec nv = adj[v];
            for (int u : nv) {
                merged.clear();
                std::set_union(adj[u].begin(), adj[u].end(), nv.begin(), nv.end(), std::back_inserter(merged));
                ivec &au = adj[u]; au.clear();
                for (int w : merged) if (w != u && w != v) au.push_back(w);
            }
            adj[v].clear(); adj[v].shrink_to_fit();
        }
        return order;
    }

    void analyze(const Csc &K) {
        N = K.cols;
        perm = min_degree(K);
        iperm.assign(N, 0);
        for (int k = 0; k < N; k++) iperm[perm[k]] = k;
        /* C = upper(P K P') with value map */
        C.rows = C.cols = N; C.ptr.assign(N + 1, 0);
        const int nz = K.nnz();
        ivec ci(nz), cj(nz);
        int e = 0;
        for (int j = 0; j < N; j++)
            for (int k = K.ptr[j]; k < K.ptr[j + 1]; k++, e++) {
                int a = iperm[K.idx[k]], b = iperm[j];
                ci[e] = std::min(a, b); cj[e] = std::max(a, b);
                C.ptr[cj[e] + 1]++;
            }
        for (int j = 0; j < N; j++) C.ptr[j + 1] += C.ptr[j];
        C.idx.resize(nz); C.val.assign(nz, 0.0); kmap.resize(nz);
        ivec next(C.ptr.begin(), C.ptr.end() - 1);
        for (e = 0; e < nz; e++) { int d = next[cj[e]]++; C.idx[d] = ci[e]; kmap[e] = d; }
        /* elimination tree and column counts (up-looking symbolic) */
        parent.assign(N, -1); Lnz.assign(N, 0); flag.assign(N, -1); Lp.assign(N + 1, 0);
        for (int k = 0; k < N; k++) {
            flag[k] = k;
            for (int p = C.ptr[k]; p < C.ptr[k + 1]; p++) {
                int i = C.idx[p];
                if (i < k)
                    for (; flag[i] != k; i = parent[i]) {
                        if (parent[i] < 0) parent[i] = k;
                        Lnz[i]++; flag[i] = k;
                    }
            }
        }
        for (int k = 0; k < N; k++) Lp[k + 1] = Lp[k] + Lnz[k];
        Li.assign(Lp[N], 0); Lx.assign(Lp[N], 0.0);
        D.assign(N, 0.0); Y.assign(N, 0.0); pattern.assign(N, 0);
    }

    /* numeric up-looking LDL' (ref: ldlt.factorize(K), src/eicos.cpp:900,1164) */
    bool factorize(const vec &Kval) {
        for (size_t e = 0; e < Kval.size(); e++) C.val[kmap[e]] = Kval[e];
        ok = true;
        for (int k = 0; k < N; k++) {
            Y[k] = 0.0; int top = N; flag[k] = k; Lnz[k] = 0;
            for (int p = C.ptr[k]; p < C.ptr[k + 1]; p++) {
                int i = C.idx[p];
                Y[i] += C.val[p];
                int len = 0;
                for (; flag[i] != k; i = parent[i]) { pattern[len++] = i; flag[i] = k; }
                while (len > 0) pattern[--top] = pattern[--len];
            }
            double dk = Y[k]; Y[k] = 0.0;
            for (; top < N; top++) {
                const int i = pattern[top];
                const double yi = Y[i]; Y[i] = 0.0;
                const int p2 = Lp[i] + Lnz[i];
                for (int p = Lp[i]; p < p2; p++) Y[Li[p]] -= Lx[p] * yi;
                const double lki = yi / D[i];
                dk -= lki * yi;
                Li[p2] = k; Lx[p2] = lki; Lnz[i]++;
            }
            if (dyn_delta > 0. && !sign.empty()) { const double sg = sign[perm[k]]; if (sg * dk <= dyn_eps) dk = sg * dyn_delta; }
            D[k] = dk;
            if (dk == 0.0) { ok = false; return false; }
        }
        return true;
    }

    /* x = P' L^-T D^-1 L^-1 P b   (ref: ldlt.solve, src/eicos.cpp:1477,1599) */
    mutable vec wtmp;
    void solve(const vec &b, vec &x) const {
        if ((int)wtmp.size() != N) wtmp.assign(N, 0.0);
        vec &w = wtmp;
        for (int k = 0; k < N; k++) w[k] = b[perm[k]];
        for (int j = 0; j < N; j++) {
            const double wj = w[j];
            for (int p = Lp[j]; p < Lp[j] + Lnz[j]; p++) w[Li[p]] -= Lx[p] * wj;
        }
        for (int j = 0; j < N; j++) w[j] /= D[j];
        for (int j = N - 1; j >= 0; j--) {
            double s = w[j];
            for (int p = Lp[j]; p < Lp[j] + Lnz[j]; p++) s -= Lx[p] * w[Li[p]];
            w[j] = s;
        }
        x.resize(N);
        for (int k = 0; k < N; k++) x[perm[k]] = w[k];
    }
};

struct Cone { /* ref SOCone include/eicos.hpp:81-95 */
    int dim = 0;
    double a = 0, d1 = 0, w = 0, eta = 0, eta2 = 0, u0 = 0, u1 = 0, v1 = 0;
    vec q;
};

struct Info { /* ref Information include/eicos.hpp:49-73 */
    double pcost = 0, dcost = 0, pres = 0, dres = 0, gap = 0, relgap = 0, sigma = 0, mu = 0,
           step = 0, step_aff = 0, kapovert = 0, pinfres = 0, dinfres = 0;
    bool has_relgap = false, has_pinfres = false, has_dinfres = false, pinf = false, dinf = false;
    int iter = 0, nitref1 = 0, nitref2 = 0, nitref3 = 0;
};

/* ref Information::isBetterThan src/eicos.cpp:23-68 */
bool better_than(const Info &a, const Info &o) {
    const bool gap_ok = a.gap > 0. && o.gap > 0. && a.gap < o.gap;
    const bool mu_ok = a.mu > 0. && a.mu < o.mu;
    if (a.has_pinfres && a.kapovert > 1.) {
        if (o.has_pinfres) return gap_ok && (a.pinfres > 0. && a.pinfres < o.pres) && mu_ok;
        return gap_ok && mu_ok;
    }
    return gap_ok && (a.pres > 0. && a.pres < o.pres) && (a.dres > 0. && a.dres < o.dres) &&
           (a.kapovert > 0. && a.kapovert < o.kapovert) && mu_ok;
}

struct Work { /* ref Work include/eicos.hpp:97-114 */
    vec x, y, z, s, lambda;
    double kap = 0, tau = 0, cx = 0, by = 0, hz = 0;
    Info i;
};

struct Solver {
    int n = 0, p = 0, m = 0, l = 0, nc = 0, N = 0, mt = 0;
    std::vector<Cone> cones;
    vec lpw, lpv;
    Csc G, A, Gt, At;
    ivec Gtpos, Atpos;
    vec c, h, b, xeq, aeq, geq;
    bool equilibrated = false;
    Work w, wbest;
    vec rx, ry, rz;
    double hresx = 0, hresy = 0, hresz = 0, rt = 0, nx = 0, ny = 0, nz = 0, ns = 0;
    double resx0 = 1, resy0 = 1, resz0 = 1;
    vec dsaff_by_W, W_times_dzaff, dsaff, rhs1, rhs2;
    Csc K;
    ivec slotAG; /* K positions of the A' and G' entries, in At/Gt storage order */
    ivec slotV;  /* K positions of the scaling entries, order of ref cacheIndices :1944-1987 */
    Ldl ldl;
    vec kx, kdxref, ke, kGdx; /* solve_kkt scratch */
    int n_factor = 0, n_ldlsolve = 0, last_exit = EX_FATAL;
    std::vector<std::array<double, 12>> history; /* one row per pass of the main loop */
    bool trace = std::getenv("ORACLE_TRACE") != nullptr; /* per-iteration table like the reference's verbose mode (ref :733-753) */

    /* ---------- construction: ref build() src/eicos.cpp:132-187 ---------- */
    void build(int n_, int m_, int p_, int ncones, const int *q,
               const double *Gpr, const int *Gjc, const int *Gir,
               const double *Apr, const int *Ajc, const int *Air,
               const double *c_, const double *h_, const double *b_) {
        n = n_; m = m_; p = p_; nc = ncones;
        const bool haveG = Gpr && Gjc && Gir, haveA = Apr && Ajc && Air; /* ref :103-117 */
        if (!haveG) { m = 0; nc = 0; }
        if (!haveA) p = 0;
        if (!c_) n = 0;
        G.rows = m; G.cols = n; G.ptr.assign(n + 1, 0);
        A.rows = p; A.cols = n; A.ptr.assign(n + 1, 0);
        if (haveG) { G.ptr.assign(Gjc, Gjc + n + 1); G.idx.assign(Gir, Gir + Gjc[n]); G.val.assign(Gpr, Gpr + Gjc[n]); }
        if (haveA) { A.ptr.assign(Ajc, Ajc + n + 1); A.idx.assign(Air, Air + Ajc[n]); A.val.assign(Apr, Apr + Ajc[n]); }
        c.assign(c_ ? c_ : nullptr, c_ ? c_ + n : nullptr);
        h.assign(haveG ? h_ : nullptr, haveG ? h_ + m : nullptr);
        b.assign(haveA ? b_ : nullptr, haveA ? b_ + p : nullptr);
        int qsum = 0;
        cones.resize(nc);
        for (int i = 0; i < nc; i++) { cones[i].dim = q[i]; cones[i].q.assign(q[i] - 1, 0.0); qsum += q[i]; }
        l = m - qsum;                       /* ref :155 */
        N = n + p + m + 2 * nc; mt = m + 2 * nc; /* ref :165 */
        lpw.assign(l, 0); lpv.assign(l, 0);
        w.x.assign(n, 0); w.y.assign(p, 0); w.z.assign(m, 0); w.s.assign(m, 0); w.lambda.assign(m, 0);
        wbest = w;
        rx.assign(n, 0); ry.assign(p, 0); rz.assign(m, 0);
        dsaff_by_W.assign(m, 0); W_times_dzaff.assign(m, 0); dsaff.assign(m, 0);
        rhs1.assign(N, 0); rhs2.assign(N, 0);
        set_equilibration();
        Gt = transpose(G, &Gtpos); At = transpose(A, &Atpos);
        setup_kkt();
        ldl.analyze(K); /* hoisted: the reference redoes analyzePattern in every solve() (:897);
                           the pattern never changes so the result is identical */
    }

    /* ---------- ref setEquilibration src/eicos.cpp:302-374 (+ helpers :256-300) ---------- */
    void set_equilibration() {
        xeq.assign(n, 1.0); aeq.assign(p, 1.0); geq.assign(m, 1.0);
        vec xt(n), at(p), gt(m);
        auto sq = [](double a) { return std::fabs(a) < 1e-6 ? 1.0 : std::sqrt(a); };
        for (int it = 0; it < EQUIL_ITERS; it++) {
            std::fill(xt.begin(), xt.end(), 0.0); std::fill(at.begin(), at.end(), 0.0); std::fill(gt.begin(), gt.end(), 0.0);
            for (const Csc *M : {&A, &G})
                for (int j = 0; j < M->cols; j++)
                    for (int k = M->ptr[j]; k < M->ptr[j + 1]; k++) xt[j] = std::max(std::fabs(M->val[k]), xt[j]);
            for (int k = 0; k < A.nnz(); k++) at[A.idx[k]] = std::max(std::fabs(A.val[k]), at[A.idx[k]]);
            for (int k = 0; k < G.nnz(); k++) gt[G.idx[k]] = std::max(std::fabs(G.val[k]), gt[G.idx[k]]);
            int ind = l;
            for (const Cone &sc : cones) { /* cone rows share the SUM of their maxima, ref :338-344 */
                double tot = 0; for (int k = 0; k < sc.dim; k++) tot += gt[ind + k];
                for (int k = 0; k < sc.dim; k++) gt[ind + k] = tot;
                ind += sc.dim;
            }
            for (double &v : xt) v = sq(v);
            for (double &v : at) v = sq(v);
            for (double &v : gt) v = sq(v);
            for (int k = 0; k < A.nnz(); k++) A.val[k] /= at[A.idx[k]];
            for (int k = 0; k < G.nnz(); k++) G.val[k] /= gt[G.idx[k]];
            for (int j = 0; j < n; j++) {
                for (int k = A.ptr[j]; k < A.ptr[j + 1]; k++) A.val[k] /= xt[j];
                for (int k = G.ptr[j]; k < G.ptr[j + 1]; k++) G.val[k] /= xt[j];
            }
            for (int j = 0; j < n; j++) xeq[j] *= xt[j];
            for (int j = 0; j < p; j++) aeq[j] *= at[j];
            for (int j = 0; j < m; j++) geq[j] *= gt[j];
        }
        for (int j = 0; j < n; j++) c[j] /= xeq[j];
        for (int j = 0; j < p; j++) b[j] /= aeq[j];
        for (int j = 0; j < m; j++) h[j] /= geq[j];
        equilibrated = true;
    }

    /* ref unsetEquilibration/restore src/eicos.cpp:376-404 */
    void unset_equilibration() {
        for (int j = 0; j < n; j++) {
            for (int k = A.ptr[j]; k < A.ptr[j + 1]; k++) A.val[k] *= aeq[A.idx[k]] * xeq[j];
            for (int k = G.ptr[j]; k < G.ptr[j + 1]; k++) G.val[k] *= geq[G.idx[k]] * xeq[j];
        }
        for (int j = 0; j < n; j++) c[j] *= xeq[j];
        for (int j = 0; j < p; j++) b[j] *= aeq[j];
        for (int j = 0; j < m; j++) h[j] *= geq[j];
        equilibrated = false;
    }

    /* ---------- ref setupKKT + cacheIndices src/eicos.cpp:1734-1988 ----------
     * K (upper CSC): columns [0,n): +delta; [n,n+p): row j of A then -delta; one column per
     * G row (row of G then the -V diagonal); after each cone its v- and u-expansion columns. */
    void setup_kkt() {
        K.rows = K.cols = N; K.ptr.assign(1, 0); K.idx.clear(); K.val.clear();
        slotAG.clear(); slotV.clear();
        auto push = [&](int r, double v) { K.idx.push_back(r); K.val.push_back(v); return (int)K.idx.size() - 1; };
        auto endcol = [&]() { K.ptr.push_back((int)K.idx.size()); };
        for (int j = 0; j < n; j++) { push(j, DELTASTAT); endcol(); }
        for (int j = 0; j < p; j++) {
            for (int k = At.ptr[j]; k < At.ptr[j + 1]; k++) slotAG.push_back(push(At.idx[k], At.val[k]));
            push(n + j, -DELTASTAT); endcol();
        }
        int grow = 0, col = n + p;
        ivec diagslot(l);
        for (int i = 0; i < l; i++, grow++, col++) {
            for (int k = Gt.ptr[grow]; k < Gt.ptr[grow + 1]; k++) slotAG.push_back(push(Gt.idx[k], Gt.val[k]));
            slotV.push_back(push(col, -1.0)); endcol();
        }
        for (const Cone &sc : cones) {
            const int c0 = col;
            ivec dslot(sc.dim), vslot(sc.dim - 1), uslot(sc.dim);
            for (int i = 0; i < sc.dim; i++, grow++, col++) {
                for (int k = Gt.ptr[grow]; k < Gt.ptr[grow + 1]; k++) slotAG.push_back(push(Gt.idx[k], Gt.val[k]));
                dslot[i] = push(col, -1.0); endcol();
            }
            for (int i = 1; i < sc.dim; i++) vslot[i - 1] = push(c0 + i, 0.0);
            const int vdiag = push(col, -1.0); endcol(); col++;
            for (int i = 0; i < sc.dim; i++) uslot[i] = push(c0 + i, 0.0);
            const int udiag = push(col, 1.0); endcol(); col++;
            /* slot order of ref cacheIndices :1955-1986: D[dim], vdiag, v[dim-1], udiag, u[dim] */
            for (int s : dslot) slotV.push_back(s);
            slotV.push_back(vdiag);
            for (int s : vslot) slotV.push_back(s);
            slotV.push_back(udiag);
            for (int s : uslot) slotV.push_back(s);
        }
    }

    /* ref updateKKTAG src/eicos.cpp:1990-2030 */
    void update_kkt_ag() {
        int s = 0;
        for (int k = 0; k < At.nnz(); k++) K.val[slotAG[s++]] = At.val[k];
        for (int k = 0; k < Gt.nnz(); k++) K.val[slotAG[s++]] = Gt.val[k];
    }

    /* ref resetKKTScalings src/eicos.cpp:807-846 */
    void reset_kkt_scalings() {
        int s = 0;
        for (int k = 0; k < l; k++) K.val[slotV[s++]] = -1.0;
        for (const Cone &sc : cones) {
            for (int k = 0; k < sc.dim; k++) K.val[slotV[s++]] = -1.0;
            K.val[slotV[s++]] = -1.0;
            for (int k = 1; k < sc.dim; k++) K.val[slotV[s++]] = 0.0;
            K.val[slotV[s++]] = 1.0;
            for (int k = 0; k < sc.dim; k++) K.val[slotV[s++]] = 0.0;
        }
    }

    /* ref updateKKTScalings src/eicos.cpp:1691-1732 */
    void update_kkt_scalings() {
        int s = 0;
        for (int k = 0; k < l; k++) K.val[slotV[s++]] = -lpv[k] - DELTASTAT;
        for (const Cone &sc : cones) {
            K.val[slotV[s++]] = -sc.eta2 * sc.d1 - DELTASTAT;
            for (int k = 1; k < sc.dim; k++) K.val[slotV[s++]] = -sc.eta2 - DELTASTAT;
            K.val[slotV[s++]] = -sc.eta2;
            for (int k = 1; k < sc.dim; k++) K.val[slotV[s++]] = -sc.eta2 * sc.v1 * sc.q[k - 1];
            K.val[slotV[s++]] = sc.eta2 + DELTASTAT;
            K.val[slotV[s++]] = -sc.eta2 * sc.u0;
            for (int k = 1; k < sc.dim; k++) K.val[slotV[s++]] = -sc.eta2 * sc.u1 * sc.q[k - 1];
        }
    }

    /* ---------- ref updateData(double*...) src/eicos.cpp:2053-2082 ---------- */
    void update(const double *Gpr, const double *Apr, const double *c_, const double *h_, const double *b_) {
        if (equilibrated) unset_equilibration();
        if (Gpr) { std::copy(Gpr, Gpr + G.nnz(), G.val.begin()); std::copy(h_, h_ + m, h.begin()); }
        if (Apr) { std::copy(Apr, Apr + A.nnz(), A.val.begin()); std::copy(b_, b_ + p, b.begin()); }
        if (c_) std::copy(c_, c_ + n, c.begin());
        set_equilibration();
        for (int k = 0; k < G.nnz(); k++) Gt.val[Gtpos[k]] = G.val[k];
        for (int k = 0; k < A.nnz(); k++) At.val[Atpos[k]] = A.val[k];
        update_kkt_ag();
    }

    /* ---------- ref scale src/eicos.cpp:485-507 : lambda = W z ---------- */
    void scale(const vec &zz, vec &lam) const {
        for (int i = 0; i < l; i++) lam[i] = lpw[i] * zz[i];
        int cs = l;
        for (const Cone &sc : cones) {
            double zeta = 0; for (int k = 1; k < sc.dim; k++) zeta += sc.q[k - 1] * zz[cs + k];
            const double factor = zz[cs] + zeta / (1. + sc.a);
            lam[cs] = sc.eta * (sc.a * zz[cs] + zeta);
            for (int k = 1; k < sc.dim; k++) lam[cs + k] = sc.eta * (zz[cs + k] + factor * sc.q[k - 1]);
            cs += sc.dim;
        }
    }

    /* ---------- ref updateScalings src/eicos.cpp:411-479 ---------- */
    bool update_scalings(const vec &s, const vec &z, vec &lam) {
        for (int i = 0; i < l; i++) { lpv[i] = s[i] / z[i]; lpw[i] = std::sqrt(lpv[i]); }
        int cs = l;
        for (Cone &sc : cones) {
            double s1 = 0, z1 = 0;
            for (int k = 1; k < sc.dim; k++) { s1 += s[cs + k] * s[cs + k]; z1 += z[cs + k] * z[cs + k]; }
            const double sres = s[cs] * s[cs] - s1, zres = z[cs] * z[cs] - z1;
            if (sres <= 0 || zres <= 0) return false;
            const double snorm = std::sqrt(sres), znorm = std::sqrt(zres);
            sc.eta2 = snorm / znorm; sc.eta = std::sqrt(sc.eta2);
            double sz = 0; for (int k = 0; k < sc.dim; k++) sz += (s[cs + k] / snorm) * (z[cs + k] / znorm);
            const double gam = std::sqrt(0.5 * (1. + sz));
            const double a = (0.5 / gam) * (s[cs] / snorm + z[cs] / znorm);
            double ww = 0;
            for (int k = 1; k < sc.dim; k++) {
                sc.q[k - 1] = (0.5 / gam) * (s[cs + k] / snorm - z[cs + k] / znorm);
                ww += sc.q[k - 1] * sc.q[k - 1];
            }
            const double cc = (1. + a) + ww / (1. + a);
            const double dd = 1. + 2. / (1. + a) + ww / ((1. + a) * (1. + a));
            const double d1 = std::max(0., 0.5 * (a * a + ww * (1. - cc * cc / (1. + ww * dd))));
            const double u0sq = a * a + ww - d1;
            const double c2byu02 = cc * cc / u0sq;
            if (c2byu02 - dd <= 0) return false;
            sc.d1 = d1; sc.u0 = std::sqrt(u0sq); sc.u1 = std::sqrt(c2byu02); sc.v1 = std::sqrt(c2byu02 - dd);
            sc.a = a; sc.w = ww;
            cs += sc.dim;
        }
        scale(z, lam);
        return true;
    }

    /* ---------- ref scale2add src/eicos.cpp:1629-1662 : y += W^2 x (expanded) ---------- */
    void scale2add(const double *x, double *y) const {
        for (int i = 0; i < l; i++) y[i] += lpv[i] * x[i];
        int cs = l;
        for (const Cone &sc : cones) {
            const int i1 = cs, i2 = i1 + 1, i3 = i2 + sc.dim - 1, i4 = i3 + 1;
            y[i1] += sc.eta2 * (sc.d1 * x[i1] + sc.u0 * x[i4]);
            const double t = sc.v1 * x[i3] + sc.u1 * x[i4];
            double qtx = 0;
            for (int k = 0; k < sc.dim - 1; k++) {
                y[i2 + k] += sc.eta2 * (x[i2 + k] + t * sc.q[k]);
                qtx += sc.q[k] * x[i2 + k];
            }
            y[i3] += sc.eta2 * (sc.v1 * qtx + x[i3]);
            y[i4] = sc.eta2 * (sc.u0 * x[i1] + sc.u1 * qtx - x[i4]); /* assignment, ref :1657 */
            cs += sc.dim + 2;
        }
    }

    /* ---------- ref bringToCone src/eicos.cpp:761-805 ---------- */
    void bring_to_cone(const vec &r, vec &s) const {
        double alpha = -GAMMA;
        for (int i = 0; i < l; i++) if (r[i] <= 0 && -r[i] > alpha) alpha = -r[i];
        int cs = l;
        for (const Cone &sc : cones) {
            double t = 0; for (int k = 1; k < sc.dim; k++) t += r[cs + k] * r[cs + k];
            const double cres = r[cs] - std::sqrt(t);
            cs += sc.dim;
            if (cres <= 0 && -cres > alpha) alpha = -cres;
        }
        alpha += 1.;
        s = r;
        for (int i = 0; i < l; i++) s[i] += alpha;
        cs = l;
        for (const Cone &sc : cones) { s[cs] += alpha; cs += sc.dim; }
    }

    /* ---------- ref solveKKT src/eicos.cpp:1471-1620 ---------- */
    int solve_kkt(const vec &rhs, vec &dx, vec &dy, vec &dz, bool init) {
        vec &x = kx; ldl.solve(rhs, x); n_ldlsolve++;
        const double thr = (1. + norminf(rhs.data(), N)) * LINSYSACC;
        double nerr_prev = std::numeric_limits<double>::max();
        if ((int)kdxref.size() != N) { kdxref.assign(N, 0.0); ke.assign(N, 0.0); kGdx.assign(m, 0.0); }
        vec &dxref = kdxref, &e = ke, &Gdx = kGdx;
        std::fill(dxref.begin(), dxref.end(), 0.0);
        const double *bx = rhs.data(), *by = rhs.data() + n, *bz = rhs.data() + n + p;
        auto unpack = [&]() {
            std::copy(x.begin(), x.begin() + n, dx.begin());
            std::copy(x.begin() + n, x.begin() + n + p, dy.begin());
            std::copy(x.begin() + n + p, x.begin() + n + p + l, dz.begin());
            int di = l, xi = n + p + l;
            for (const Cone &sc : cones) {
                std::copy(x.begin() + xi, x.begin() + xi + sc.dim, dz.begin() + di);
                di += sc.dim; xi += sc.dim + 2;
            }
        };
        int k;
        for (k = 0; k <= NITREF; k++) {
            unpack();
            double *ex = e.data(), *ey = e.data() + n, *ez = e.data() + n + p;
            /* ex = bx - G'dz - A'dy - delta dx   (ref :1515-1521) */
            for (int j = 0; j < n; j++) ex[j] = bx[j];
            spmv(Gt, dz.data(), ex, -1.0, true);
            if (p > 0) spmv(At, dy.data(), ex, -1.0, true);
            for (int j = 0; j < n; j++) ex[j] -= DELTASTAT * dx[j];
            const double nex = norminf(ex, n);
            /* ey = by - A dx + delta dy   (ref :1525-1531) */
            for (int j = 0; j < p; j++) ey[j] = by[j];
            if (p > 0) spmv(A, dx.data(), ey, -1.0, true);
            for (int j = 0; j < p; j++) ey[j] += DELTASTAT * dy[j];
            const double ney = norminf(ey, p);
            /* ez = bz - G dx + delta-terms + V dz_true   (ref :1535-1567) */
            spmv(G, dx.data(), Gdx.data(), 1.0, false);
            for (int i = 0; i < l; i++) ez[i] = bz[i] - Gdx[i] + DELTASTAT * dz[i];
            int ei = l, di = l;
            for (const Cone &sc : cones) {
                for (int q2 = 0; q2 < sc.dim; q2++) ez[ei + q2] = bz[ei + q2] - Gdx[di + q2];
                for (int q2 = 0; q2 < sc.dim - 1; q2++) ez[ei + q2] += DELTASTAT * dz[di + q2];
                di += sc.dim; ei += sc.dim;
                ez[ei - 1] -= DELTASTAT * dz[di - 1];
                ez[ei++] = 0.; ez[ei++] = 0.;
            }
            const double *dzt = x.data() + n + p;
            if (init) for (int i = 0; i < mt; i++) ez[i] += dzt[i];
            else scale2add(dzt, ez);
            const double nez = norminf(ez, mt);
            double nerr = std::max(nex, nez);
            if (p > 0) nerr = std::max(nerr, ney);
            if (k > 0 && nerr > nerr_prev) { /* got worse: undo and quit (ref :1579-1585) */
                for (int i = 0; i < N; i++) x[i] -= dxref[i];
                k--; break;
            }
            if (k == NITREF || nerr < thr || (k > 0 && nerr_prev < IRERRFACT * nerr)) break;
            nerr_prev = nerr;
            ldl.solve(e, dxref); n_ldlsolve++;
            for (int i = 0; i < N; i++) x[i] += dxref[i];
        }
        unpack();
        return k;
    }

    /* ---------- ref computeResiduals src/eicos.cpp:643-689 ---------- */
    void compute_residuals() {
        spmv(Gt, w.z.data(), rx.data(), -1.0, false);
        if (p > 0) spmv(At, w.y.data(), rx.data(), -1.0, true);
        hresx = norm2(rx);
        for (int j = 0; j < n; j++) rx[j] -= w.tau * c[j];
        if (p > 0) {
            spmv(A, w.x.data(), ry.data(), 1.0, false);
            hresy = norm2(ry);
            for (int j = 0; j < p; j++) ry[j] -= w.tau * b[j];
        } else hresy = 0.;
        spmv(G, w.x.data(), rz.data(), 1.0, false);
        for (int j = 0; j < m; j++) rz[j] += w.s[j];
        hresz = norm2(rz);
        for (int j = 0; j < m; j++) rz[j] -= w.tau * h[j];
        w.cx = dot(c.data(), w.x.data(), n);
        w.by = p > 0 ? dot(b.data(), w.y.data(), p) : 0.;
        w.hz = dot(h.data(), w.z.data(), m);
        rt = w.kap + w.cx + w.by + w.hz;
        nx = norm2(w.x); ny = norm2(w.y); nz = norm2(w.z); ns = norm2(w.s);
    }

    /* ---------- ref updateStatistics src/eicos.cpp:691-754 ---------- */
    void update_statistics() {
        Info &i = w.i;
        i.gap = dot(w.s.data(), w.z.data(), m);
        i.mu = (i.gap + w.kap * w.tau) / ((l + nc) + 1);
        i.kapovert = w.kap / w.tau;
        i.pcost = w.cx / w.tau;
        i.dcost = -(w.hz + w.by) / w.tau;
        if (i.pcost < 0.) { i.relgap = i.gap / (-i.pcost); i.has_relgap = true; }
        else if (i.dcost > 0.) { i.relgap = i.gap / i.dcost; i.has_relgap = true; }
        else i.has_relgap = false;
        const double nry = p > 0 ? norm2(ry) / std::max(resy0 + nx, 1.) : 0.;
        const double nrz = norm2(rz) / std::max(resz0 + nx + ns, 1.);
        i.pres = std::max(nry, nrz) / w.tau;
        i.dres = norm2(rx) / std::max(resx0 + ny + nz, 1.) / w.tau;
        /* sticky: only ever set, never cleared (ref :720-728) */
        if ((w.hz + w.by) / std::max(ny + nz, 1.) < -RELTOL) { i.pinfres = hresx / std::max(ny + nz, 1.); i.has_pinfres = true; }
        if (w.cx / std::max(nx, 1.) < -RELTOL) {
            i.dinfres = std::max(hresy / std::max(nx, 1.), hresz / std::max(nx + ns, 1.)); i.has_dinfres = true;
        }
    }

    /* ---------- ref checkExitConditions src/eicos.cpp:526-641 ---------- */
    int check_exit(bool reduced) {
        const double feastol = reduced ? FEASTOL_INACC : FEASTOL;
        const double abstol = reduced ? ABSTOL_INACC : ABSTOL;
        const double reltol = reduced ? RELTOL_INACC : RELTOL;
        Info &i = w.i;
        /* optional<double> < double is true for nullopt (C++17) */
        const bool relgap_lt = !i.has_relgap || i.relgap < reltol;
        const bool pinfres_lt = !i.has_pinfres || i.pinfres < feastol;
        if ((-w.cx > 0. || -w.by - w.hz >= -abstol) && (i.pres < feastol && i.dres < feastol) &&
            (i.gap < abstol || relgap_lt)) {
            i.pinf = false; i.dinf = false;
            return EX_OPTIMAL + (reduced ? EX_INACC : 0);
        }
        if (i.has_dinfres && i.dinfres < feastol && w.tau < w.kap) {
            i.pinf = false; i.dinf = true;
            return EX_DINF + (reduced ? EX_INACC : 0);
        }
        if ((i.has_pinfres && i.pinfres < feastol && w.tau < w.kap) ||
            (w.tau < feastol && w.kap < feastol && pinfres_lt)) {
            i.pinf = true; i.dinf = false;
            return EX_PINF + (reduced ? EX_INACC : 0);
        }
        return EX_NOT_CONVERGED;
    }

    /* ---------- ref conicProduct :1357-1378 / conicDivision :1330-1351 ---------- */
    void conic_product(const vec &u, const vec &v, vec &o) const {
        for (int i = 0; i < l; i++) o[i] = u[i] * v[i];
        int cs = l;
        for (const Cone &sc : cones) {
            double d = 0; for (int k = 0; k < sc.dim; k++) d += u[cs + k] * v[cs + k];
            const double u0 = u[cs], v0 = v[cs];
            o[cs] = d;
            for (int k = 1; k < sc.dim; k++) o[cs + k] = u0 * v[cs + k] + v0 * u[cs + k];
            cs += sc.dim;
        }
    }
    void conic_division(const vec &u, const vec &ww, vec &v) const {
        for (int i = 0; i < l; i++) v[i] = ww[i] / u[i];
        int cs = l;
        for (const Cone &sc : cones) {
            const double u0 = u[cs], w0 = ww[cs];
            double u1sq = 0, zeta = 0;
            for (int k = 1; k < sc.dim; k++) { u1sq += u[cs + k] * u[cs + k]; zeta += u[cs + k] * ww[cs + k]; }
            const double rho = u0 * u0 - u1sq;
            const double factor = (zeta / u0 - w0) / rho;
            v[cs] = (u0 * w0 - zeta) / rho;
            for (int k = 1; k < sc.dim; k++) v[cs + k] = factor * u[cs + k] + ww[cs + k] / u0;
            cs += sc.dim;
        }
    }

    /* ---------- ref RHSaffine src/eicos.cpp:1670-1689 ---------- */
    void rhs_affine() {
        for (int j = 0; j < n; j++) rhs2[j] = rx[j];
        for (int j = 0; j < p; j++) rhs2[n + j] = -ry[j];
        for (int i = 0; i < l; i++) rhs2[n + p + i] = w.s[i] - rz[i];
        int ri = n + p + l, zi = l;
        for (const Cone &sc : cones) {
            for (int k = 0; k < sc.dim; k++) rhs2[ri + k] = w.s[zi + k] - rz[zi + k];
            zi += sc.dim; ri += sc.dim;
            rhs2[ri++] = 0.; rhs2[ri++] = 0.;
        }
    }

    /* ---------- ref RHScombined src/eicos.cpp:1282-1325 ---------- */
    void rhs_combined() {
        vec ds1(m), ds2(m);
        conic_product(w.lambda, w.lambda, ds1);
        conic_product(dsaff_by_W, W_times_dzaff, ds2);
        const double sigmamu = w.i.sigma * w.i.mu;
        for (int i = 0; i < l; i++) ds1[i] += ds2[i] - sigmamu;
        int k = l;
        for (const Cone &sc : cones) {
            ds1[k] -= sigmamu;
            for (int j = 0; j < sc.dim; j++) ds1[k + j] += ds2[k + j];
            k += sc.dim;
        }
        conic_division(w.lambda, ds1, dsaff_by_W);
        scale(dsaff_by_W, ds1);
        const double oms = 1. - w.i.sigma;
        for (int j = 0; j < n + p; j++) rhs2[j] *= oms;
        for (int i = 0; i < l; i++) rhs2[n + p + i] = -oms * rz[i] + ds1[i];
        int ri = n + p + l; k = l;
        for (const Cone &sc : cones) {
            for (int j = 0; j < sc.dim; j++) rhs2[ri + j] = -oms * rz[k + j] + ds1[k + j];
            k += sc.dim; ri += sc.dim;
            rhs2[ri++] = 0.; rhs2[ri++] = 0.;
        }
    }

    /* ---------- ref lineSearch src/eicos.cpp:1380-1469 ---------- */
    double line_search(const vec &lam, const vec &ds, const vec &dz, double tau, double dtau, double kap, double dkap) const {
        double alpha;
        if (l > 0) {
            double rhomin = ds[0] / lam[0], sigmin = dz[0] / lam[0];
            for (int i = 1; i < l; i++) { rhomin = std::min(rhomin, ds[i] / lam[i]); sigmin = std::min(sigmin, dz[i] / lam[i]); }
            const double eps = 1e-13;
            if (-sigmin > -rhomin) alpha = sigmin < 0. ? 1. / (-sigmin) : 1. / eps;
            else alpha = rhomin < 0. ? 1. / (-rhomin) : 1. / eps;
        } else alpha = 10.;
        const double mt_ = -tau / dtau, mk_ = -kap / dkap;
        if (mt_ > 0. && mt_ < alpha) alpha = mt_;
        if (mk_ > 0. && mk_ < alpha) alpha = mk_;
        int cs = l;
        for (const Cone &sc : cones) {
            double l1 = 0; for (int k = 1; k < sc.dim; k++) l1 += lam[cs + k] * lam[cs + k];
            const double lknorm2 = lam[cs] * lam[cs] - l1;
            if (lknorm2 <= 0.) continue; /* NB: cone_start is NOT advanced (ref :1423-1424) */
            const double lknorm = std::sqrt(lknorm2), inv = 1. / lknorm;
            const double lk0 = lam[cs] / lknorm;
            double ld = 0, lz = 0;
            for (int k = 1; k < sc.dim; k++) { ld += (lam[cs + k] / lknorm) * ds[cs + k]; lz += (lam[cs + k] / lknorm) * dz[cs + k]; }
            const double lds = lk0 * ds[cs] - ld, ldz = lk0 * dz[cs] - lz;
            const double rho0 = inv * lds, fr = (lds + ds[cs]) / (lk0 + 1.);
            const double sig0 = inv * ldz, fs = (ldz + dz[cs]) / (lk0 + 1.);
            double rn = 0, sn = 0;
            for (int k = 1; k < sc.dim; k++) {
                const double lb = lam[cs + k] / lknorm;
                const double r = inv * (ds[cs + k] - fr * lb), s = inv * (dz[cs + k] - fs * lb);
                rn += r * r; sn += s * s;
            }
            const double rhonorm = std::sqrt(rn) - rho0, signorm = std::sqrt(sn) - sig0;
            const double conic_step = std::max({0., signorm, rhonorm});
            if (conic_step != 0.) alpha = std::min(1. / conic_step, alpha);
            cs += sc.dim;
        }
        return std::clamp(alpha, STEPMIN, STEPMAX);
    }

    /* ref backscale src/eicos.cpp:1271-1277 */
    void backscale() {
        for (int j = 0; j < n; j++) w.x[j] /= (xeq[j] * w.tau);
        for (int j = 0; j < p; j++) w.y[j] /= (aeq[j] * w.tau);
        for (int j = 0; j < m; j++) w.z[j] /= (geq[j] * w.tau);
        for (int j = 0; j < m; j++) w.s[j] *= (geq[j] / w.tau);
    }

    /* ---------- N3 of SURVEY.md 8f (NOT in the reference, off by default): warm start ----------
     * With warm_shift > 0 and a previous OPTIMAL solve on this object, solve() skips the two initialisation solves
     * (ref :929-972) and starts from the previous (x, y, z, s), re-equilibrated, with s and z pushed into the cone:
     * LP rows s_i = max(s_i, a_s), z_i = max(z_i, a_z); SOC: head = max(head, ||tail|| + a), where
     * a_s = warm_shift * mean|s|, a_z = warm_shift * mean|z|.  tau = kap = 1. */
    double warm_shift = 0.;
    vec px, py, pz, ps; bool have_prev = false;
    void warm_init() {
        for (int j = 0; j < n; j++) w.x[j] = px[j] * xeq[j];
        for (int j = 0; j < p; j++) w.y[j] = py[j] * aeq[j];
        double sm = 0, zm = 0;
        for (int j = 0; j < m; j++) { w.z[j] = pz[j] * geq[j]; w.s[j] = ps[j] / geq[j]; sm += std::fabs(w.s[j]); zm += std::fabs(w.z[j]); }
        const double as = warm_shift * sm / std::max(1, m), az = warm_shift * zm / std::max(1, m);
        for (int j = 0; j < l; j++) { w.s[j] = std::max(w.s[j], as); w.z[j] = std::max(w.z[j], az); }
        int cs = l;
        for (const Cone &sc : cones) {
            double ts = 0, tz = 0;
            for (int k = 1; k < sc.dim; k++) { ts += w.s[cs + k] * w.s[cs + k]; tz += w.z[cs + k] * w.z[cs + k]; }
            w.s[cs] = std::max(w.s[cs], std::sqrt(ts) + as); w.z[cs] = std::max(w.z[cs], std::sqrt(tz) + az);
            cs += sc.dim;
        }
    }

    /* ---------- ref solve src/eicos.cpp:848-1262 ---------- */
    int solve() {
        int code = EX_FATAL;
        n_factor = 0; n_ldlsolve = 0; history.clear();
        reset_kkt_scalings();
        /* rhs1 = [0; b; h expanded], rhs2 = [-c; 0; 0]   (ref :865-886) */
        std::fill(rhs1.begin(), rhs1.end(), 0.0);
        for (int j = 0; j < p; j++) rhs1[n + j] = b[j];
        for (int i = 0; i < l; i++) rhs1[n + p + i] = h[i];
        { int hi = l, ri = n + p + l;
          for (const Cone &sc : cones) { for (int k = 0; k < sc.dim; k++) rhs1[ri + k] = h[hi + k]; hi += sc.dim; ri += sc.dim + 2; } }
        std::fill(rhs2.begin(), rhs2.end(), 0.0);
        for (int j = 0; j < n; j++) rhs2[j] = -c[j];
        resx0 = std::max(1., norm2(c)); resy0 = std::max(1., norm2(b)); resz0 = std::max(1., norm2(h));
        vec dx1(n), dy1(p), dz1(m), dx2(n), dy2(p), dz2(m), neg(m);
        if (warm_shift > 0. && have_prev && (last_exit == EX_OPTIMAL || last_exit == EX_OPTIMAL + EX_INACC)) {
            warm_init();
            w.i.nitref1 = 0; w.i.nitref2 = 0;
        } else {
        n_factor++;
        if (!ldl.factorize(K.val)) return last_exit = EX_FATAL; /* ref :900-905 */
        w.i.nitref1 = solve_kkt(rhs1, dx1, dy1, dz1, true);
        w.x = dx1;
        for (int i = 0; i < m; i++) neg[i] = -dz1[i];
        bring_to_cone(neg, w.s);
        w.i.nitref2 = solve_kkt(rhs2, dx2, dy2, dz2, true);
        w.y = dy2;
        bring_to_cone(dz2, w.z);
        }
        for (int j = 0; j < n; j++) rhs1[j] = -c[j];
        w.kap = 1.; w.tau = 1.;
        w.i.step = 0.; w.i.step_aff = 0.; w.i.pinf = false; w.i.dinf = false;
        double pres_prev = std::numeric_limits<double>::max();

        for (w.i.iter = 0; w.i.iter <= ITER_MAX; w.i.iter++) {
            compute_residuals();
            update_statistics();
            history.push_back({w.i.pcost, w.i.dcost, w.i.gap, w.i.pres, w.i.dres, w.i.kapovert, w.i.mu, w.i.step,
                               w.i.sigma, w.tau, w.kap, (double)w.i.nitref3});
            if (trace)
                std::printf("[oracle] it %2d pcost %+.6e dcost %+.6e gap %+.2e pres %.2e dres %.2e k/t %.2e mu %.2e step %.4f sigma %.2e tau %.3e kap %.3e\n",
                            w.i.iter, w.i.pcost, w.i.dcost, w.i.gap, w.i.pres, w.i.dres, w.i.kapovert, w.i.mu, w.i.step, w.i.sigma, w.tau, w.kap);
            /* safeguard, ref :1010-1041 */
            if (w.i.iter > 0 && (w.i.pres > SAFEGUARD * pres_prev || w.i.gap < 0.)) {
                w = wbest;
                code = check_exit(true);
                if (code == EX_NOT_CONVERGED) code = EX_NUMERICS;
                break;
            }
            pres_prev = w.i.pres;
            code = check_exit(false);
            if (code == EX_NOT_CONVERGED) {
                if (w.i.iter > 0 && w.i.step == STEPMIN * GAMMA) { /* ref :1058-1081 */
                    w = wbest;
                    code = check_exit(true);
                    if (code == EX_NOT_CONVERGED) code = EX_NUMERICS;
                    break;
                } else if (w.i.iter == ITER_MAX) { /* ref :1083-1109 */
                    if (!better_than(w.i, wbest.i)) w = wbest;
                    code = check_exit(true);
                    if (code == EX_NOT_CONVERGED) code = EX_MAXIT;
                    break;
                } else if (std::isnan(w.i.pcost)) { /* ref :1111-1137 */
                    if (!(w.i.iter == 0 || better_than(w.i, wbest.i))) {
                        w = wbest;
                        code = check_exit(true);
                        if (code == EX_NOT_CONVERGED) code = EX_NUMERICS;
                    }
                    break;
                }
            } else break;
            if (w.i.iter == 0 || better_than(w.i, wbest.i)) wbest = w; /* ref :1150-1158 */

            update_scalings(w.s, w.z, w.lambda); /* failure ignored, ref :1160 */
            update_kkt_scalings();
            n_factor++;
            if (!ldl.factorize(K.val)) return last_exit = EX_FATAL; /* no backscale, ref :1166-1170 */
            solve_kkt(rhs1, dx1, dy1, dz1, false);
            rhs_affine();
            solve_kkt(rhs2, dx2, dy2, dz2, false);
            const double dtau_denom = w.kap / w.tau - dot(c.data(), dx1.data(), n) - dot(b.data(), dy1.data(), p) - dot(h.data(), dz1.data(), m);
            const double dtauaff = (rt - w.kap + dot(c.data(), dx2.data(), n) + dot(b.data(), dy2.data(), p) + dot(h.data(), dz2.data(), m)) / dtau_denom;
            for (int i = 0; i < m; i++) dz2[i] += dtauaff * dz1[i];
            scale(dz2, W_times_dzaff);
            for (int i = 0; i < m; i++) dsaff_by_W[i] = -W_times_dzaff[i] - w.lambda[i];
            const double dkapaff = -w.kap - w.kap / w.tau * dtauaff;
            w.i.step_aff = line_search(w.lambda, dsaff_by_W, W_times_dzaff, w.tau, dtauaff, w.kap, dkapaff);
            const double sigma = std::clamp(std::pow(1. - w.i.step_aff, 3), SIGMAMIN, SIGMAMAX);
            w.i.sigma = sigma;
            rhs_combined();
            w.i.nitref3 = solve_kkt(rhs2, dx2, dy2, dz2, false);
            const double bkap = w.kap * w.tau + dkapaff * dtauaff - sigma * w.i.mu;
            const double dtau = ((1. - sigma) * rt - bkap / w.tau + dot(c.data(), dx2.data(), n) + dot(b.data(), dy2.data(), p) + dot(h.data(), dz2.data(), m)) / dtau_denom;
            for (int j = 0; j < n; j++) dx2[j] += dtau * dx1[j];
            for (int j = 0; j < p; j++) dy2[j] += dtau * dy1[j];
            for (int j = 0; j < m; j++) dz2[j] += dtau * dz1[j];
            scale(dz2, W_times_dzaff);
            for (int i = 0; i < m; i++) dsaff_by_W[i] = -(dsaff_by_W[i] + W_times_dzaff[i]);
            const double dkap = -(bkap + w.kap * dtau) / w.tau;
            w.i.step = GAMMA * line_search(w.lambda, dsaff_by_W, W_times_dzaff, w.tau, dtau, w.kap, dkap);
            scale(dsaff_by_W, dsaff);
            for (int j = 0; j < n; j++) w.x[j] += w.i.step * dx2[j];
            for (int j = 0; j < p; j++) w.y[j] += w.i.step * dy2[j];
            for (int j = 0; j < m; j++) w.z[j] += w.i.step * dz2[j];
            for (int j = 0; j < m; j++) w.s[j] += w.i.step * dsaff[j];
            w.kap += w.i.step * dkap;
            w.tau += w.i.step * dtau;
        }
        backscale();
        px = w.x; py = w.y; pz = w.z; ps = w.s; have_prev = true;
        return last_exit = code;
    }
};

} // namespace

extern "C" {

void *oracle_create(int n, int m, int p, int /*l*/, int ncones, const int *q,
                    const double *Gpr, const int *Gjc, const int *Gir,
                    const double *Apr, const int *Ajc, const int *Air,
                    const double *c, const double *h, const double *b) {
    Solver *s = new Solver();
    s->build(n, m, p, ncones, q, Gpr, Gjc, Gir, Apr, Ajc, Air, c, h, b);
    return s;
}
void oracle_update(void *s, const double *Gpr, const double *Apr, const double *c, const double *h, const double *b) {
    static_cast<Solver *>(s)->update(Gpr, Apr, c, h, b);
}
int oracle_solve(void *s) { return static_cast<Solver *>(s)->solve(); }
void oracle_get_info(void *sv, oracle_info *o) {
    Solver *s = static_cast<Solver *>(sv);
    const Info &i = s->w.i;
    o->pcost = i.pcost; o->dcost = i.dcost; o->pres = i.pres; o->dres = i.dres; o->gap = i.gap;
    o->relgap = i.relgap; o->sigma = i.sigma; o->mu = i.mu; o->step = i.step; o->step_aff = i.step_aff;
    o->kapovert = i.kapovert; o->pinfres = i.pinfres; o->dinfres = i.dinfres; o->tau = s->w.tau; o->kap = s->w.kap;
    o->has_relgap = i.has_relgap; o->has_pinfres = i.has_pinfres; o->has_dinfres = i.has_dinfres;
    o->pinf = i.pinf; o->dinf = i.dinf; o->iter = i.iter; o->nitref1 = i.nitref1; o->nitref2 = i.nitref2;
    o->nitref3 = i.nitref3; o->exitcode = s->last_exit; o->n_factor = s->n_factor; o->n_ldlsolve = s->n_ldlsolve;
}
void oracle_get_x(void *s, double *x) { Solver *S = static_cast<Solver *>(s); std::copy(S->w.x.begin(), S->w.x.end(), x); }
void oracle_get_yzs(void *s, double *y, double *z, double *sl) {
    Solver *S = static_cast<Solver *>(s);
    if (y) std::copy(S->w.y.begin(), S->w.y.end(), y);
    if (z) std::copy(S->w.z.begin(), S->w.z.end(), z);
    if (sl) std::copy(S->w.s.begin(), S->w.s.end(), sl);
}
void oracle_get_dims(void *s, int *dimK, int *nnzK, int *nnzL) {
    Solver *S = static_cast<Solver *>(s);
    if (dimK) *dimK = S->N;
    if (nnzK) *nnzK = S->K.nnz();
    if (nnzL) *nnzL = S->ldl.Lp.empty() ? 0 : S->ldl.Lp[S->N];
}
int oracle_get_trace(void *s, double *out, int max_rows) {
    Solver *S = static_cast<Solver *>(s);
    const int rows = std::min<int>(max_rows, (int)S->history.size());
    for (int r = 0; r < rows; r++) std::copy(S->history[r].begin(), S->history[r].end(), out + 12 * r);
    return (int)S->history.size();
}
void oracle_set_warm_start(void *s, double shift) { static_cast<Solver *>(s)->warm_shift = shift; }
void oracle_set_dynamic_regularization(void *s, double delta, double eps) {
    Solver *S = static_cast<Solver *>(s);
    S->ldl.dyn_delta = delta; S->ldl.dyn_eps = eps;
    /* sign pattern of the KKT matrix (ref setupKKT :1734-1890): +delta block of x, negative y / z blocks, per cone the
     * q rows and the v expansion are negative, the u expansion is positive */
    S->ldl.sign.assign(S->N, -1);
    for (int j = 0; j < S->n; j++) S->ldl.sign[j] = 1;
    int k = S->n + S->p + S->l;
    for (const Cone &sc : S->cones) { S->ldl.sign[k + sc.dim + 1] = 1; k += sc.dim + 2; }
}
void oracle_destroy(void *s) { delete static_cast<Solver *>(s); }

/* Test hook: what solve() does at ref :1160-1162 for given (s, z): updateScalings (return value ignored) followed
 * by updateKKTScalings; V_out receives the scaling block of K in the slot order of ref cacheIndices :1944-1987.
 * The cone structs keep their state between calls, as they do between iterations.  Returns updateScalings' bool. */
int oracle_debug_scalings(void *sv, const double *s_in, const double *z_in, double *V_out) {
    Solver *S = static_cast<Solver *>(sv);
    vec s(s_in, s_in + S->m), z(z_in, z_in + S->m), lam(S->m, 0.0);
    const bool ok = S->update_scalings(s, z, lam);
    S->update_kkt_scalings();
    for (size_t k = 0; k < S->slotV.size(); k++) V_out[k] = S->K.val[S->slotV[k]];
    return ok ? 1 : 0;
}
/* Test hook: the KKT matrix as it stands (upper triangle, CSC, reference column layout ref :1734-1890). */
int oracle_debug_kkt(void *sv, int *Kp, int *Ki, double *Kx) {
    Solver *S = static_cast<Solver *>(sv);
    if (Kp) std::copy(S->K.ptr.begin(), S->K.ptr.end(), Kp);
    if (Ki) std::copy(S->K.idx.begin(), S->K.idx.end(), Ki);
    if (Kx) std::copy(S->K.val.begin(), S->K.val.end(), Kx);
    return S->K.nnz();
}

double oracle_batch_solve(int n, int m, int p, int ncones, const int *q,
                          const int *Gjc, const int *Gir, const int *Ajc, const int *Air,
                          int batch, const double *Gpr, const double *Apr,
                          const double *c, const double *h, const double *b,
                          int nthreads, int *exitcodes, int *iters, double *pcost,
                          double *x_out, double *update_s, long long *total_ldlsolves) {
    const size_t nnzG = Gjc ? Gjc[n] : 0, nnzA = Ajc ? Ajc[n] : 0;
    nthreads = std::max(1, std::min(nthreads, batch));
    std::atomic<int> next(0), ready(0);
    std::atomic<bool> go(false);
    std::atomic<long long> upd_ns(0), slv_ns(0), nsolves(0);
    auto ptrs = [&](int i, const double *&g, const double *&a, const double *&ci, const double *&hi, const double *&bi) {
        g = Gpr + (size_t)i * nnzG; a = Apr + (size_t)i * nnzA;
        ci = c + (size_t)i * n; hi = h + (size_t)i * m; bi = b + (size_t)i * p;
    };
    auto worker = [&](int tid) {
        // pattern setup (constructor: ordering + symbolic analysis) is once-per-pattern work and is NOT
        // timed: every thread builds its solver on instance `tid`, then all start together and every
        // instance -- including the first ones again -- goes through updateData + solve.
        const double *g, *a, *ci, *hi, *bi;
        ptrs(tid, g, a, ci, hi, bi);
        Solver *S = new Solver();
        S->build(n, m, p, ncones, q, g, Gjc, Gir, nnzA ? a : nullptr, Ajc, Air, ci, hi, bi);
        ready++;
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= batch) break;
            ptrs(i, g, a, ci, hi, bi);
            // Every instance of a batch is its own problem = its own Solver object as far as the reference's sticky state
            // goes (pinfres / dinfres are never cleared and survive solve() calls on one object; on the GPU
            // they persist per INSTANCE).  The thread's solver object is reused for speed only, so the flags an earlier,
            // different instance left behind are dropped -- otherwise one infeasible instance turns every later instance
            // of that thread into an immediate DINF / PINF exit.
            S->w.i.has_pinfres = 0; S->w.i.has_dinfres = 0;
            auto ta = std::chrono::steady_clock::now();
            S->update(nnzG ? g : nullptr, nnzA ? a : nullptr, ci, hi, bi);
            auto tb = std::chrono::steady_clock::now();
            exitcodes[i] = S->solve();
            auto tc = std::chrono::steady_clock::now();
            iters[i] = S->w.i.iter;
            if (pcost) pcost[i] = S->w.i.pcost;
            if (x_out) std::copy(S->w.x.begin(), S->w.x.end(), x_out + (size_t)i * n);
            nsolves += S->n_ldlsolve;
            upd_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(tb - ta).count();
            slv_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(tc - tb).count();
        }
        delete S;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
    while (ready.load() < nthreads) std::this_thread::yield();
    auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto &t : th) t.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double tot = (double)(upd_ns + slv_ns);
    if (update_s) *update_s = tot > 0 ? wall * (double)upd_ns / tot : 0.;
    if (total_ldlsolves) *total_ldlsolves = nsolves;
    return tot > 0 ? wall * (double)slv_ns / tot : wall;
}

} // extern "C"
