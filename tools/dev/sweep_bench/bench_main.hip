// Micro-benchmark of the level-scheduled triangular sweeps on the REAL plan of a fixture (default MPC02), round 3.
//   build: tools/dev/sweep_bench/build.sh      run (GPU box): build_exp/sweep_bench tests/golden/MPC02.epb [grid] [reps]
// Every workgroup owns one copy of the factor (UF, UB, 1/D) in HBM and solves `reps` right-hand sides:
// rhs -> LDS, forward sweep, backward sweep, result -> HBM; the result is checked against a host emulation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../../eicos_amd/csrc/device_types.hpp"
#include "../../../eicos_amd/csrc/plans.hpp"
#include "../../../eicos_amd/csrc/symbolic.hpp"
#include "bench_base.hpp"
#include "lean.hpp"

using namespace eicos;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static bool read_epb(const std::string &path, ProblemPattern &P) {
    std::ifstream f(path, std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), {});
    if (raw.size() < 36 || std::memcmp(raw.data(), "EPB1", 4)) return false;
    const int *hd = reinterpret_cast<const int *>(raw.data() + 4);
    P.n = hd[0]; P.m = hd[1]; P.p = hd[2]; P.l = hd[3]; P.nc = hd[4];
    const int nnzG = hd[5], nnzA = hd[6];
    const int *ip = hd + 8;
    auto take = [&](std::vector<int> &v, int cnt) { v.assign(ip, ip + cnt); ip += cnt; };
    take(P.q, P.nc); take(P.Gjc, P.n + 1); take(P.Gir, nnzG); take(P.Ajc, P.n + 1); take(P.Air, nnzA);
    return true;
}

extern __shared__ double l_dyn[];

struct LeanDev { lean::cdesc_p fdesc, bdesc; int nfs, nfs_solo, nbs_solo, nbs; lean::gbytes_p fidx, bidx; };

template <int T, int Q>
__global__ __launch_bounds__(T, (T == 256 ? 3 : (T == 512 ? 2 : 4))) void k_lean(LeanDev L, const double *UF, const double *UB, const double *invD, size_t sUF, size_t sUB, size_t sD,
                                                       const double *rhs, double *out, int N, int Npad, int reps) {
    const lean::gbytes_p uf = (lean::gbytes_p)(UF + (size_t)blockIdx.x * sUF), ub = (lean::gbytes_p)(UB + (size_t)blockIdx.x * sUB);
    const lean::gbytes_p id = (lean::gbytes_p)(invD + (size_t)blockIdx.x * sD);
    double *ws = l_dyn;
    const bool wave0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0;
    for (int r = 0; r < reps; r++) {
        for (int i = threadIdx.x; i < Npad; i += T) ws[i] = i < N ? rhs[i] * (1. + 1e-3 * r) : 0.;
        __syncthreads();
        lean::sweep<T, true, false, Q>(L.fdesc, L.nfs, L.fidx, uf, id, ws);
        if (wave0) {
            lean::sweep<T, true, true, Q>(L.fdesc + L.nfs, L.nfs_solo, L.fidx, uf, id, ws);
            lean::sweep<T, false, true, Q>(L.bdesc, L.nbs_solo, L.bidx, ub, id, ws);
        }
        __syncthreads();
        lean::sweep<T, false, false, Q>(L.bdesc + L.nbs_solo, L.nbs, L.bidx, ub, id, ws);
        if (r == reps - 1) for (int i = threadIdx.x; i < N; i += T) out[(size_t)blockIdx.x * N + i] = ws[i];
        __syncthreads();
    }
}

template <int T, int Q>
__global__ __launch_bounds__(T, (T == 256 ? 3 : (T == 512 ? 2 : 4))) void k_lean2(LeanDev L, const double *UF, const double *UB, const double *invD, size_t sUF, size_t sUB, size_t sD,
                                                        const double *rhs, double *out, int N, int Npad, int reps) {
    const lean::gbytes_p uf = (lean::gbytes_p)(UF + (size_t)blockIdx.x * sUF), ub = (lean::gbytes_p)(UB + (size_t)blockIdx.x * sUB);
    const lean::gbytes_p id = (lean::gbytes_p)(invD + (size_t)blockIdx.x * sD);
    double *ws = l_dyn;
    const unsigned dummy = (unsigned)Npad * 8u; // T doubles of scratch behind the vector
    const bool wave0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0;
    for (int r = 0; r < reps; r++) {
        for (int i = threadIdx.x; i < Npad; i += T) ws[i] = i < N ? rhs[i] * (1. + 1e-3 * r) : 0.;
        __syncthreads();
        lean::sweep2<T, true, false, Q>(L.fdesc, L.nfs, L.fidx, uf, id, ws, dummy);
        if (wave0) {
            lean::sweep2<T, true, true, Q>(L.fdesc + L.nfs, L.nfs_solo, L.fidx, uf, id, ws, dummy);
            lean::sweep2<T, false, true, Q>(L.bdesc, L.nbs_solo, L.bidx, ub, id, ws, dummy);
        }
        __syncthreads();
        lean::sweep2<T, false, false, Q>(L.bdesc + L.nbs_solo, L.nbs, L.bidx, ub, id, ws, dummy);
        if (r == reps - 1) for (int i = threadIdx.x; i < N; i += T) out[(size_t)blockIdx.x * N + i] = ws[i];
        __syncthreads();
    }
}

int main(int argc, char **argv) {
    const std::string path = argc > 1 ? argv[1] : "tests/golden/MPC02.epb";
    const int grid = argc > 2 ? atoi(argv[2]) : 512, reps = argc > 3 ? atoi(argv[3]) : 200;
    const int T = argc > 4 ? atoi(argv[4]) : 256;
    setvbuf(stdout, nullptr, _IONBF, 0);
    ProblemPattern P;
    if (!read_epb(path, P)) { fprintf(stderr, "cannot read %s\n", path.c_str()); return 2; }
    Symbolic S = analyze(P, -1, 0);
    const int N = S.N, Npad = (N + 1 + 15) & ~15;
    TriPlan pf = build_tri_plan(S, T, true), pb = build_tri_plan(S, T, false);
    printf("%s: N %d nnzL %d levels %d | T %d: forward %d wide + %d solo slices (%d slots), backward %d solo + %d wide (%d slots)\n", path.c_str(), N, S.nnzL, S.nlev,
           T, pf.n_wide, pf.n_solo, pf.slots, pb.n_solo, pb.n_wide, pb.slots);
    // ---- values: a well-conditioned random unit-lower L (forward slot order), U = L D (backward slot order), 1/D ----
    unsigned long long st = 12345;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return ((st >> 11) & 0xFFFFFFFFFFFFull) / (double)(1ull << 48); };
    std::vector<double> Lv(S.nnzL), D(N), invD(N + 8 + 512, 0.0), rhs(N);
    std::vector<int> rowlen(N, 0);
    for (int j = 0; j < N; j++) for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) rowlen[S.Li[e]]++;
    for (int j = 0; j < N; j++) for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) Lv[e] = (rnd() - 0.5) / std::max(1, rowlen[S.Li[e]]);
    for (int j = 0; j < N; j++) { D[j] = (rnd() < 0.5 ? -1. : 1.) * (0.5 + 1.5 * rnd()); invD[j] = 1. / D[j]; rhs[j] = rnd() - 0.5; }
    const size_t padF = (size_t)pf.slots + 2 * T + 8, padB = (size_t)pb.slots + 2 * T + 8; // [slots, slots+T): overrun of inactive lanes; [slots+T, slots+2T): zero region
    std::vector<double> UF(padF, 0.0), UB(padB, 0.0);
    for (int j = 0; j < N; j++) for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) { UF[pf.pos[e]] = Lv[e]; UB[pb.pos[e]] = Lv[e] * D[j]; }
    // ---- host emulation of the two sweeps over the plans (same arithmetic order per row as the kernels) ----
    std::vector<double> ref(N + 1, 0.0);
    for (int i = 0; i < N; i++) ref[i] = rhs[i] * (1. + 1e-3 * (reps - 1));
    auto hsweep = [&](const TriPlan &pl, const std::vector<double> &val, bool fwd) {
        std::vector<double> carry;
        for (const SliceMeta &m : pl.sl) {
            const int g = 1 << m.lg, lanes = m.cnt * g;
            std::vector<double> acc(m.cnt, 0.0);
            for (int r = 0; r < m.cnt; r++) {
                // lane partial sums (k ascending), then the fold ladder
                std::vector<double> part(g, 0.0);
                for (int q = 0; q < g; q++) { double a = 0; for (int kk = 0; kk < m.K; kk++) { const int slot = m.off + kk * lanes + r * g + q; a = (kk == 0) ? val[slot] * ref[pl.idx[slot]] : a + val[slot] * ref[pl.idx[slot]]; } part[q] = a; }
                for (int w = g / 2; w >= 1; w /= 2) for (int q = 0; q < w; q++) part[q] += part[q + w];
                acc[r] = part[0];
            }
            if (!m.cont) carry.assign(m.cnt, 0.0);
            for (int r = 0; r < m.cnt; r++) {
                double a = acc[r];
                if (m.cont) a += carry[r];
                if (m.more) { carry[r] = a; continue; }
                const int i = m.row0 + r;
                ref[i] = fwd ? ref[i] - a : (ref[i] - a) * invD[i];
            }
        }
    };
    hsweep(pf, UF, true); hsweep(pb, UB, false);

    // ---- device buffers: one factor copy per workgroup ----
    double *dUF, *dUB, *dD, *drhs, *dout;
    CK(hipMalloc(&dUF, padF * sizeof(double) * grid)); CK(hipMalloc(&dUB, padB * sizeof(double) * grid)); CK(hipMalloc(&dD, (size_t)(N + 8 + 512) * sizeof(double) * grid));
    for (int g = 0; g < grid; g++) {
        CK(hipMemcpy(dUF + (size_t)g * padF, UF.data(), padF * sizeof(double), hipMemcpyHostToDevice));
        CK(hipMemcpy(dUB + (size_t)g * padB, UB.data(), padB * sizeof(double), hipMemcpyHostToDevice));
        CK(hipMemcpy(dD + (size_t)g * (N + 8 + 512), invD.data(), (size_t)(N + 8 + 512) * sizeof(double), hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&drhs, N * sizeof(double))); CK(hipMemcpy(drhs, rhs.data(), N * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, (size_t)grid * N * sizeof(double)));
    auto check = [&](const char *tag, float ms) {
        std::vector<double> o((size_t)grid * N);
        CK(hipMemcpy(o.data(), dout, o.size() * sizeof(double), hipMemcpyDeviceToHost));
        double err = 0, nrm = 0;
        for (int g : {0, grid - 1}) for (int i = 0; i < N; i++) { err = std::max(err, std::fabs(o[(size_t)g * N + i] - ref[i])); nrm = std::max(nrm, std::fabs(ref[i])); }
        const double bytes = (double)grid * reps * 8.0 * (2.0 * S.nnzL + 3.0 * N);
        printf("%-28s %8.3f ms  %7.2f us per solve per workgroup  alg %.2f TB/s  max err %.2e (|x| %.2e)\n", tag, ms, ms * 1e3 / reps, bytes / (ms * 1e-3) / 1e12, err, nrm);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ---- baseline: the library's tri_sweep ----
    {
        BasePlan bp = make_base_plan(pf, pb, N);
        for (int it = 0; it < 2; it++) {
            CK(hipMemset(dout, 0, (size_t)grid * N * sizeof(double)));
            CK(hipEventRecord(e0));
            launch_base(T, grid, bp, dUF, dUB, dD, padF, padB, (size_t)(N + 8 + 512), drhs, dout, N, Npad, pf.slots, pb.slots, reps);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) check("baseline tri_sweep", ms);
        }
        free_base_plan(bp);
    }
    // ---- lean: descriptors + byte-offset indices ----
    auto run_lean = [&](int Q, int ver) {
        auto build = [&](const TriPlan &pl, std::vector<int> &desc, std::vector<unsigned short> &idx) {
            // pad the slice list to a multiple of Q per part is already done for TRI_DEPTH; re-pad to Q
            std::vector<SliceMeta> sl = pl.sl;
            (void)sl;
            desc.clear(); idx.clear();
            int pos = 0;
            for (const SliceMeta &m : pl.sl) {
                const int lanes = m.cnt << m.lg;
                int d[8] = {pos * 8, 0, 0, 0, 0, m.row0 * 8, lanes | (m.lg << lean::LF_LG) | ((m.newlev & 1) << lean::LF_NEWLEV) | (m.more << lean::LF_MORE) | (m.cont << lean::LF_CONT), 0};
                for (int k = 0; k < 4; k++) d[1 + k] = (k < m.K ? m.off + k * lanes : pl.slots + T) * 8;
                desc.insert(desc.end(), d, d + 8);
                for (int t = 0; t < lanes; t++) for (int k = 0; k < 4; k++) idx.push_back((unsigned short)((k < m.K ? pl.idx[(size_t)m.off + (size_t)k * lanes + t] : N) * 8));
                pos += lanes;
            }
            for (int t = 0; t < T * 4; t++) idx.push_back((unsigned short)(N * 8)); // overrun of inactive lanes
        };
        std::vector<int> fd, bd; std::vector<unsigned short> fi, bi;
        build(pf, fd, fi); build(pb, bd, bi);
        // parts must be multiples of Q: wide / solo parts are padded to TRI_DEPTH by the plan builder; pad each part here
        auto pad_part = [&](std::vector<int> &desc, int &n_a, int &n_b, const TriPlan &pl) {
            // split at n_a, pad both to multiples of Q with empty slices (lanes 0, all offsets -> zero region)
            std::vector<int> a(desc.begin(), desc.begin() + (size_t)n_a * 8), b(desc.begin() + (size_t)n_a * 8, desc.begin() + (size_t)(n_a + n_b) * 8);
            int e[8] = {0, (pl.slots + T) * 8, (pl.slots + T) * 8, (pl.slots + T) * 8, (pl.slots + T) * 8, 0, 0, 0};
            const int mult = ver == 2 ? 12 : Q;
            while ((a.size() / 8) % mult) a.insert(a.end(), e, e + 8);
            while ((b.size() / 8) % mult) b.insert(b.end(), e, e + 8);
            n_a = (int)a.size() / 8; n_b = (int)b.size() / 8;
            desc = a; desc.insert(desc.end(), b.begin(), b.end());
        };
        int nfs = pf.n_wide, nfs_solo = pf.n_solo, nbs_solo = pb.n_solo, nbs = pb.n_wide;
        pad_part(fd, nfs, nfs_solo, pf); pad_part(bd, nbs_solo, nbs, pb);
        if ((size_t)Npad * 8 > 65535) { printf("lean: N too large for 16-bit byte offsets\n"); return; }
        int *dfd, *dbd; unsigned short *dfi, *dbi;
        CK(hipMalloc(&dfd, fd.size() * 4)); CK(hipMalloc(&dbd, bd.size() * 4)); CK(hipMalloc(&dfi, fi.size() * 2)); CK(hipMalloc(&dbi, bi.size() * 2));
        CK(hipMemcpy(dfd, fd.data(), fd.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbd, bd.data(), bd.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dfi, fi.data(), fi.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dbi, bi.data(), bi.size() * 2, hipMemcpyHostToDevice));
        LeanDev L{(lean::cdesc_p)(unsigned long long)dfd, (lean::cdesc_p)(unsigned long long)dbd, nfs, nfs_solo, nbs_solo, nbs, (lean::gbytes_p)dfi, (lean::gbytes_p)dbi};
        const size_t lds = (size_t)Npad * 8 + std::max(base_table_bytes(pf, pb), (size_t)T * 8); // same LDS footprint as the baseline (occupancy); scratch for the masked-off stores
        for (int it = 0; it < 2; it++) {
            CK(hipMemset(dout, 0, (size_t)grid * N * sizeof(double)));
            CK(hipEventRecord(e0));
            auto go = [&](auto kern) {
                CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, 0, L, dUF, dUB, dD, padF, padB, (size_t)(N + 8 + 512), drhs, dout, N, Npad, reps);
            };
            if (ver == 1) {
            if (T == 256) { if (Q == 3) go(k_lean<256, 3>); else if (Q == 4) go(k_lean<256, 4>); else go(k_lean<256, 5>); }
            else if (T == 512) { if (Q == 3) go(k_lean<512, 3>); else if (Q == 4) go(k_lean<512, 4>); else go(k_lean<512, 5>); }
            else { if (Q == 3) go(k_lean<128, 3>); else if (Q == 4) go(k_lean<128, 4>); else go(k_lean<128, 5>); }
            } else {
            if (T == 256) { if (Q == 3) go(k_lean2<256, 3>); else if (Q == 4) go(k_lean2<256, 4>); else go(k_lean2<256, 6>); }
            else if (T == 512) { if (Q == 3) go(k_lean2<512, 3>); else if (Q == 4) go(k_lean2<512, 4>); else go(k_lean2<512, 6>); }
            else { if (Q == 3) go(k_lean2<128, 3>); else if (Q == 4) go(k_lean2<128, 4>); else go(k_lean2<128, 6>); }
            }
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            char tag[64]; snprintf(tag, sizeof tag, "lean v%d Q=%d", ver, Q);
            if (it) check(tag, ms);
        }
        hipFree(dfd); hipFree(dbd); hipFree(dfi); hipFree(dbi);
    };
    for (int Q : {4}) run_lean(Q, 1);
    for (int Q : {3, 4, 6}) run_lean(Q, 2);
    return 0;
}
