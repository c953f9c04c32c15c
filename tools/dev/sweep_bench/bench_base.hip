// Baseline leg of the sweep micro-benchmark: the library's own tri_sweep (kernels.hip included as it stands).
#include "../../../eicos_amd/csrc/kernels.hip"
#include "bench_base.hpp"
#include <cstring>
#include <vector>
using namespace eicos;
#define CKB(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace eicos {
template <int T>
__global__ __launch_bounds__(T, (waves_per_eu<T>())) void k_bench_base(const PackedSlice *fsl, const PackedSlice *bsl, int nfs, int nfs_solo, int nbs_solo, int nbs,
        const int *fidx16, const int *bidx16, int f_d16, int b_d16, const double *UF, const double *UB, const double *invD, size_t sUF, size_t sUB, size_t sD,
        const double *rhs, double *out, int N, int Npad, int nUF, int nUB, int reps) {
    gcdbl_p uf = (gcdbl_p)(UF + (size_t)blockIdx.x * sUF), ub = (gcdbl_p)(UB + (size_t)blockIdx.x * sUB), id = (gcdbl_p)(invD + (size_t)blockIdx.x * sD);
    double *ws = g_dyn;
    PackedSlice *tab = reinterpret_cast<PackedSlice *>(g_dyn + Npad);
    { int *dst = reinterpret_cast<int *>(tab);
      const int *s1 = reinterpret_cast<const int *>(fsl), *s2 = reinterpret_cast<const int *>(bsl);
      const int c1 = (nfs + nfs_solo) * 4, c2 = (nbs + nbs_solo) * 4;
      for (int q = threadIdx.x; q < c1; q += T) dst[q] = s1[q];
      for (int q = threadIdx.x; q < c2; q += T) dst[c1 + q] = s2[q];
      __syncthreads(); }
    const PackedSlice *tf = tab, *tb = tab + nfs + nfs_solo;
    const bool wave0 = uni((int)threadIdx.x >> 6) == 0;
    for (int r = 0; r < reps; r++) {
        for (int i = threadIdx.x; i < Npad; i += T) ws[i] = i < N ? rhs[i] * (1. + 1e-3 * r) : 0.;
        __syncthreads();
        tri_sweep<T, true, true, false, true, 1>(tf, nfs, (gint_p)nullptr, (gint_p)fidx16, f_d16, uf, id, ws, nUF);
        if (wave0) {
            tri_sweep<T, true, true, true, true, 1>(tf + nfs, nfs_solo, (gint_p)nullptr, (gint_p)fidx16, f_d16, uf, id, ws, nUF);
            tri_sweep<T, false, true, true, true, 1>(tb, nbs_solo, (gint_p)nullptr, (gint_p)bidx16, b_d16, ub, id, ws, nUB);
        }
        __syncthreads();
        tri_sweep<T, false, true, false, true, 1>(tb + nbs_solo, nbs, (gint_p)nullptr, (gint_p)bidx16, b_d16, ub, id, ws, nUB);
        if (r == reps - 1) for (int i = threadIdx.x; i < N; i += T) out[(size_t)blockIdx.x * N + i] = ws[i];
        __syncthreads();
    }
}
} // namespace eicos

static std::vector<int> lane_offsets(const std::vector<SliceMeta> &sl, int &dummy) {
    std::vector<int> off16(sl.size());
    int pos = 0;
    for (size_t i = 0; i < sl.size(); i++) { off16[i] = pos; pos += sl[i].cnt << sl[i].lg; }
    dummy = pos;
    return off16;
}
static std::vector<int> pack16(const std::vector<SliceMeta> &sl, const std::vector<int> &off16, int dummy, const std::vector<int> &idx, int pad) {
    std::vector<int> words(((size_t)dummy + 1) * 2, 0);
    auto set = [&](size_t entry, int kk, int v) { words[entry * 2 + (kk >> 1)] |= v << (16 * (kk & 1)); };
    for (size_t i = 0; i < sl.size(); i++) {
        const int lanes = sl[i].cnt << sl[i].lg;
        for (int t = 0; t < lanes; t++)
            for (int kk = 0; kk < ELL_KMAX; kk++) set((size_t)off16[i] + t, kk, kk < sl[i].K ? idx[(size_t)sl[i].off + (size_t)kk * lanes + t] : pad);
    }
    for (int kk = 0; kk < ELL_KMAX; kk++) set((size_t)dummy, kk, pad);
    return words;
}
static std::vector<int> meta_ints(const std::vector<SliceMeta> &v, const std::vector<int> &off16) {
    std::vector<int> o(v.size() * 4);
    for (size_t i = 0; i < v.size(); i++) { const PackedSlice ps = pack_slice(v[i], off16[i]); std::memcpy(o.data() + 4 * i, &ps, sizeof ps); }
    return o;
}
static int *upload(const std::vector<int> &v) { int *d; CKB(hipMalloc(&d, (v.size() + 4) * 4)); CKB(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice)); return d; }

BasePlan make_base_plan(const TriPlan &pf, const TriPlan &pb, int N) {
    BasePlan b{};
    const std::vector<int> fo = lane_offsets(pf.sl, b.f_d16), bo = lane_offsets(pb.sl, b.b_d16);
    b.fidx16 = upload(pack16(pf.sl, fo, b.f_d16, pf.idx, N)); b.bidx16 = upload(pack16(pb.sl, bo, b.b_d16, pb.idx, N));
    b.fsl = upload(meta_ints(pf.sl, fo)); b.bsl = upload(meta_ints(pb.sl, bo));
    b.nfs = pf.n_wide; b.nfs_solo = pf.n_solo; b.nbs_solo = pb.n_solo; b.nbs = pb.n_wide;
    return b;
}
void free_base_plan(BasePlan &b) { hipFree(b.fsl); hipFree(b.bsl); hipFree(b.fidx16); hipFree(b.bidx16); }
size_t base_table_bytes(const TriPlan &pf, const TriPlan &pb) { return (pf.sl.size() + pb.sl.size()) * sizeof(PackedSlice); }
void launch_base(int T, int grid, const BasePlan &bp, const double *UF, const double *UB, const double *invD, size_t sUF, size_t sUB, size_t sD,
                 const double *rhs, double *out, int N, int Npad, int nUF, int nUB, int reps) {
    const size_t lds = (size_t)Npad * 8 + (size_t)(bp.nfs + bp.nfs_solo + bp.nbs + bp.nbs_solo) * sizeof(PackedSlice);
    auto go = [&](auto kern) {
        CKB(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, 0, (const PackedSlice *)bp.fsl, (const PackedSlice *)bp.bsl, bp.nfs, bp.nfs_solo, bp.nbs_solo, bp.nbs,
                           (const int *)bp.fidx16, (const int *)bp.bidx16, bp.f_d16, bp.b_d16, UF, UB, invD, sUF, sUB, sD, rhs, out, N, Npad, nUF, nUB, reps);
    };
    if (T == 256) go(k_bench_base<256>); else if (T == 512) go(k_bench_base<512>); else go(k_bench_base<128>);
    CKB(hipGetLastError());
}
