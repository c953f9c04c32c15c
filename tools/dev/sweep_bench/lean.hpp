// Lean sliced-ELL triangular sweep (round 3 experiment): per-slice descriptors in SGPRs (s_load_dwordx8), unconditional
// loads with precomputed byte offsets, gather indices stored as LDS byte offsets, partial sums of long rows carried in a
// register.  See tools/dev/sweep_bench/README.md.
#pragma once
#include <hip/hip_runtime.h>

namespace lean {
#if defined(__HIP_DEVICE_COMPILE__)
#define LEAN_G __attribute__((address_space(1)))
#define LEAN_C __attribute__((address_space(4)))
#else
#define LEAN_G
#define LEAN_C
#endif
typedef int i8_t __attribute__((ext_vector_type(8)));
typedef const i8_t LEAN_C *cdesc_p;
typedef const char LEAN_G *gbytes_p;

// descriptor words: 0 idx byte offset | 1..4 value byte offsets of entries 0..3 (absent entries -> the zero region) |
// 5 row0 * 8 | 6 flags: lanes [0,12) | lg [12,15) | newlev 16 | more 17 | cont 18 | 7 unused
constexpr int LF_LG = 12, LF_NEWLEV = 16, LF_MORE = 17, LF_CONT = 18;

__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
template <int CTRL> __device__ __forceinline__ double dpp_shl_add(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
    return v + __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double grp_reduce_to_lane0(double v, int lg) { // lg is wavefront-uniform
    if (lg == 0) return v;
    if (lg >= 3) {
        if (lg >= 6) v += __shfl_xor(v, 32, 64);
        if (lg >= 5) v += __shfl_xor(v, 16, 64);
        if (lg >= 4) v = dpp_shl_add<0x108>(v);
        v = dpp_shl_add<0x104>(v);
    }
    if (lg >= 2) v = dpp_shl_add<0x102>(v);
    return dpp_shl_add<0x101>(v);
}

// ws: the sweep vector in LDS (byte offsets in the index stream are relative to it).  Q = register slots: Q - 1 slices
// in flight.  ns is a multiple of Q (host pads with empty slices).
template <int T, bool FORWARD, bool SOLO, int Q>
__device__ __forceinline__ void sweep(cdesc_p desc, int ns, gbytes_p idx_base, gbytes_p val_base, gbytes_p invd_base, double *ws) {
    if (ns == 0) { if (!SOLO) __syncthreads(); return; }
    const unsigned t = threadIdx.x, t8 = t * 8u;
    char *wsb = reinterpret_cast<char *>(ws);
    struct Slot { uint2 ix; double v[4]; double d, own; } q[Q];
    int s_row[Q], s_fl[Q]; // consume-time descriptor words (SGPRs)
    auto issue = [&](const i8_t &D, Slot &o, int &row, int &fl) {
        o.ix = *reinterpret_cast<const uint2 LEAN_G *>(idx_base + ((unsigned)D[0] + t8));
#pragma unroll
        for (int k = 0; k < 4; k++) o.v[k] = __builtin_nontemporal_load(reinterpret_cast<const double LEAN_G *>(val_base + ((unsigned)D[1 + k] + t8)));
        row = D[5]; fl = D[6];
        const unsigned r8 = (unsigned)D[5] + ((t8 >> ((fl >> LF_LG) & 7)) & ~7u);
        if constexpr (!FORWARD) o.d = *reinterpret_cast<const double LEAN_G *>(invd_base + r8);
        else o.d = 0.;
        o.own = *reinterpret_cast<double *>(wsb + r8);
    };
#pragma unroll
    for (int d = 0; d < Q - 1; d++) { const i8_t D = desc[min(d, ns - 1)]; issue(D, q[d], s_row[d], s_fl[d]); }
    i8_t dn = desc[min(Q - 1, ns - 1)];
    double carry = 0.;
    for (int s0 = 0; s0 < ns; s0 += Q) {
#pragma unroll
        for (int d = 0; d < Q; d++) {
            const int s = s0 + d;
            constexpr int nxt = 0; (void)nxt;
            const int pd = (d + Q - 1) % Q;
            const int fl = s_fl[d], row = s_row[d];
            if (fl & (1 << LF_NEWLEV)) { if constexpr (!SOLO) lds_barrier(); }
            const Slot c = q[d];
            double x0 = *reinterpret_cast<double *>(wsb + (c.ix.x & 0xffffu));
            double x1 = *reinterpret_cast<double *>(wsb + (c.ix.x >> 16));
            double x2 = *reinterpret_cast<double *>(wsb + (c.ix.y & 0xffffu));
            double x3 = *reinterpret_cast<double *>(wsb + (c.ix.y >> 16));
            __builtin_amdgcn_sched_barrier(0);
            issue(dn, q[pd], s_row[pd], s_fl[pd]); // the slice Q - 1 ahead, into the slot consumed in the previous step
            __builtin_amdgcn_sched_barrier(0);
            double a = c.v[0] * x0;
            a = a + c.v[1] * x1; a = a + c.v[2] * x2; a = a + c.v[3] * x3;
            dn = desc[min(s + Q, ns - 1)];
            const int lg = (fl >> LF_LG) & 7;
            a = grp_reduce_to_lane0(a, lg);
            if (fl & (1 << LF_CONT)) a += carry;
            if (fl & (1 << LF_MORE)) carry = a;
            else {
                const unsigned lanes = fl & 0xfff;
                if (t < lanes && (t & ((1u << lg) - 1)) == 0) {
                    double o = c.own - a;
                    if constexpr (!FORWARD) o *= c.d;
                    *reinterpret_cast<double *>(wsb + ((unsigned)row + ((t8 >> lg) & ~7u))) = o;
                }
            }
        }
    }
    if (!SOLO) __syncthreads();
}

// ---- variant 2.  What the ISA of variant 1 showed (and what this one does about it):
//  * the compiler's s_waitcnt insertion gives up at the head of a software-pipelined loop whose body has many basic blocks
//    (s_waitcnt vmcnt(0): the first step of every trip then waits for the loads issued one step earlier) -> unroll U = 3 Q
//    steps per trip, so that this happens once per 12 steps instead of once per 4;
//  * the divergent region around the store is replaced by a select of the store address (inactive lanes and the lanes of a
//    row group that do not hold the sum write a scratch area behind the vector);
//  * the descriptor of the slice after next is fetched (scalar load, same counter as LDS) at the START of a step, ahead of
//    the step's LDS traffic, into a double buffer: its latency hides behind the gathers instead of adding to them;
//  * partial sums of rows cut into sub-slices are carried branch-free (carry is 0 unless the previous slice said `more`).
template <int T, bool FORWARD, bool SOLO, int Q>
__device__ __forceinline__ void sweep2(cdesc_p desc, int ns, gbytes_p idx_base, gbytes_p val_base, gbytes_p invd_base, double *ws, unsigned dummy_off) {
    if (ns == 0) { if (!SOLO) __syncthreads(); return; }
    constexpr int U = 12; // steps per trip; ns is a multiple of U
    static_assert(U % Q == 0 && U % 2 == 0, "slot rotation and descriptor double buffer");
    const unsigned t = threadIdx.x, t8 = t * 8u;
    char *wsb = reinterpret_cast<char *>(ws);
    struct Slot { uint2 ix; double v[4]; double d, own; } q[Q];
    int s_row[Q], s_fl[Q];
    auto issue = [&](const i8_t &D, Slot &o, int &row, int &fl) {
        o.ix = *reinterpret_cast<const uint2 LEAN_G *>(idx_base + ((unsigned)D[0] + t8));
#pragma unroll
        for (int k = 0; k < 4; k++) o.v[k] = __builtin_nontemporal_load(reinterpret_cast<const double LEAN_G *>(val_base + ((unsigned)D[1 + k] + t8)));
        row = D[5]; fl = D[6];
        const unsigned r8 = (unsigned)D[5] + ((t8 >> ((fl >> LF_LG) & 7)) & ~7u);
        if constexpr (!FORWARD) o.d = *reinterpret_cast<const double LEAN_G *>(invd_base + r8);
        else o.d = 0.;
        o.own = *reinterpret_cast<double *>(wsb + r8);
    };
#pragma unroll
    for (int d = 0; d < Q - 1; d++) { const i8_t D = desc[min(d, ns - 1)]; issue(D, q[d], s_row[d], s_fl[d]); }
    i8_t dq[2]; // descriptors of the next two slices to be issued
    dq[0] = desc[min(Q - 1, ns - 1)];
    double carry = 0.;
    for (int s0 = 0; s0 < ns; s0 += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int d = u % Q, pd = (d + Q - 1) % Q, s = s0 + u;
            const int fl = s_fl[d], row = s_row[d];
            i8_t &dn = dq[u & 1];
            asm volatile("" :: "s"(dn[0]), "s"(dn[1]), "s"(dn[2]), "s"(dn[3]), "s"(dn[4]), "s"(dn[5]), "s"(dn[6])); // (it has landed before the gathers go out)
            __builtin_amdgcn_sched_barrier(0);
            if (fl & (1 << LF_NEWLEV)) { if constexpr (!SOLO) lds_barrier(); }
            dq[(u + 1) & 1] = desc[min(s + Q, ns - 1)];
            __builtin_amdgcn_sched_barrier(0);
            const Slot c = q[d];
            double x0 = *reinterpret_cast<double *>(wsb + (c.ix.x & 0xffffu));
            double x1 = *reinterpret_cast<double *>(wsb + (c.ix.x >> 16));
            double x2 = *reinterpret_cast<double *>(wsb + (c.ix.y & 0xffffu));
            double x3 = *reinterpret_cast<double *>(wsb + (c.ix.y >> 16));
            __builtin_amdgcn_sched_barrier(0);
            issue(dn, q[pd], s_row[pd], s_fl[pd]); // the slice Q - 1 ahead, into the slot consumed in the previous step
            __builtin_amdgcn_sched_barrier(0);
            double a = c.v[0] * x0;
            a = a + c.v[1] * x1; a = a + c.v[2] * x2; a = a + c.v[3] * x3;
            const int lg = (fl >> LF_LG) & 7;
            a = grp_reduce_to_lane0(a, lg);
            a = a + carry;
            const bool more = (fl & (1 << LF_MORE)) != 0;
            carry = more ? a : 0.;
            const unsigned lanes = fl & 0xfff;
            const bool wr = (t < lanes) & ((t & ((1u << lg) - 1)) == 0) & !more;
            double o = c.own - a;
            if constexpr (!FORWARD) o *= c.d;
            const unsigned addr = wr ? ((unsigned)row + ((t8 >> lg) & ~7u)) : dummy_off + t8;
            *reinterpret_cast<double *>(wsb + addr) = o;
        }
    }
    if (!SOLO) __syncthreads();
}
} // namespace lean
