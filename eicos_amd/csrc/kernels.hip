// HIP kernels (gfx950 / CDNA4) of the batched SOCP interior-point solver.
//
// Execution model: ONE WORKGROUP PER PROBLEM INSTANCE, persistent over the whole
// interior-point solve.  The grid is sized to what is co-resident on the chip
// (256 CUs x 1-3 workgroups); each workgroup solves instance blockIdx.x and then pulls further
// instances from a queue, longest first by the work of their previous solve (k_order).
// Every step of the reference's solve() (reference src/eicos.cpp:848-1262) is executed by
// all threads of the workgroup with workgroup-uniform control flow.  The scalar state
// (struct Information, tau/kappa, step lengths, exit decisions) lives in LDS and is advanced
// by thread 0 between barriers, so the vector code keeps few registers live and several
// workgroups share a CU to hide the latency of the dependent level-by-level sparse solves.
// Per-instance control flow (exit tests, refinement counts, safeguards) therefore needs no
// inter-workgroup communication and no active-mask compaction.
//
// Data: per-instance values live in slabs in HBM; the index arrays are shared by all instances.
// Every sparse operation -- the three matrix-vector products, the numeric LDL' program and the two
// level-scheduled triangular sweeps -- walks a "sliced ELL" plan (SliceMeta, device_types.hpp): g lanes
// per row, unit-stride index/value loads with no row pointers, DPP row reductions, and the loads of
// the next slices are issued before the current one is consumed (register queue; across the LDS-only
// level barriers of the sweeps).  KKT-space vectors live in elimination order, the solve vector (and
// the current solution when one workgroup owns a CU) in LDS.  The narrow top of the elimination tree is
// swept by a single wavefront without workgroup barriers; elementwise passes issue the loads of several
// iterations before the first use (for_t_pre) because the compiler will not.
//
// Code shape: the solve is a small state machine so that the three big pieces -- numeric
// factorisation, LDL' solve, KKT solve with iterative refinement -- each have exactly ONE call site,
// and every stage is a non-inlined function (the straight-line form of the reference calls solveKKT
// at five places; inlining that blows the instruction cache and the register budget).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "device_types.hpp"
#include "launch.hpp"

// This file is ONE source compiled FIVE times (kernels.hip, kernels_t128.hip, kernels_t512.hip, kernels_ldsres.hip, kernels_w2.hip: translation
// units that build side by side; launch.hpp: solve_build() picks a handle's).  EICOS_LDSRES = 0: the slabs of an instance live in HBM.  EICOS_LDSRES = 1
// (kernels_ldsres.o, namespace eicos::ldsres): the LDS-RESIDENT variant for small patterns -- k_solve copies the
// instance slab and the workspace slab into LDS, solves there and copies both back, so every "slab" pointer below is an
// LDS pointer (address space 3, ds_read / ds_write) and a dependent step of the sparse programs costs an LDS round trip
// instead of an L2 one.  The shared pattern tables stay global in every build.
#ifndef EICOS_LDSRES
#define EICOS_LDSRES 0
#endif
// A third compilation (kernels_w2.o, namespace eicos::w2, EICOS_W2 = 1): the 256-thread solve kernel with the register budget of TWO
// waves per SIMD (256 VGPRs) for launches that run at most two workgroups per CU -- the default 256-thread build is held to 168
// VGPRs so that three fit (batches >= 1536).  Same box: +3.4 % at batch 1024, +7.6 % on lp_adlittle (profiles/r03_log_wpe.log).
#ifndef EICOS_W2
#define EICOS_W2 0
#endif
// Two more compilations split the default build by workgroup size, so that the five translation units compile side by side (a clean
// build is bounded by the slowest one instead of by the sum of eighteen k_solve instantiations): EICOS_TSPLIT = 128 (kernels_t128.o,
// namespace eicos::t128) and 512 (kernels_t512.o, namespace eicos::t512) hold the 128- / 512-thread solve kernels at their default
// register budgets; kernels.o keeps the 256-thread one (168 VGPRs), updateData and the debug kernels.
#ifndef EICOS_TSPLIT
#define EICOS_TSPLIT 0
#endif
// Two more (kernels_ubl256.o / kernels_ubl512.o, namespaces eicos::ubl256 / ubl512, EICOS_UBL = the workgroup size): the 256- / 512-thread solve
// kernel for launches of ONE workgroup per CU (batch <= CUs: BASELINE configs[3]) whose factor operand array U = L.*D (DevPat::w_UB) fits the
// LDS that a lone workgroup leaves idle.  There the numeric factorisation is a chain of dependent levels whose operand gathers and
// stores each cost an L2 round trip (lp_bandm: 73 levels, 1.7 us per level); with U in LDS (`ubdbl_p` below is an LDS pointer in this
// build; DevPat::ub_lds = its offset in the dynamic LDS) a level costs LDS round trips, and the backward sweep streams U from LDS as well.
// Same program, same order of operations: results are bit-identical to the HBM-slab kernels.  256 VGPRs (two waves per SIMD).
#ifndef EICOS_UBL
#define EICOS_UBL 0
#endif
#define EICOS_MAIN_BUILD (!EICOS_LDSRES && !EICOS_W2 && !EICOS_TSPLIT && !EICOS_UBL)

namespace eicos {
#if EICOS_LDSRES
namespace ldsres {
#if defined(__HIP_DEVICE_COMPILE__)
#define EICOS_DATA __attribute__((address_space(3)))
#else
#define EICOS_DATA
#endif
typedef double EICOS_DATA *gdbl_p;        // (shadow the global-memory typedefs of device_types.hpp)
typedef const double EICOS_DATA *gcdbl_p;
#elif EICOS_W2
namespace w2 {
#define EICOS_DATA EICOS_GLOBAL
#elif EICOS_UBL == 256
namespace ubl256 {
#define EICOS_DATA EICOS_GLOBAL
#elif EICOS_UBL == 512
namespace ubl512 {
#define EICOS_DATA EICOS_GLOBAL
#elif EICOS_TSPLIT == 128
namespace t128 {
#define EICOS_DATA EICOS_GLOBAL
#elif EICOS_TSPLIT == 512
namespace t512 {
#define EICOS_DATA EICOS_GLOBAL
#else
#define EICOS_DATA EICOS_GLOBAL
#endif

// ---- constants of struct Settings (reference include/eicos.hpp:23-47) ----
__device__ constexpr double GAMMA = 0.99, DELTASTAT = 7e-8;
__device__ constexpr double FEASTOL = 1e-8, ABSTOL = 1e-8, RELTOL = 1e-8;
__device__ constexpr double FEASTOL_INACC = 1e-4, ABSTOL_INACC = 5e-5, RELTOL_INACC = 5e-5;
__device__ constexpr int NITREF = 9, EQUIL_ITERS = 3, ITER_MAX = 100;
__device__ constexpr double LINSYSACC = 1e-14, IRERRFACT = 6., STEPMIN = 1e-6, STEPMAX = 0.999;
__device__ constexpr double SIGMAMIN = 1e-4, SIGMAMAX = 1.0, SAFEGUARD = 500.;
constexpr int EX_NOT_CONVERGED = -87;

// Register budget per workgroup size = waves per SIMD the kernel is compiled for: 512 threads -> 2 (256 VGPRs, one
// workgroup per CU -- what large patterns and the tile path get anyway; the tile factorisation keeps three operations
// of two tiles in flight per wavefront and spills at 128), 256 threads -> 3 (168 VGPRs, three workgroups per CU: three
// 48 KB solve vectors are what the 160 KB of LDS hold), 128 threads -> 4 (small patterns, up to eight workgroups per CU).  The AMDGPU
// attributor propagates the kernel's budget to the non-inlined stage functions.
#if EICOS_LDSRES
template <int T> constexpr int waves_per_eu() { return 2; } // LDS allows at most three small workgroups per CU
#elif EICOS_W2 || EICOS_UBL
template <int T> constexpr int waves_per_eu() { return 2; }
#else
template <int T> constexpr int waves_per_eu() { return T == 256 ? 3 : (T == 512 ? 2 : 4); }
#endif

// Multiply-accumulate of the sparse inner loops (products, sweeps, factor program).  The file is compiled with -ffp-contract=off and the
// product is NOT fused: the arithmetic then rounds like the CPU oracle's.  (One v_fma_f64 instead of v_mul_f64 + v_add_f64 was measured
// in round 2: +0.2 ... +0.7 % on every config -- the loops wait for their operands, not for issue slots.)
__device__ __forceinline__ double madd(double acc, double a, double b) { return acc + a * b; }
constexpr int RED_SLOTS = 8 * 8; // up to 8 wavefronts (512 threads) x 8 values per reduction
constexpr bool ELL_DQ = EICOS_W2 != 0 || EICOS_UBL != 0; // product loops keep decoded slice descriptors in their queue (ell_dots, ell_dots_k)

// scalar slots in LDS (written by thread 0 only)
enum { SV_RESX0 = 0, SV_RESY0, SV_RESZ0, SV_PRESPREV, SV_RT, SV_DTAUDEN, SV_DTAUAFF, SV_DKAPAFF, SV_BKAP,
       SV_DTAU, SV_DKAP, SV_ALPHA, SV_COUNT };
enum { FL_FATAL = 0, FL_ACTION, FL_RESTORE, FL_SAVE, FL_CODE, FL_WARM, FL_COUNT };
enum { ACT_CONTINUE = 0, ACT_BREAK = 1 };

// Pattern descriptors live in the constant address space: field loads are scalar (s_load) and
// need no generic pointer.  One slot per live handle (eicos_batch), written at creation.
constexpr int MAX_PATTERNS = 64;
__constant__ DevPat c_pat[MAX_PATTERNS];

// Per-instance scalar state of the instance the workgroup is solving (LDS, advanced by thread 0 between barriers).
struct ShI {
    DevInfo wi, bi;
    double sv[SV_COUNT];
    int fl[FL_COUNT];
    // The iterate (x, y, z, s) has two buffer sets: 0 = the instance slab (where the host reads the result), 1 = the workspace slab.
    // `cur` holds the current iterate, `best` the reference's best iterate (-1: none yet).  Saving the best iterate is `best = cur`;
    // the next update then writes the new iterate into the OTHER set instead of in place -- no copy of the iterate per pass.
    int cur, best;
    double dots[2][3]; // c'dx, b'dy, h'dz of the last solve with right-hand side 1 / 2 (kkt_solve's epilogue; kkt_post's d tau needs them)
    int kref, kref2, done; // refinement steps of the last KKT solve (kref2: of the second right-hand side of a dual solve); 1 = this instance has finished
    unsigned long long tick[12]; // per-phase time of the current solve (100 MHz ticks), thread 0; [7] = start
};
struct alignas(16) Sh : ShI {
    double red[2 * RED_SLOTS];
    double dyn_delta, dyn_eps; // dynamic regularisation of the pivots (extension; 0 = off), set by k_solve
    int next; // next instance of this workgroup (k_solve's queue)
};
constexpr int KI_MAX = 2; // right-hand sides of a dual solve (kkt_solve<..., 2, true>)
enum { TK_FACTOR = 0, TK_LDL, TK_KRES, TK_KPOST, TK_RESID, TK_FWD, TK_COUNT, TK_FA = 8, TK_FW1, TK_FB, TK_FW2 }; // 8..11: inside the factor
#define TICK_BEGIN unsigned long long tk0_ = (threadIdx.x == 0) ? wall_clock64() : 0ull
#define TICK_END(slot) do { if (threadIdx.x == 0) { const unsigned long long t1_ = wall_clock64(); g_S.tick[slot] += t1_ - tk0_; tk0_ = t1_; } } while (0)
static_assert(sizeof(Sh) <= 4096, "api.cpp budgets 4 KB of static LDS per workgroup");
__shared__ __attribute__((aligned(16))) Sh g_S;
struct IterBuf { gdbl_p x, y, z, s; };
// (16-byte aligned: the dual right-hand-side vectors are read and written as 16-byte LDS accesses)
extern __shared__ __attribute__((aligned(16))) double g_dyn[]; // E[Npad] (NLDS>=1) | X[Npad] (NLDS>=2) | slice tables of both sweeps

// Arguments of non-inlined device functions arrive in VGPRs; the values below are workgroup-uniform,
// so move them to SGPRs: address math and branches on them become scalar (s_load, s_cbranch) instead
// of per-lane loads and exec-masked regions.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Element `i` of a workgroup-uniform global array through a 32-bit byte offset: the address is SGPR base + VGPR
// offset (global_load ... v_off, s[base:base+1]), one shift instead of a sign extension + 64-bit shift-add per
// load.  Every array addressed this way is smaller than 4 GB.
template <class E> __device__ __forceinline__ E ld_u32(const E EICOS_GLOBAL *base, int i) {
    return *reinterpret_cast<const E EICOS_GLOBAL *>(reinterpret_cast<const char EICOS_GLOBAL *>(base) + (unsigned)i * (unsigned)sizeof(E));
}
template <class E> __device__ __forceinline__ E ld_u32_nt(const E EICOS_GLOBAL *base, int i) {
    return __builtin_nontemporal_load(reinterpret_cast<const E EICOS_GLOBAL *>(reinterpret_cast<const char EICOS_GLOBAL *>(base) + (unsigned)i * (unsigned)sizeof(E)));
}
template <class Ptr> __device__ __forceinline__ Ptr uni_ptr(Ptr p) {
    if constexpr (sizeof(Ptr) == 4) return (Ptr)__builtin_amdgcn_readfirstlane((unsigned)p); // LDS pointer
    else {
        const unsigned long long a = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return (Ptr)(((unsigned long long)hi << 32) | lo);
    }
}
typedef double d2_t __attribute__((ext_vector_type(2)));
#if EICOS_LDSRES // slab arrays are LDS arrays: plain ds_read
__device__ __forceinline__ double ld_u32(const double EICOS_DATA *base, int i) { return base[i]; }
__device__ __forceinline__ double ld_u32_nt(const double EICOS_DATA *base, int i) { return base[i]; }
__device__ __forceinline__ d2_t ld_u32(const d2_t EICOS_DATA *base, int i) { return base[i]; }
__device__ __forceinline__ d2_t ld_u32_nt(const d2_t EICOS_DATA *base, int i) { return base[i]; }
#endif
// the factor operand array U (DevPat::w_UB): in the workspace slab, or (EICOS_UBL builds) in the dynamic LDS at DevPat::ub_lds
#if EICOS_UBL && defined(__HIP_DEVICE_COMPILE__)
typedef double __attribute__((address_space(3))) *ubdbl_p;
typedef const double __attribute__((address_space(3))) *ubcdbl_p;
__device__ __forceinline__ double ld_u32(ubcdbl_p base, int i) { return base[i]; }
#else
typedef gdbl_p ubdbl_p;
typedef gcdbl_p ubcdbl_p;
#endif

// ---- KI-interleaved arrays (the two right-hand sides of a dual solve): element (i, k) sits at i * KI + k, so the
// KI values of one slot are ONE load / store of 8 KI bytes (KI = 2: 16 bytes per lane, the width the memory system likes best).
template <class P> struct vec2_of;
template <> struct vec2_of<double *> { typedef d2_t *type; };
template <> struct vec2_of<const double *> { typedef const d2_t *type; };
#if defined(__HIP_DEVICE_COMPILE__)
template <> struct vec2_of<gdbl_p> { typedef d2_t EICOS_DATA *type; };
template <> struct vec2_of<gcdbl_p> { typedef const d2_t EICOS_DATA *type; };
#endif
template <int KI, class P> __device__ __forceinline__ void ldK(P base, int i, double (&o)[KI]) {
    if constexpr (KI == 1) o[0] = base[i];
    else { static_assert(KI == 2, "KI"); const d2_t v = reinterpret_cast<typename vec2_of<P>::type>(base)[i]; o[0] = v.x; o[1] = v.y; }
}
template <int KI, class P> __device__ __forceinline__ void stK(P base, int i, const double (&o)[KI]) {
    if constexpr (KI == 1) base[i] = o[0];
    else { static_assert(KI == 2, "KI"); reinterpret_cast<typename vec2_of<P>::type>(base)[i] = d2_t{o[0], o[1]}; }
}
// global array, 32-bit byte offset addressing (ld_u32), optionally non-temporal
#if EICOS_UBL && defined(__HIP_DEVICE_COMPILE__)
template <int KI, bool NT> __device__ __forceinline__ void ldK_g(ubcdbl_p base, int i, double (&o)[KI]) { // (U in LDS: the backward sweep's value stream)
    if constexpr (KI == 1) o[0] = base[i];
    else { static_assert(KI == 2, "KI"); const d2_t v = reinterpret_cast<const d2_t __attribute__((address_space(3))) *>(base)[i]; o[0] = v.x; o[1] = v.y; }
}
#endif
template <int KI, bool NT> __device__ __forceinline__ void ldK_g(gcdbl_p base, int i, double (&o)[KI]) {
    if constexpr (KI == 1) o[0] = NT ? ld_u32_nt(base, i) : ld_u32(base, i);
    else {
        static_assert(KI == 2, "KI");
        const d2_t EICOS_DATA *b2 = reinterpret_cast<const d2_t EICOS_DATA *>(base);
        const d2_t v = NT ? ld_u32_nt(b2, i) : ld_u32(b2, i);
        o[0] = v.x; o[1] = v.y;
    }
}

// Sum over an aligned group of g = 1<<lg adjacent lanes, result valid in the group's lane 0.
// Steps of 1..8 lanes stay inside a DPP row (row_shl: lane i reads lane i+n of its 16-lane row, 0 past
// the end) and run at VALU speed; only g = 32/64 needs the LDS-crossbar shuffles for its last steps.
template <int CTRL> __device__ __forceinline__ double dpp_shl_add(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
    return v + __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// lane C of every 16-lane DPP row, broadcast to the whole row: ONE v_mov_b64_dpp row_newbcast:C (the only DPP control
// the 64-bit data path of gfx90a+ has)
template <int C> __device__ __forceinline__ double dpp_row_bcast(double v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + C, 0xF, 0xF, false); // (every lane has a source: `old` is never used)
}
// compile-time loop C = BEGIN, BEGIN + STEP, ... (exclusive END): the DPP controls above are immediates
template <int C, int END, int STEP, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr ((STEP > 0 && C < END) || (STEP < 0 && C > END)) { f(std::integral_constant<int, C>{}); static_for<C + STEP, END, STEP>(f); }
}
__device__ __forceinline__ double grp_reduce_to_lane0(double v, int lg) { // lg is wavefront-uniform
    if (lg == 0) return v; // one lane per row (most slices): one scalar branch instead of the six of the ladder
    if (lg >= 3) {
        if (lg >= 6) v += __shfl_xor(v, 32, 64);
        if (lg >= 5) v += __shfl_xor(v, 16, 64);
        if (lg >= 4) v = dpp_shl_add<0x108>(v); // row_shl:8
        v = dpp_shl_add<0x104>(v);              // row_shl:4
    }
    if (lg >= 2) v = dpp_shl_add<0x102>(v); // row_shl:2
    return dpp_shl_add<0x101>(v);           // row_shl:1
}

struct OpSum { __device__ static double f(double a, double b) { return a + b; } };
struct OpMax { __device__ static double f(double a, double b) { return fmax(a, b); } };
struct OpMin { __device__ static double f(double a, double b) { return fmin(a, b); } };

template <class Op>
__device__ __forceinline__ double wave_reduce(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = Op::f(v, __shfl_xor(v, o, 64));
    return v;
}

// Workgroup-wide reduction of NV values, result to EVERY thread.  One barrier per call: the
// LDS scratch is ping-ponged by `phase` (a workgroup-uniform register toggled per call).
template <class Op, int T, int NV>
__device__ __forceinline__ void blk_reduce(int &phase, double (&v)[NV]) {
    static_assert(NV <= 8, "too many values");
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = wave_reduce<Op>(v[i]);
    auto buf = g_S.red + phase * RED_SLOTS;
    phase ^= 1;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) buf[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) {
        double r = buf[i];
#pragma unroll
        for (int w = 1; w < NW; w++) r = Op::f(r, buf[w * NV + i]);
        v[i] = r;
    }
}
template <class Op, int T>
__device__ __forceinline__ double blk_reduce1(int &phase, double x) {
    double v[1] = {x};
    blk_reduce<Op, T, 1>(phase, v);
    return v[0];
}

template <int G> __device__ __forceinline__ double grp_sum(double v) {
    if constexpr (G == 64) return wave_reduce<OpSum>(v);
    else return v;
}

// Run body(c, integral_constant<G>, lane) once per cone: small cones one thread each (G=1),
// big cones one wavefront each (G=64).
template <int T, class Body>
__device__ __forceinline__ void for_cones(int ps, Body &&body) {
    const DevPat &P = c_pat[ps];
    const int tid = threadIdx.x;
    for (int q = tid; q < P.n_small; q += T) body(P.cone_small[q], std::integral_constant<int, 1>{}, 0);
    for (int q = tid >> 6; q < P.n_big; q += T / 64) body(P.cone_big[q], std::integral_constant<int, 64>{}, tid & 63);
}

// The cones neither for_tiny nor for_wave covers: DevPat::cone_mid (TINY_D < dimension < CONE_BIG) one thread each, cone_huge
// (dimension > 64) one wavefront each.
template <int T, class Body>
__device__ __forceinline__ void for_cones_rest(int ps, Body &&body) {
    const DevPat &P = c_pat[ps];
    const int tid = threadIdx.x;
    for (int q = tid; q < P.n_mid; q += T) body(P.cone_mid[q], std::integral_constant<int, 1>{}, 0);
    for (int q = tid >> 6; q < P.n_huge; q += T / 64) body(P.cone_huge[q], std::integral_constant<int, 64>{}, tid & 63);
}

// "Wave" cones (CONE_BIG <= dimension <= 64: the 32 cones of dimension 64 of the dense-front config), one wavefront per cone,
// LANE-PER-ROW REGISTERS: the generic wavefront body is a chain of dependent loads per cone (cone id -> first row / dimension -> row
// slots -> values) and a workgroup of 8 wavefronts walks 32 cones in 4 such rounds; here the descriptors of a wavefront's U cones are
// loaded first (`pre`: integers the value loads need as addresses), then all values (`ld`), then the bodies run (`fn`, wavefront
// reductions).  Lane L holds row L ("shift 0") and / or row L + 1 ("shift 1") of its cone, exactly the rows the generic loops
// `for (k = lane; ...)` / `for (k = 1 + lane; ...)` give that lane, so every wavefront reduction sums the same values on the same
// lanes: results are bit-identical to the generic bodies.
struct WCone { int c, o, d; };
template <int T, int U, class PreF, class L, class F>
__device__ __forceinline__ void for_wave(const DevPat &P, PreF &&pre, L &&ld, F &&fn) {
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, wave = uni((int)threadIdx.x >> 6);
    for (int q0 = wave; q0 < P.n_wave; q0 += U * NW) {
        WCone b[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int c = P.cone_wave[min(q0 + u * NW, P.n_wave - 1)]; b[u] = WCone{c, P.cone_off[c], P.cq[c]}; }
        decltype(pre(b[0], 0)) a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = pre(b[u], lane);
        decltype(ld(b[0], 0, a[0])) r[U];
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = ld(b[u], lane, a[u]);
#pragma unroll
        for (int u = 0; u < U; u++) if (q0 + u * NW < P.n_wave) fn(b[u], lane, a[u], r[u]);
    }
}
struct NoPre {};
// the lane's value of array a in cone b: row lane + s (clamped to the last row; lanes past the dimension load a valid duplicate)
template <class Ptr> __device__ __forceinline__ double wrow(Ptr a, const WCone &b, int lane, int s) { return a[b.o + min(lane + s, b.d - 1)]; }
template <int T> constexpr int wave_cones_in_flight() { return waves_per_eu<T>() <= 2 ? 4 : 2; }

// Tiny cones (dimension <= TINY_D = 4: the 332 cones of dimension 3 of an MPC-SOC pattern), one thread per cone, REGISTER-RESIDENT:
// `ld(t)` loads everything the cone's body reads (into a small struct), `fn(t, loaded)` does the arithmetic and the stores.  The
// descriptors of a thread's U cones are loaded first, then all their data, then the bodies run: a generic per-cone body is a chain of
// 4..6 dependent global loads (cone id -> first row / dimension -> elimination slots -> values, and a loop over the rows that the
// compiler will not unroll), repeated for every round of T cones; here a pass over the cones costs two round trips.  The bodies
// spell out the generic loops with the rows in registers, IN THE SAME ORDER of operations (results are bit-identical).
typedef int ti4_t __attribute__((ext_vector_type(4)));
struct Tiny { int o, d, c, ev, eu, ez[TINY_D], vb; }; // first row, dimension, cone id, elimination slots of v / u / the rows, first scaling-block slot
__device__ __forceinline__ Tiny tiny_at(gint_p tab, int q) {
    typedef const ti4_t EICOS_GLOBAL *gi4;
    gi4 p = reinterpret_cast<gi4>(tab + (size_t)q * TINY_INTS);
    const ti4_t a = p[0], b = p[1], c = p[2];
    return Tiny{a.x, a.y, a.z, a.w, b.x, {b.y, b.z, b.w, c.x}, c.y};
}
template <int T, int U = 2, class L, class F>
__device__ __forceinline__ void for_tiny(const DevPat &P, L &&ld, F &&fn) {
    const int nt = P.n_tiny;
    for (int q0 = threadIdx.x; q0 < nt; q0 += U * T) {
        Tiny t[U];
#pragma unroll
        for (int u = 0; u < U; u++) t[u] = tiny_at(P.cone_tiny, min(q0 + u * T, nt - 1)); // (clamped: unconditional loads)
        decltype(ld(t[0])) r[U];
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = ld(t[u]);
#pragma unroll
        for (int u = 0; u < U; u++) if (q0 + u * T < nt) fn(t[u], r[u]);
    }
}
// the TINY_D consecutive doubles of array a at the rows of tiny cone t (indices past the dimension are clamped to the last row)
struct D4 { double v[TINY_D]; };
template <class Ptr> __device__ __forceinline__ D4 tiny_rows(Ptr a, const Tiny &t) {
    D4 r;
#pragma unroll
    for (int k = 0; k < TINY_D; k++) r.v[k] = a[t.o + min(k, t.d - 1)];
    return r;
}

#define FOR_T(i, cnt) for (int i = threadIdx.x; i < (cnt); i += T)

// Elementwise pass over [0, cnt) whose global loads are issued U iterations at a time: `ld(i)` returns what
// iteration i reads (a small struct), `fn(i, loaded)` does the arithmetic and the stores.  A plain FOR_T loop is
// not unrolled by the compiler (unknown trip count, possibly aliasing stores) and waits for every iteration's
// loads before it issues the next ones: cnt/T serial memory round trips per pass.
struct V1 { double a; };
struct V2 { double a, b; };
struct V3 { double a, b, c; };
struct V4 { double a, b, c, d; };
template <int KI> struct VKI { double v[KI]; };
template <int KI> struct IVK { int o; double v[KI]; };
struct IV1 { int i; double a; };
struct IV2 { int i; double a, b; };
struct IIV { int i, j; double a; };
struct IIV3 { int i, j; double a, b, c; };
template <int T, int U0 = 4, class L, class F>
__device__ __forceinline__ void for_t_pre(int cnt, L &&ld, F &&fn) {
    constexpr int U = U0; // elements per thread in flight (x2 / x3 measured in round 3: +-0 / -6 %)
    for (int i0 = threadIdx.x; i0 < cnt; i0 += U * T) {
        decltype(ld(0)) r[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * T; r[u] = ld(i < cnt ? i : i0); } // clamped: unconditional loads
#pragma unroll
        for (int u = 0; u < U; u++) { const int i = i0 + u * T; if (i < cnt) fn(i, r[u]); }
    }
}

// The same over three index ranges (the x, y and z parts of a KKT-space quantity) as ONE software-pipelined loop: a step covers T
// consecutive elements of one range (the range of a step is workgroup-uniform), U steps are in flight -- the short x and y ranges no
// longer cost a memory round trip each.  Element i of a range goes to thread i mod T, as in three separate for_t_pre loops (per-thread
// partial sums are bit-identical).  ld / fn get the range and the index in the CONCATENATED array [range 0 | range 1 | range 2].
template <int T, int U, class L, class F>
__device__ __forceinline__ void for_t_pre3(int n0, int n1, int n2, L &&ld, F &&fn) {
    const int K0 = (n0 + T - 1) / T, K1 = (n1 + T - 1) / T, K2 = (n2 + T - 1) / T, K = K0 + K1 + K2;
    const int tid = threadIdx.x;
    // step kk -> range g (workgroup-uniform), index in the concatenated array [range 0 | range 1 | range 2], valid or not; all by
    // arithmetic on SGPRs (a three-way select of pointers is turned into a table in scratch memory by the compiler)
    auto locate = [&](int kk, int &g, int &j, bool &ok) {
        const int a = kk >= K0 ? 1 : 0, b = kk >= K0 + K1 ? 1 : 0;
        g = a + b;
        const int i = tid + (kk - a * K0 - b * K1) * T;            // index inside its range
        const int cnt = n0 + a * (n1 - n0) + b * (n2 - n1);
        ok = kk < K && i < cnt;
        j = ok ? i + a * n0 + b * n1 : 0;                           // (clamped: unconditional loads)
    };
    for (int kb = 0; kb < K; kb += U) {
        decltype(ld(0, 0)) r[U];
#pragma unroll
        for (int u = 0; u < U; u++) { int g, j; bool ok; locate(kb + u, g, j, ok); r[u] = ld(g, j); }
#pragma unroll
        for (int u = 0; u < U; u++) { int g, j; bool ok; locate(kb + u, g, j, ok); if (ok) fn(g, j, r[u]); }
    }
}

// Row products over a sliced-ELL plan (no levels): sum_r = sum_k val[r,k] * x[idx[r,k]], unit-stride
// index/value loads, g lanes per row, epi(row, sum) on one lane per row.  Loads run ELL_DEPTH slices
// ahead of the arithmetic (same register queue as tri_sweep, no barriers); plans are padded to a
// multiple of ELL_DEPTH slices and hold at most ELL_KMAX entries per lane and slice (longer rows are
// cut into sub-slices whose partial sums are carried in a register).  `sm` may live in LDS (staged by k_solve) or in global memory.
// Decoded slice descriptor; every field is workgroup-uniform (SGPRs).
struct Sl { int row0, off, off16, cnt, lg, K, newlev, last, more, cont; };
__device__ __forceinline__ Sl slice_decode(const PackedSlice &w) {
    const int b = uni(w.bits);
    return Sl{uni(w.row0), uni(w.off), uni(w.off16), b & ((1 << PS_LG) - 1), (b >> PS_LG) & 7, (b >> PS_K) & 7,
              (b >> PS_NEWLEV) & 1, (b >> PS_LAST) & 1, (b >> PS_MORE) & 1, (b >> PS_CONT) & 1};
}
template <class Tab> __device__ __forceinline__ Sl slice_at(const Tab *tab, int s) {
    const PackedSlice w = tab[s]; // 16 bytes: one ds_read_b128 (LDS copy) or one global/scalar load
    const int b = uni(w.bits);
    return Sl{uni(w.row0), uni(w.off), uni(w.off16), b & ((1 << PS_LG) - 1), (b >> PS_LG) & 7, (b >> PS_K) & 7,
              (b >> PS_NEWLEV) & 1, (b >> PS_LAST) & 1, (b >> PS_MORE) & 1, (b >> PS_CONT) & 1};
}
// The ELL_KMAX gather indices of one lane: four 32-bit loads, or (I16) one 8-byte load of four packed 16-bit indices.
typedef const uint2 EICOS_GLOBAL *gidx16_p;
template <bool I16> __device__ __forceinline__ void load_indices(int (&ni)[ELL_KMAX], gint_p eidx, gint_p eidx16, bool act, int K, int off,
                                                                 int lanes, int t, int dummy_slot, int off16, int d16) {
    if constexpr (I16) {
        const uint2 w = ld_u32(reinterpret_cast<gidx16_p>(eidx16), act ? off16 + t : d16);
        ni[0] = w.x & 0xffffu; ni[1] = w.x >> 16; ni[2] = w.y & 0xffffu; ni[3] = w.y >> 16;
    } else {
#pragma unroll
        for (int kk = 0; kk < ELL_KMAX; kk++) ni[kk] = ld_u32(eidx, (act && kk < K) ? off + kk * lanes + t : dummy_slot);
    }
}

// `pre(row)` loads whatever the epilogue needs per row (rhs entry, destination index, ...); it is issued with the
// slice's index/value loads, ELL_DEPTH slices ahead, so that `epi(row, sum, pre(row))` starts no global load itself.
template <int T, bool I16, class SM, class V, class X, class Pre, class Epi>
__device__ __forceinline__ void ell_dots(const SM *sm, int ns, int nr, gint_p eidx, gint_p eidx16, int d16, V eval, X x, int dummy_slot,
                                         Pre &&pre, Epi &&epi) {
    if (nr == 0) return; // no rows  (ns: slices incl. the host's padding, nr: the real ones -- see tri_sweep)
    const int t = threadIdx.x;
    using R = decltype(pre(0));
    int qi[ELL_DEPTH][ELL_KMAX]; double qv[ELL_DEPTH][ELL_KMAX]; R qr[ELL_DEPTH]; Sl qm[ELL_DEPTH]; // (qm: the slice's decoded descriptor, SGPRs)
    double carry = 0.; // partial sum of rows cut into sub-slices (SliceMeta::more / cont)
    auto load = [&](const Sl &nm, int (&ni)[ELL_KMAX], double (&nv)[ELL_KMAX], R &nr, Sl &om) {
        if constexpr (ELL_DQ) om = nm;
        const int lanes = nm.cnt << nm.lg;
        const bool act = t < lanes;
        load_indices<I16>(ni, eidx, eidx16, act, nm.K, nm.off, lanes, t, dummy_slot, nm.off16, d16);
#pragma unroll
        for (int kk = 0; kk < ELL_KMAX; kk++) {
            const int slot = (act && kk < nm.K) ? nm.off + kk * lanes + t : dummy_slot;
#ifdef EICOS_PROBE_NOVAL // (dev probe: what would the products cost with their value stream free?  wrong numerics, timing only)
            nv[kk] = 1e-3 * (double)(slot & 7);
#else
            nv[kk] = ld_u32_nt(eval, slot); // streamed once per pass: keep the shared index arrays in L2
#endif
        }
        nr = pre(act ? nm.row0 + (t >> nm.lg) : 0);
    };
#pragma unroll
    for (int d = 0; d < ELL_DEPTH; d++) load(slice_at(sm, d < ns ? d : 0), qi[d], qv[d], qr[d], qm[d]);
    // (ELL_DQ, the 256-VGPR build only: the queue keeps the decoded descriptors and the next one is read a step before it is decoded, see tri_sweep --
    // round 5: headline +2 ... 3 %; in the register-tight builds the extra live scalars cost more than the round trip: batch 4096 and dense-front -2.5 %)
    PackedSlice raw = sm[min(ELL_DEPTH, ns - 1)];
    // one slice step; `d` = the slice's slot of the register queue.  A trip of ELL_TRIP slices is unrolled (the compiler's s_waitcnt insertion is
    // exact inside a trip and drains the load queue at every loop head: DESIGN.md 4.2); the remainder runs in trips of ELL_DEPTH, so
    // plans are padded to a multiple of ELL_DEPTH only -- a padded slice costs a full step of the dependent chain on small patterns
    auto step = [&](const int d, const int s) __attribute__((always_inline)) {
        const Sl m = ELL_DQ ? qm[d] : slice_at(sm, s);
        int ci[ELL_KMAX]; double cv[ELL_KMAX];
        const R cr = qr[d];
#pragma unroll
        for (int kk = 0; kk < ELL_KMAX; kk++) { ci[kk] = qi[d][kk]; cv[kk] = qv[d][kk]; }
        if constexpr (ELL_DQ) { load(slice_decode(raw), qi[d], qv[d], qr[d], qm[d]); raw = sm[min(s + ELL_DEPTH + 1, ns - 1)]; }
        else load(slice_at(sm, min(s + ELL_DEPTH, ns - 1)), qi[d], qv[d], qr[d], qm[d]);
        const int lanes = m.cnt << m.lg;
        const bool act = t < lanes;
        double xg[ELL_KMAX];
#pragma unroll
        for (int kk = 0; kk < ELL_KMAX; kk++) xg[kk] = x[ci[kk]];
        __builtin_amdgcn_sched_barrier(0);
        double acc = 0.;
#pragma unroll
        for (int kk = 0; kk < ELL_KMAX; kk++) acc = madd(acc, cv[kk], xg[kk]);
        acc = grp_reduce_to_lane0(acc, m.lg);
        if (m.cont) acc += carry;
        if (m.more) carry = acc;
        else if (act && (t & ((1 << m.lg) - 1)) == 0) epi(m.row0 + (t >> m.lg), acc, cr);
    };
    int s0 = 0;
    for (; s0 + ELL_TRIP <= nr; s0 += ELL_TRIP) {
#pragma unroll
        for (int u = 0; u < ELL_TRIP; u++) step(u % ELL_DEPTH, s0 + u);
    }
    for (; s0 < nr; s0 += ELL_DEPTH) { // (stopping at the last real slice)
#pragma unroll
        for (int u = 0; u < ELL_DEPTH; u++) { if (s0 + u < nr) step(u, s0 + u); }
    }
}

// ell_dots for KI right-hand sides at once: the slice descriptors, gather indices and (SHARED: the dual solve of ONE instance)
// the matrix values are loaded once, the gathered vector x is KI-interleaved; `pre(k, row)` / `epi(k, row, sum, pre)` get the
// number of the right-hand side.
template <int T, bool I16, int KI, bool SHARED, class SM, class X, class Pre, class Epi>
__device__ __forceinline__ void ell_dots_k(const SM *sm, int ns, int nr, gint_p eidx, gint_p eidx16, int d16, const gcdbl_p (&eval)[KI], X x, int dummy_slot,
                                           Pre &&pre, Epi &&epi) {
    if (nr == 0) return; // no rows
    const int t = threadIdx.x;
    using R = decltype(pre(0, 0));
    int qi[ELL_DEPTH][ELL_KMAX]; double qv[ELL_DEPTH][ELL_KMAX][KI]; R qr[ELL_DEPTH][KI]; Sl qm[ELL_DEPTH]; // (qm: the slice's decoded descriptor, SGPRs)
    double carry[KI]; // partial sum of rows cut into sub-slices (SliceMeta::more / cont)
#pragma unroll
    for (int k = 0; k < KI; k++) carry[k] = 0.;
    auto load = [&](const Sl &nm, int (&ni)[ELL_KMAX], double (&nv)[ELL_KMAX][KI], R (&nr)[KI], Sl &om) {
        if constexpr (ELL_DQ) om = nm;
        const int lanes = nm.cnt << nm.lg;
        const bool act = t < lanes;
        load_indices<I16>(ni, eidx, eidx16, act, nm.K, nm.off, lanes, t, dummy_slot, nm.off16, d16);
#pragma unroll
        for (int kk = 0; kk < ELL_KMAX; kk++) {
            const int slot = (act && kk < nm.K) ? nm.off + kk * lanes + t : dummy_slot;
            // streamed once per pass: keep the shared index arrays in L2.  SHARED (dual right-hand sides of ONE instance):
            // the KI vectors are multiplied by the same matrix values -> one load
#ifdef EICOS_PROBE_NOVAL
            nv[kk][0] = 1e-3 * (double)(slot & 7);
#else
            nv[kk][0] = ld_u32_nt(eval[0], slot);
#endif
#pragma unroll
            for (int k = 1; k < KI; k++) nv[kk][k] = SHARED ? nv[kk][0] : ld_u32_nt(eval[k], slot);
        }
#pragma unroll
        for (int k = 0; k < KI; k++) nr[k] = pre(k, act ? nm.row0 + (t >> nm.lg) : 0);
    };
#pragma unroll
    for (int d = 0; d < ELL_DEPTH; d++) load(slice_at(sm, d < ns ? d : 0), qi[d], qv[d], qr[d], qm[d]);
    // (ELL_DQ, the 256-VGPR build only: the queue keeps the decoded descriptors and the next one is read a step before it is decoded, see tri_sweep --
    // round 5: headline +2 ... 3 %; in the register-tight builds the extra live scalars cost more than the round trip: batch 4096 and dense-front -2.5 %)
    PackedSlice raw = sm[min(ELL_DEPTH, ns - 1)];
    // one slice step; `d` = the slice's slot of the register queue.  A trip of ELL_TRIP slices is unrolled (the compiler's s_waitcnt insertion is
    // exact inside a trip and drains the load queue at every loop head: DESIGN.md 4.2); the remainder runs in trips of ELL_DEPTH, so
    // plans are padded to a multiple of ELL_DEPTH only -- a padded slice costs a full step of the dependent chain on small patterns
    auto step = [&](const int d, const int s) __attribute__((always_inline)) {
        const Sl m = ELL_DQ ? qm[d] : slice_at(sm, s);
        int ci[ELL_KMAX]; double cv[ELL_KMAX][KI]; R cr[KI];
#pragma unroll
        for (int k = 0; k < KI; k++) cr[k] = qr[d][k];
#pragma unroll
        for (int kk = 0; kk < ELL_KMAX; kk++) {
            ci[kk] = qi[d][kk];
#pragma unroll
            for (int k = 0; k < KI; k++) cv[kk][k] = qv[d][kk][k];
        }
        if constexpr (ELL_DQ) { load(slice_decode(raw), qi[d], qv[d], qr[d], qm[d]); raw = sm[min(s + ELL_DEPTH + 1, ns - 1)]; }
        else load(slice_at(sm, min(s + ELL_DEPTH, ns - 1)), qi[d], qv[d], qr[d], qm[d]);
        const int lanes = m.cnt << m.lg;
        const bool act = t < lanes;
        double xg[ELL_KMAX][KI];
#pragma unroll
        for (int kk = 0; kk < ELL_KMAX; kk++) ldK<KI>(x, ci[kk], xg[kk]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KI; k++) {
            double acc = 0.;
#pragma unroll
            for (int kk = 0; kk < ELL_KMAX; kk++) acc = madd(acc, cv[kk][k], xg[kk][k]);
            acc = grp_reduce_to_lane0(acc, m.lg);
            if (m.cont) acc += carry[k];
            if (m.more) carry[k] = acc;
            else if (act && (t & ((1 << m.lg) - 1)) == 0) epi(k, m.row0 + (t >> m.lg), acc, cr[k]);
        }
    };
    int s0 = 0;
    for (; s0 + ELL_TRIP <= nr; s0 += ELL_TRIP) {
#pragma unroll
        for (int u = 0; u < ELL_TRIP; u++) step(u % ELL_DEPTH, s0 + u);
    }
    for (; s0 < nr; s0 += ELL_DEPTH) { // (stopping at the last real slice)
#pragma unroll
        for (int u = 0; u < ELL_DEPTH; u++) { if (s0 + u < nr) step(u, s0 + u); }
    }
}

// Workgroup barrier that orders LDS traffic only.  On gfx9-family parts loads and stores share
// the vmcnt counter, so __syncthreads() (release fence) drains every outstanding global LOAD as
// well -- which would serialise the software prefetch below behind each level barrier.  The
// level-to-level hand-off of the triangular sweeps goes through LDS only, so waiting for
// lgkmcnt(0) before s_barrier is sufficient there.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// One triangular sweep of the level-scheduled LDL' solve over a sliced-ELL plan (SliceMeta):
// forward:  ws[i] =  ws[i] - sum_k L[i,k] ws[k]              (L y = b, unit-lower L in UF, rows by level)
// backward: ws[j] = (ws[j] - sum_i U[i,j] ws[i]) / D[j]        (x = L^-T D^-1 y with U = L.*D in UB, columns by level, top down)
// Index/value/invD loads run TRI_DEPTH slices ahead of their use: they do not depend on ws, so
// their HBM/L2 latency overlaps earlier levels and the dependent part of a level is only LDS
// gathers + one LDS store + the barrier.  `sm` points to the slice table (LDS copy when staged).
// SOLO: the narrow top of the elimination tree, laid out for 64 lanes and run by wavefront 0 alone -- no
// workgroup barrier between its levels (one wavefront's LDS accesses execute in order), the other wavefronts
// wait at the caller's barrier instead of issuing a full slice of masked-off instructions per level.
// VSH (dual right-hand sides of ONE instance): the vector ws is KI-interleaved, the factor (eval, invD) is a single one.
template <int T, bool FORWARD, bool LDSBAR, bool SOLO, bool I16, int KI, bool VSH = false, class SM, class EV, class WS>
__device__ __forceinline__ void tri_sweep(const SM *sm, int ns, int nr, gint_p eidx, gint_p eidx16, int d16, EV eval, gcdbl_p invD, WS ws,
                                          int dummy_slot) {
    // KI = 2 (with VSH): the two right-hand sides of a dual solve -- one factor, the sweep vector ws 2-interleaved, so every
    // gather / store of the pair is one 16-byte LDS access
    // ns = slices of this section of the plan INCLUDING the empty ones the host pads it with (to a multiple of the queue depth: the refills
    // past the end stay inside the table), nr = the real ones: the padding is never stepped through -- on a small pattern a padded slice was a
    // full step of the dependent chain (lp_afiro: three of the five steps of its forward sweep)
    if (nr == 0) { if (!SOLO) __syncthreads(); return; }
    const int t = threadIdx.x;
    // the narrow tree top (SOLO, one wavefront) runs short steps: the same lead time needs a deeper queue than the workgroup-wide levels
    constexpr int DEPTH = SOLO ? TRI_DEPTH_SOLO : TRI_DEPTH;
    constexpr int TRIP = SOLO ? ((TRI_TRIP + TRI_DEPTH_SOLO - 1) / TRI_DEPTH_SOLO) * TRI_DEPTH_SOLO : TRI_TRIP; // (a multiple of the queue depth)
    struct Slot { // one prefetched slice: descriptor (SGPRs), ELL_KMAX (index, value) pairs, 1/D and old value of the own row
        int row0, lg, K, off, lanes, newlev, more, cont;
        int idx[ELL_KMAX]; double val[ELL_KMAX][KI]; double d[KI], own[KI];
    } q[DEPTH];
    // every slice issues the same number of global loads per lane (inactive lanes / padding read the plan's dummy
    // slot: index N, value 0) so the compiler can count them in s_waitcnt vmcnt(n)
    auto load_m = [&](const Sl &nm, Slot &o) {
        o.row0 = nm.row0; o.lg = nm.lg; o.K = nm.K; o.off = nm.off; o.newlev = nm.newlev;
        o.more = nm.more; o.cont = nm.cont;
        o.lanes = nm.cnt << o.lg;
        const bool act = t < o.lanes;
        load_indices<I16>(o.idx, eidx, eidx16, act, o.K, o.off, o.lanes, t, dummy_slot, nm.off16, d16);
#pragma unroll
        for (int kk = 0; kk < ELL_KMAX; kk++) {
            const int slot = (act && kk < o.K) ? o.off + kk * o.lanes + t : dummy_slot;
            if constexpr (VSH) { // one value for both right-hand sides
                double v1[1];
                ldK_g<1, true>(eval, slot, v1);
#pragma unroll
                for (int k = 0; k < KI; k++) o.val[kk][k] = v1[0];
            } else ldK_g<KI, true>(eval, slot, o.val[kk]); // streamed once per sweep: do not displace the index arrays in L2
        }
        const int r = act ? o.row0 + (t >> o.lg) : 0;
        if constexpr (FORWARD) { // forward is L y = b with unit-lower L: no pivot needed
#pragma unroll
            for (int k = 0; k < KI; k++) o.d[k] = 0.;
        } else if constexpr (VSH) {
            double d1[1];
            ldK_g<1, false>(invD, r, d1);
#pragma unroll
            for (int k = 0; k < KI; k++) o.d[k] = d1[0];
        } else ldK_g<KI, false>(invD, r, o.d);
        ldK<KI>(ws, r, o.own); // rows of later slices are not written before their own slice runs
    };
    auto load = [&](int s, Slot &o) { load_m(slice_at(sm, s), o); };
    // ns is a multiple of TRI_DEPTH (the host pads plans with empty slices) and refills past the end
    // re-read the last slice, so the steady-state loop has no data-dependent branch around its loads
#pragma unroll
    for (int d = 0; d < DEPTH; d++) load(d < ns ? d : 0, q[d]);
    // one slice step; `d` = the slice's slot of the register queue.  A trip of TRI_TRIP slices is unrolled (the compiler's s_waitcnt insertion is
    // exact inside a trip and drains the load queue at every loop head: DESIGN.md 4.2); the remainder runs in trips of TRI_DEPTH, so
    // plans are padded to a multiple of TRI_DEPTH only -- a padded slice costs a full step of the dependent chain on small patterns
    // The descriptor of the next slice to be prefetched is READ one step before it is decoded (readfirstlane needs the data: a descriptor
    // read at the point of use is an LDS round trip on the dependent chain of every step; round 5: sweeps 44.8 -> 42.9 us per MPC02 solve).
    PackedSlice raw = sm[min(DEPTH, ns - 1)];
    auto step = [&](const int d, const int s) __attribute__((always_inline)) {
        const Slot c = q[d];
        load_m(slice_decode(raw), q[d]);
        raw = sm[min(s + DEPTH + 1, ns - 1)];
        if (c.newlev) {
            if (SOLO) { if (!LDSBAR) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); } // slab vector: drain the level's stores
            else if (LDSBAR) lds_barrier();
            else __syncthreads();
        }
        double xg[ELL_KMAX][KI]; // all gathers in flight before the first multiply (the scheduler would serialise them)
#pragma unroll
        for (int kk = 0; kk < ELL_KMAX; kk++) ldK<KI>(ws, c.idx[kk], xg[kk]);
        __builtin_amdgcn_sched_barrier(0);
        double acc[KI];
#pragma unroll
        for (int k = 0; k < KI; k++) {
            double a = 0.;
#pragma unroll
            for (int kk = 0; kk < ELL_KMAX; kk++) a = madd(a, c.val[kk][k], xg[kk][k]);
            acc[k] = grp_reduce_to_lane0(a, c.lg);
        }
        if (t < c.lanes && (t & ((1 << c.lg) - 1)) == 0) {
            const int r = c.row0 + (t >> c.lg);
            // rows cut into sub-slices: a continuation works on what the previous sub-slice left in ws[r]
            // (same lane, program order), and only the last one applies the pivot
            double cur[KI], out[KI];
            if (c.cont) ldK<KI>(ws, r, cur);
#pragma unroll
            for (int k = 0; k < KI; k++) {
                const double v = (c.cont ? cur[k] : c.own[k]) - acc[k];
                out[k] = (FORWARD || c.more) ? v         // y_i = b_i - sum_k L[i,k] y_k
                                             : v * c.d[k]; // x_j = (y_j - sum_i U[i,j] x_i) / D_j
            }
            stK<KI>(ws, r, out);
        }
    
    };
    int s0 = 0;
    for (; s0 + TRIP <= nr; s0 += TRIP) {
#pragma unroll
        for (int u = 0; u < TRIP; u++) step(u % DEPTH, s0 + u);
    }
    for (; s0 < nr; s0 += DEPTH) { // remainder in trips of the queue depth, stopping at the last real slice
#pragma unroll
        for (int u = 0; u < DEPTH; u++) { if (s0 + u < nr) step(u, s0 + u); }
    }
    if (!SOLO) __syncthreads();
}

// The dense apex of the elimination tree (symbolic.hpp: Symbolic::apex0; the last levels of the schedule, <= 64 nodes in all) -- both
// sweeps over it by ONE wavefront, lane = node, the iterate in a register:
//   forward  (unit-lower L):  for k = 0 .. na-2:  y_i -= L[i,k] y_k  on the lanes i > k   -- column k of the block across the lanes
//   backward (U = L.*D):      for i = na-1 .. 0:  x_i = acc_i / D_i;  acc_k -= U[i,k] x_i  on the lanes k < i  -- row i across the lanes
// na dependent steps of ~30 cycles replace (levels of the apex) slice steps of ~1100 (MPC02: 63 nodes = levels 10..20 of 21).  Same operations
// as a column-oriented substitution (the reference's, Eigen's, order); the level-scheduled sweeps sum a row first and subtract once.
// apex_solve below is the FALLBACK for handles without an LDS vector (patterns too large for LDS): it gathers the entries, APEX_QD steps ahead,
// from the folded images behind the sweep plans' value arrays in the workspace slab (device_types.hpp: apex_img_at; zero where L has no entry)
// -- a memory round trip per APEX_QD steps, slower than the level schedule on MPC02.  The product path is apex_solve_lds.
constexpr int APEX_QD = 8;
// (128-thread workgroups carry an apex only in the LDS-resident build -- 256 VGPRs, the images inside the LDS copy of the workspace slab: api.cpp)
template <int T> __device__ __forceinline__ bool apex_on(const DevPat &P) { if constexpr (T >= 256 || EICOS_LDSRES != 0) return P.apex_na > 0; else return false; }
// the parts of the split row (DevPat::apex_split_*): their sum -- every lane reads the same few slots -- and the slots zeroed for the next sweep
template <int KI, class WS>
__device__ __forceinline__ void apex_take_split(const DevPat &P, WS ws, int lane, double (&s)[KI]) {
#pragma unroll
    for (int r = 0; r < KI; r++) s[r] = 0.;
    const int n = uni(P.apex_split_n), slot = uni(P.apex_split_slot);
    if (n == 0) return;
    for (int q = 0; q < n; q++) {
        double t[KI];
        ldK<KI>(ws, slot + q, t);
#pragma unroll
        for (int r = 0; r < KI; r++) s[r] += t[r];
    }
    double z[KI];
#pragma unroll
    for (int r = 0; r < KI; r++) z[r] = 0.;
    if (lane < n) stK<KI>(ws, slot + lane, z);
}
__device__ __forceinline__ double rdlane_d(double v, int l) { // v of lane l (wavefront-uniform l) to every lane, through SGPRs
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int KI, class WS>
__device__ __forceinline__ void apex_solve(const DevPat &P, gcdbl_p UF, gcdbl_p UB, gcdbl_p invD, WS ws) {
    const int lane = threadIdx.x & 63;
    const int na = uni(P.apex_na), n0 = uni(P.apex_n0);
    gcdbl_p F = UF + uni(P.apex_f), R = UB + uni(P.apex_b);
    const int me = n0 + min(lane, na); // (lanes beyond the block: slot N, the always-zero slot of the vector and of 1/D; their image entries are zero)
    double x[KI], q[APEX_QD], sp[KI];
    ldK<KI>(ws, me, x);
    apex_take_split<KI>(P, ws, lane, sp);
#pragma unroll
    for (int r = 0; r < KI; r++) x[r] = lane == uni(P.apex_split_lane) ? x[r] + sp[r] : x[r];
    auto fcol = [&](int k) { return lane > k && lane < na ? ld_u32(F, apex_img_at(lane, k)) : 0.; }; // (the forward image is the folded one: a gather here)
#pragma unroll
    for (int d = 0; d < APEX_QD; d++) q[d] = fcol(min(d, na - 1));
    for (int k0 = 0; k0 < na - 1; k0 += APEX_QD) { // (the last column has nothing below it; a trip may run past it: those steps re-read it -- all zeros)
#pragma unroll
        for (int d = 0; d < APEX_QD; d++) {
            const int k = k0 + d;
            const double lk = q[d];
            q[d] = fcol(min(k + APEX_QD, na - 1));
#pragma unroll
            for (int r = 0; r < KI; r++) x[r] = x[r] - lk * rdlane_d(x[r], min(k, 63));
        }
    }
    const double dinv = ld_u32(invD, me);
    auto brow = [&](int i) { return lane < i ? ld_u32(R, apex_img_at(max(i, 1), lane)) : 0.; }; // (U[i, lane]: the folded image again)
#pragma unroll
    for (int d = 0; d < APEX_QD; d++) q[d] = brow(max(na - 1 - d, 0));
    for (int i0 = na - 1; i0 >= 0; i0 -= APEX_QD) { // (steps below row 0 of the last trip re-read row 0 -- all zeros -- and finalise no lane)
#pragma unroll
        for (int d = 0; d < APEX_QD; d++) {
            const int i = i0 - d;
            const double ui = q[d];
            q[d] = brow(max(i - APEX_QD, 0));
#pragma unroll
            for (int r = 0; r < KI; r++) {
                x[r] = (lane == i) ? x[r] * dinv : x[r]; // lane i: every row above it has been subtracted
                x[r] = x[r] - ui * rdlane_d(x[r], max(i, 0));
            }
        }
    }
    if (lane < na) stK<KI>(ws, n0 + lane, x);
}
// The same from the LDS image of the block's unit-lower L (DevPat::apex_lds; filled by stage_factor).  The image is the strictly lower triangle FOLDED into
// 32 rows of 65 doubles (device_types.hpp: apex_img_at): row i >= 32 lies in image row 63 - i at columns 0 .. i-1, row i < 32 in image row i
// from column 63 DOWN -- a row of L is a run of consecutive addresses (backward: row i across the lanes k < i) and a column of L visits
// consecutive image rows at stride 65 +- 1 (forward: lane i reads its own row at column k): both sweeps read without bank conflicts beyond
// the two passes a 64-lane 8-byte read takes anyway.  Backward works on z = D^-1 y with L itself -- x_k = z_k - sum_{i > k} L[i,k] x_i -- so no
// step carries a scaling.
// NO MASKS: at forward step k the lanes <= k read some other slot of the image (finite: an entry of L or a zero) and their register is
// overwritten with garbage -- but a lane's value is dead once it has been the pivot: step k broadcasts lane k BEFORE the update, and the
// broadcast value (two SGPRs) is written into lane k of a result register with v_writelane.  Backward likewise (lanes >= i are dead at step i).
// Both loops are FULLY unrolled in blocks of APEX_BLK steps -- the step number is a compile-time constant: pivot lane, written lane and the
// row offsets of the backward reads are immediates -- and block b + 1 is read while block b's chain of broadcast / multiply / subtract
// runs.  (Measured on the way: a rolled loop with a four-deep queue 186 cycles per step; masked reads under constant lane masks made the
// compiler keep 63 exec masks in spilled SGPRs: 2400 instructions.)
constexpr int APEX_BLK = 16;
template <int L> __device__ __forceinline__ double wrlane_d(double into, double uniform_v) { // lane L of `into` = uniform_v (wavefront-uniform: SGPRs)
    unsigned long long o = (unsigned long long)__double_as_longlong(into);
    const unsigned long long v = (unsigned long long)__double_as_longlong(uniform_v);
    unsigned lo = (unsigned)o, hi = (unsigned)(o >> 32);
    // (no v_writelane builtin in this compiler)
    asm("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"((unsigned)v), "n"(L));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"((unsigned)(v >> 32)), "n"(L));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int KI, class WS>
__device__ __forceinline__ void apex_solve_lds(const DevPat &P, gcdbl_p invD, WS ws) {
    constexpr int NB = 64 / APEX_BLK;
    const int lane = threadIdx.x & 63;
    const int na = uni(P.apex_na), n0 = uni(P.apex_n0), img = uni(P.apex_lds);
    const int me = n0 + min(lane, na);
    double x[KI], y[KI], sp[KI];
    ldK<KI>(ws, me, x); // (lanes beyond the block: the zero slot N; the rows >= na of the image hold zeros, so they stay zero)
    apex_take_split<KI>(P, ws, lane, sp);
#pragma unroll
    for (int r = 0; r < KI; r++) x[r] = lane == uni(P.apex_split_lane) ? x[r] + sp[r] : x[r];
    const double dinv = ld_u32(invD, me);
#pragma unroll
    for (int r = 0; r < KI; r++) y[r] = 0.;
    // ---- forward: lane i reads its own row of L at column k ----
    const int cbase = img + (lane >= 32 ? (63 - lane) * 65 : lane * 65 + 63), cstep = lane >= 32 ? 1 : -1;
    double a[2][APEX_BLK];
    auto load_cols = [&](int k0, double (&o)[APEX_BLK]) {
#pragma unroll
        for (int d = 0; d < APEX_BLK; d++) o[d] = g_dyn[cbase + cstep * min(k0 + d, 62)];
    };
    load_cols(0, a[0]);
    int done = 0; // pivots captured in y: [0, done)
    static_for<0, NB, 1>([&](auto Bc) {
        constexpr int b = decltype(Bc)::value;
        if (b * APEX_BLK < na - 1) { // (wavefront-uniform; the last column has nothing below it)
            if constexpr (b + 1 < NB) load_cols((b + 1) * APEX_BLK, a[(b + 1) & 1]);
            static_for<0, APEX_BLK, 1>([&](auto Dc) {
                constexpr int d = decltype(Dc)::value, k = b * APEX_BLK + d;
#pragma unroll
                for (int r = 0; r < KI; r++) {
                    const double xk = rdlane_d(x[r], k);
                    y[r] = wrlane_d<k>(y[r], xk);
                    x[r] = x[r] - a[b & 1][d] * xk;
                }
            });
            done = (b + 1) * APEX_BLK;
        }
    });
    // z = D^-1 y: captured pivots, the live values of the lanes the loop did not reach
#pragma unroll
    for (int r = 0; r < KI; r++) { x[r] = (lane < done ? y[r] : x[r]) * dinv; y[r] = x[r]; }
    // ---- backward: row i of L across the lanes k < i ----
    const int rpos = img + lane, rneg = img + 63 - lane;
    auto load_rows = [&](int i0, double (&o)[APEX_BLK]) { // rows i0, i0 - 1, ..., i0 - APEX_BLK + 1 (row 0 has no entries: any slot)
#pragma unroll
        for (int d = 0; d < APEX_BLK; d++) { const int i = max(i0 - d, 1); o[d] = i >= 32 ? g_dyn[rpos + (63 - i) * 65] : g_dyn[rneg + i * 65]; }
    };
    load_rows(63, a[0]);
    static_for<0, NB, 1>([&](auto Bc) {
        constexpr int b = decltype(Bc)::value, i0 = 63 - b * APEX_BLK;
        if constexpr (b + 1 < NB) load_rows(i0 - APEX_BLK, a[(b + 1) & 1]);
        if (i0 - APEX_BLK + 1 < na) { // (wavefront-uniform; else every row of this block lies beyond the apex: y keeps z there -- zeros)
            static_for<0, APEX_BLK, 1>([&](auto Dc) {
                constexpr int d = decltype(Dc)::value, i = i0 - d;
                if constexpr (i >= 1) {
#pragma unroll
                    for (int r = 0; r < KI; r++) {
                        const double xi = rdlane_d(x[r], i);
                        y[r] = wrlane_d<i>(y[r], xi);
                        x[r] = x[r] - a[b & 1][d] * xi;
                    }
                }
            });
        }
    });
#pragma unroll
    for (int r = 0; r < KI; r++) y[r] = lane == 0 ? x[r] : y[r]; // (lane 0 is never a pivot of the backward loop and never dead)
    if (lane < na) stK<KI>(ws, n0 + lane, y);
}

// ---------------- lambda = W z (ref scale :485-507); ends with a barrier ----------------
// CONES_ONLY: the LP rows were done by the caller inside a fused pass (same product, out[i] = lpw[i] * zz[i])
template <int T, bool CONES_ONLY = false>
static __device__ __noinline__ __attribute__((not_tail_called)) void dev_scale(int ps, gcdbl_p W, gcdbl_p zz, gdbl_p out) {
    ps = uni(ps); W = uni_ptr(W); zz = uni_ptr(zz); out = uni_ptr(out);
    const DevPat &P = c_pat[ps];
    gcdbl_p lpw = W + P.w_lpw, csc = W + P.w_csc, qv = W + P.w_qv;
    if constexpr (!CONES_ONLY) for_t_pre<T, 8>(P.l, [&](int i) { return V2{lpw[i], zz[i]}; }, [&](int i, const V2 &r) { out[i] = r.a * r.b; });
    struct SC { D4 z, q; double a, eta; };
    for_tiny<T>(P, [&](const Tiny &t) { gcdbl_p cs = csc + t.c * CSC_STRIDE; return SC{tiny_rows(zz, t), tiny_rows(qv, t), cs[CS_A], cs[CS_ETA]}; },
                [&](const Tiny &t, const SC &r) { // (the generic body below with the rows in registers)
        double zeta = 0.;
#pragma unroll
        for (int k = 1; k < TINY_D; k++) if (k < t.d) zeta += r.q.v[k] * r.z.v[k];
        const double z0 = r.z.v[0], factor = z0 + zeta / (1. + r.a);
#pragma unroll
        for (int k = 1; k < TINY_D; k++) if (k < t.d) out[t.o + k] = r.eta * (r.z.v[k] + factor * r.q.v[k]);
        out[t.o] = r.eta * (r.a * z0 + zeta);
    });
    struct SW { double z1, q1, z0, a, eta; };
    for_wave<T, wave_cones_in_flight<T>()>(P, [](const WCone &, int) { return NoPre{}; },
        [&](const WCone &b, int lane, NoPre) { gcdbl_p cs = csc + b.c * CSC_STRIDE; return SW{wrow(zz, b, lane, 1), wrow(qv, b, lane, 1), zz[b.o], cs[CS_A], cs[CS_ETA]}; },
        [&](const WCone &b, int lane, NoPre, const SW &r) {
        const int k = 1 + lane;
        double zeta = 0.;
        if (k < b.d) zeta += r.q1 * r.z1;
        zeta = wave_reduce<OpSum>(zeta);
        const double factor = r.z0 + zeta / (1. + r.a);
        if (k < b.d) out[b.o + k] = r.eta * (r.z1 + factor * r.q1);
        if (lane == 0) out[b.o] = r.eta * (r.a * r.z0 + zeta);
    });
    for_cones_rest<T>(ps, [&](int c, auto G, int lane) {
        constexpr int g = decltype(G)::value;
        const int o = P.cone_off[c], d = P.cq[c];
        gcdbl_p cs = csc + c * CSC_STRIDE;
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
}

// ---------------- bringToCone (ref :761-805): s = sgn*r shifted into the cone ----------------
template <int T>
static __device__ __noinline__ __attribute__((not_tail_called)) void dev_bring_to_cone(int ps, gcdbl_p r, double sgn, gdbl_p s) {
    ps = uni(ps); r = uni_ptr(r); s = uni_ptr(s);
    const DevPat &P = c_pat[ps];
    int phase = 0;
    __syncthreads();
    double a = -GAMMA;
    FOR_T(i, P.l) { const double ri = sgn * r[i]; if (ri <= 0. && -ri > a) a = -ri; }
    for_cones<T>(ps, [&](int c, auto G, int lane) {
        constexpr int g = decltype(G)::value;
        const int o = P.cone_off[c], d = P.cq[c];
        double t = 0.;
        for (int k = 1 + lane; k < d; k += g) t += r[o + k] * r[o + k];
        t = grp_sum<g>(t);
        const double cres = sgn * r[o] - sqrt(t);
        if (cres <= 0. && -cres > a) a = -cres;
    });
    a = blk_reduce1<OpMax, T>(phase, a) + 1.;
    FOR_T(i, P.m) s[i] = sgn * r[i] + (i < P.l ? a : 0.);
    __syncthreads();
    FOR_T(c, P.nc) s[P.cone_off[c]] += a;
    __syncthreads();
}

// ---------------- lineSearch (ref :1380-1469); result to every thread ----------------
// LP_DONE: the caller has formed the LP rows' min(ds / lam), min(dz / lam) per thread inside a fused pass (rmin0, smin0)
template <int T, bool LP_DONE = false>
static __device__ __noinline__ __attribute__((not_tail_called)) double dev_line_search(int ps, gcdbl_p W, double tau, double dtau,
                                               double kap, double dkap, double rmin0 = DBL_MAX, double smin0 = DBL_MAX) {
    ps = uni(ps); W = uni_ptr(W);
    const DevPat &P = c_pat[ps];
    gcdbl_p lam = W + P.w_lam, ds = W + P.w_dsw, dz = W + P.w_wdz;
    const int l = P.l;
    int phase = 0;
    __syncthreads();
    double rmin = rmin0, smin = smin0, cstep = 0., bad = 0.;
    if constexpr (!LP_DONE) for_t_pre<T, 4>(l, [&](int i) { return V3{lam[i], ds[i], dz[i]}; },
                    [&](int i, const V3 &r) { rmin = fmin(rmin, r.b / r.a); smin = fmin(smin, r.c / r.a); });
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
    struct LS { D4 lam, ds, dz; };
    for_tiny<T>(P, [&](const Tiny &t) { return LS{tiny_rows(lam, t), tiny_rows(ds, t), tiny_rows(dz, t)}; }, [&](const Tiny &t, const LS &r) {
        // cone_step with the rows in registers (same operations in the same order)
        const int d = t.d;
        double l1 = 0.;
#pragma unroll
        for (int k = 1; k < TINY_D; k++) if (k < d) l1 += r.lam.v[k] * r.lam.v[k];
        const double lknorm2 = r.lam.v[0] * r.lam.v[0] - l1;
        if (lknorm2 <= 0.) { bad = 1.; return; }
        const double lknorm = sqrt(lknorm2), inv = 1. / lknorm, lk0 = r.lam.v[0] / lknorm;
        double ld = 0., lz = 0.;
#pragma unroll
        for (int k = 1; k < TINY_D; k++) if (k < d) { const double lb = r.lam.v[k] / lknorm; ld += lb * r.ds.v[k]; lz += lb * r.dz.v[k]; }
        const double lds = lk0 * r.ds.v[0] - ld, ldz = lk0 * r.dz.v[0] - lz;
        const double rho0 = inv * lds, fr = (lds + r.ds.v[0]) / (lk0 + 1.);
        const double sig0 = inv * ldz, fs = (ldz + r.dz.v[0]) / (lk0 + 1.);
        double rn = 0., sn = 0.;
#pragma unroll
        for (int k = 1; k < TINY_D; k++) if (k < d) {
            const double lb = r.lam.v[k] / lknorm;
            const double rr = inv * (r.ds.v[k] - fr * lb), ss = inv * (r.dz.v[k] - fs * lb);
            rn += rr * rr; sn += ss * ss;
        }
        cstep = fmax(cstep, fmax(0., fmax(sqrt(sn) - sig0, sqrt(rn) - rho0)));
    });
    struct LW { double lam1, ds1, dz1, lam0, ds0, dz0; };
    for_wave<T, wave_cones_in_flight<T>()>(P, [](const WCone &, int) { return NoPre{}; },
        [&](const WCone &b, int lane, NoPre) { return LW{wrow(lam, b, lane, 1), wrow(ds, b, lane, 1), wrow(dz, b, lane, 1), lam[b.o], ds[b.o], dz[b.o]}; },
        [&](const WCone &b, int lane, NoPre, const LW &r) { // cone_step with the lane's row in registers
        const bool act = 1 + lane < b.d;
        double l1 = 0.;
        if (act) l1 += r.lam1 * r.lam1;
        l1 = wave_reduce<OpSum>(l1);
        const double lknorm2 = r.lam0 * r.lam0 - l1;
        if (lknorm2 <= 0.) { bad = 1.; return; }
        const double lknorm = sqrt(lknorm2), inv = 1. / lknorm, lk0 = r.lam0 / lknorm;
        double ld = 0., lz = 0.;
        if (act) { const double lb = r.lam1 / lknorm; ld += lb * r.ds1; lz += lb * r.dz1; }
        ld = wave_reduce<OpSum>(ld); lz = wave_reduce<OpSum>(lz);
        const double lds = lk0 * r.ds0 - ld, ldz = lk0 * r.dz0 - lz;
        const double rho0 = inv * lds, fr = (lds + r.ds0) / (lk0 + 1.);
        const double sig0 = inv * ldz, fs = (ldz + r.dz0) / (lk0 + 1.);
        double rn = 0., sn = 0.;
        if (act) {
            const double lb = r.lam1 / lknorm;
            const double rr = inv * (r.ds1 - fr * lb), ss = inv * (r.dz1 - fs * lb);
            rn += rr * rr; sn += ss * ss;
        }
        rn = wave_reduce<OpSum>(rn); sn = wave_reduce<OpSum>(sn);
        cstep = fmax(cstep, fmax(0., fmax(sqrt(sn) - sig0, sqrt(rn) - rho0)));
    });
    for_cones_rest<T>(ps, [&](int c, auto G, int lane) {
        bool sk;
        const double st = cone_step(P.cone_off[c], P.cq[c], G, lane, sk);
        if (sk) bad = 1.; else cstep = fmax(cstep, st);
    });
    double v4[4] = {-rmin, -smin, cstep, bad};
    blk_reduce<OpMax, T, 4>(phase, v4);
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
    __syncthreads();
    return fmin(fmax(alpha, STEPMIN), STEPMAX);
}

// lineSearch for a pattern WITHOUT second-order cones, inlined (no call: the caller keeps per-row values in registers across it).
// rmin, smin: the thread's min(ds / lam), min(dz / lam) over its LP rows; same arithmetic as dev_line_search with nc = 0.
template <int T>
__device__ __forceinline__ double lp_line_search(double rmin, double smin, int l, double tau, double dtau, double kap, double dkap) {
    int phase = 0;
    __syncthreads();
    double v4[4] = {-rmin, -smin, 0., 0.};
    blk_reduce<OpMax, T, 4>(phase, v4);
    rmin = -v4[0]; smin = -v4[1];
    double alpha;
    if (l > 0) {
        const double eps = 1e-13;
        if (-smin > -rmin) alpha = smin < 0. ? 1. / (-smin) : 1. / eps;
        else alpha = rmin < 0. ? 1. / (-rmin) : 1. / eps;
    } else alpha = 10.;
    const double mtd = -tau / dtau, mkd = -kap / dkap;
    if (mtd > 0. && mtd < alpha) alpha = mtd;
    if (mkd > 0. && mkd < alpha) alpha = mkd;
    __syncthreads();
    return fmin(fmax(alpha, STEPMIN), STEPMAX);
}

// ---------------- checkExitConditions (ref :526-641), thread 0 only, on g_S.wi ----------------
__device__ __forceinline__ int dev_check_exit(bool reduced) {
    DevInfo &wi = g_S.wi;
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
}
// Information::isBetterThan (ref :23-68): is g_S.wi better than g_S.bi ?
__device__ __forceinline__ bool dev_better_than() {
    const DevInfo &a = g_S.wi, &o = g_S.bi;
    const bool gap_ok = a.gap > 0. && o.gap > 0. && a.gap < o.gap;
    const bool mu_ok = a.mu > 0. && a.mu < o.mu;
    if (a.has_pinfres && a.kapovert > 1.) {
        if (o.has_pinfres) return gap_ok && (a.pinfres > 0. && a.pinfres < o.pres) && mu_ok;
        return gap_ok && mu_ok;
    }
    return gap_ok && (a.pres > 0. && a.pres < o.pres) && (a.dres > 0. && a.dres < o.dres) &&
           (a.kapovert > 0. && a.kapovert < o.kapovert) && mu_ok;
}
__device__ __forceinline__ void restore_scalars() { // w = w_best (scalars), counters kept
    const int nf = g_S.wi.n_factor, ns = g_S.wi.n_ldlsolve, nw = g_S.wi.n_sweep;
    g_S.wi = g_S.bi; g_S.wi.n_factor = nf; g_S.wi.n_ldlsolve = ns; g_S.wi.n_sweep = nw;
}

// Slice table `which` of the current pattern: the LDS copy (NLDS >= 1) or the one in global memory.
#define LDS_TABLE(at) (reinterpret_cast<const PackedSlice *>(g_dyn + P.lds_tab) + (at))

// states of the solve program
enum Stage { ST_FACTOR = 0, ST_KKT_INIT1, ST_KKT_INIT2, ST_RESID, ST_KKT1, ST_KKT_AFF, ST_KKT_COMB, ST_DONE };

// ============================================================================================
// One instance, whole solve.  Follows reference Solver::solve (src/eicos.cpp:848-1262).
// The solve is a state machine over "stages"; every stage is its own (single call site,
// noinline) function so that the register allocator sees one stage at a time -- the
// straight-line form keeps ~45 per-thread array base addresses live and spills heavily.
// ============================================================================================
#define STAGE_PROLOGUE                                                                                  \
    ps = uni(ps); I = uni_ptr(I); W = uni_ptr(W);                                                       \
    const DevPat &P = c_pat[ps];                                                                        \
    const int n = P.n, p = P.p, m = P.m, l = P.l, N = P.N, np = P.n + P.p;                              \
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                                      \
    DevInfo &wi = g_S.wi;                                                                               \
    int phase = 0;                                                                                      \
    (void)n; (void)p; (void)m; (void)l; (void)N; (void)np; (void)lane; (void)wave; (void)phase; (void)wi;

// ---------------- ST_FACTOR: numeric LDL' (replaces ldlt.factorize, ref :900,1164) ----------------
// Wg = the workgroup's workspace slab.  Left-looking sliced-ELL program over the level schedule (host: plans.cpp, api.cpp).
template <int T, int NLDS, bool I16, bool DEFER>
static __device__ __noinline__ __attribute__((not_tail_called)) void stage_factor(int ps, gdbl_p Wg) {
    ps = uni(ps); Wg = uni_ptr(Wg);
    const DevPat &P = c_pat[ps];
    const int tid = threadIdx.x;
    gdbl_p UF = Wg + P.w_UF, D = Wg + P.w_D, invD = Wg + P.w_invD;
    ubdbl_p U = [&] { if constexpr (EICOS_UBL != 0) return (ubdbl_p)(g_dyn + P.ub_lds); else return (ubdbl_p)(Wg + P.w_UB); }(); // pa/pb index UB slots
    gcdbl_p Kt = Wg + P.w_Kt;    // KKT entries in target order (solve prologue + updateKKTScalings)
    gdbl_p Kimg = Wg + P.w_Kimg; // hybrid: targets of the top block go to its tile image
    __syncthreads();
    unsigned long long tk0_ = (tid == 0) ? wall_clock64() : 0ull;
    const int ns = P.fac_ns;
    if (ns > 0) { // (empty problem, dim_K = 0: nothing to factorise, and no slice table to decode)
    // Phase A of a level: target value = K entry - sum over pairs U[i,k] * L[j,k]  (U = UB slots, L = UF slots);
    // diagonal targets give D and 1/D, the others U[i,j].  Phase B (after a barrier, D of the level is known):
    // L[i,j] = U[i,j] / D[j] into the forward slots.  Slices of one level are independent.
    // Everything that does not depend on factor values (pair indices, the K entry, destinations) is loaded
    // FAC_DEPTH slices ahead, across the level barriers, so a level costs one dependent gather round trip
    // per phase instead of an index load + gather + source-index load + value load chain.
    struct FSlot { int row0, cnt, lg, K, off, lanes, newlev, last, more, cont; int ia[ELL_KMAX], ib[ELL_KMAX], ik[ELL_KMAX]; double kv; int dst; } q[FAC_DEPTH];
    // Deferred L (DevPat::fac_defer; needs the LDS solve vector, idle during the factorisation, as a mirror of 1/D): a pair is
    // U[i,k] * (U[j,k] * (1/D[k])) with 1/D[k] gathered from LDS -- the product in brackets is exactly the stored L[j,k] -- so the
    // forward-order copy of L is not needed before the sweeps: no phase B, ONE barrier per level, L written in one pass at the end
    // (DEFER = DevPat::fac_defer, a template parameter: the stored-L form must not pay for the pivot-column indices and the third operand)
    constexpr bool defer = DEFER;
    // 1/D[k] of a pair's pivot column: from the LDS mirror ([0, N] at the start of the dynamic LDS, slot fac_kpad = 0 for padding
    // pairs), or -- kernels without an LDS vector: only the debug entry, api.cpp sets fac_defer with NLDS >= 1 -- from the array itself
    auto inv_of = [&](int k) -> double { if constexpr (NLDS >= 1) return g_dyn[k]; else return invD[k]; };
    // the pair's second operand: deferred -- U[j,k] (times 1/D[k] from the mirror); stored L -- the forward array
    auto ld_l = [&](int i) -> double { if constexpr (defer) return ld_u32((ubcdbl_p)U, i); else return ld_u32((gcdbl_p)UF, i); };
    double carry = 0.; // partial sum of targets cut into sub-slices
    const bool tab_lds = NLDS >= 1 && P.lm_fac >= 0; // slice table staged in LDS by k_solve (no global round trip per slice)
    // slice descriptors: from the LDS copy, or (table not staged) from global memory one slice further ahead than
    // the loads that need them, so that their round trip is not on the path either
    auto fmeta = [&](int sidx) { return tab_lds ? slice_at(reinterpret_cast<const PackedSlice *>(g_dyn + P.lds_tab) + P.lm_fac, sidx) : slice_at(P.fac_sl, sidx); };
    auto fload = [&](const Sl &nm, FSlot &o) {
        o.row0 = nm.row0; o.cnt = nm.cnt; o.lg = nm.lg; o.K = nm.K; o.off = nm.off;
        o.newlev = nm.newlev; o.last = nm.last;
        o.more = nm.more; o.cont = nm.cont;
        o.lanes = o.cnt << o.lg;
        const bool act = tid < o.lanes;
        if constexpr (I16) { // the lane's four (pa, pb) pairs packed as eight 16-bit slot numbers: one 16-byte load (+ 8 bytes: the pivot columns)
            const uint4 w = ld_u32(reinterpret_cast<const uint4 EICOS_GLOBAL *>(P.fac_p16), act ? nm.off16 + tid : P.fac_d16);
            uint2 wk = {0u, 0u};
            if constexpr (defer) wk = ld_u32(reinterpret_cast<const uint2 EICOS_GLOBAL *>(P.fac_k16), act ? nm.off16 + tid : P.fac_d16);
            o.ia[0] = w.x & 0xffffu; o.ia[1] = w.x >> 16; o.ia[2] = w.y & 0xffffu; o.ia[3] = w.y >> 16;
            o.ib[0] = w.z & 0xffffu; o.ib[1] = w.z >> 16; o.ib[2] = w.w & 0xffffu; o.ib[3] = w.w >> 16;
            o.ik[0] = wk.x & 0xffffu; o.ik[1] = wk.x >> 16; o.ik[2] = wk.y & 0xffffu; o.ik[3] = wk.y >> 16;
        } else {
#pragma unroll
            for (int u = 0; u < ELL_KMAX; u++) {
                const int slot = (act && u < o.K) ? o.off + u * o.lanes + tid : P.fac_slots; // dummy pair: 0 * 0
                o.ia[u] = P.fac_pa[slot]; o.ib[u] = P.fac_pb[slot]; o.ik[u] = defer ? P.fac_pk[slot] : 0;
            }
        }
        const int t = act ? o.row0 + (tid >> o.lg) : 0;
        o.kv = ld_u32(Kt, t);
        o.dst = ld_u32(P.fac_dst, t);
    };
    // a diagonal target: pivot (with the optional dynamic regularisation, extension N4) -> D, 1/D; zero pivot -> fatal
    auto pivot = [&](int dst, double val) {
        const int e = -dst - 1, j = e & (DIAG_POS - 1); // -(j+1), plus DIAG_POS when the quasi-definite sign of pivot j is +
        if (g_S.dyn_delta > 0.) { // ECOS-style dynamic regularisation, off by default
            const double sg = (e & DIAG_POS) ? 1. : -1.;
#ifdef EICOS_TRACE_DYNREG
            if (sg * val <= g_S.dyn_eps) printf("[dynreg scalar] blk %d j %d val %.6e sg %.0f\n", (int)blockIdx.x, j, val, sg);
#endif
            if (sg * val <= g_S.dyn_eps) val = sg * g_S.dyn_delta;
        }
        if (val == 0.) g_S.fl[FL_FATAL] = 1; // (Eigen: NumericalIssue)
        const double iv = 1. / val;
        D[j] = val; invD[j] = iv;
        if constexpr (NLDS >= 1) { if (defer) g_dyn[j] = iv; }
    };
    // ---- level 0 (the leaves of the elimination tree: two thirds of the nodes of an MPC pattern) has no pairs at all:
    // D_j = K_jj, U_ij = K_ij, L_ij = K_ij / D_j.  Two streaming passes over its targets (diagonals first, host: api.cpp)
    // instead of a slice step per 256 targets: coalesced reads of the K stream, eight targets per thread in flight ----
    const int sbeg = P.fac_s1;
    if constexpr (NLDS >= 1) { if (defer) { if (tid == 0) g_dyn[P.fac_kpad] = 0.; __syncthreads(); } } // (what padding pairs read; the global array keeps a 0 there)
    if (P.fac_nt0 > 0) {
        for_t_pre<T, 8>(P.fac_nd0, [&](int t) { return IV1{ld_u32(P.fac_dst, t), ld_u32(Kt, t)}; }, [&](int t, const IV1 &r) {
            if (r.i >= IMG_BASE) Kimg[r.i - IMG_BASE] = r.a; else pivot(r.i, r.a);
        });
        __syncthreads();
        struct F1 { int dst, dstF; double kv, d; };
        for_t_pre<T, 8>(P.fac_nt0 - P.fac_nd0, [&](int q_) {
            const int t = P.fac_nd0 + q_;
            const int col = ld_u32(P.fac_col, t);
            return F1{ld_u32(P.fac_dst, t), ld_u32(P.fac_dstF, t), ld_u32(Kt, t), defer ? inv_of(col) : invD[col]};
        }, [&](int, const F1 &r) {
            if (r.dst >= IMG_BASE) { Kimg[r.dst - IMG_BASE] = r.kv; return; }
            U[r.dst] = r.kv;
            if (r.dstF >= 0) UF[r.dstF] = r.kv * r.d;
        });
        __syncthreads();
    }
#ifdef EICOS_FAC_TICKS
    unsigned long long tq_ = (tid == 0) ? wall_clock64() : 0ull;
    if (tid == 0) g_S.tick[8] += tq_ - tk0_;
#define FTICK(slot) do { if (tid == 0) { const unsigned long long t1_ = wall_clock64(); g_S.tick[slot] += t1_ - tq_; tq_ = t1_; } } while (0)
#else
#define FTICK(slot) do {} while (0)
#endif
    if (sbeg < ns) {
#pragma unroll
    for (int d = 0; d < FAC_DEPTH; d++) fload(fmeta(min(sbeg + d, ns - 1)), q[d]);
    Sl pm = fmeta(min(sbeg + FAC_DEPTH, ns - 1)); // descriptor of the next slice to be loaded
    int lvl_t0 = 0;
    double gu[2][ELL_KMAX], gl[2][ELL_KMAX], gk[2][ELL_KMAX];
    static_assert(FAC_DEPTH % 2 == 0, "operand register sets alternate with the queue slot");
    bool have = false;
    for (int s0 = sbeg; s0 < ns; s0 += FAC_DEPTH) {
#pragma unroll
        for (int d = 0; d < FAC_DEPTH; d++) {
            const int sidx = s0 + d;
            if (sidx >= ns) break;
            const FSlot c = q[d];
            fload(pm, q[d]);
            pm = fmeta(min(sidx + FAC_DEPTH + 1, ns - 1));
            if (c.newlev) lvl_t0 = c.row0;
            const int lvl_t1 = c.row0 + c.cnt; // targets of a level are one contiguous range
            const bool act = tid < c.lanes;
            // gathered factor values: slices of one level are independent, so the next slice's gathers are issued before this
            // slice's arithmetic waits on its own (one L2 round trip per level, not per slice).  Two operand register sets, by
            // the parity of the slice's queue slot: no copies between a slice's gathers and their use
            double (&cu)[ELL_KMAX] = gu[d & 1], (&cl)[ELL_KMAX] = gl[d & 1], (&ck)[ELL_KMAX] = gk[d & 1];
            if (!have) {
#pragma unroll
                for (int u = 0; u < ELL_KMAX; u++) { cu[u] = ld_u32((ubcdbl_p)U, c.ia[u]); cl[u] = ld_l(c.ib[u]); ck[u] = defer ? inv_of(c.ik[u]) : 1.; }
            }
            have = !c.last;
            { // UNCONDITIONAL (when the next slice opens a new level its operands are not final yet: they are fetched again
              // after the barrier, `have` = false): with the same number of loads in flight on every path the compiler can
              // wait for this slice's operands with s_waitcnt vmcnt(8) instead of draining the queue at every slice
                const FSlot &nx = q[(d + 1) % FAC_DEPTH];
#pragma unroll
                for (int u = 0; u < ELL_KMAX; u++) {
                    gu[(d + 1) & 1][u] = ld_u32((ubcdbl_p)U, nx.ia[u]); gl[(d + 1) & 1][u] = ld_l(nx.ib[u]);
                    gk[(d + 1) & 1][u] = defer ? inv_of(nx.ik[u]) : 1.;
                }
            }
            double acc = 0.;
#pragma unroll
            for (int u = 0; u < ELL_KMAX; u++) acc = madd(acc, cu[u], defer ? cl[u] * ck[u] : cl[u]); // (deferred: cl * ck IS the stored L[j,k])
            acc = grp_reduce_to_lane0(acc, c.lg);
            if (c.cont) acc += carry;
            if (c.more) carry = acc;
            else if (act && (tid & ((1 << c.lg) - 1)) == 0) {
                const double val = c.kv - acc;
                if (c.dst >= IMG_BASE) Kimg[c.dst - IMG_BASE] = val; // hybrid: K - (updates from the columns below the top block) -> input of the tile factorisation
                else if (c.dst < 0) pivot(c.dst, val);
                else U[c.dst] = val;
            }
            if (c.last && defer) { // the level's U, D and the pivot mirror are final
                FTICK(9);
                // (U in LDS: everything the next level reads -- U, the mirror of 1/D -- went through LDS, so the barrier need not drain the global
                // stores of D / 1/D (read by the sweeps, after the stage's last barrier) nor the static prefetch of the next slices)
                if constexpr (EICOS_UBL != 0 && NLDS >= 1) lds_barrier(); else __syncthreads();
                FTICK(10);
            }
            else if (c.last) {
                FTICK(9);
                __syncthreads();
                FTICK(10);
                // phase B over the level's targets, four per thread in flight.  Two-stage pipeline: the (static) destination
                // indices of the next four targets are loaded behind this round's gathers, so a round exposes ONE memory
                // round trip (the gathers of the just-written U and 1/D), not two
                {
                    constexpr int UBN = 4;
                    const int cnt = lvl_t1 - lvl_t0;
                    int xd[UBN], xf[UBN], xc[UBN];
                    auto lidx = [&](int i0) {
#pragma unroll
                        for (int u = 0; u < UBN; u++) {
                            const int t = lvl_t0 + min(i0 + u * T, cnt - 1);
                            xd[u] = ld_u32(P.fac_dst, t); xf[u] = ld_u32(P.fac_dstF, t); xc[u] = ld_u32(P.fac_col, t);
                        }
                    };
                    lidx(tid);
                    for (int i0 = tid; i0 < cnt; i0 += UBN * T) {
                        int cd[UBN], cf[UBN]; double pu[UBN], pd[UBN];
#pragma unroll
                        for (int u = 0; u < UBN; u++) {
                            cd[u] = xd[u]; cf[u] = xf[u];
                            pu[u] = U[(cd[u] >= 0 && cd[u] < IMG_BASE) ? cd[u] : 0]; pd[u] = invD[xc[u]];
                        }
                        lidx(i0 + UBN * T);
#pragma unroll
                        for (int u = 0; u < UBN; u++) {
                            asm volatile("" : "+v"(pu[u]), "+v"(pd[u])); // (pins the gathers above the branch: a load whose only use sits under a branch is sunk into it)
                            const double o = pu[u] * pd[u];
                            if (i0 + u * T < cnt && cd[u] >= 0 && cf[u] >= 0) UF[cf[u]] = o;
                        }
                    }
                }
                FTICK(11);
                __syncthreads();
                FTICK(10);
            }
        }
    }
    }
    if (defer) { // L = U / D into the forward slots for every entry target beyond level 0 (level 0 wrote its own above): one pass
        struct F2 { int dst, dstF; double u, d; };
        for_t_pre<T, 8>(P.fac_nt - P.fac_nt0, [&](int q_) {
            const int t = P.fac_nt0 + q_, dst = ld_u32(P.fac_dst, t);
            return F2{dst, ld_u32(P.fac_dstF, t), U[(dst >= 0 && dst < IMG_BASE) ? dst : 0], inv_of(ld_u32(P.fac_col, t))};
        }, [&](int, const F2 &r) {
            if (r.dst >= 0 && r.dst < IMG_BASE && r.dstF >= 0) UF[r.dstF] = r.u * r.d;
        });
        FTICK(11);
    }
    }
    if constexpr (NLDS >= 1) { // dense apex: the LDS image of the block's L -- a straight copy of the forward image the passes above have just written
        if (P.apex_na > 0 && P.apex_lds >= 0 && !P.apex_inplace) { // (measured: sharing the region with the head of E -- dead during the sweeps -- and copying the image in at every solve costs the sweeps what it gives the residuals)
            __syncthreads();
            for (int e = tid; e < APEX_IMG; e += T) g_dyn[P.apex_lds + e] = ld_u32((gcdbl_p)UF, P.apex_f + e);
        }
    }
    if (tid == 0) { g_S.wi.n_factor++; g_S.tick[TK_FACTOR] += wall_clock64() - tk0_; }
    __syncthreads();
}

// ============================================================================================
// Tile mode (dense fronts): L is a block-sparse matrix of dense 16 x 16 fp64 tiles (host: tiles.hpp).
// One WAVEFRONT per tile operation; levels of the block dependency graph are separated by workgroup barriers.
//   LC[t] : unit-lower L tile t = (I, J), column-major ((r, c) at 16 c + r)  -- forward sweep + both MFMA operands
//   LR[t] : the same tile row-major ((r, c) at 16 r + c)                      -- backward sweep
//   DL[J] : the strictly lower part of the unit-lower diagonal tile L_JJ, plain row-major; D, invD per slot.  No inverse of a
//           diagonal tile is ever formed: the factorisation (phase 2) and both sweeps solve with L_JJ by substitution.
// v_mfma_f64_16x16x4_f64 lane maps (checked on gfx950, tools/dev/mfma_f64_layout.hip): operand A: lane l holds
// A[row l&15][k l>>4], operand B: B[k l>>4][col l&15], result: C[row (l>>4) + 4 reg][col l&15].  A column-major tile is
// therefore read as four fully coalesced 512-byte loads (element s*64 + l for K-step s), for A and for B alike.
// ============================================================================================
typedef double d4_t __attribute__((ext_vector_type(4)));
// Workgroup-uniform reads of the (read-only) pattern arrays through the CONSTANT address space: s_load into SGPRs.  As a
// vector load the descriptor of the next tile operation would sit in the in-order vmcnt queue BEHIND the tile loads issued
// before it, and waiting for it would drain the whole prefetch queue at every operation.
typedef const int __attribute__((address_space(4))) *cint_p;
typedef int i4_t __attribute__((ext_vector_type(4)));
typedef const i4_t __attribute__((address_space(4))) *cint4_p;
__device__ __forceinline__ cint_p as_const(gint_p p) { return (cint_p)(unsigned long long)p; }
typedef const d4_t EICOS_DATA *gcd4_p;
typedef d4_t EICOS_DATA *gd4_p;
// the four doubles of lane `lane` of tile `t` (tile-internal order of device_types.hpp: two 16-byte loads)
__device__ __forceinline__ d4_t tile_ld(gcdbl_p base, int t, int lane) { return *reinterpret_cast<gcd4_p>(base + (size_t)t * 256 + lane * 4); }

// ---------------- ST_FACTOR, tile mode: left-looking block LDL' (replaces ldlt.factorize, ref :900,1164) ----------------
template <int T, int NLDS>
static __device__ __noinline__ __attribute__((not_tail_called)) int stage_factor_tiles(int ps, gdbl_p I, gdbl_p W, int iter) {
    STAGE_PROLOGUE
    iter = uni(iter);
    constexpr int NW = T / 64;
    // (D, invD of the blocks start at slot tl_base: hybrid keeps the scalar part of the vectors in front)
    gdbl_p LC = W + P.w_LC, LR = W + P.w_LR, DL = W + P.w_DL, D = W + P.w_D + P.tl_base, invD = W + P.w_invD + P.tl_base;
    gcdbl_p Kt = W + P.w_Kimg;
    double *scr = g_dyn + P.tl_scratch + uni(wave) * TILE_SCR; // wave-private 16 x 17 tile in LDS
    const int nbk = P.nb, kq = lane >> 4, lc = lane & 15;
    cint4_p c_fops = (cint4_p)(unsigned long long)P.tl_facops;
    cint_p c_fptr = as_const(P.tl_facptr), c_fin = as_const(P.tl_fin), c_tcol = as_const(P.tl_tcol), c_fl = as_const(P.tl_fin_lev);
    __syncthreads();
    TICK_BEGIN;
    for (int v = 0; v < P.nblev; v++) {
        // ---- phase 1: T = K - sum_K L_IK D_K L_JK' for every target of the level; diagonal targets are factorised.
        // Every wavefront walks ITS flat list of operations (host: build_tile_factor_ops): per target an INIT operation (the
        // K tile, already in the MFMA result order) and its pairs, the last one flagged END; the loads of the next TILE_FPF
        // operations are in flight across target boundaries (unconditional: padding and INIT operations load tile 0 too) ----
        const int o0 = c_fptr[v * NW + wave], o1 = c_fptr[v * NW + wave + 1];
        d4_t qa[TILE_FPF], qb[TILE_FPF], qd[TILE_FPF];
        auto load = [&](int o, d4_t &xa, d4_t &xb, d4_t &xd) {
            const i4_t op = c_fops[min(o, o1 - 1)];
            xa = tile_ld((op.w & FOP_INIT) ? Kt : (gcdbl_p)LC, op.x, lane); xb = tile_ld(LC, op.y, lane);
#pragma unroll
            for (int st = 0; st < 4; st++) xd[st] = D[op.z * 16 + 4 * st + kq];
        };
        if (o0 < o1) {
#pragma unroll
            for (int u = 0; u < TILE_FPF; u++) load(o0 + u, qa[u], qb[u], qd[u]);
        }
        d4_t acc = {0., 0., 0., 0.};
        for (int o = o0; o < o1; o += TILE_FTRIP) {
#pragma unroll
          for (int uu = 0; uu < TILE_FTRIP; uu++) {
            const int u = uu % TILE_FPF;
            const i4_t op = c_fops[o + uu];
            const d4_t a = qa[u], b = qb[u], dd = qd[u];
            load(o + uu + TILE_FPF, qa[u], qb[u], qd[u]);
            const int fl = op.w, tg = fl >> FOP_SHIFT;
            if (fl & FOP_INIT) acc = (fl & FOP_ZERO) ? d4_t{0., 0., 0., 0.} : a; // (a structurally zero K tile: its load went to image tile 0, which stays cached)
            else if (!(fl & FOP_PAD)) {
#pragma unroll
                for (int st = 0; st < 4; st++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[st], b[st] * dd[st], acc, 0, 0, 0);
            }
            if (!(fl & FOP_END)) continue;
            if (tg >= nbk) { // off-diagonal target: park T in its own L slot, result order = the accumulator as it stands
                *reinterpret_cast<gd4_p>(LR + (size_t)(tg - nbk) * 256 + lane * 4) = acc;
                continue;
            }
            // ---- diagonal target: dense LDL' of the 16 x 16 tile in LDS (lower triangle) ----
            const int J = tg;
#pragma unroll
            for (int r = 0; r < 4; r++) scr[(kq + 4 * r) * 17 + lc] = acc[r];
            for (int j = 0; j < 16; j++) {
                double dj = scr[j * 17 + j];
                if (g_S.dyn_delta > 0.) { // extension (N4): ECOS-style dynamic regularisation, off by default
                    const double sg = (double)P.tl_psign[P.tl_base + J * 16 + j]; // (per slot of the KKT-space vectors)
#ifdef EICOS_TRACE_DYNREG
                    if (sg * dj <= g_S.dyn_eps && lane == 0) printf("[dynreg tile] blk %d level %d J %d j %d slot %d dj %.6e sg %.0f iter %d\n", (int)blockIdx.x, v, J, j, P.tl_base + J * 16 + j, dj, sg, iter);
#endif
                    if (sg * dj <= g_S.dyn_eps) { dj = sg * g_S.dyn_delta; if (lane == 0) scr[j * 17 + j] = dj; }
                }
                if (dj == 0. && lane == 0) g_S.fl[FL_FATAL] = 1; // zero pivot -> fatal (Eigen NumericalIssue)
                const double idj = 1. / dj; // ONE division per column (an fp64 division is ~40 instructions; the scalar program also multiplies by 1/D)
#pragma unroll
                for (int r = 0; r < 4; r++) { // trailing update of the lower triangle with column j of L times the unscaled column j
                    const int rr = kq + 4 * r;
                    if (lc > j && rr >= lc) scr[rr * 17 + lc] -= (scr[rr * 17 + j] * idj) * scr[lc * 17 + j];
                }
                if (lane > j && lane < 16) scr[lane * 17 + j] = scr[lane * 17 + j] * idj; // column j of L
            }
            if (lane < 16) {
                const double dv = scr[lane * 17 + lane];
                D[J * 16 + lane] = dv; invD[J * 16 + lane] = 1. / dv;
            }
            { // L_JJ (strictly lower part, row-major: element e = 16 r + k) for the triangular solves of phase 2 and of the sweeps
                d4_t lv;
#pragma unroll
                for (int i = 0; i < 4; i++) { const int e = lane * 4 + i, r = e >> 4, k = e & 15; lv[i] = (k < r) ? scr[r * 17 + k] : 0.; }
                *reinterpret_cast<gd4_p>(DL + (size_t)J * 256 + lane * 4) = lv;
            }
          }
        }
        __syncthreads();
        // ---- phase 2: L_IJ = T_IJ L_JJ^-T D_J^-1 for the off-diagonal tiles of the level's block columns, as a triangular
        // solve X L_JJ' = T by substitution over the columns (backward stable like the scalar program's dot products; the
        // product with the explicit inverse of L_JJ loses cond(L_JJ) digits, which flips delta-sized pivots of the last
        // blocks).  Lane l holds X[(l >> 4) + 4 reg][l & 15]: column c of X reaches the other columns of its rows through a
        // DPP row broadcast, the multiplier L_JJ[l & 15][c] is zero for c >= l & 15. ----
        const int f1 = c_fl[v + 1];
        struct Fin { d4_t x, l; double idc; } nx; // one tile's inputs: T, L_JJ (2 KB per wavefront, row-major), 1/D of column l & 15
        auto load2 = [&](int q, Fin &o) {
            const int t = c_fin[q], J = c_tcol[t];
            o.x = tile_ld(LR, t, lane); o.l = tile_ld(DL, J, lane);
            o.idc = invD[J * 16 + lc];
        };
        const int q0 = c_fl[v] + uni(wave);
        if (q0 < f1) load2(q0, nx);
        for (int q = q0; q < f1; q += NW) {
            const int t = c_fin[q];
            const Fin cu = nx;
            load2(min(q + NW, f1 - 1), nx); // the next tile's inputs in flight behind this tile's substitution (unconditional, clamped)
            d4_t acc = cu.x;
            // every lane needs row l & 15 of L_JJ: through the wavefront's LDS tile (one global load of the tile per
            // wavefront instead of four; a wavefront's LDS accesses execute in order)
#pragma unroll
            for (int i = 0; i < 4; i++) { const int e = lane * 4 + i; scr[(e >> 4) * 17 + (e & 15)] = cu.l[i]; }
            double lrow[15];
#pragma unroll
            for (int c = 0; c < 15; c++) lrow[c] = scr[lc * 17 + c];
            const double idc = cu.idc;
            auto step = [&](auto cc) {
                constexpr int c = decltype(cc)::value;
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = __builtin_fma(-dpp_row_bcast<c>(acc[r]), lrow[c], acc[r]);
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
            step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
            step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{});
            acc *= idc;
            *reinterpret_cast<gd4_p>(LR + (size_t)t * 256 + lane * 4) = acc; // result order = the accumulator layout
#pragma unroll
            for (int r = 0; r < 4; r++) LC[(size_t)t * 256 + tile_op(kq + 4 * r, lc)] = acc[r];
        }
        __syncthreads();
    }
    if (tid == 0 && P.tile == 1) wi.n_factor++; // (hybrid: counted by the scalar part)
    __syncthreads();
    if (threadIdx.x == 0) { // (hybrid: the tile part is also booked on its own slot, for the phase breakdown of tools/dev/gpu_sweep.py)
        const unsigned long long t1_ = wall_clock64();
        g_S.tick[TK_FACTOR] += t1_ - tk0_;
        if (P.tile == 2) g_S.tick[TK_FA] += t1_ - tk0_;
    }
    if (g_S.fl[FL_FATAL]) return ST_DONE; // ref :901-905,1166-1170 (no backscale)
    return (iter < 0) ? ST_KKT_INIT1 : ST_KKT1;
}

// ---------------- LDL' solve, tile mode: ws <- L^-T D^-1 L^-1 ws in the padded elimination order ----------------
// Forward, block row I:  y_I = Linv_II (b_I - sum_K L_IK y_K);  backward, block column J:  x_J = Linv_JJ' (y_J / D_J - sum_I L_IJ' x_I).
// A tile mat-vec: lane l multiplies its four tile elements by the vector entries 4 s + (l >> 4), s = 0..3, and the four
// lane groups are folded by two cross-lane adds: 2 KB contiguous per wavefront and tile, no index arrays at all.
// The loads of TILE_PF tiles are in flight per wavefront; blocks whose diagonal tile is the identity skip its product.
// NR = 2: two right-hand sides at once (the KKT1 and affine systems of one pass): ws is NR-interleaved, a
// tile is loaded ONCE and multiplies both vectors -- the sweeps stream L once instead of twice.
template <int T, bool LDSBAR, int NR, class WS>
__device__ __forceinline__ void tile_solve(const DevPat &P, gdbl_p W, WS ws0) {
    auto ws = ws0 + (size_t)P.tl_base * NR; // the blocks start at slot tl_base (hybrid: behind the scalar part of the vector)
    constexpr int NW = T / 64;
    gcdbl_p LC = W + P.w_LC, LR = W + P.w_LR, DL = W + P.w_DL, invD = W + P.w_invD + P.tl_base;
    const int lane = threadIdx.x & 63, wave = uni((int)threadIdx.x >> 6), kq = lane >> 4, lc = lane & 15;
    double *scr = g_dyn + P.tl_scratch + wave * TILE_SCR; // wave-private 16 x 17 tile in LDS: L_JJ of the block being closed
    double *part = g_dyn + P.tl_part;                     // partial sums of split blocks: slot q at (q * 16 + row) * NR
    auto bar = [&] { if constexpr (LDSBAR) lds_barrier(); else __syncthreads(); };
    auto fold = [](double v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); };
    // One sweep: per level every wavefront walks ITS flat list of tile operations (host: build_tile_sweeps) -- the tiles
    // of its blocks, each block closed by its diagonal operation -- with the tile loads of the next TILE_PF operations in
    // flight across block boundaries (they do not depend on ws).
    // A level has a second phase when it holds SPLIT blocks (device_types.hpp: TOP_PART): phase 0 = tiles, whole blocks and the parts of
    // split blocks (partial sums -> LDS slots), barrier, phase 1 = the split blocks' diagonal operations (partial sums added in slot order).
    auto sweep = [&](gint_p ops_i, gint_p ptr_g, gint_p end_g, gint_p split_g, gcdbl_p val, gcdbl_p dia, auto bwd) {
        constexpr bool scale = decltype(bwd)::value; // backward: x_J = L_JJ^-T (y_J / D_J - ...), forward: y_J = L_JJ^-1 (b_J - ...)
        cint4_p ops = (cint4_p)(unsigned long long)ops_i;
        cint_p ptr = as_const(ptr_g), endr = as_const(end_g), split = as_const(split_g);
        for (int v = 0; v < P.nblev; v++) {
          const int nph = split[v] ? 2 : 1; // (workgroup-uniform: a scalar load)
          for (int ph = 0; ph < nph; ph++) {
            if (ph) bar(); // the partial sums of phase 0 are in LDS
            const int o0 = ptr[(v * 2 + ph) * NW + wave], o1 = ptr[(v * 2 + ph) * NW + wave + 1]; // (o1 - o0) is a multiple of TILE_STRIP
            const int o1r = endr[(v * 2 + ph) * NW + wave]; // end of the REAL operations: [o1r, o1) is padding for the queue's refills, never executed
            d4_t qv[TILE_PF];
            // unconditional: a conditional load would force s_waitcnt vmcnt(0) at every join and serialise the queue
            auto load = [&](int o, d4_t &x) {
                const i4_t op = ops[min(o, o1 - 1)];
                x = tile_ld((op.w & TOP_DIAG) ? dia : val, op.x, lane);
            };
            double acc[NR];
#pragma unroll
            for (int k = 0; k < NR; k++) acc[k] = 0.;
            if (o0 < o1) {
#pragma unroll
                for (int u = 0; u < TILE_PF; u++) load(o0 + u, qv[u]);
            }
            auto op_step = [&](const int uu, const int o) __attribute__((always_inline)) {
                    const int u = uu % TILE_PF;
                    const i4_t op = ops[o + uu];
                    const d4_t cv = qv[u];
                    load(o + uu + TILE_PF, qv[u]);
                    const int fl = op.w, vb = op.y;
                    if (fl & TOP_PART) { // a part of a split block ends: the partial sum goes to its LDS slot
#pragma unroll
                        for (int k = 0; k < NR; k++) {
                            const double pv = fold(acc[k]); // (every 16-lane row holds the whole block vector)
                            if (lane < 16) part[(op.z * 16 + lane) * NR + k] = pv;
                            acc[k] = 0.;
                        }
                    } else if (!(fl & TOP_DIAG)) {
#pragma unroll
                        for (int st = 0; st < 4; st++) {
                            double y[NR];
                            ldK<NR>(ws, vb * 16 + 4 * st + kq, y);
#pragma unroll
                            for (int k = 0; k < NR; k++) acc[k] = madd(acc[k], cv[st], y[k]);
                        }
                    } else { // close block vb: r = b_B - acc, then the diagonal tile
                        double own[NR], res[NR];
                        ldK<NR>(ws, vb * 16 + lc, own);
                        const double idv = scale ? invD[vb * 16 + lc] : 1.;
                        const int np = fl >> TOP_NPART_SHIFT; // split block: its partial sums, added in slot order (0 for a whole block)
#pragma unroll
                        for (int k = 0; k < NR; k++) {
                            double sum = fold(acc[k]); // (every 16-lane row holds the whole block vector)
                            for (int q = 0; q < np; q++) sum += part[((op.z + q) * 16 + lc) * NR + k];
                            res[k] = (scale ? own[k] * idv : own[k]) - sum;
                            acc[k] = 0.;
                        }
                        if (!(fl & TOP_IDENT)) {
                            // triangular solve with the unit-lower L_JJ by substitution inside the 16-lane row (no explicit inverse:
                            // it costs cond(L_JJ) digits exactly when the scalings spread, in the last passes).  The tile arrives
                            // row-major through the prefetch queue and goes through the wavefront's LDS tile: lane l & 15 needs
                            // its ROW of L_JJ (forward: b_r -= L[r][c] y_c) or its COLUMN (backward: y_k -= L[c][k] x_c).
#pragma unroll
                            for (int i = 0; i < 4; i++) { const int e = lane * 4 + i; scr[(e >> 4) * 17 + (e & 15)] = cv[i]; }
                            double lv[16];
#pragma unroll
                            for (int c = 0; c < 16; c++) lv[c] = scale ? scr[c * 17 + lc] : scr[lc * 17 + c];
                            if constexpr (scale) static_for<15, 0, -1>([&](auto cc) {
                                constexpr int c = decltype(cc)::value;
#pragma unroll
                                for (int k = 0; k < NR; k++) res[k] = __builtin_fma(-dpp_row_bcast<c>(res[k]), lv[c], res[k]);
                            });
                            else static_for<0, 15, 1>([&](auto cc) {
                                constexpr int c = decltype(cc)::value;
#pragma unroll
                                for (int k = 0; k < NR; k++) res[k] = __builtin_fma(-dpp_row_bcast<c>(res[k]), lv[c], res[k]);
                            });
                        }
                        if (lane < 16) stK<NR>(ws, vb * 16 + lane, res);
                        // (a vector in the workspace slab: a serially swept block system -- tiles.cpp -- reads this block's entries from the SAME
                        // wavefront's next operations with no barrier in between: its stores must have landed)
                        if constexpr (!LDSBAR) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    }
            };
            int o = o0;
            for (; o + TILE_STRIP <= o1r; o += TILE_STRIP) { // whole trips of real operations (the loads inside are unconditional)
#pragma unroll
                for (int uu = 0; uu < TILE_STRIP; uu++) op_step(uu, o);
            }
            if (o < o1r) { // the last, partial trip stops at the last real operation (a small block system -- the top block of a hybrid pattern: seven
                           // levels of one block each on lp_bandm -- spent 42 % of its critical path on padding)
#pragma unroll
                for (int uu = 0; uu < TILE_STRIP; uu++) if (o + uu < o1r) op_step(uu, o);
            }
          }
          bar();
        }
    };
#ifdef EICOS_TILE_TICKS // (dev builds: forward against backward sweep, thread 0's clock, on the trace slots of the factor's inner timers)
    unsigned long long tq_ = threadIdx.x == 0 ? wall_clock64() : 0ull;
#endif
    sweep(P.tl_fops, P.tl_fptr, P.tl_fend, P.tl_fsplit, LC, DL, std::false_type{}); // forward: block rows, levels up
#ifdef EICOS_TILE_TICKS
    if (threadIdx.x == 0) { const unsigned long long t1_ = wall_clock64(); g_S.tick[TK_FW1] += t1_ - tq_; tq_ = t1_; }
#endif
    sweep(P.tl_bops, P.tl_bptr, P.tl_bend, P.tl_bsplit, LR, DL, std::true_type{});  // backward: block columns, levels down
#ifdef EICOS_TILE_TICKS
    if (threadIdx.x == 0) g_S.tick[TK_FB] += wall_clock64() - tq_;
#endif
}

// ---------------- G in dense tiles: G x and G' z in ONE pass over the values (DevPat::gt_on, host: api.cpp) ----------------
// One wavefront per row block of 16 rows: lane l = 16 kq + r holds G[r][4 reg + kq] of the current tile (tile-internal
// operand order, two 16-byte loads), gathers the vector entries of its four columns and the z entry of its row, and
//   * accumulates row r of G x over the tiles of the row block (folded over kq at the end: two cross-lane adds),
//   * forms G[r][c] z_r and sums it over the 16 rows of the tile inside the DPP row (column c's partial sum, one per tile).
// A second pass adds the partial sums of every column in a fixed order (gt_cidx): deterministic, no atomics.
// NR right-hand sides share one load of the tile.  getx(column entry of gt_col / gt_colk, k), getz(row, k): the vectors.
// Outputs: gz[i * NR + k] = (G x)_i, gx[j * NR + k] = (G' z)_j.  Ends with a barrier.
template <int T, int NR, class GetX, class GetZ>
__device__ __forceinline__ void g_tile_products(const DevPat &P, gcdbl_p Gt, gint_p colidx, GetX &&getx, GetZ &&getz, gdbl_p gpart, gdbl_p gx, gdbl_p gz) {
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, wave = uni((int)threadIdx.x >> 6), kq = lane >> 4, r = lane & 15;
    cint_p rbptr = as_const(P.gt_rbptr);
    auto fold = [](double v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); };
    // every wavefront owns a contiguous run of row blocks = a contiguous run of tiles: one flat loop with the loads of the
    // next GT_PF tiles in flight (across row-block boundaries; clamped, unconditional)
    constexpr int GT_PF = 4;
    const int rb0 = (int)((long long)wave * P.gt_nrb / NW), rb1 = (int)((long long)(wave + 1) * P.gt_nrb / NW);
    const int tb = rbptr[rb0], te = rbptr[rb1];
    d4_t qg[GT_PF]; int qc[GT_PF][4];
    auto load = [&](int t, d4_t &g, int (&c)[4]) {
        g = tile_ld(Gt, t, lane);
#pragma unroll
        for (int q = 0; q < 4; q++) c[q] = colidx[t * 16 + 4 * q + kq];
    };
    if (tb < te) {
#pragma unroll
        for (int d = 0; d < GT_PF; d++) load(min(tb + d, te - 1), qg[d], qc[d]);
    }
    int rb = rb0, tend = rb0 < rb1 ? rbptr[rb0 + 1] : 0;
    double zr[NR], racc[NR];
    auto open_rb = [&] { // z entries of the row block's 16 rows
#pragma unroll
        for (int k = 0; k < NR; k++) { zr[k] = (rb < rb1 && rb * 16 + r < P.m) ? getz(rb * 16 + r, k) : 0.; racc[k] = 0.; }
    };
    auto close_rb = [&] { // (G x) of the row block: fold the four column groups, one store per row
#pragma unroll
        for (int k = 0; k < NR; k++) { const double v = fold(racc[k]); if (lane < 16 && rb * 16 + r < P.m) gz[(size_t)(rb * 16 + r) * NR + k] = v; }
    };
    open_rb();
    for (int t0 = tb; t0 < te; t0 += GT_PF) {
#pragma unroll
        for (int d = 0; d < GT_PF; d++) {
            const int t = t0 + d;
            if (t >= te) break;
            while (t == tend) { close_rb(); rb++; tend = rbptr[rb + 1]; open_rb(); } // (row blocks without tiles write zeros)
            const d4_t g = qg[d];
            int c[4];
#pragma unroll
            for (int q = 0; q < 4; q++) c[q] = qc[d][q];
            load(min(t + GT_PF, te - 1), qg[d], qc[d]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int k = 0; k < NR; k++) {
                    racc[k] = madd(racc[k], g[q], getx(c[q], k));
                    const double cs = grp_reduce_to_lane0(g[q] * zr[k], 4); // over the 16 rows of the tile: lane r = 0 of the DPP row
                    if (r == 0) gpart[(size_t)(t * 16 + 4 * q + kq) * NR + k] = cs;
                }
            }
        }
    }
    for (; rb < rb1; ) { close_rb(); rb++; if (rb < rb1) open_rb(); } // the last row block (and trailing ones without tiles)
    __syncthreads();
    typedef const i4_t EICOS_GLOBAL *gi4_p;
    FOR_T(j, P.n) { // eight contributions per column (host pads gt_cidx with -1): all loads in flight, fixed order of the adds
        const i4_t e0 = *reinterpret_cast<gi4_p>(P.gt_cidx + (size_t)j * 8), e1 = *reinterpret_cast<gi4_p>(P.gt_cidx + (size_t)j * 8 + 4);
        double v[8][NR];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int e = q < 4 ? e0[q] : e1[q - 4]; // (-1: padding -- the load is unconditional, the value is dropped)
#pragma unroll
            for (int k = 0; k < NR; k++) { const double w = gpart[(size_t)max(e, 0) * NR + k]; v[q][k] = e < 0 ? 0. : w; }
        }
#pragma unroll
        for (int k = 0; k < NR; k++) {
            double a = 0.;
#pragma unroll
            for (int q = 0; q < 8; q++) a += v[q][k];
            gx[(size_t)j * NR + k] = a;
        }
    }
    __syncthreads();
}

// the buffer set `which` of the iterate (ShI::cur / ShI::best): 0 = instance slab, 1 = workspace slab (same spacing of y and z in both)
__device__ __forceinline__ IterBuf iter_buf(const DevPat &P, gdbl_p I, gdbl_p W, int which) {
    which = uni(which);
    return which ? IterBuf{W + P.w_bx, W + P.w_by, W + P.w_bz, W + P.w_bs} : IterBuf{I + P.i_x, I + P.i_y, I + P.i_z, I + P.i_s};
}

// ---------------- ST_RESID: residuals, statistics, exit logic, scalings ----------------
template <int T, int NLDS, bool I16>
static __device__ __noinline__ __attribute__((not_tail_called)) int stage_resid(int ps, gdbl_p I, gdbl_p W, int iter) {
    STAGE_PROLOGUE
    iter = uni(iter);
    gcdbl_p cagv = I + P.i_cag, rAv = I + P.i_rA, rGv = I + P.i_rG;
    gdbl_p cv = I + P.i_c, hv = I + P.i_h, bv = I + P.i_b, Vv = I + P.i_Vv;
    gdbl_p lam = W + P.w_lam, rz = W + P.w_rz, rhs2k = W + P.w_rhs2k;
    gdbl_p lpw = W + P.w_lpw, lpv = W + P.w_lpv, csc = W + P.w_csc, qv = W + P.w_qv;
    __syncthreads();
    const IterBuf it = iter_buf(P, I, W, g_S.cur); // (after the barrier: the previous stage's thread 0 may just have switched sets)
    gdbl_p wx = it.x, wy = it.y, wz = it.z, wsl = it.s;
    TICK_BEGIN;
    // ---- computeResiduals (ref :643-689) + updateStatistics (ref :691-754) ----
    const double tau = wi.tau;
    double r8[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // hresx2 rx2 cx nx2 | hresy2 ry2 by ny2
    auto tab_cag = [&] { if constexpr (NLDS >= 1) return LDS_TABLE(P.lm_cag); else return P.cag_sl; }();
    auto tab_rA = [&] { if constexpr (NLDS >= 1) return LDS_TABLE(P.lm_rA); else return P.rA_sl; }();
    auto tab_rG = [&] { if constexpr (NLDS >= 1) return LDS_TABLE(P.lm_rG); else return P.rG_sl; }();
    struct Pre2 { double a, b, g; };
    struct Pre3 { double a, b, c, g; };
    const bool gt = P.gt_on != 0; // G in tiles: one pass gives G x (gzv) and G' z (gxv); the ELL plans below then hold A only
    gdbl_p gxv = W + P.w_gx, gzv = W + P.w_gz;
    if (gt) g_tile_products<T, 1>(P, I + P.i_Gt, P.gt_col, [&](int c, int) { return c < 0 ? 0. : wx[c]; }, [&](int i, int) { return wz[i]; },
                                  W + P.w_gpart, gxv, gzv);
    ell_dots<T, I16>(tab_cag, P.cag_ns, P.cag_ns_r, P.cag_idx_yz, P.cag_yz16, P.cag_d16, cagv, wy, P.cag_slots, [&](int j) { return Pre2{cv[j], wx[j], gt ? gxv[j] : 0.}; },
                [&](int j, double s, const Pre2 &pr) { // -G'z - A'y: (y,z) contiguous
        const double hr = -(s + pr.g), c_ = pr.a, xj = pr.b;
        const double r = hr - tau * c_;
        rhs2k[j] = r; // (rx, only ever read as the x part of RHSaffine)
        r8[0] += hr * hr; r8[1] += r * r; r8[2] += c_ * xj; r8[3] += xj * xj;
    });
    ell_dots<T, I16>(tab_rA, P.rA_ns, P.rA_ns_r, P.rA_idx, P.rA_16, P.rA_d16, rAv, wx, P.rA_slots, [&](int r) { return Pre2{bv[r], wy[r], 0.}; },
                [&](int r, double s, const Pre2 &pr) {
        const double b_ = pr.a, yr = pr.b;
        const double rr = s - tau * b_;
        rhs2k[n + r] = -rr; // (-ry: the y part of RHSaffine)
        r8[4] += s * s; r8[5] += rr * rr; r8[6] += b_ * yr; r8[7] += yr * yr;
    });
    blk_reduce<OpSum, T, 8>(phase, r8);
    double q6[6] = {0, 0, 0, 0, 0, 0}; // hresz2 rz2 hz nz2 ns2 gap
    ell_dots<T, I16>(tab_rG, P.rG_ns, P.rG_ns_r, P.rG_idx, P.rG_16, P.rG_d16, rGv, wx, P.rG_slots, [&](int i) { return Pre3{wsl[i], wz[i], hv[i], gt ? gzv[i] : 0.}; },
                [&](int i, double s, const Pre3 &pr) {
        const double si = pr.a, zi = pr.b, h_ = pr.c;
        const double hr = si + (s + pr.g), r = hr - tau * h_;
        rz[i] = r; rhs2k[np + i] = si - r; // (s - rz: the z part of RHSaffine)
        q6[0] += hr * hr; q6[1] += r * r; q6[2] += h_ * zi; q6[3] += zi * zi; q6[4] += si * si; q6[5] += si * zi;
    });
    blk_reduce<OpSum, T, 6>(phase, q6);
    if (tid == 0) {
        wi.iter = iter;
        const double resx0 = g_S.sv[SV_RESX0], resy0 = g_S.sv[SV_RESY0], resz0 = g_S.sv[SV_RESZ0];
        const double hresx = sqrt(r8[0]), nrx = sqrt(r8[1]), nx = sqrt(r8[3]);
        const double hresy = p > 0 ? sqrt(r8[4]) : 0., nry2 = sqrt(r8[5]), ny = sqrt(r8[7]);
        const double hresz = sqrt(q6[0]), nrz2 = sqrt(q6[1]), nz = sqrt(q6[3]), ns = sqrt(q6[4]);
        wi.cx = r8[2]; wi.by = p > 0 ? r8[6] : 0.; wi.hz = q6[2];
        g_S.sv[SV_RT] = wi.kap + wi.cx + wi.by + wi.hz;
        wi.gap = q6[5];
        wi.mu = (wi.gap + wi.kap * wi.tau) / (double)((l + P.nc) + 1);
        wi.kapovert = wi.kap / wi.tau;
        wi.pcost = wi.cx / wi.tau;
        wi.dcost = -(wi.hz + wi.by) / wi.tau;
        if (wi.pcost < 0.) { wi.relgap = wi.gap / (-wi.pcost); wi.has_relgap = 1; }
        else if (wi.dcost > 0.) { wi.relgap = wi.gap / wi.dcost; wi.has_relgap = 1; }
        else wi.has_relgap = 0;
        const double nry = p > 0 ? nry2 / fmax(resy0 + nx, 1.) : 0.;
        const double nrz = nrz2 / fmax(resz0 + nx + ns, 1.);
        wi.pres = fmax(nry, nrz) / wi.tau;
        wi.dres = nrx / fmax(resx0 + ny + nz, 1.) / wi.tau;
        if ((wi.hz + wi.by) / fmax(ny + nz, 1.) < -RELTOL) { wi.pinfres = hresx / fmax(ny + nz, 1.); wi.has_pinfres = 1; }
        if (wi.cx / fmax(nx, 1.) < -RELTOL) {
            wi.dinfres = fmax(hresy / fmax(nx, 1.), hresz / fmax(nx + ns, 1.)); wi.has_dinfres = 1;
        }
        { // per-iteration history (the reference's verbose table, ref :733-753), kept per workspace slot
            gdbl_p tr = W + P.w_trace + (size_t)iter * TRACE_COLS;
            tr[0] = wi.pcost; tr[1] = wi.dcost; tr[2] = wi.gap; tr[3] = wi.pres; tr[4] = wi.dres; tr[5] = wi.kapovert;
            tr[6] = wi.mu; tr[7] = wi.step; tr[8] = wi.sigma; tr[9] = wi.tau; tr[10] = wi.kap; tr[11] = (double)wi.nitref3;
        }
        // ---- safeguard / exit logic (ref :1010-1158) ----
        int action = ACT_CONTINUE, restore = 0, save = 0, code;
        if (iter > 0 && (wi.pres > SAFEGUARD * g_S.sv[SV_PRESPREV] || wi.gap < 0.)) {
            restore_scalars(); restore = 1;
            code = dev_check_exit(true);
            if (code == EX_NOT_CONVERGED) code = -2;
            action = ACT_BREAK;
        } else {
            g_S.sv[SV_PRESPREV] = wi.pres;
            code = dev_check_exit(false);
            if (code == EX_NOT_CONVERGED) {
                if (iter > 0 && wi.step == STEPMIN * GAMMA) {
                    restore_scalars(); restore = 1;
                    code = dev_check_exit(true);
                    if (code == EX_NOT_CONVERGED) code = -2;
                    action = ACT_BREAK;
                } else if (iter == ITER_MAX) {
                    if (!dev_better_than()) { restore_scalars(); restore = 1; }
                    code = dev_check_exit(true);
                    if (code == EX_NOT_CONVERGED) code = -1;
                    action = ACT_BREAK;
                } else if (isnan(wi.pcost)) {
                    if (!(iter == 0 || dev_better_than())) {
                        restore_scalars(); restore = 1;
                        code = dev_check_exit(true);
                        if (code == EX_NOT_CONVERGED) code = -2;
                    }
                    action = ACT_BREAK;
                }
            } else action = ACT_BREAK;
            if (action == ACT_CONTINUE && (iter == 0 || dev_better_than())) { g_S.bi = wi; save = 1; g_S.best = g_S.cur; } // ref :1150-1158: w_best = w, by reference
        }
        g_S.fl[FL_ACTION] = action; g_S.fl[FL_RESTORE] = restore; g_S.fl[FL_SAVE] = save; g_S.fl[FL_CODE] = code;
    }
    __syncthreads();
    if (g_S.fl[FL_ACTION] == ACT_BREAK) {
        // w = w_best (ref :1014 ff.: only ever followed by the exit) = read the best buffer set; backscale (ref :1271-1277) into the
        // instance slab, where the host reads the result.  (lambda is not restored: nothing reads it after the exit.)
        const IterBuf src = iter_buf(P, I, W, g_S.fl[FL_RESTORE] ? g_S.best : g_S.cur), dst = iter_buf(P, I, W, 0);
        gcdbl_p xe = I + P.i_xe, ae = I + P.i_ae, ge = I + P.i_ge;
        const double tau2 = wi.tau;
        FOR_T(j, n) dst.x[j] = src.x[j] / (xe[j] * tau2);
        FOR_T(r, p) dst.y[r] = src.y[r] / (ae[r] * tau2);
        FOR_T(i, m) { dst.z[i] = src.z[i] / (ge[i] * tau2); dst.s[i] = src.s[i] * (ge[i] / tau2); }
        __syncthreads();
        if (tid == 0) g_S.cur = 0;
        return ST_DONE;
    }
    // ---- updateScalings (ref :411-479) + updateKKTScalings (ref :1691-1732) ----
    // the scaling block goes to the instance slab (Vv) and to the factor's target-ordered value stream (Kt)
    gdbl_p Kt = W + P.w_Kt;
    const bool lp_only = P.nc == 0;
    for_t_pre<T, 4>(l, [&](int i) { return IV2{P.v2t[i], wsl[i], wz[i]}; }, [&](int i, const IV2 &r) {
        const double v = r.a / r.b;
        const double w = sqrt(v);
        lpv[i] = v; lpw[i] = w; Vv[i] = -v - DELTASTAT; Kt[r.i] = -v - DELTASTAT;
        if (lp_only) lam[i] = w * r.b; // lambda = W z (ref :476) in the same pass when there is no cone that could fail
    });
    double firstfail = 1e300;
    if (P.nc > 0) {
        // tiny cones: both phases with the cone's rows in registers (same operations in the same order as the generic bodies below)
        struct S1 { D4 s, z; };
        for_tiny<T>(P, [&](const Tiny &t) { return S1{tiny_rows(wsl, t), tiny_rows(wz, t)}; }, [&](const Tiny &t, const S1 &r) {
            const int d = t.d;
            gdbl_p cs = csc + t.c * CSC_STRIDE;
            double s1 = 0., z1 = 0.;
#pragma unroll
            for (int k = 1; k < TINY_D; k++) if (k < d) { s1 += r.s.v[k] * r.s.v[k]; z1 += r.z.v[k] * r.z.v[k]; }
            const double s0 = r.s.v[0], z0 = r.z.v[0];
            const double sres = s0 * s0 - s1, zres = z0 * z0 - z1;
            bool fail = (sres <= 0. || zres <= 0.);
            if (!fail) {
                const double snorm = sqrt(sres), znorm = sqrt(zres);
                double sz = 0., ww = 0.;
#pragma unroll
                for (int k = 0; k < TINY_D; k++) if (k < d) sz += (r.s.v[k] / snorm) * (r.z.v[k] / znorm);
                const double gam = sqrt(0.5 * (1. + sz));
                const double a = (0.5 / gam) * (s0 / snorm + z0 / znorm);
#pragma unroll
                for (int k = 1; k < TINY_D; k++) if (k < d) { const double qk = (0.5 / gam) * (r.s.v[k] / snorm - r.z.v[k] / znorm); ww += qk * qk; }
                const double cc = (1. + a) + ww / (1. + a);
                const double dd = 1. + 2. / (1. + a) + ww / ((1. + a) * (1. + a));
                const double d1 = fmax(0., 0.5 * (a * a + ww * (1. - cc * cc / (1. + ww * dd))));
                const double u0sq = a * a + ww - d1;
                const double c2 = cc * cc / u0sq;
                const bool late = (c2 - dd <= 0.);
                if (late) fail = true;
                cs[CN_ETA2] = snorm / znorm; cs[CN_SN] = snorm; cs[CN_ZN] = znorm; cs[CN_GAM] = gam;
                cs[CN_MODE] = late ? 2. : 0.;
                if (!late) {
                    cs[CN_A] = a; cs[CN_D1] = d1; cs[CN_W] = ww;
                    cs[CN_U0] = sqrt(u0sq); cs[CN_U1] = sqrt(c2); cs[CN_V1] = sqrt(c2 - dd);
                }
            } else cs[CN_MODE] = 1.;
            if (fail) firstfail = fmin(firstfail, (double)t.c);
        });
        // wave cones: lane-per-row registers (same operations in the same order and on the same lanes as the generic bodies below)
        struct S1W { double sA, zA, sB, zB, s0, z0; };
        for_wave<T, wave_cones_in_flight<T>()>(P, [](const WCone &, int) { return NoPre{}; },
            [&](const WCone &b, int lane, NoPre) { return S1W{wrow(wsl, b, lane, 0), wrow(wz, b, lane, 0), wrow(wsl, b, lane, 1), wrow(wz, b, lane, 1), wsl[b.o], wz[b.o]}; },
            [&](const WCone &b, int lane, NoPre, const S1W &r) {
            const bool actA = lane < b.d, actB = 1 + lane < b.d;
            gdbl_p cs = csc + b.c * CSC_STRIDE;
            double s1 = 0., z1 = 0.;
            if (actB) { s1 += r.sB * r.sB; z1 += r.zB * r.zB; }
            s1 = wave_reduce<OpSum>(s1); z1 = wave_reduce<OpSum>(z1);
            const double s0 = r.s0, z0 = r.z0;
            const double sres = s0 * s0 - s1, zres = z0 * z0 - z1;
            bool fail = (sres <= 0. || zres <= 0.);
            if (!fail) {
                const double snorm = sqrt(sres), znorm = sqrt(zres);
                double sz = 0., ww = 0.;
                if (actA) sz += (r.sA / snorm) * (r.zA / znorm);
                sz = wave_reduce<OpSum>(sz);
                const double gam = sqrt(0.5 * (1. + sz));
                const double a = (0.5 / gam) * (s0 / snorm + z0 / znorm);
                if (actB) { const double qk = (0.5 / gam) * (r.sB / snorm - r.zB / znorm); ww += qk * qk; }
                ww = wave_reduce<OpSum>(ww);
                const double cc = (1. + a) + ww / (1. + a);
                const double dd = 1. + 2. / (1. + a) + ww / ((1. + a) * (1. + a));
                const double d1 = fmax(0., 0.5 * (a * a + ww * (1. - cc * cc / (1. + ww * dd))));
                const double u0sq = a * a + ww - d1;
                const double c2 = cc * cc / u0sq;
                const bool late = (c2 - dd <= 0.);
                if (late) fail = true;
                if (lane == 0) {
                    cs[CN_ETA2] = snorm / znorm; cs[CN_SN] = snorm; cs[CN_ZN] = znorm; cs[CN_GAM] = gam;
                    cs[CN_MODE] = late ? 2. : 0.;
                    if (!late) {
                        cs[CN_A] = a; cs[CN_D1] = d1; cs[CN_W] = ww;
                        cs[CN_U0] = sqrt(u0sq); cs[CN_U1] = sqrt(c2); cs[CN_V1] = sqrt(c2 - dd);
                    }
                }
            } else if (lane == 0) cs[CN_MODE] = 1.;
            if (fail) firstfail = fmin(firstfail, (double)b.c);
        });
        for_cones_rest<T>(ps, [&](int c, auto G, int ln) { // phase 1: candidate scalings per cone
            constexpr int g = decltype(G)::value;
            const int o = P.cone_off[c], d = P.cq[c];
            gdbl_p cs = csc + c * CSC_STRIDE;
            double s1 = 0., z1 = 0.;
            for (int k = 1 + ln; k < d; k += g) { s1 += wsl[o + k] * wsl[o + k]; z1 += wz[o + k] * wz[o + k]; }
            s1 = grp_sum<g>(s1); z1 = grp_sum<g>(z1);
            const double s0 = wsl[o], z0 = wz[o];
            const double sres = s0 * s0 - s1, zres = z0 * z0 - z1;
            bool fail = (sres <= 0. || zres <= 0.); // uniform across the cone's lanes
            if (!fail) {
                const double snorm = sqrt(sres), znorm = sqrt(zres);
                double sz = 0., ww = 0.;
                for (int k = ln; k < d; k += g) sz += (wsl[o + k] / snorm) * (wz[o + k] / znorm);
                sz = grp_sum<g>(sz);
                const double gam = sqrt(0.5 * (1. + sz));
                const double a = (0.5 / gam) * (s0 / snorm + z0 / znorm);
                for (int k = 1 + ln; k < d; k += g) {
                    const double qk = (0.5 / gam) * (wsl[o + k] / snorm - wz[o + k] / znorm);
                    ww += qk * qk;
                }
                ww = grp_sum<g>(ww);
                const double cc = (1. + a) + ww / (1. + a);
                const double dd = 1. + 2. / (1. + a) + ww / ((1. + a) * (1. + a));
                const double d1 = fmax(0., 0.5 * (a * a + ww * (1. - cc * cc / (1. + ww * dd))));
                const double u0sq = a * a + ww - d1;
                const double c2 = cc * cc / u0sq;
                const bool late = (c2 - dd <= 0.); // ref :460-463: returns AFTER eta_square, eta (:440-441) and q (:448) were overwritten
                if (late) fail = true;
                if (ln == 0) {
                    cs[CN_ETA2] = snorm / znorm; cs[CN_SN] = snorm; cs[CN_ZN] = znorm; cs[CN_GAM] = gam;
                    cs[CN_MODE] = late ? 2. : 0.;
                    if (!late) {
                        cs[CN_A] = a; cs[CN_D1] = d1; cs[CN_W] = ww;
                        cs[CN_U0] = sqrt(u0sq); cs[CN_U1] = sqrt(c2); cs[CN_V1] = sqrt(c2 - dd);
                    }
                }
            } else if (ln == 0) cs[CN_MODE] = 1.; // left the cone: nothing of this cone was touched (ref :428-431)
            if (fail) firstfail = fmin(firstfail, (double)c);
        });
        // index of the first cone that failed; 1e300 if none.  The reference returns at that cone (ref :428-431,
        // :460-463): earlier cones keep their new scalings, later ones their old ones, lambda is not refreshed --
        // and updateKKTScalings (ref :1162, return value of updateScalings ignored) then writes whatever the structs
        // hold: for a cone that failed the LATE test that is the new eta^2 and q with the old d1, u0, u1, v1.
        firstfail = blk_reduce1<OpMin, T>(phase, firstfail);
        {
            constexpr int NVT = 3 * TINY_D + 1;
            struct S2 { D4 s, z; double eta2, mode, sn, zn, gam, nA, nW, nD1, nU0, nU1, nV1, sD1, sU0, sU1, sV1; int vt[NVT]; };
            constexpr int U2 = waves_per_eu<T>() <= 2 ? 2 : 1; // (~60 registers per cone in flight)
            for_tiny<T, U2>(P, [&](const Tiny &t) {
                S2 r;
                gcdbl_p cs = csc + t.c * CSC_STRIDE;
                r.s = tiny_rows(wsl, t); r.z = tiny_rows(wz, t);
                r.eta2 = cs[CN_ETA2]; r.mode = cs[CN_MODE]; r.sn = cs[CN_SN]; r.zn = cs[CN_ZN]; r.gam = cs[CN_GAM];
                r.nA = cs[CN_A]; r.nW = cs[CN_W]; r.nD1 = cs[CN_D1]; r.nU0 = cs[CN_U0]; r.nU1 = cs[CN_U1]; r.nV1 = cs[CN_V1];
                r.sD1 = cs[CS_D1]; r.sU0 = cs[CS_U0]; r.sU1 = cs[CS_U1]; r.sV1 = cs[CS_V1];
#pragma unroll
                for (int j = 0; j < NVT; j++) r.vt[j] = P.v2t[t.vb + min(j, 3 * t.d)];
                return r;
            }, [&](const Tiny &t, const S2 &r) {
                if ((double)t.c > firstfail) return;
                const bool partial = ((double)t.c == firstfail);
                if (partial && r.mode != 2.) return;
                const int d = t.d, o = t.o;
                gdbl_p cs = csc + t.c * CSC_STRIDE;
                gdbl_p v = Vv + t.vb;
                const double eta2 = r.eta2;
                const double d1 = partial ? r.sD1 : r.nD1, u0 = partial ? r.sU0 : r.nU0, u1 = partial ? r.sU1 : r.nU1, v1 = partial ? r.sV1 : r.nV1;
                const double snorm = r.sn, znorm = r.zn, gam = r.gam;
                cs[CS_ETA2] = eta2; cs[CS_ETA] = sqrt(eta2);
                if (!partial) { cs[CS_A] = r.nA; cs[CS_D1] = d1; cs[CS_W] = r.nW; cs[CS_U0] = u0; cs[CS_U1] = u1; cs[CS_V1] = v1; }
                // (vt[j] = target of scaling-block slot j; d is runtime, so the slots d + k and 2 d + 1 + k are picked by compile-time loops)
                auto vt_at = [&](int j) { int x = r.vt[0];
#pragma unroll
                    for (int q = 1; q < NVT; q++) x = (j == q) ? r.vt[q] : x;
                    return x; };
#pragma unroll
                for (int k = 0; k < TINY_D; k++) if (k < d) {
                    const double qk = (k >= 1) ? (0.5 / gam) * (r.s.v[k] / snorm - r.z.v[k] / znorm) : 0.;
                    if (k >= 1) { qv[o + k] = qk; const double e = -eta2 * v1 * qk; v[d + k] = e; Kt[vt_at(d + k)] = e; }
                    const double e0 = (k == 0) ? -eta2 * d1 - DELTASTAT : -eta2 - DELTASTAT;
                    const double e2 = (k == 0) ? -eta2 * u0 : -eta2 * u1 * qk;
                    v[k] = e0; Kt[r.vt[k]] = e0;
                    v[2 * d + 1 + k] = e2; Kt[vt_at(2 * d + 1 + k)] = e2;
                }
                v[d] = -eta2; Kt[vt_at(d)] = -eta2; v[2 * d] = eta2 + DELTASTAT; Kt[vt_at(2 * d)] = eta2 + DELTASTAT;
            });
        }
        {
            struct S2A { int vb; };
            struct S2W { double sA, zA, eta2, mode, sn, zn, gam, nA, nW, nD1, nU0, nU1, nV1, sD1, sU0, sU1, sV1; int t0, t1, t2, td, t2d; };
            for_wave<T, wave_cones_in_flight<T>()>(P, [&](const WCone &b, int) { return S2A{P.cone_vbase[b.c]}; },
                [&](const WCone &b, int lane, const S2A &a) {
                gcdbl_p cs = csc + b.c * CSC_STRIDE;
                gint_p vt = P.v2t + a.vb;
                const int k = min(lane, b.d - 1), d = b.d;
                return S2W{wrow(wsl, b, lane, 0), wrow(wz, b, lane, 0), cs[CN_ETA2], cs[CN_MODE], cs[CN_SN], cs[CN_ZN], cs[CN_GAM], cs[CN_A], cs[CN_W], cs[CN_D1], cs[CN_U0],
                           cs[CN_U1], cs[CN_V1], cs[CS_D1], cs[CS_U0], cs[CS_U1], cs[CS_V1], vt[k], vt[d + k], vt[2 * d + 1 + k], vt[d], vt[2 * d]};
            }, [&](const WCone &b, int lane, const S2A &a, const S2W &r) {
                if ((double)b.c > firstfail) return;
                const bool partial = ((double)b.c == firstfail);
                if (partial && r.mode != 2.) return;
                const int d = b.d, o = b.o, k = lane;
                gdbl_p cs = csc + b.c * CSC_STRIDE;
                gdbl_p v = Vv + a.vb;
                const double eta2 = r.eta2;
                const double d1 = partial ? r.sD1 : r.nD1, u0 = partial ? r.sU0 : r.nU0, u1 = partial ? r.sU1 : r.nU1, v1 = partial ? r.sV1 : r.nV1;
                const double snorm = r.sn, znorm = r.zn, gam = r.gam;
                if (lane == 0) {
                    cs[CS_ETA2] = eta2; cs[CS_ETA] = sqrt(eta2);
                    if (!partial) { cs[CS_A] = r.nA; cs[CS_D1] = d1; cs[CS_W] = r.nW; cs[CS_U0] = u0; cs[CS_U1] = u1; cs[CS_V1] = v1; }
                }
                if (k < d) {
                    const double qk = (k >= 1) ? (0.5 / gam) * (r.sA / snorm - r.zA / znorm) : 0.;
                    if (k >= 1) { qv[o + k] = qk; const double e = -eta2 * v1 * qk; v[d + k] = e; Kt[r.t1] = e; }
                    const double e0 = (k == 0) ? -eta2 * d1 - DELTASTAT : -eta2 - DELTASTAT;
                    const double e2 = (k == 0) ? -eta2 * u0 : -eta2 * u1 * qk;
                    v[k] = e0; Kt[r.t0] = e0;
                    v[2 * d + 1 + k] = e2; Kt[r.t2] = e2;
                }
                if (lane == 0) { v[d] = -eta2; Kt[r.td] = -eta2; v[2 * d] = eta2 + DELTASTAT; Kt[r.t2d] = eta2 + DELTASTAT; }
            });
        }
        for_cones_rest<T>(ps, [&](int c, auto G, int ln) { // phase 2: commit + updateKKTScalings
            constexpr int g = decltype(G)::value;
            if ((double)c > firstfail) return;
            const int o = P.cone_off[c], d = P.cq[c];
            gdbl_p cs = csc + c * CSC_STRIDE;
            const bool partial = ((double)c == firstfail);
            if (partial && cs[CN_MODE] != 2.) return;
            gdbl_p v = Vv + P.cone_vbase[c];
            gint_p vt = P.v2t + P.cone_vbase[c];
            const double eta2 = cs[CN_ETA2];
            const double d1 = partial ? cs[CS_D1] : cs[CN_D1], u0 = partial ? cs[CS_U0] : cs[CN_U0];
            const double u1 = partial ? cs[CS_U1] : cs[CN_U1], v1 = partial ? cs[CS_V1] : cs[CN_V1];
            const double snorm = cs[CN_SN], znorm = cs[CN_ZN], gam = cs[CN_GAM];
            if (ln == 0) {
                cs[CS_ETA2] = eta2; cs[CS_ETA] = sqrt(eta2);
                if (!partial) { cs[CS_A] = cs[CN_A]; cs[CS_D1] = d1; cs[CS_W] = cs[CN_W]; cs[CS_U0] = u0; cs[CS_U1] = u1; cs[CS_V1] = v1; }
            }
            // KKT scaling block, slot order of ref cacheIndices :1955-1986: D[d], vdiag, v[d-1], udiag, u[d]
            for (int k = ln; k < d; k += g) {
                const double qk = (k >= 1) ? (0.5 / gam) * (wsl[o + k] / snorm - wz[o + k] / znorm) : 0.;
                if (k >= 1) { qv[o + k] = qk; const double e = -eta2 * v1 * qk; v[d + k] = e; Kt[vt[d + k]] = e; }
                const double e0 = (k == 0) ? -eta2 * d1 - DELTASTAT : -eta2 - DELTASTAT;
                const double e2 = (k == 0) ? -eta2 * u0 : -eta2 * u1 * qk;
                v[k] = e0; Kt[vt[k]] = e0;
                v[2 * d + 1 + k] = e2; Kt[vt[2 * d + 1 + k]] = e2;
            }
            if (ln == 0) { v[d] = -eta2; Kt[vt[d]] = -eta2; v[2 * d] = eta2 + DELTASTAT; Kt[vt[2 * d]] = eta2 + DELTASTAT; }
        });
    }
    __syncthreads();
    if (!lp_only && firstfail >= 1e299) dev_scale<T>(ps, W, wz, lam); // lambda = W z only when every cone succeeded (ref :476)
    TICK_END(TK_RESID);
    return ST_FACTOR;
}

// ---------------- the KKT stages, part 1: solveKKT (ref :1471-1620) ----------------
// KI = 1: one right-hand side.  KI = 2 with DUAL: the TWO INDEPENDENT right-hand sides of one instance -- rhs1 and rhs2 of
// the initialisation (ref :933, :966) or of a pass (KKT1 and the affine system, ref :1173-1179: RHSaffine does not depend on
// the first solution) -- solved together: the sweeps and the refinement residuals stream L, A and G once for both.  The
// KKT-space vectors (sweep vector SV, iterate X, residual E, parked iterate Xg) are then 2-interleaved (element i of
// right-hand side k at 2 i + k); every right-hand side keeps its own refinement state (step count, previous error, done
// flag): the loop runs until both have stopped, one that has stopped keeps its iterate while the other takes further
// steps (its lanes still compute, the result is discarded).  amask: bit k set = right-hand side k takes part.
template <int T, int NLDS, bool I16, int KI, bool DUAL = false>
static __device__ __noinline__ __attribute__((not_tail_called)) void kkt_solve(int ps, gdbl_p I0, gdbl_p I1, gdbl_p Wg, int stage, int amask) {
    ps = uni(ps); I0 = uni_ptr(I0); I1 = uni_ptr(I1); Wg = uni_ptr(Wg); stage = uni(stage); amask = uni(amask);
    const DevPat &P = c_pat[ps];
    const int n = P.n, p = P.p, m = P.m, l = P.l, N = P.N, np = P.n + P.p;
    const int tid = threadIdx.x;
    int phase = 0;
    static_assert(DUAL == (KI == 2), "KI = 2 is the dual right-hand-side solve");
    static_assert(KI == 1 || NLDS == 1, "a dual solve keeps its two sweep vectors in LDS");
    gdbl_p Ik[KI_MAX] = {I0, I1};
    const bool init = (stage == ST_KKT_INIT1 || stage == ST_KKT_INIT2);
    const bool first = (stage == ST_KKT_INIT1 || stage == ST_KKT1);
    gcdbl_p cagv[KI], rAv[KI], rGv[KI], bx[KI], by[KI], bz[KI], lpv[KI], csc[KI], qv[KI];
    gdbl_p dx[KI], dy[KI], dz[KI];
#pragma unroll
    for (int k = 0; k < KI; k++) {
        gdbl_p I = Ik[k], W = Wg;
        const bool fk = DUAL ? (k == 0) : first; // dual: right-hand side 0 is rhs1 -> (dx1, dy1, dz1), 1 is rhs2 -> (dx2, dy2, dz2)
        cagv[k] = I + P.i_cag; rAv[k] = I + P.i_rA; rGv[k] = I + P.i_rG;
        gcdbl_p rhsk = W + (fk ? P.w_rhs1k : P.w_rhs2k);               // the right-hand side as [x | y | z]
        bx[k] = rhsk; by[k] = rhsk + n; bz[k] = rhsk + np;
        lpv[k] = W + P.w_lpv; csc[k] = W + P.w_csc; qv[k] = W + P.w_qv;
        dx[k] = W + (fk ? P.w_dx1 : P.w_dx2); dy[k] = W + (fk ? P.w_dy1 : P.w_dy2); dz[k] = W + (fk ? P.w_dz1 : P.w_dz2);
    }
    auto state = [&](int) -> ShI & { return g_S; };
    // KKT-space vectors, in elimination order: X = current solution, E = rhs / residual / solve vector.
    // Both in LDS (NLDS = 2) or both in the workspace slab (NLDS = 0, patterns too large for LDS).
    // Roles: SV = vector the triangular sweeps run on; X = solution the residual gathers from; E = where the
    // residual is written.  NLDS = 2: SV = E and X both in LDS.  NLDS = 1 (one LDS vector, several workgroups
    // per CU): the LDS vector alternates between the sweep vector and X -- the ~25k gathers of a residual hit
    // LDS, its ~N scattered stores go to E in the workspace slab, and X is parked in the slab (Xg) only while a
    // further refinement step borrows the LDS vector.  NLDS = 0: everything in the workspace slab.
    auto SV = [&] { if constexpr (NLDS >= 1) return g_dyn; else return Wg + P.w_ek; }();
    auto X = [&] { if constexpr (NLDS >= 2) return g_dyn + P.Npad; else if constexpr (NLDS == 1) return g_dyn; else return Wg + P.w_xk; }();
    auto E = [&] { if constexpr (DUAL) return Wg + P.w_dual_ek; else if constexpr (NLDS == 1) return Wg + P.w_ek; else return SV; }();
    gdbl_p Xg = DUAL ? Wg + P.w_dual_xk : Wg + P.w_xk; // NLDS = 1: the iterate while the LDS vector serves the triangular sweeps
    // NLDS = 1, one right-hand side: the first e_lds elimination positions of E live in the LDS the launch shape leaves free
    // behind the tables (api.cpp) -- E is written by scattered 8-byte stores (partial lines in HBM) and read back once
    constexpr bool ESPLIT = (NLDS == 1 && !DUAL);
    const int e_lds = ESPLIT ? P.e_lds : 0;
    double *EL = g_dyn + P.e_off;
    auto stE = [&](int o, int k, double v) { if constexpr (ESPLIT) { if (o < e_lds) EL[o] = v; else E[o] = v; } else E[o * KI + k] = v; };
    auto ldE = [&](int o, int k) -> double { if constexpr (ESPLIT) return o < e_lds ? EL[o] : (double)E[o]; else return E[o * KI + k]; };
    const PackedSlice *tabs = reinterpret_cast<const PackedSlice *>(g_dyn + P.lds_tab);
    auto tab_cag = [&] { if constexpr (NLDS >= 1) return tabs + P.lm_cag; else return P.cag_sl; }();
    auto tab_rA = [&] { if constexpr (NLDS >= 1) return tabs + P.lm_rA; else return P.rA_sl; }();
    auto tab_rG = [&] { if constexpr (NLDS >= 1) return tabs + P.lm_rG; else return P.rG_sl; }();
    gdbl_p dxr = Wg + P.w_dxr;
    gdbl_p UF = Wg + P.w_UF, invD = Wg + P.w_invD; // (one factor, also for two right-hand sides)
    ubcdbl_p UB = [&] { if constexpr (EICOS_UBL != 0) return (ubcdbl_p)(g_dyn + P.ub_lds); else return (ubcdbl_p)(Wg + P.w_UB); }();
    __syncthreads();
    unsigned long long tk0_ = (tid == 0) ? wall_clock64() : 0ull;
    auto tick = [&](int slot) {
        if (tid == 0) { const unsigned long long t1_ = wall_clock64(); g_S.tick[slot] += t1_ - tk0_; tk0_ = t1_; }
    };
    // per-instance refinement state (workgroup-uniform: every thread sees the same reduction results)
    int kcnt[KI]; double nerr_prev[KI], thr[KI]; bool rdone[KI];
    {
        double nr[KI];
#pragma unroll
        for (int k = 0; k < KI; k++) nr[k] = 0.;
        // the sweep vector: the right-hand side [x | y | z] scattered to its elimination positions (one pass: P.ipk = [ipx | ipy | ipz]), zero
        // in the expansion rows of the cones and in the padding (tile layouts pad inside the blocks)
        {
            double z[KI];
#pragma unroll
            for (int k = 0; k < KI; k++) z[k] = 0.;
            FOR_T(q_, P.nzpos) stK<KI>(SV, P.zpos[q_], z); // (host list: every slot of [0, Npad) that no entry of ipk names)
        }
        for_t_pre<T, 6>(np + m, [&](int j) {
            IVK<KI> r;
            r.o = ld_u32(P.ipk, j);
#pragma unroll
            for (int k = 0; k < KI; k++) r.v[k] = ld_u32(bx[k], j); // (bx = the whole [x | y | z] vector)
            return r;
        }, [&](int, const IVK<KI> &r) {
#pragma unroll
            for (int k = 0; k < KI; k++) nr[k] = fmax(nr[k], fabs(r.v[k]));
            stK<KI>(SV, r.o, r.v);
        });
        blk_reduce<OpMax, T, KI>(phase, nr);
#pragma unroll
        for (int k = 0; k < KI; k++) { thr[k] = (1. + nr[k]) * LINSYSACC; nerr_prev[k] = DBL_MAX; kcnt[k] = -1; rdone[k] = !((amask >> k) & 1); }
    }
    for (int pass = 0;; pass++) {
        // -------- SV <- L^-T D^-1 L^-1 SV in elimination order (replaces ldlt.solve, ref :1477,1599) --------
        tick(TK_KRES);
        __syncthreads();
        // forward: workgroup-wide levels, then the narrow top of the tree on wavefront 0; backward: the top first
        const bool wave0 = uni(tid >> 6) == 0;
        if (P.tile == 1) { // dense fronts: tile mat-vecs over the block levels (single-instance workgroups only)
            if constexpr (NLDS >= 1) tile_solve<T, true, KI>(P, Wg, SV); else tile_solve<T, false, KI>(P, Wg, SV);
        } else if constexpr (NLDS >= 1) { // slice tables staged in LDS behind the vectors (k_solve prologue)
            tri_sweep<T, true, true, false, I16, KI, DUAL>(tabs + P.lm_f, P.nfs, P.nfs_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF); // barriers at level starts + end
            if (P.tile == 2) { // hybrid: levels below the cut, the top block's rows against them, both tile sweeps on the block, back down
                {
                    if (wave0) tri_sweep<T, true, true, true, I16, KI, DUAL>(tabs + P.lm_f + P.nfs, P.nfs_solo, P.nfs_solo_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF);
                    __syncthreads();
                    tri_sweep<T, true, true, false, I16, KI, DUAL>(tabs + P.lm_f + P.nfs + P.nfs_solo, P.nfs_ext, P.nfs_ext_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF);
                    tile_solve<T, true, KI>(P, Wg, SV);
                    if (wave0) tri_sweep<T, false, true, true, I16, KI, DUAL>(tabs + P.lm_b, P.nbs_solo, P.nbs_solo_r, P.b_idx, P.b_idx16, P.b_d16, UB, invD, SV, P.nUB);
                }
            } else if (apex_on<T>(P)) { // dense apex: the narrow levels below it on wavefront 0, its rows against everything below (all wavefronts), the apex, back down
                if (P.nfs_solo) {
                    if (wave0) tri_sweep<T, true, true, true, I16, KI, DUAL>(tabs + P.lm_f + P.nfs, P.nfs_solo, P.nfs_solo_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF);
                    __syncthreads();
                }
                tri_sweep<T, true, true, false, I16, KI, DUAL>(tabs + P.lm_f + P.nfs + P.nfs_solo, P.nfs_ext, P.nfs_ext_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF);
                if (wave0) {
#ifdef EICOS_SOLO_TICKS
                    tick(TK_LDL);
#endif
                    if (P.apex_lds >= 0) apex_solve_lds<KI>(P, invD, SV);
                    else if constexpr (EICOS_UBL == 0) apex_solve<KI>(P, UF, UB, invD, SV); // (UBL handles always carry the LDS image: api.cpp)
                    tri_sweep<T, false, true, true, I16, KI, DUAL>(tabs + P.lm_b, P.nbs_solo, P.nbs_solo_r, P.b_idx, P.b_idx16, P.b_d16, UB, invD, SV, P.nUB);
#ifdef EICOS_SOLO_TICKS
                    tick(TK_FWD);
#endif
                }
            } else if (wave0) {
#ifdef EICOS_SOLO_TICKS
                tick(TK_LDL);
#endif
                tri_sweep<T, true, true, true, I16, KI, DUAL>(tabs + P.lm_f + P.nfs, P.nfs_solo, P.nfs_solo_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF);
                tri_sweep<T, false, true, true, I16, KI, DUAL>(tabs + P.lm_b, P.nbs_solo, P.nbs_solo_r, P.b_idx, P.b_idx16, P.b_d16, UB, invD, SV, P.nUB);
#ifdef EICOS_SOLO_TICKS
                tick(TK_FWD); // (dev builds: the narrow tree top of both sweeps, wavefront 0 alone, shows up as "fwd" in the phase timers)
#endif
            }
            __syncthreads();
            tri_sweep<T, false, true, false, I16, KI, DUAL>(tabs + P.lm_b + P.nbs_solo, P.nbs, P.nbs_r, P.b_idx, P.b_idx16, P.b_d16, UB, invD, SV, P.nUB);
        } else {
            tri_sweep<T, true, false, false, I16, KI, DUAL>(P.fsl, P.nfs, P.nfs_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF);
            if (P.tile == 2) {
                {
                    if (wave0) tri_sweep<T, true, false, true, I16, KI, DUAL>(P.fsl + P.nfs, P.nfs_solo, P.nfs_solo_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF);
                    __syncthreads();
                    tri_sweep<T, true, false, false, I16, KI, DUAL>(P.fsl + P.nfs + P.nfs_solo, P.nfs_ext, P.nfs_ext_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF);
                    tile_solve<T, false, KI>(P, Wg, SV);
                    if (wave0) tri_sweep<T, false, false, true, I16, KI, DUAL>(P.bsl, P.nbs_solo, P.nbs_solo_r, P.b_idx, P.b_idx16, P.b_d16, UB, invD, SV, P.nUB);
                }
            } else if (apex_on<T>(P)) {
                if (P.nfs_solo) {
                    if (wave0) tri_sweep<T, true, false, true, I16, KI, DUAL>(P.fsl + P.nfs, P.nfs_solo, P.nfs_solo_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF);
                    __syncthreads();
                }
                tri_sweep<T, true, false, false, I16, KI, DUAL>(P.fsl + P.nfs + P.nfs_solo, P.nfs_ext, P.nfs_ext_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF);
                if (wave0) {
                    apex_solve<KI>(P, UF, UB, invD, SV);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // (slab vector: the apex's stores before the gathers of the levels below)
                    tri_sweep<T, false, false, true, I16, KI, DUAL>(P.bsl, P.nbs_solo, P.nbs_solo_r, P.b_idx, P.b_idx16, P.b_d16, UB, invD, SV, P.nUB);
                }
            } else if (wave0) {
                tri_sweep<T, true, false, true, I16, KI, DUAL>(P.fsl + P.nfs, P.nfs_solo, P.nfs_solo_r, P.f_idx, P.f_idx16, P.f_d16, UF, invD, SV, P.nUF);
                tri_sweep<T, false, false, true, I16, KI, DUAL>(P.bsl, P.nbs_solo, P.nbs_solo_r, P.b_idx, P.b_idx16, P.b_d16, UB, invD, SV, P.nUB);
            }
            __syncthreads();
            tri_sweep<T, false, false, false, I16, KI, DUAL>(P.bsl + P.nbs_solo, P.nbs, P.nbs_r, P.b_idx, P.b_idx16, P.b_d16, UB, invD, SV, P.nUB);
        }
        // x = first solve / x += dx_ref (ref :1602); an instance that has stopped keeps its iterate
        if constexpr (NLDS == 1) { // the LDS vector becomes X again: previous iterate (slab copy Xg) + increment
            if (pass > 0) for_t_pre<T, 6>(N, [&](int i) { VKI<KI> r; ldK<KI>(Xg, i, r.v); return r; }, [&](int i, const VKI<KI> &r) {
                double c[KI], o[KI];
                ldK<KI>(SV, i, c);
#pragma unroll
                for (int k = 0; k < KI; k++) o[k] = rdone[k] ? r.v[k] : r.v[k] + c[k];
                stK<KI>(SV, i, o);
            });
        } else { // (KI = 1)
            if (pass == 0) for_t_pre<T, 6>(P.Npad, [&](int i) { return V1{SV[i]}; }, [&](int i, const V1 &r) { X[i] = r.a; }); // x = first solve (slot N.. = 0)
            else for_t_pre<T, 6>(N, [&](int i) { return V2{SV[i], X[i]}; }, [&](int i, const V2 &r) { dxr[i] = r.a; X[i] = r.b + r.a; });
        }
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < KI; k++) if (!rdone[k]) state(k).wi.n_ldlsolve++;
            g_S.wi.n_sweep++; // (one pass over L, whatever the number of right-hand sides)
        }
#pragma unroll
        for (int k = 0; k < KI; k++) if (!rdone[k]) kcnt[k]++;
        __syncthreads();
        tick(TK_LDL);
        // ---- residual e = rhs - K~ x, matrix-free (ref :1511-1567), written in elimination order into E ----
        double nex[KI], ney[KI], nez[KI];
#pragma unroll
        for (int k = 0; k < KI; k++) nex[k] = ney[k] = nez[k] = 0.;
        struct PreK { double b, w; int o, sg; double g; }; // rhs entry, LP scaling, elimination-order slot, sign of the regularisation, G-tile term
        const bool gt = P.gt_on != 0; // G in tiles: one pass gives G dx and G' dz for all KI right-hand sides
        gdbl_p gxv = Wg + P.w_gx, gzv = Wg + P.w_gz;
        if (gt) g_tile_products<T, KI>(P, I0 + P.i_Gt, P.gt_colk, [&](int c, int k) { return X[c * KI + k]; },
                                       [&](int i, int k) { return X[P.gt_zslot[i] * KI + k]; }, Wg + P.w_gpart, gxv, gzv);
        ell_dots_k<T, I16, KI, DUAL>(tab_cag, P.cag_ns, P.cag_ns_r, P.cag_idx_k, P.cag_k16, P.cag_d16, cagv, X, P.cag_slots,
                    [&](int k, int j) { return PreK{ld_u32(bx[k], j), 0., ld_u32(P.ipx, j), 0, gt ? gxv[j * KI + k] : 0.}; },
                    [&](int k, int j, double s, const PreK &pr) {
            const int o = pr.o;
            const double e = pr.b - (s + pr.g) - DELTASTAT * X[o * KI + k]; // ex = bx - G'dz - A'dy - delta dx
            stE(o, k, e); nex[k] = fmax(nex[k], fabs(e));
        });
        ell_dots_k<T, I16, KI, DUAL>(tab_rA, P.rA_ns, P.rA_ns_r, P.rA_idx_k, P.rA_k16, P.rA_d16, rAv, X, P.rA_slots,
                    [&](int k, int r) { return PreK{ld_u32(by[k], r), 0., ld_u32(P.ipy, r), 0, 0.}; },
                    [&](int k, int r, double s, const PreK &pr) {
            const int o = pr.o;
            const double e = pr.b - s + DELTASTAT * X[o * KI + k]; // ey = by - A dx + delta dy
            stE(o, k, e); ney[k] = fmax(ney[k], fabs(e));
        });
        ell_dots_k<T, I16, KI, DUAL>(tab_rG, P.rG_ns, P.rG_ns_r, P.rG_idx_k, P.rG_k16, P.rG_d16, rGv, X, P.rG_slots,
                    [&](int k, int i) { return PreK{ld_u32(bz[k], i), ld_u32(lpv[k], i < l ? i : 0), ld_u32(P.ipz, i), ld_u32(P.zdsign, i), gt ? gzv[i * KI + k] : 0.}; },
                    [&](int k, int i, double s, const PreK &pr) {
            const int o = pr.o;
            const double xo = X[o * KI + k];
            double v = pr.b - (s + pr.g) + (double)pr.sg * DELTASTAT * xo; // ez = bz - G dx +/- delta dz ...
            if (i < l) { v += init ? xo : pr.w * xo; nez[k] = fmax(nez[k], fabs(v)); } // ... + V dz (LP part)
            stE(o, k, v);
        });
        if (P.nc > 0) {
            __syncthreads();
            // cone blocks (expanded): ez += dz_true (init) or scale2add (ref :1629-1662)
#pragma unroll
            for (int k = 0; k < KI; k++) {
                // tiny cones: the body below with the cone's rows in registers (for_tiny: descriptors, then every value, then the arithmetic)
                struct RK { double e[TINY_D], x[TINY_D], q[TINY_D], x3, x4, eta2, v1, u1, d1, u0; };
                for_tiny<T>(P, [&](const Tiny &t) {
                    RK r;
                    gcdbl_p cs = csc[k] + t.c * CSC_STRIDE;
#pragma unroll
                    for (int q = 0; q < TINY_D; q++) { r.e[q] = ldE(t.ez[q], k); r.x[q] = X[t.ez[q] * KI + k]; r.q[q] = qv[k][t.o + min(q, t.d - 1)]; }
                    r.x3 = X[t.ev * KI + k]; r.x4 = X[t.eu * KI + k];
                    r.eta2 = cs[CS_ETA2]; r.v1 = cs[CS_V1]; r.u1 = cs[CS_U1]; r.d1 = cs[CS_D1]; r.u0 = cs[CS_U0];
                    return r;
                }, [&](const Tiny &t, const RK &r) {
                    double mx = 0.;
                    if (init) {
#pragma unroll
                        for (int q = 0; q < TINY_D; q++) if (q < t.d) { const double v = r.e[q] + r.x[q]; stE(t.ez[q], k, v); mx = fmax(mx, fabs(v)); }
                        stE(t.ev, k, r.x3); stE(t.eu, k, r.x4); mx = fmax(mx, fmax(fabs(r.x3), fabs(r.x4)));
                    } else {
                        const double tt = r.v1 * r.x3 + r.u1 * r.x4;
                        double qtx = 0.;
#pragma unroll
                        for (int q = 1; q < TINY_D; q++) if (q < t.d) {
                            const double v = r.e[q] + r.eta2 * (r.x[q] + tt * r.q[q]);
                            stE(t.ez[q], k, v); mx = fmax(mx, fabs(v));
                            qtx += r.q[q] * r.x[q];
                        }
                        const double v1 = r.e[0] + r.eta2 * (r.d1 * r.x[0] + r.u0 * r.x4);
                        const double v3 = r.eta2 * (r.v1 * qtx + r.x3);
                        const double v4 = r.eta2 * (r.u0 * r.x[0] + r.u1 * qtx - r.x4);
                        stE(t.ez[0], k, v1); stE(t.ev, k, v3); stE(t.eu, k, v4);
                        mx = fmax(mx, fmax(fabs(v1), fmax(fabs(v3), fabs(v4))));
                    }
                    nez[k] = fmax(nez[k], mx);
                });
                // wave cones, refinement operator proper (not the initialisation solves): lane-per-row registers
                struct RWA { int eq, e1, e3, e4; };
                struct RW { double e, xq, qq, e1v, x1, x3, x4, eta2, v1, u1, d1, u0; };
                if (!init) for_wave<T, wave_cones_in_flight<T>()>(P, [&](const WCone &b, int lane) {
                    return RWA{P.ipz[b.o + min(1 + lane, b.d - 1)], P.ipz[b.o], P.ipv[b.c], P.ipu[b.c]};
                }, [&](const WCone &b, int lane, const RWA &a) {
                    gcdbl_p cs = csc[k] + b.c * CSC_STRIDE;
                    return RW{ldE(a.eq, k), X[a.eq * KI + k], qv[k][b.o + min(1 + lane, b.d - 1)], ldE(a.e1, k), X[a.e1 * KI + k], X[a.e3 * KI + k], X[a.e4 * KI + k],
                              cs[CS_ETA2], cs[CS_V1], cs[CS_U1], cs[CS_D1], cs[CS_U0]};
                }, [&](const WCone &b, int lane, const RWA &a, const RW &r) {
                    const bool act = 1 + lane < b.d;
                    const double tt = r.v1 * r.x3 + r.u1 * r.x4;
                    double mx = 0., qtx = 0.;
                    if (act) {
                        const double v = r.e + r.eta2 * (r.xq + tt * r.qq);
                        stE(a.eq, k, v); mx = fmax(mx, fabs(v));
                        qtx += r.qq * r.xq;
                    }
                    qtx = wave_reduce<OpSum>(qtx);
                    if (lane == 0) {
                        const double v1 = r.e1v + r.eta2 * (r.d1 * r.x1 + r.u0 * r.x4);
                        const double v3 = r.eta2 * (r.v1 * qtx + r.x3);
                        const double v4 = r.eta2 * (r.u0 * r.x1 + r.u1 * qtx - r.x4);
                        stE(a.e1, k, v1); stE(a.e3, k, v3); stE(a.e4, k, v4);
                        mx = fmax(mx, fmax(fabs(v1), fmax(fabs(v3), fabs(v4))));
                    }
                    nez[k] = fmax(nez[k], mx);
                });
                else for (int q_ = threadIdx.x >> 6; q_ < P.n_wave; q_ += T / 64) { // (initialisation solves: ez += dz on the cone's rows, the generic body)
                    const int c = P.cone_wave[q_], ln = threadIdx.x & 63, d = P.cq[c], o = P.cone_off[c];
                    double mx = 0.;
                    for (int q = ln; q < d; q += 64) { const int eq = P.ipz[o + q], pq = eq * KI + k; const double v = ldE(eq, k) + X[pq]; stE(eq, k, v); mx = fmax(mx, fabs(v)); }
                    if (ln == 0) { const int e3 = P.ipv[c], e4 = P.ipu[c]; const double x3 = X[e3 * KI + k], x4 = X[e4 * KI + k]; stE(e3, k, x3); stE(e4, k, x4); mx = fmax(mx, fmax(fabs(x3), fabs(x4))); }
                    nez[k] = fmax(nez[k], mx);
                }
                for_cones_rest<T>(ps, [&](int c, auto G, int ln) {
                    constexpr int g = decltype(G)::value;
                    const int d = P.cq[c], o = P.cone_off[c];
                    const int p1 = P.ipz[o] * KI + k, p3 = P.ipv[c] * KI + k, p4 = P.ipu[c] * KI + k;
                    const int e1 = P.ipz[o], e3 = P.ipv[c], e4 = P.ipu[c];
                    double mx = 0.;
                    if (init) {
                        for (int q = ln; q < d; q += g) { const int eq = P.ipz[o + q], pq = eq * KI + k; const double v = ldE(eq, k) + X[pq]; stE(eq, k, v); mx = fmax(mx, fabs(v)); }
                        if (ln == 0) { const double x3 = X[p3], x4 = X[p4]; stE(e3, k, x3); stE(e4, k, x4); mx = fmax(mx, fmax(fabs(x3), fabs(x4))); }
                    } else {
                        gcdbl_p cs = csc[k] + c * CSC_STRIDE;
                        const double eta2 = cs[CS_ETA2], x1 = X[p1], x3 = X[p3], x4 = X[p4];
                        const double tt = cs[CS_V1] * x3 + cs[CS_U1] * x4;
                        double qtx = 0.;
                        for (int q = 1 + ln; q < d; q += g) {
                            const int eq = P.ipz[o + q], pq = eq * KI + k;
                            const double qq = qv[k][o + q], xq = X[pq];
                            const double v = ldE(eq, k) + eta2 * (xq + tt * qq);
                            stE(eq, k, v); mx = fmax(mx, fabs(v));
                            qtx += qq * xq;
                        }
                        qtx = grp_sum<g>(qtx);
                        if (ln == 0) {
                            const double v1 = ldE(e1, k) + eta2 * (cs[CS_D1] * x1 + cs[CS_U0] * x4);
                            const double v3 = eta2 * (cs[CS_V1] * qtx + x3);
                            const double v4 = eta2 * (cs[CS_U0] * x1 + cs[CS_U1] * qtx - x4);
                            stE(e1, k, v1); stE(e3, k, v3); stE(e4, k, v4);
                            mx = fmax(mx, fmax(fabs(v1), fmax(fabs(v3), fabs(v4))));
                        }
                    }
                    nez[k] = fmax(nez[k], mx);
                });
            }
        }
        double nv[3 * KI];
#pragma unroll
        for (int k = 0; k < KI; k++) { nv[3 * k] = nex[k]; nv[3 * k + 1] = ney[k]; nv[3 * k + 2] = nez[k]; }
        blk_reduce<OpMax, T, 3 * KI>(phase, nv);
        bool undo[KI], all_done = true;
#pragma unroll
        for (int k = 0; k < KI; k++) {
            undo[k] = false;
            if (rdone[k]) continue;
            double nerr = fmax(nv[3 * k], nv[3 * k + 2]);
            if (p > 0) nerr = fmax(nerr, nv[3 * k + 1]);
            if (kcnt[k] > 0 && nerr > nerr_prev[k]) { undo[k] = true; kcnt[k]--; rdone[k] = true; } // got worse: undo and quit (ref :1579-1585)
            else if (kcnt[k] == NITREF || nerr < thr[k] || (kcnt[k] > 0 && nerr_prev[k] < IRERRFACT * nerr)) rdone[k] = true;
            else { nerr_prev[k] = nerr; all_done = false; }
        }
#pragma unroll
        for (int k = 0; k < KI; k++) {
            if (!undo[k]) continue;
            if constexpr (NLDS == 1) { FOR_T(i, N) X[i * KI + k] = Xg[i * KI + k]; } // Xg still holds the previous iterate
            else { FOR_T(i, N) X[i] -= dxr[i]; }
        }
        if (all_done) break;
        if constexpr (NLDS == 1) { // another step: park the iterate in the slab, residual -> sweep vector (unit stride)
            __syncthreads();
            for_t_pre<T, 6>(N, [&](int i) {
                VKI<KI> r;
                if constexpr (ESPLIT) r.v[0] = ldE(i, 0); else ldK<KI>(E, i, r.v);
                return r;
            }, [&](int i, const VKI<KI> &r) {
                double c[KI];
                ldK<KI>(X, i, c); stK<KI>(Xg, i, c); stK<KI>(SV, i, r.v);
            });
        }
    }
    __syncthreads();
    double d3k[3 * KI]; // ... with c'dx, b'dy, h'dz on the way (every pass of the main loop needs them: ref :1185-1190, :1216-1219)
#pragma unroll
    for (int k = 0; k < KI; k++) {
        double d3[3] = {0., 0., 0.};
        if ((amask >> k) & 1) {
        gcdbl_p cv = Ik[k] + P.i_c;
        for_t_pre3<T, 8>(n, p, m, [&](int, int j) { // (j: index in [x | y | z]; c, b, h and dx, dy, dz are laid out in that order)
            return V2{X[ld_u32(P.ipk, j) * KI + k], init ? 0. : ld_u32(cv, j)};
        }, [&](int rg, int j, const V2 &r) {
            dx[k][j] = r.a;
            const double t = r.b * r.a;
            d3[0] += rg == 0 ? t : 0.; d3[1] += rg == 1 ? t : 0.; d3[2] += rg == 2 ? t : 0.; // (x + 0 = x: the partial sums are those of three separate loops)
        });
        if (tid == 0) { if (DUAL && k == 1) g_S.kref2 = kcnt[k]; else state(k).kref = kcnt[k]; }
        }
        d3k[3 * k] = d3[0]; d3k[3 * k + 1] = d3[1]; d3k[3 * k + 2] = d3[2];
    }
    if (!init) { // (outside the loop over the right-hand sides: a barrier inside keeps the compiler from unrolling it)
        blk_reduce<OpSum, T, 3 * KI>(phase, d3k);
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < KI; k++) if ((amask >> k) & 1) {
                const int which = (DUAL ? k == 0 : first) ? 0 : 1;
                g_S.dots[which][0] = d3k[3 * k]; g_S.dots[which][1] = d3k[3 * k + 1]; g_S.dots[which][2] = d3k[3 * k + 2];
            }
        }
    }
    __syncthreads();
    tick(TK_KRES);
}

// ---------------- the KKT stages, part 2: post-processing of one instance (its state is in g_S) ----------------
// RF: the register-resident fast path (below) for the affine / combined stages -- its own instantiation, so that the generic one keeps
// the code (and instruction-cache footprint) it had; kkt_post_any picks per pattern.
template <int T, bool RF>
static __device__ __noinline__ __attribute__((not_tail_called)) int kkt_post(int ps, gdbl_p I, gdbl_p W, int stage) {
    ps = uni(ps); I = uni_ptr(I); W = uni_ptr(W); stage = uni(stage);
    const DevPat &P = c_pat[ps];
    const int n = P.n, p = P.p, m = P.m, l = P.l, np = P.n + P.p;
    const int tid = threadIdx.x;
    DevInfo &wi = g_S.wi;
    gdbl_p cv = I + P.i_c;
    const IterBuf it = iter_buf(P, I, W, g_S.cur); // the current iterate's buffer set (switched only at the end of a stage, before its last barrier)
    gdbl_p wx = it.x, wy = it.y, wz = it.z, wsl = it.s;
    gdbl_p lam = W + P.w_lam, rz = W + P.w_rz;
    gdbl_p rhs1k = W + P.w_rhs1k, rhs2k = W + P.w_rhs2k; // the right-hand sides as [x | y | z] (kkt_solve scatters them into the sweep vector)
    gdbl_p dx1 = W + P.w_dx1, dy1 = W + P.w_dy1, dz1 = W + P.w_dz1, dx2 = W + P.w_dx2, dy2 = W + P.w_dy2, dz2 = W + P.w_dz2;
    gdbl_p dsw = W + P.w_dsw, wdz = W + P.w_wdz, dsa = W + P.w_dsa, t1 = W + P.w_t1, t2 = W + P.w_t2;
    gdbl_p lpw = W + P.w_lpw, csc = W + P.w_csc, qv = W + P.w_qv;
    const int kref = g_S.kref;
    // Register-resident fast path of the affine / combined post-processing: no second-order cone and at most RL LP rows per thread --
    // W dz, ds (and the combined dz) of a thread's rows stay in registers across the line search instead of going through HBM.
    // Only in builds compiled for <= 2 waves per SIMD (256 VGPRs).
    constexpr int RL = 16, RH = 8;
    // (and only when at least a quarter of the RL slots of a thread are used: on small patterns the unrolled rounds mostly issue clamped loads)
    constexpr bool regfast = RF; // (kkt_post_any: builds for <= 2 waves per SIMD, no cone, RL T / 4 < l <= RL T)
    auto for_rl = [&](auto &&ld, auto &&fn) __attribute__((always_inline)) { // (two half rounds: RH rows of a thread in flight)
#pragma unroll
        for (int h = 0; h < RL / RH; h++) {
            decltype(ld(0)) r[RH];
#pragma unroll
            for (int u = 0; u < RH; u++) { const int i = tid + (h * RH + u) * T; r[u] = ld(i < l ? i : 0); }
#pragma unroll
            for (int u = 0; u < RH; u++) { const int i = tid + (h * RH + u) * T; if (i < l) fn(h * RH + u, i, r[u]); }
        }
    };
    __syncthreads();
    TICK_BEGIN;
    if (stage == ST_KKT_INIT1) { // ref :933-939
        if (tid == 0) wi.nitref1 = kref;
        for_t_pre<T, 4>(n, [&](int j) { return V1{dx1[j]}; }, [&](int j, const V1 &r) { wx[j] = r.a; });
        dev_bring_to_cone<T>(ps, dz1, -1., wsl);
        stage = ST_KKT_INIT2;
    } else if (stage == ST_KKT_INIT2) { // ref :966-992
        for_t_pre<T, 4>(p, [&](int j) { return V1{dy2[j]}; }, [&](int j, const V1 &r) { wy[j] = r.a; });
        dev_bring_to_cone<T>(ps, dz2, 1., wz);
        for_t_pre<T, 4>(n, [&](int j) { return V1{cv[j]}; }, [&](int j, const V1 &r) { rhs1k[j] = -r.a; });
        if (tid == 0) {
            wi.nitref2 = kref;
            wi.kap = 1.; wi.tau = 1.; wi.step = 0.; wi.step_aff = 0.; wi.pinf = 0; wi.dinf = 0;
            g_S.sv[SV_PRESPREV] = DBL_MAX;
        }
        __syncthreads();
        stage = ST_RESID;
    } else if (stage == ST_KKT1) { // RHSaffine (ref :1670-1689) = [rx; -ry; s - rz]: written by stage_resid's epilogues
        stage = ST_KKT_AFF;
    } else if (stage == ST_KKT_AFF) { // ref :1181-1210
        // c.dx1 b.dy1 h.dz1 c.dx2 b.dy2 h.dz2: formed by the epilogues of the two solves
        const double d6[6] = {g_S.dots[0][0], g_S.dots[0][1], g_S.dots[0][2], g_S.dots[1][0], g_S.dots[1][1], g_S.dots[1][2]};
        const double kap = wi.kap, tau = wi.tau;
        const double dtau_denom = kap / tau - d6[0] - d6[1] - d6[2];
        const double dtauaff = (g_S.sv[SV_RT] - kap + d6[3] + d6[4] + d6[5]) / dtau_denom;
        const double dkapaff = -kap - kap / tau * dtauaff;
        // LP rows, one pass: dz2 += dtauaff dz1 (kept in registers: nothing reads the affine dz2 of an LP row again), wdz = W dz2,
        // dsw = -wdz - lam and the line search's ratios -- the element-wise operations of the four separate passes, unchanged
        double rmin = DBL_MAX, smin = DBL_MAX;
        double wv[RL], dv[RL]; // fast path: W dz and ds of the thread's LP rows stay in registers across the line search
        double step_aff;
        if constexpr (regfast) {
            for_rl([&](int i) { return V4{dz2[i], dz1[i], lpw[i], lam[i]}; }, [&](int k, int, const V4 &r) {
                const double z2 = r.a + dtauaff * r.b, w = r.c * z2, d = -w - r.d;
                wv[k] = w; dv[k] = d;
                rmin = fmin(rmin, d / r.d); smin = fmin(smin, w / r.d);
            });
            if (tid == 0) { g_S.sv[SV_DTAUDEN] = dtau_denom; g_S.sv[SV_DTAUAFF] = dtauaff; g_S.sv[SV_DKAPAFF] = dkapaff; }
            step_aff = lp_line_search<T>(rmin, smin, l, tau, dtauaff, kap, dkapaff);
        } else {
        for_t_pre<T, 4>(l, [&](int i) { return V4{dz2[i], dz1[i], lpw[i], lam[i]}; }, [&](int i, const V4 &r) {
            const double z2 = r.a + dtauaff * r.b, w = r.c * z2, d = -w - r.d;
            wdz[i] = w; dsw[i] = d;
            rmin = fmin(rmin, d / r.d); smin = fmin(smin, w / r.d);
        });
        for_t_pre<T, 8>(m - l, [&](int i) { return V2{dz2[l + i], dz1[l + i]}; }, [&](int i, const V2 &r) { dz2[l + i] = r.a + dtauaff * r.b; });
        __syncthreads();
        if (tid == 0) { g_S.sv[SV_DTAUDEN] = dtau_denom; g_S.sv[SV_DTAUAFF] = dtauaff; g_S.sv[SV_DKAPAFF] = dkapaff; }
        if (P.nc > 0) {
            dev_scale<T, true>(ps, W, dz2, wdz);
            for_t_pre<T, 8>(m - l, [&](int i) { return V2{wdz[l + i], lam[l + i]}; }, [&](int i, const V2 &r) { dsw[l + i] = -r.a - r.b; });
        }
        step_aff = dev_line_search<T, true>(ps, W, tau, dtauaff, kap, dkapaff, rmin, smin);
        }
        const double oms_ = 1. - step_aff;
        const double sigma = fmin(fmax(oms_ * oms_ * oms_, SIGMAMIN), SIGMAMAX);
        const double mu = wi.mu;
        __syncthreads();
        if (tid == 0) { wi.step_aff = step_aff; wi.sigma = sigma; }
        // ---- RHScombined (ref :1282-1325) ----
        const double sigmamu = sigma * mu, oms = 1. - sigma;
        struct RC { double lam, ds, wz, w, rz; };
        if constexpr (regfast) for_rl([&](int i) { return V3{lam[i], lpw[i], rz[i]}; }, [&](int k, int i, const V3 &r) {
            const double d1_ = r.a * r.a + dv[k] * wv[k] - sigmamu; // (the same expressions as below, ds and W dz from the registers)
            const double q_ = d1_ / r.a;
            dsw[i] = q_;
            rhs2k[np + i] = -oms * r.c + r.b * q_;
        });
        else for_t_pre<T, 4>(l, [&](int i) { return RC{lam[i], dsw[i], wdz[i], lpw[i], rz[i]}; }, [&](int i, const RC &r) {
            // LP part: ds1 = lam*lam + dsw*wdz - sigmamu ; dsw = ds1/lam ; t1 = w*dsw (in registers) ; rhs2 row = -(1 - sigma) rz + t1
            const double d1_ = r.lam * r.lam + r.ds * r.wz - sigmamu;
            const double q_ = d1_ / r.lam;
            dsw[i] = q_;
            const double v = -oms * r.rz + r.w * q_;
            rhs2k[np + i] = v;
        });
        if (P.nc > 0) {
            // tiny cones: both loops below in one register-resident pass (a cone belongs to one thread: no barrier between them; the
            // ds1 tail t2 stays in registers)
            struct RC4 { D4 lam, ds, wz, q; double a, eta; };
            for_tiny<T>(P, [&](const Tiny &t) { gcdbl_p cs = csc + t.c * CSC_STRIDE; return RC4{tiny_rows(lam, t), tiny_rows(dsw, t), tiny_rows(wdz, t), tiny_rows(qv, t), cs[CS_A], cs[CS_ETA]}; },
                        [&](const Tiny &t, const RC4 &r) {
                const int d = t.d, o = t.o;
                double ll = 0., dw = 0., u1sq = 0.;
#pragma unroll
                for (int k = 0; k < TINY_D; k++) if (k < d) { ll += r.lam.v[k] * r.lam.v[k]; dw += r.ds.v[k] * r.wz.v[k]; }
                const double l0 = r.lam.v[0], a0 = r.ds.v[0], w0_ = r.wz.v[0];
                const double p0 = ll - sigmamu + dw;
                double zeta = 0., pk[TINY_D], nd[TINY_D];
#pragma unroll
                for (int k = 1; k < TINY_D; k++) if (k < d) {
                    const double lk = r.lam.v[k];
                    pk[k] = (l0 * lk + l0 * lk) + (a0 * r.wz.v[k] + w0_ * r.ds.v[k]);
                    u1sq += lk * lk; zeta += lk * pk[k];
                }
                const double rho = l0 * l0 - u1sq;
                const double factor = (zeta / l0 - p0) / rho;
#pragma unroll
                for (int k = 1; k < TINY_D; k++) if (k < d) { nd[k] = factor * r.lam.v[k] + pk[k] / l0; dsw[o + k] = nd[k]; }
                nd[0] = (l0 * p0 - zeta) / rho; dsw[o] = nd[0];
                // t1 = W * (lam \ ds) on the cone
                double zeta2 = 0.;
#pragma unroll
                for (int k = 1; k < TINY_D; k++) if (k < d) zeta2 += r.q.v[k] * nd[k];
                const double z0 = nd[0], factor2 = z0 + zeta2 / (1. + r.a);
#pragma unroll
                for (int k = 1; k < TINY_D; k++) if (k < d) t1[o + k] = r.eta * (nd[k] + factor2 * r.q.v[k]);
                t1[o] = r.eta * (r.a * z0 + zeta2);
            });
            // wave cones: both loops in one lane-per-row pass (a row's intermediate values stay on its lane; the head's new ds is formed by
            // every lane from the reduced sums)
            struct RCW { double lamA, dsA, wzA, lamB, dsB, wzB, qB, l0, a0, w0, a, eta; };
            for_wave<T, wave_cones_in_flight<T>()>(P, [](const WCone &, int) { return NoPre{}; },
                [&](const WCone &b, int lane, NoPre) {
                gcdbl_p cs = csc + b.c * CSC_STRIDE;
                return RCW{wrow(lam, b, lane, 0), wrow(dsw, b, lane, 0), wrow(wdz, b, lane, 0), wrow(lam, b, lane, 1), wrow(dsw, b, lane, 1), wrow(wdz, b, lane, 1),
                           wrow(qv, b, lane, 1), lam[b.o], dsw[b.o], wdz[b.o], cs[CS_A], cs[CS_ETA]};
            }, [&](const WCone &b, int lane, NoPre, const RCW &r) {
                const int o = b.o, k = 1 + lane;
                const bool actA = lane < b.d, actB = k < b.d;
                double ll = 0., dw = 0., u1sq = 0.;
                if (actA) { ll += r.lamA * r.lamA; dw += r.dsA * r.wzA; }
                ll = wave_reduce<OpSum>(ll); dw = wave_reduce<OpSum>(dw);
                const double l0 = r.l0, a0 = r.a0, w0_ = r.w0;
                const double p0 = ll - sigmamu + dw;
                double zeta = 0., pk_ = 0.;
                if (actB) {
                    const double lk = r.lamB;
                    pk_ = (l0 * lk + l0 * lk) + (a0 * r.wzB + w0_ * r.dsB);
                    u1sq += lk * lk; zeta += lk * pk_;
                }
                u1sq = wave_reduce<OpSum>(u1sq); zeta = wave_reduce<OpSum>(zeta);
                const double rho = l0 * l0 - u1sq;
                const double factor = (zeta / l0 - p0) / rho;
                const double ndB = factor * r.lamB + pk_ / l0; // the new ds of row k
                if (actB) dsw[o + k] = ndB;
                const double nd0 = (l0 * p0 - zeta) / rho;       // ... and of the head
                if (lane == 0) dsw[o] = nd0;
                // t1 = W * (lam \ ds) on the cone
                double zeta2 = 0.;
                if (actB) zeta2 += r.qB * ndB;
                zeta2 = wave_reduce<OpSum>(zeta2);
                const double factor2 = nd0 + zeta2 / (1. + r.a);
                if (actB) t1[o + k] = r.eta * (ndB + factor2 * r.qB);
                if (lane == 0) t1[o] = r.eta * (r.a * nd0 + zeta2);
            });
            for_cones_rest<T>(ps, [&](int c, auto G, int ln) {
                constexpr int g = decltype(G)::value;
                const int o = P.cone_off[c], d = P.cq[c];
                // conic products (ref :1357-1378): ds1 = lam o lam + dsw o wdz - sigmamu e
                double ll = 0., dw = 0., u1sq = 0.;
                for (int k = ln; k < d; k += g) { ll += lam[o + k] * lam[o + k]; dw += dsw[o + k] * wdz[o + k]; }
                ll = grp_sum<g>(ll); dw = grp_sum<g>(dw);
                const double l0 = lam[o], a0 = dsw[o], w0_ = wdz[o];
                const double p0 = ll - sigmamu + dw;
                // conic division v = lam \ ds1 (ref :1330-1351)
                double zeta = 0.;
                for (int k = 1 + ln; k < d; k += g) {
                    const double lk = lam[o + k];
                    const double pk_ = (l0 * lk + l0 * lk) + (a0 * wdz[o + k] + w0_ * dsw[o + k]);
                    t2[o + k] = pk_; // ds1 tail
                    u1sq += lk * lk; zeta += lk * pk_;
                }
                u1sq = grp_sum<g>(u1sq); zeta = grp_sum<g>(zeta);
                const double rho = l0 * l0 - u1sq;
                const double factor = (zeta / l0 - p0) / rho;
                for (int k = 1 + ln; k < d; k += g) dsw[o + k] = factor * lam[o + k] + t2[o + k] / l0;
                if (ln == 0) dsw[o] = (l0 * p0 - zeta) / rho;
            });
            __syncthreads();
            for_cones_rest<T>(ps, [&](int c, auto G, int ln) { // t1 = W * (lam \ ds) on the cone part
                constexpr int g = decltype(G)::value;
                const int o = P.cone_off[c], d = P.cq[c];
                gcdbl_p cs = csc + c * CSC_STRIDE;
                double zeta = 0.;
                for (int k = 1 + ln; k < d; k += g) zeta += qv[o + k] * dsw[o + k];
                zeta = grp_sum<g>(zeta);
                const double z0 = dsw[o], factor = z0 + zeta / (1. + cs[CS_A]), eta = cs[CS_ETA];
                for (int k = 1 + ln; k < d; k += g) t1[o + k] = eta * (dsw[o + k] + factor * qv[o + k]);
                if (ln == 0) t1[o] = eta * (cs[CS_A] * z0 + zeta);
            });
        }
        __syncthreads();
        for_t_pre<T, 8>(np, [&](int j) { return V1{rhs2k[j]}; }, [&](int j, const V1 &r) { rhs2k[j] = r.a * oms; });
        for_t_pre<T, 8>(m - l, [&](int i) { return V2{rz[l + i], t1[l + i]}; }, [&](int i, const V2 &r) { rhs2k[np + l + i] = -oms * r.a + r.b; });
        __syncthreads();
        stage = ST_KKT_COMB;
    } else { // ST_KKT_COMB, ref :1212-1252
        const double e3[3] = {g_S.dots[1][0], g_S.dots[1][1], g_S.dots[1][2]}; // c.dx2 b.dy2 h.dz2 (epilogue of the combined solve)
        const double kap = wi.kap, tau = wi.tau, sigma = wi.sigma;
        const double dtauaff = g_S.sv[SV_DTAUAFF], dkapaff = g_S.sv[SV_DKAPAFF];
        const double bkap = kap * tau + dkapaff * dtauaff - sigma * wi.mu;
        const double dtau = ((1. - sigma) * g_S.sv[SV_RT] - bkap / tau + e3[0] + e3[1] + e3[2]) / g_S.sv[SV_DTAUDEN];
        const double dkap = -(bkap + kap * dtau) / tau;
        // (dx2 += dtau dx1, dy2 += dtau dy1 are formed inside the update of x and y below: nothing else reads the combined dx, dy)
        // LP rows, one pass: dz2 += dtau dz1, wdz = W dz2 (registers), dsw = -(dsw + wdz) and the line search's ratios
        double rmin = DBL_MAX, smin = DBL_MAX;
        struct CB { double z2, z1, w, ds, lam; };
        double zv[RL], dv[RL]; // fast path: the combined dz and ds of the thread's LP rows stay in registers across the line search
        double st;
        if constexpr (regfast) {
            for_rl([&](int i) { return CB{dz2[i], dz1[i], lpw[i], dsw[i], lam[i]}; }, [&](int k, int, const CB &r) {
                const double z2 = r.z2 + dtau * r.z1, w = r.w * z2, d = -(r.ds + w);
                zv[k] = z2; dv[k] = d;
                rmin = fmin(rmin, d / r.lam); smin = fmin(smin, w / r.lam);
            });
            st = GAMMA * lp_line_search<T>(rmin, smin, l, tau, dtau, kap, dkap);
        } else {
        for_t_pre<T, 4>(l, [&](int i) { return CB{dz2[i], dz1[i], lpw[i], dsw[i], lam[i]}; }, [&](int i, const CB &r) {
            const double z2 = r.z2 + dtau * r.z1, w = r.w * z2, d = -(r.ds + w);
            dz2[i] = z2; dsw[i] = d;
            rmin = fmin(rmin, d / r.lam); smin = fmin(smin, w / r.lam);
        });
        for_t_pre<T, 8>(m - l, [&](int i) { return V2{dz2[l + i], dz1[l + i]}; }, [&](int i, const V2 &r) { dz2[l + i] = r.a + dtau * r.b; });
        __syncthreads();
        if (P.nc > 0) {
            dev_scale<T, true>(ps, W, dz2, wdz);
            for_t_pre<T, 8>(m - l, [&](int i) { return V2{dsw[l + i], wdz[l + i]}; }, [&](int i, const V2 &r) { dsw[l + i] = -(r.a + r.b); });
        }
        st = GAMMA * dev_line_search<T, true>(ps, W, tau, dtau, kap, dkap, rmin, smin);
        if (P.nc > 0) dev_scale<T, true>(ps, W, dsw, dsa);
        }
        // the new iterate goes to the OTHER buffer set when the current one is the saved best iterate (ShI::best), else in place
        const int tgt = uni(g_S.best == g_S.cur ? 1 - g_S.cur : g_S.cur);
        const IterBuf nw = iter_buf(P, I, W, tgt);
        for_t_pre<T, 4>(n, [&](int j) { return V3{wx[j], dx2[j], dx1[j]}; }, [&](int j, const V3 &r) { nw.x[j] = r.a + st * (r.b + dtau * r.c); });
        for_t_pre<T, 4>(p, [&](int j) { return V3{wy[j], dy2[j], dy1[j]}; }, [&](int j, const V3 &r) { nw.y[j] = r.a + st * (r.b + dtau * r.c); });
        // LP rows: ds = W dsw formed in registers
        if constexpr (regfast) for_rl([&](int i) { return V3{wz[i], lpw[i], wsl[i]}; }, [&](int k, int i, const V3 &r) { nw.z[i] = r.a + st * zv[k]; nw.s[i] = r.c + st * (r.b * dv[k]); });
        else for_t_pre<T, 4>(l, [&](int i) { return CB{wz[i], dz2[i], lpw[i], dsw[i], wsl[i]}; }, [&](int i, const CB &r) { nw.z[i] = r.z2 + st * r.z1; nw.s[i] = r.lam + st * (r.w * r.ds); });
        for_t_pre<T, 4>(m - l, [&](int i) { return V4{wz[l + i], dz2[l + i], wsl[l + i], dsa[l + i]}; }, [&](int i, const V4 &r) { nw.z[l + i] = r.a + st * r.b; nw.s[l + i] = r.c + st * r.d; });
        __syncthreads(); // (every thread has read g_S.cur / best)
        if (tid == 0) {
            g_S.cur = tgt;
            wi.nitref3 = kref; wi.step = st;
            wi.kap = kap + st * dkap;
            wi.tau = tau + st * dtau;
        }
        __syncthreads();
        stage = ST_RESID;
    }
    TICK_END(TK_KPOST);
    return stage;
}

template <int T>
__device__ __forceinline__ int kkt_post_any(int ps, gdbl_p I, gdbl_p W, int stage) {
    const DevPat &P = c_pat[ps];
    constexpr int RL = 16; // (= kkt_post's)
    const bool rf = waves_per_eu<T>() <= 2 && P.nc == 0 && 4 * P.l > RL * T && P.l <= RL * T && (stage == ST_KKT_AFF || stage == ST_KKT_COMB);
    if (rf) return kkt_post<T, true>(ps, I, W, stage);
    return kkt_post<T, false>(ps, I, W, stage);
}

// ---------------- per-instance prologue of a solve (its state ends up in g_S); returns 1 if it was warm-started ----------------
template <int T>
static __device__ __noinline__ __attribute__((not_tail_called)) int instance_begin(int ps, gdbl_p I, gdbl_p W, double warm) {
    ps = uni(ps); I = uni_ptr(I); W = uni_ptr(W);
    const DevPat &P = c_pat[ps];
    const int n = P.n, p = P.p, m = P.m, l = P.l, np = P.n + P.p;
    const int tid = threadIdx.x;
    DevInfo *ginfo = (DevInfo *)(I + P.i_info); // (C-style cast: in the LDS-resident build the slab pointer is an LDS pointer)
    DevInfo &wi = g_S.wi;
    {
        gdbl_p cv = I + P.i_c, hv = I + P.i_h, bv = I + P.i_b, Vv = I + P.i_Vv;
        gdbl_p rhs1k = W + P.w_rhs1k, rhs2k = W + P.w_rhs2k;
        int phase = 0;
        __syncthreads();
        if (tid == 0) { // sticky across solve() calls like the reference's w.i
            wi = *ginfo; g_S.bi = wi;
            // warm start (N3, not in the reference): needs a previous OPTIMAL solve of this instance
            g_S.fl[FL_WARM] = (warm > 0. && wi.n_factor > 0 && (wi.exitcode == 0 || wi.exitcode == 10)) ? 1 : 0;
            wi.n_factor = 0; wi.n_ldlsolve = 0; wi.n_sweep = 0;
            g_S.fl[FL_FATAL] = 0; g_S.fl[FL_CODE] = -7; g_S.done = 0; g_S.kref = 0;
            g_S.cur = 0; g_S.best = -1; // the iterate starts in the instance slab (warm start: the previous solution is there), no best iterate yet
            for (int q = 0; q < 12; q++) g_S.tick[q] = 0;
            g_S.tick[7] = wall_clock64();
        }
        // resetKKTScalings (ref :807-846)
        FOR_T(i, l) Vv[i] = -1.;
        for_cones<T>(ps, [&](int c, auto G, int ln) {
            constexpr int g = decltype(G)::value;
            const int d = P.cq[c];
            gdbl_p v = Vv + P.cone_vbase[c];
            for (int k = ln; k < d; k += g) { v[k] = -1.; v[2 * d + 1 + k] = 0.; if (k >= 1) v[d + k] = 0.; }
            if (ln == 0) { v[d] = -1.; v[2 * d] = 1.; }
        });
        // rhs1 = [0; b; h expanded], rhs2 = [-c; 0; 0]   (ref :865-886)
        FOR_T(i, np + m) { rhs1k[i] = 0.; rhs2k[i] = 0.; }
        __syncthreads();
        { // KKT entries in the factor's target order (one gather per solve; the scaling part is refreshed per iteration)
            gdbl_p Kt = W + P.w_Kt;
            if (P.tile) { // dense tile image of K: zero, then scatter the structural entries (+ 1 on the padding diagonals);
                          // hybrid: only the padding -- the scalar factor program writes the structural entries every pass
                gdbl_p Kimg = W + P.w_Kimg;
                const int nimg = (P.nb + P.nt) * 256;
                FOR_T(t, nimg) Kimg[t] = 0.;
                __syncthreads();
                for_t_pre<T, 8>(P.tl_nimg, [&](int e) { return IV1{P.tl_img_dst[e], I[P.tl_img_src[e]]}; }, [&](int e, const IV1 &r) { Kimg[r.i] = r.a; });
            }
            if (P.tile != 1)
            for_t_pre<T, 8>(P.fac_nt, [&](int t) { return V1{I[P.fac_src[t]]}; }, [&](int t, const V1 &r) { Kt[t] = r.a; });
        }
        {
            double nr3[3] = {0., 0., 0.};
            for_t_pre<T, 4>(n, [&](int j) { return V1{cv[j]}; }, [&](int j, const V1 &r) { rhs2k[j] = -r.a; nr3[0] += r.a * r.a; });
            for_t_pre<T, 4>(p, [&](int j) { return V1{bv[j]}; }, [&](int j, const V1 &r) { rhs1k[n + j] = r.a; nr3[1] += r.a * r.a; });
            for_t_pre<T, 8>(m, [&](int i) { return V1{hv[i]}; }, [&](int i, const V1 &r) { rhs1k[np + i] = r.a; nr3[2] += r.a * r.a; });
            blk_reduce<OpSum, T, 3>(phase, nr3);
            if (tid == 0) {
                g_S.sv[SV_RESX0] = fmax(1., sqrt(nr3[0])); g_S.sv[SV_RESY0] = fmax(1., sqrt(nr3[1])); g_S.sv[SV_RESZ0] = fmax(1., sqrt(nr3[2]));
            }
        }
    }
    __syncthreads();
    if (!g_S.fl[FL_WARM]) return 0;
    {
        // ---- warm start: previous (x, y, z, s) of this instance (still in its slab, backscaled) re-equilibrated and
        // pushed into the cone -- LP rows floored at warm * mean|.|, cone heads at ||tail|| + the same margin --
        // instead of the two initialisation solves (ref :929-972); tau = kap = 1, first pass = iteration 0.
        gdbl_p wx = I + P.i_x, wy = I + P.i_y, wz = I + P.i_z, wsl = I + P.i_s, cv = I + P.i_c;
        gcdbl_p xe = I + P.i_xe, ae = I + P.i_ae, ge = I + P.i_ge;
        gdbl_p rhs1k = W + P.w_rhs1k;
        int phase = 0;
        FOR_T(j, n) wx[j] *= xe[j];
        FOR_T(r, p) wy[r] *= ae[r];
        double sz2[2] = {0., 0.};
        FOR_T(i, m) { const double g = ge[i], zv = wz[i] * g, sv = wsl[i] / g; wz[i] = zv; wsl[i] = sv; sz2[0] += fabs(sv); sz2[1] += fabs(zv); }
        blk_reduce<OpSum, T, 2>(phase, sz2);
        const double as = warm * sz2[0] / (double)max(1, m), az = warm * sz2[1] / (double)max(1, m);
        __syncthreads();
        FOR_T(i, l) { wsl[i] = fmax(wsl[i], as); wz[i] = fmax(wz[i], az); }
        for_cones<T>(ps, [&](int c, auto G, int ln) {
            constexpr int g = decltype(G)::value;
            const int o = P.cone_off[c], d = P.cq[c];
            double ts = 0., tz = 0.;
            for (int k = 1 + ln; k < d; k += g) { ts += wsl[o + k] * wsl[o + k]; tz += wz[o + k] * wz[o + k]; }
            ts = grp_sum<g>(ts); tz = grp_sum<g>(tz);
            if (ln == 0) { wsl[o] = fmax(wsl[o], sqrt(ts) + as); wz[o] = fmax(wz[o], sqrt(tz) + az); }
        });
        FOR_T(j, n) rhs1k[j] = -cv[j]; // as after the second init solve (ref :966-972)
        if (tid == 0) {
            wi.nitref1 = 0; wi.nitref2 = 0;
            wi.kap = 1.; wi.tau = 1.; wi.step = 0.; wi.step_aff = 0.; wi.pinf = 0; wi.dinf = 0;
            g_S.sv[SV_PRESPREV] = DBL_MAX;
        }
        __syncthreads();
    }
    return 1;
}

// an instance has finished (its state is in g_S): exit code, Information and the phase timers go back to its slab / workspace
__device__ __forceinline__ void instance_end(const DevPat &P, gdbl_p I, gdbl_p W) {
    if (threadIdx.x == 0) {
        DevInfo &wi = g_S.wi;
        wi.exitcode = g_S.fl[FL_FATAL] ? -7 : g_S.fl[FL_CODE];
        wi.solve_us = (double)(wall_clock64() - g_S.tick[7]) * 0.01;
        *(DevInfo *)(I + P.i_info) = wi;
        // phase timers (microseconds) into the last row of the trace buffer: factor, LDL solves, refinement
        // residuals, KKT post-processing, residual/statistics/scalings stage, forward part of the solves, [6] total
        gdbl_p tr = W + P.w_trace + (size_t)(TRACE_ROWS - 1) * TRACE_COLS;
        for (int q = 0; q < TK_COUNT; q++) tr[q] = (double)g_S.tick[q] * 0.01;
        tr[6] = (double)(wall_clock64() - g_S.tick[7]) * 0.01;
        for (int q = 8; q < 12; q++) tr[q - 1] = (double)g_S.tick[q] * 0.01;
        g_S.done = 1;
    }
}

// ---------------- one instance, whole solve (reference Solver::solve, src/eicos.cpp:848-1262): the stage machine ----------------
template <int T, int NLDS, bool I16>
__device__ __forceinline__ void solve_instance(int ps, gdbl_p I, gdbl_p W, double warm) {
    const DevPat &P = c_pat[ps];
    int stage = ST_FACTOR, iter = -1; // iter = -1 while initialising
    if (instance_begin<T>(ps, I, W, warm)) { stage = ST_RESID; iter = 0; } // warm start: no initialisation solves
    __syncthreads();
    while (stage != ST_DONE) {
        if (stage == ST_FACTOR) {
            if (P.tile != 1) { if (P.fac_defer) stage_factor<T, NLDS, I16, true>(ps, W); else stage_factor<T, NLDS, I16, false>(ps, W); } // scalar program (hybrid: everything below the top block + its image)
            if (P.tile) stage_factor_tiles<T, NLDS>(ps, I, W, iter);
            if (g_S.fl[FL_FATAL]) { // zero pivot -> fatal, no backscale (ref :901-905,1166-1170): the iterate as it stands is the result
                if (g_S.cur != 0) {
                    const IterBuf src = iter_buf(P, I, W, 1), dst = iter_buf(P, I, W, 0);
                    for (int j = threadIdx.x; j < P.n; j += T) dst.x[j] = src.x[j];
                    for (int j = threadIdx.x; j < P.p; j += T) dst.y[j] = src.y[j];
                    for (int i = threadIdx.x; i < P.m; i += T) { dst.z[i] = src.z[i]; dst.s[i] = src.s[i]; }
                    __syncthreads();
                }
                instance_end(P, I, W); stage = ST_DONE;
            }
            else stage = (iter < 0) ? ST_KKT_INIT1 : ST_KKT1;
        } else if (stage == ST_RESID) {
            if (stage_resid<T, NLDS, I16>(ps, I, W, iter) == ST_DONE) { __syncthreads(); instance_end(P, I, W); stage = ST_DONE; }
            else stage = ST_FACTOR;
        } else if (NLDS == 1 && P.dual && (stage == ST_KKT_INIT1 || stage == ST_KKT1)) {
            // the two right-hand sides of this point of the algorithm do not depend on each other: one dual solve
            if constexpr (NLDS == 1) {
                if (stage == ST_KKT1) kkt_post_any<T>(ps, I, W, ST_KKT1); // RHSaffine (ref :1176) needs the residuals only
                kkt_solve<T, 1, I16, 2, true>(ps, I, I, W, stage, 3);
                const int second = (stage == ST_KKT1) ? ST_KKT_AFF : ST_KKT_INIT2;
                if (stage == ST_KKT_INIT1) kkt_post_any<T>(ps, I, W, ST_KKT_INIT1);
                __syncthreads();
                if (threadIdx.x == 0) g_S.kref = g_S.kref2;
                __syncthreads();
                const int next = kkt_post_any<T>(ps, I, W, second);
                if (next == ST_RESID) iter = 0; // (after the initialisation pair)
                stage = next;
            }
        } else {
            kkt_solve<T, NLDS, I16, 1>(ps, I, I, W, stage, 1);
            const int next = kkt_post_any<T>(ps, I, W, stage);
            if (next == ST_RESID) iter = (stage == ST_KKT_INIT2) ? 0 : iter + 1; // a pass of the main loop completed
            stage = next;
        }
    }
    __syncthreads();
}

// The same updateData for patterns whose A and G values fit LDS (most: MPC02 needs 76 KB for the values + 48 KB for the
// row / column maxima): ENTRY-parallel instead of thread-per-column.  The equilibrated working copy of the values and the
// running maxima stay in LDS across the three sweeps, every pass is unit-stride over the entries, the maxima are integer
// atomic maxima on the bit patterns of |a| (non-negative doubles order like their bit patterns: exact), and HBM sees
// each input once plus the outputs.  Same arithmetic in the same order per entry as k_update (rows, then columns;
// cone rows share the SUM of their row maxima; |a| < 1e-6 -> 1): bit-identical results.
// LDS: [ xt (n) | at (p) | gt (m) | Av | Gv ] doubles.
// LDSV = false (values too large for LDS, e.g. the dense-front config: 131 k entries): the same entry-parallel passes with
// the working copy of the values IN PLACE in the instance slab (it is their destination anyway) and only the maxima in
// LDS; every pass over the values then streams them from HBM with batched loads (for_t_pre), several workgroups per CU.
// update_instance = the per-instance body: called by k_update_lds (one launch over a range of instances) and -- the fused path,
// eicos_batch_update_solve -- by k_solve itself at the start of an instance's solve, the row / column maxima in the then idle sweep vector,
// so that a batch handed over in (pinned) HOST memory is pulled over PCIe by the solve's own workgroups while other workgroups compute:
// the registers and the LDS of a CU are fully owned by its resident solve workgroups, no other kernel could run beside them.
// `q` = row of this instance in the input arrays (NULL = keep that group), I = its slab in HBM.
typedef double EICOS_GLOBAL *hbm_p; // (global memory in every build: the LDS-resident build's gdbl_p is an LDS pointer)
template <int T, bool LDSV>
static __device__ __noinline__ __attribute__((not_tail_called)) void update_instance(int ps, hbm_p I, size_t q, const double *Gpr, const double *Apr, const double *cin,
                                                                                      const double *hin, const double *bin) {
    ps = uni(ps); I = uni_ptr(I);
    const DevPat &P = c_pat[ps];
    const int n = P.n, p = P.p, m = P.m, l = P.l, nnzA = P.nnzA, nnzG = P.nnzG;
    double *xt = g_dyn, *at = xt + n, *gt = at + p;
    unsigned long long *xtb = reinterpret_cast<unsigned long long *>(xt), *atb = reinterpret_cast<unsigned long long *>(at), *gtb = reinterpret_cast<unsigned long long *>(gt);
    auto sq = [](double a) { return fabs(a) < 1e-6 ? 1. : sqrt(a); };
    hbm_p Av = I + P.i_Av, Gv = I + P.i_Gv, cagv = I + P.i_cag, rAv = I + P.i_rA, rGv = I + P.i_rG;
    hbm_p cv = I + P.i_c, hv = I + P.i_h, bv = I + P.i_b, xe = I + P.i_xe, ae = I + P.i_ae, ge = I + P.i_ge;
    DevInfo *ginfo = reinterpret_cast<DevInfo *>(I + P.i_info);
    const bool was_eq = ginfo->equilibrated != 0;
    auto sA = [&] { if constexpr (LDSV) return gt + m; else return Av; }(); // working copy of the values: LDS, or in place
    auto sG = [&] { if constexpr (LDSV) return gt + m + nnzA; else return Gv; }();
    __syncthreads();
    // un-equilibrate what is kept, overwrite what is given (ref :2053-2074, :389-404) -> working copy of the values
    for_t_pre<T, 4>(nnzA, [&](int k) {
        if (Apr) return V3{Apr[(size_t)q * nnzA + k], 1., 1.};
        return was_eq ? V3{Av[k], ae[P.Air[k]], xe[P.Acol[k]]} : V3{Av[k], 1., 1.};
    }, [&](int k, const V3 &r) { sA[k] = (Apr || !was_eq) ? r.a : r.a * (r.b * r.c); });
    for_t_pre<T, 4>(nnzG, [&](int k) {
        if (Gpr) return V3{Gpr[(size_t)q * nnzG + k], 1., 1.};
        return was_eq ? V3{Gv[k], ge[P.Gir[k]], xe[P.Gcol[k]]} : V3{Gv[k], 1., 1.};
    }, [&](int k, const V3 &r) { sG[k] = (Gpr || !was_eq) ? r.a : r.a * (r.b * r.c); });
    FOR_T(j, n) cv[j] = cin ? cin[(size_t)q * n + j] : (was_eq ? cv[j] * xe[j] : cv[j]);
    FOR_T(r, p) bv[r] = Apr ? bin[(size_t)q * p + r] : (was_eq ? bv[r] * ae[r] : bv[r]);
    FOR_T(i, m) hv[i] = Gpr ? hin[(size_t)q * m + i] : (was_eq ? hv[i] * ge[i] : hv[i]);
    __syncthreads();
    // the accumulated scalings live in registers of the thread that owns the index (fixed FOR_T mapping); they are
    // written once at the end.  Up to 8 indices per thread and vector: patterns beyond that take the generic kernel.
    constexpr int OWN = 8;
    double xacc[OWN], aacc[OWN], gacc[2 * OWN];
#pragma unroll
    for (int u = 0; u < OWN; u++) { xacc[u] = 1.; aacc[u] = 1.; gacc[2 * u] = 1.; gacc[2 * u + 1] = 1.; }
    for (int it = 0; it < EQUIL_ITERS; it++) {
        FOR_T(j, n) xtb[j] = 0ull;
        FOR_T(r, p) atb[r] = 0ull;
        FOR_T(i, m) gtb[i] = 0ull;
        __syncthreads();
        // column maxima over A and G, row maxima of A and of G: one pass over the entries
        for_t_pre<T, 8>(nnzA, [&](int k) { return IIV{P.Acol[k], P.Air[k], sA[k]}; }, [&](int k, const IIV &r) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(r.a)); atomicMax(&xtb[r.i], b); atomicMax(&atb[r.j], b); });
        for_t_pre<T, 8>(nnzG, [&](int k) { return IIV{P.Gcol[k], P.Gir[k], sG[k]}; }, [&](int k, const IIV &r) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(r.a)); atomicMax(&xtb[r.i], b); atomicMax(&gtb[r.j], b); });
        __syncthreads();
        FOR_T(j, n) xt[j] = sq(xt[j]);
        FOR_T(r, p) at[r] = sq(at[r]);
        FOR_T(i, l) gt[i] = sq(gt[i]); // cone rows: sqrt after the per-cone sum (ref :338-350)
        __syncthreads();
        FOR_T(c, P.nc) { // cone rows share the SUM of their row maxima
            const int o = P.cone_off[c], d = P.cq[c];
            double tot = 0.;
            for (int k = 0; k < d; k++) tot += gt[o + k];
            tot = sq(tot);
            for (int k = 0; k < d; k++) gt[o + k] = tot;
        }
        __syncthreads();
        // rows first, then columns -- same division order as the reference (:353-356)
        for_t_pre<T, 8>(nnzA, [&](int k) { return IIV{P.Acol[k], P.Air[k], sA[k]}; }, [&](int k, const IIV &r) { sA[k] = (r.a / at[r.j]) / xt[r.i]; });
        for_t_pre<T, 8>(nnzG, [&](int k) { return IIV{P.Gcol[k], P.Gir[k], sG[k]}; }, [&](int k, const IIV &r) { sG[k] = (r.a / gt[r.j]) / xt[r.i]; });
#pragma unroll
        for (int u = 0; u < OWN; u++) { // (compile-time register indices: the accumulators must not go to scratch)
            const int j = threadIdx.x + u * T;
            if (j < n) xacc[u] *= xt[j];
            if (j < p) aacc[u] *= at[j];
        }
#pragma unroll
        for (int u = 0; u < 2 * OWN; u++) { const int i = threadIdx.x + u * T; if (i < m) gacc[u] *= gt[i]; }
        __syncthreads();
    }
    // write back: scalings, scaled c, b, h, the values
#pragma unroll
    for (int u = 0; u < OWN; u++) {
        const int j = threadIdx.x + u * T;
        if (j < n) { xe[j] = xacc[u]; cv[j] = cv[j] / xacc[u]; }
        if (j < p) { ae[j] = aacc[u]; bv[j] = bv[j] / aacc[u]; }
    }
#pragma unroll
    for (int u = 0; u < 2 * OWN; u++) { const int i = threadIdx.x + u * T; if (i < m) { ge[i] = gacc[u]; hv[i] = hv[i] / gacc[u]; } }
    if constexpr (LDSV) {
        FOR_T(k, nnzA) Av[k] = sA[k];
        FOR_T(k, nnzG) Gv[k] = sG[k];
    } else __syncthreads(); // (in place: the gathers below read entries other threads scaled)
    // sliced-ELL value copies for the products, straight from the working copy; *_src is relative to Av (G values follow at i_Gv - i_Av)
    const int grel = P.i_Gv - P.i_Av;
    auto ell_copy = [&](hbm_p dst, gint_p src, int cnt) {
        if constexpr (LDSV) for_t_pre<T, 8>(cnt, [&](int k) { return src[k]; }, [&](int k, int e) { dst[k] = e < 0 ? 0. : (e < grel ? sA[e] : sG[e - grel]); });
        else for_t_pre<T, 8>(cnt, [&](int k) { const int e = src[k]; return IV1{e, Av[max(e, 0)]}; }, [&](int k, const IV1 &r) { dst[k] = r.i < 0 ? 0. : r.a; });
    };
    ell_copy(cagv, P.cag_src, P.cag_slots + 1);
    ell_copy(rAv, P.rA_src, P.rA_slots + 1);
    ell_copy(rGv, P.rG_src, P.rG_slots + 1);
    if (P.gt_on) ell_copy(I + P.i_Gt, P.gt_src, P.gt_nt * 256); // G as dense tiles (tile-internal operand order)
    if (threadIdx.x == 0) { // static-regularisation constants read by the factor program
        hbm_p cst = I + P.i_cst;
        cst[0] = DELTASTAT; cst[1] = -DELTASTAT; cst[2] = 0.; cst[3] = 1.; // [3]: diagonal of the padding nodes (tile mode)
        ginfo->equilibrated = 1;
    }
    __syncthreads();
}

template <int T, int NLDS, bool I16>
__global__ __launch_bounds__(T, (waves_per_eu<T>())) void k_solve(
    int ps, double *inst, double *work, int B, int *queue, const int *order, double warm, double dyn_delta, double dyn_eps, UpdArgs upd) {
    const DevPat &P = c_pat[ps];
#if EICOS_LDSRES
    static_assert(NLDS >= 1, "LDS-resident variant: sweep vector + tables in LDS");
    gdbl_p W = (gdbl_p)(g_dyn + P.lr_work), Il = (gdbl_p)(g_dyn + P.lr_inst); // the slabs of the instance being solved
    double *Wglob = work + (size_t)blockIdx.x * P.work_stride;
    for (int q = threadIdx.x; q < (int)P.work_stride; q += T) W[q] = Wglob[q]; // (zero padding slots, cone state kept between solves)
#else
    gdbl_p W = (gdbl_p)work + (size_t)blockIdx.x * P.work_stride;
#endif
    if constexpr (NLDS >= 1) { // every slice table -> LDS, once per workgroup (same plans for every instance)
        int *dst = reinterpret_cast<int *>(g_dyn + P.lds_tab); // (doubles from the start of the dynamic LDS: behind the vectors)
        auto stage = [&](const PackedSlice EICOS_GLOBAL *src, int cnt, int at) {
            gint_p si = reinterpret_cast<gint_p>(src);
            for (int q = threadIdx.x; q < cnt * 4; q += T) dst[at * 4 + q] = si[q];
        };
        stage(P.fsl, P.nfs + P.nfs_solo + P.nfs_ext, P.lm_f); stage(P.bsl, P.nbs + P.nbs_solo, P.lm_b); stage(P.cag_sl, P.cag_ns, P.lm_cag);
        stage(P.rA_sl, P.rA_ns, P.lm_rA); stage(P.rG_sl, P.rG_ns, P.lm_rG);
        if (P.lm_fac >= 0) stage(P.fac_sl, P.fac_ns, P.lm_fac);
        __syncthreads();
    }
#if EICOS_UBL
    // U in LDS: the padding slots and the dummy slot (value 0) are never written by the factor program -- zero the array once per workgroup
    for (int q = threadIdx.x; q < P.ub_len; q += T) g_dyn[P.ub_lds + q] = 0.;
    __syncthreads();
#endif
    if (threadIdx.x == 0) { g_S.dyn_delta = dyn_delta; g_S.dyn_eps = dyn_eps; }
    // Instances differ in iteration count (12..18 on the headline batch): after its first instance (= its own index, so
    // that workspace slot g holds the history of instance g when the batch fits the grid) a workgroup pulls the next
    // unsolved instance from a queue instead of striding through the batch.
    // `order` (batches larger than one instance per CU): instances sorted by the work their previous solve took, longest first.
    for (int g = blockIdx.x; g < B;) {
        const int id = order ? order[g] : g;
        if constexpr (NLDS >= 1) { // fused updateData (launch.hpp: UpdArgs): the maxima live in the sweep vector, idle between two instances
            if (upd.on) {
                if (upd.flags) { // staged host arrays: the host is still copying -- wait for the chunk that holds this instance (launch.hpp: UpdArgs)
                    if (threadIdx.x == 0) {
                        const unsigned *f = upd.flags + id / upd.chunk;
                        const unsigned long long t0 = wall_clock64();
                        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != upd.seq) {
                            // a poll is a read over PCIe: ~30 us between two (a chunk arrives every ~250 us) -- 512 workgroups polling every
                            // 4 us were 10 GB/s of link traffic beside the rows the other workgroups are pulling
                            for (int r = 0; r < 8; r++) __builtin_amdgcn_s_sleep(127);
                            if (wall_clock64() - t0 > 500000000ull) { atomicExch(upd.err, 1); break; } // (5 s of the 100 MHz clock: the host died or lost the plot)
                        }
                    }
                    __syncthreads();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); // (system scope: the rows this workgroup reads next were written by the host)
                }
                update_instance<T, false>(ps, (hbm_p)inst + (size_t)id * P.inst_stride, (size_t)id, upd.G, upd.A, upd.c, upd.h, upd.b);
            }
        }
#if EICOS_LDSRES
        gdbl_p I = Il;
        {
            const double *Ig = inst + (size_t)id * P.inst_stride;
            __syncthreads();
            for (int q = threadIdx.x; q < (int)P.inst_stride; q += T) Il[q] = Ig[q];
            __syncthreads();
        }
#else
        gdbl_p I = (gdbl_p)inst + (size_t)id * P.inst_stride;
#endif
        solve_instance<T, NLDS, I16>(ps, I, W, warm);
        __syncthreads();
        if (upd.x) for (int j = threadIdx.x; j < P.n; j += T) upd.x[(size_t)id * P.n + j] = I[P.i_x + j]; // (fused path: the result straight to the caller's array)
#if EICOS_LDSRES
        { // results, persistent per-instance state and (for the debug readbacks) the workspace go back to HBM
            double *Ig = inst + (size_t)id * P.inst_stride;
            for (int q = threadIdx.x; q < (int)P.inst_stride; q += T) Ig[q] = Il[q];
            for (int q = threadIdx.x; q < (int)P.work_stride; q += T) Wglob[q] = W[q];
        }
#endif
        if (threadIdx.x == 0) g_S.next = (int)gridDim.x + atomicAdd(queue, 1);
        __syncthreads();
        g = g_S.next;
    }
}
// Longest-processing-time-first order for k_solve's queue: instances keyed by the number of LDL solves of their
// previous solve (DevInfo::n_ldlsolve; 0 before the first solve), counting sort, descending.  One workgroup.
// snake (one-round launches: B <= resident workgroups, several per CU): workgroup g lands on CU g mod ncu, so the instances are laid out
// boustrophedon over rows of ncu -- row 0 the ncu longest in descending order, row 1 the next ncu ASCENDING, ... -- which pairs the longest
// instance of a CU with the shortest of the next row instead of with the ncu-th longest: the pair sums are balanced, and a launch ends
// when its slowest CU does.
__global__ __launch_bounds__(1024) void k_order(int ps, const double *inst, int B, int *order, int snake, int ncu) {
    const DevPat &P = c_pat[ps];
    constexpr int NB = 1024;
    __shared__ int cnt[NB];
    for (int k = threadIdx.x; k < NB; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    auto key = [&](int i) {
        const DevInfo *di = reinterpret_cast<const DevInfo *>(inst + (size_t)i * P.inst_stride + P.i_info);
        return NB - 1 - min(max(di->n_ldlsolve, 0), NB - 1); // descending
    };
    for (int i = threadIdx.x; i < B; i += blockDim.x) atomicAdd(&cnt[key(i)], 1);
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int k = 0; k < NB; k++) { const int c = cnt[k]; cnt[k] = run; run += c; } }
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        int pos = atomicAdd(&cnt[key(i)], 1); // rank in descending work (ties in arrival order)
        if (snake && pos < snake) { // (snake = number of leading positions laid out boustrophedon: the first round of the launch)
            const int row = pos / ncu, col = pos % ncu, len = min(ncu, snake - row * ncu);
            if (row & 1) pos = row * ncu + (len - 1 - col);
        }
        order[pos] = i;
    }
}

// ============================================================================================
// updateData for a range of instances: reference updateData(double*...) src/eicos.cpp:2053-2082
// = unsetEquilibration (:389-404) -> copy-in -> setEquilibration (:302-374) -> transposes.
// (The KKT "AG" scatter of updateKKTAG :1990-2030 is implicit: the factor kernel reads the
//  equilibrated A/G values in place through DevPat::Lsrc.)
// ============================================================================================
#if EICOS_MAIN_BUILD // (updateData and the debug kernels work on the slabs in HBM: main build only)
template <int T>
__global__ __launch_bounds__(T) void k_update(int ps, double *inst, int first, int count,
                                              const double *Gpr, const double *Apr, const double *cin,
                                              const double *hin, const double *bin, double *scratch) {
    const DevPat &P = c_pat[ps];
    const int n = P.n, p = P.p, m = P.m, l = P.l;
    gdbl_p xt = (gdbl_p)scratch + (size_t)blockIdx.x * (size_t)(n + p + m), at = xt + n, gt = at + p;
    for (int q = blockIdx.x; q < count; q += gridDim.x) {
        gdbl_p I = (gdbl_p)inst + (size_t)(first + q) * P.inst_stride;
        gdbl_p Av = I + P.i_Av, Gv = I + P.i_Gv, cagv = I + P.i_cag, rAv = I + P.i_rA, rGv = I + P.i_rG;
        gdbl_p cv = I + P.i_c, hv = I + P.i_h, bv = I + P.i_b, xe = I + P.i_xe, ae = I + P.i_ae, ge = I + P.i_ge;
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
            for_t_pre<T, 4>(p, [&](int j) { return V2{ae[j], at[j]}; }, [&](int j, const V2 &r) { ae[j] = r.a * r.b; });
            for_t_pre<T, 8>(m, [&](int i) { return V2{ge[i], gt[i]}; }, [&](int i, const V2 &r) { ge[i] = r.a * r.b; });
            __syncthreads();
        }
        for_t_pre<T, 4>(n, [&](int j) { return V2{cv[j], xe[j]}; }, [&](int j, const V2 &r) { cv[j] = r.a / r.b; });
        for_t_pre<T, 4>(p, [&](int j) { return V2{bv[j], ae[j]}; }, [&](int j, const V2 &r) { bv[j] = r.a / r.b; });
        for_t_pre<T, 8>(m, [&](int i) { return V2{hv[i], ge[i]}; }, [&](int i, const V2 &r) { hv[i] = r.a / r.b; });
        // sliced-ELL value copies for the products (stand in for the reference's Gt/At, :2078-2079);
        // *_src is relative to Av (G values follow at i_Gv - i_Av); padding and the dummy slot get 0
        __syncthreads();
        auto ell_copy = [&](gdbl_p dst, gint_p src, int cnt) { // eight gathers per thread in flight (23 k slots per instance)
            for_t_pre<T, 8>(cnt, [&](int k) { const int e = src[k]; return IV1{e, Av[max(e, 0)]}; },
                            [&](int k, const IV1 &r) { dst[k] = r.i < 0 ? 0. : r.a; });
        };
        ell_copy(cagv, P.cag_src, P.cag_slots + 1);
        ell_copy(rAv, P.rA_src, P.rA_slots + 1);
        ell_copy(rGv, P.rG_src, P.rG_slots + 1);
        if (P.gt_on) ell_copy(I + P.i_Gt, P.gt_src, P.gt_nt * 256); // G as dense tiles (tile-internal operand order)
        // static-regularisation constants read by the factor program
        if (threadIdx.x == 0) {
            gdbl_p cst = I + P.i_cst;
            cst[0] = DELTASTAT; cst[1] = -DELTASTAT; cst[2] = 0.; cst[3] = 1.; // [3]: diagonal of the padding nodes (tile mode)
            ginfo->equilibrated = 1;
        }
        __syncthreads();
    }
}

// (the entry-parallel updateData: update_instance, above k_solve, which calls it too)
template <int T, bool LDSV>
__global__ __launch_bounds__(T) void k_update_lds(int ps, double *inst, int first, int count,
                                                  const double *Gpr, const double *Apr, const double *cin,
                                                  const double *hin, const double *bin) {
    const DevPat &P = c_pat[ps];
    for (int q = blockIdx.x; q < count; q += gridDim.x)
        update_instance<T, LDSV>(ps, (hbm_p)inst + (size_t)(first + q) * P.inst_stride, (size_t)q, Gpr, Apr, cin, hin, bin);
}

// Debug: factorise instance `i` with the KKT scaling block as it stands in memory (runs the solver's own stage).
template <int T>
__global__ __launch_bounds__(T, waves_per_eu<T>()) void k_debug_factor(int ps, double *inst, double *work, int i) {
    const DevPat &P = c_pat[ps];
    if (threadIdx.x == 0) { g_S.fl[FL_FATAL] = 0; g_S.wi.n_factor = 0; g_S.dyn_delta = 0.; g_S.dyn_eps = 0.; for (int k = 0; k < 12; k++) g_S.tick[k] = 0; }
    gdbl_p I = (gdbl_p)inst + (size_t)i * P.inst_stride, Kt = (gdbl_p)work + P.w_Kt;
    if (P.tile) { // launched with the solve kernel's dynamic LDS size: the per-wave scratch sits at the same offset
        gdbl_p Kimg = (gdbl_p)work + P.w_Kimg;
        for (int t = threadIdx.x; t < (P.nb + P.nt) * 256; t += T) Kimg[t] = 0.;
        __syncthreads();
        for (int e = threadIdx.x; e < P.tl_nimg; e += T) Kimg[P.tl_img_dst[e]] = I[P.tl_img_src[e]];
        __syncthreads();
    }
    if (P.tile != 1) {
        for (int t = threadIdx.x; t < P.fac_nt; t += T) Kt[t] = I[P.fac_src[t]];
        __syncthreads();
        if (c_pat[ps].fac_defer) stage_factor<T, 0, false, true>(ps, (gdbl_p)work); else stage_factor<T, 0, false, false>(ps, (gdbl_p)work);
    }
    if (P.tile) stage_factor_tiles<T, 0>(ps, I, (gdbl_p)work, -1);
}

// Debug: the solver's own residual/scaling stage (updateScalings + updateKKTScalings, ref :1160-1162) on instance `i`
// with the given (s, z) in its slab, x = y = 0, tau = kap = 1, as pass 0 of the main loop.  ok[0] = 1 if the stage went
// on to the factorisation (i.e. the scalings were executed).  The cone state of workspace 0 persists between calls.
template <int T>
__global__ __launch_bounds__(T, waves_per_eu<T>()) void k_debug_scalings(int ps, double *inst, double *work, int i, int *ok) {
    const DevPat &P = c_pat[ps];
    gdbl_p I = (gdbl_p)inst + (size_t)i * P.inst_stride;
    if (threadIdx.x == 0) {
        g_S.wi = DevInfo{}; g_S.wi.tau = 1.; g_S.wi.kap = 1.; g_S.bi = g_S.wi;
        for (int k = 0; k < SV_COUNT; k++) g_S.sv[k] = 1.;
        g_S.sv[SV_PRESPREV] = DBL_MAX;
        for (int k = 0; k < FL_COUNT; k++) g_S.fl[k] = 0;
        for (int k = 0; k < 12; k++) g_S.tick[k] = 0;
        g_S.dyn_delta = 0.; g_S.dyn_eps = 0.;
        g_S.cur = 0; g_S.best = -1;
    }
    for (int j = threadIdx.x; j < P.n; j += T) I[P.i_x + j] = 0.;
    for (int j = threadIdx.x; j < P.p; j += T) I[P.i_y + j] = 0.;
    __syncthreads();
    const int st = stage_resid<T, 0, false>(ps, I, (gdbl_p)work, 0);
    if (threadIdx.x == 0) ok[0] = (st == ST_FACTOR) ? 1 : 0;
}

#endif // EICOS_MAIN_BUILD

// ---- launchers (called from api.cpp) ----
#ifndef EICOS_ISA_PROBE // (tools/dev/isa_probe.sh compiles single stage functions without the kernel instantiations)
#if EICOS_LDSRES
template <class F> static auto dispatch_solve(int threads, int nlds, int idx16, F &&f) {
    auto byT = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if (idx16) return nlds >= 2 ? f((const void *)k_solve<T, 2, true>) : f((const void *)k_solve<T, 1, true>);
        return nlds >= 2 ? f((const void *)k_solve<T, 2, false>) : f((const void *)k_solve<T, 1, false>);
    };
    return byT(std::integral_constant<int, 128>{}); // (small patterns run 128 threads)
}
#else
template <class F> static auto dispatch_solve(int threads, int nlds, int idx16, F &&f) {
    auto byT = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if (idx16) {
            if (nlds >= 2) return f((const void *)k_solve<T, 2, true>);
            if (nlds == 1 || EICOS_UBL != 0) return f((const void *)k_solve<T, 1, true>); // (U-in-LDS builds: api.cpp never asks for nlds = 0)
#if !EICOS_UBL
            return f((const void *)k_solve<T, 0, true>);
#endif
        }
        if (nlds >= 2) return f((const void *)k_solve<T, 2, false>);
        if (nlds == 1 || EICOS_UBL != 0) return f((const void *)k_solve<T, 1, false>);
#if !EICOS_UBL
        return f((const void *)k_solve<T, 0, false>);
#endif
    };
#if EICOS_TSPLIT
    return byT(std::integral_constant<int, EICOS_TSPLIT>{}); // (this build exists for one workgroup size only)
#elif EICOS_UBL
    return byT(std::integral_constant<int, EICOS_UBL>{});
#else
    return byT(std::integral_constant<int, 256>{}); // (kernels.o / kernels_w2.o: 256 threads; 128 and 512 live in kernels_t128.o / kernels_t512.o)
#endif
}
#endif
hipError_t launch_solve(int ps, double *inst, double *work, int B, int *queue, int *order, int grid, int threads, int nlds,
                        int idx16, int order_min, double warm, double dyn_delta, double dyn_eps, size_t dyn_lds, hipStream_t st, const UpdArgs *upd_in) {
    UpdArgs upd = upd_in ? *upd_in : UpdArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 1, 0u, nullptr};
    if (upd.on && nlds < 1) return hipErrorInvalidValue; // (the fused updateData keeps its maxima in the LDS sweep vector)
    if (B <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(queue, 0, sizeof(int), st); // group queue of this launch
    if (e != hipSuccess) return e;
    // Longest-first order (by the work of the previous solve) whenever a CU gets more than one workgroup: for a batch larger
    // than the grid it is the queue's processing order; for a one-round batch it makes the dispatcher pair a long instance
    // with a short one on each CU, which then finishes the long one alone (+7 % at batch 512 on 256 CUs).
    if (B <= order_min) order = nullptr; // at most one instance per CU: identity
    else {
        // (EICOS_SNAKE=0 under EICOS_EXPERIMENT=1: plain descending order, for A/B runs.  Laying out the first round of a MULTI-round launch the
        // same way measured +-0: profiles/r05_log_snake_order.log)
        static const int snake_on = [] { const char *e = getenv("EICOS_EXPERIMENT"), *k = getenv("EICOS_SNAKE"); return !(e && k && !strcmp(e, "1") && !strcmp(k, "0")); }();
        hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, st, ps, inst, B, order, (snake_on && B <= grid) ? B : 0, order_min);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return dispatch_solve(threads, nlds, idx16, [&](const void *fn) {
        void *args[] = {(void *)&ps, (void *)&inst, (void *)&work, (void *)&B, (void *)&queue, (void *)&order, (void *)&warm, (void *)&dyn_delta,
                        (void *)&dyn_eps, (void *)&upd};
        return hipLaunchKernel(fn, dim3(grid), dim3(threads), args, dyn_lds, st);
    });
}
#if EICOS_MAIN_BUILD
hipError_t launch_update(int ps, double *inst, int first, int count, const double *Gpr, const double *Apr,
                         const double *c, const double *h, const double *b, double *scratch, int grid, size_t lds_bytes, int vals_in_lds, hipStream_t st) {
    if (count <= 0) return hipSuccess;
    if (lds_bytes > 0 && !vals_in_lds) { // entry-parallel, maxima in LDS, values streamed in place in the slab (several workgroups per CU)
        hipLaunchKernelGGL((k_update_lds<512, false>), dim3(grid), dim3(512), lds_bytes, st, ps, inst, first, count, Gpr, Apr, c, h, b);
    } else if (lds_bytes > 0) { // values + maxima fit LDS: the entry-parallel kernel, 512 threads, one workgroup per CU at a time
        hipLaunchKernelGGL((k_update_lds<512, true>), dim3(grid), dim3(512), lds_bytes, st, ps, inst, first, count, Gpr, Apr, c, h, b);
    } else hipLaunchKernelGGL(k_update<256>, dim3(grid), dim3(256), 0, st, ps, inst, first, count, Gpr, Apr, c, h, b, scratch);
    return hipGetLastError();
}
// the dynamic-LDS ceiling of the two entry-parallel updateData kernels, set once per handle on the handle's device
hipError_t update_set_max_lds() {
    hipError_t e = hipFuncSetAttribute((const void *)k_update_lds<512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *)k_update_lds<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
}
hipError_t launch_debug_factor(int ps, double *inst, double *work, int i, int threads, size_t dyn_lds, hipStream_t st) {
    auto big = [&](const void *fn) { if (dyn_lds > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds); };
    switch (threads) { // the factor program is built for the handle's workgroup size
    case 128: big((const void *)k_debug_factor<128>); hipLaunchKernelGGL(k_debug_factor<128>, dim3(1), dim3(128), dyn_lds, st, ps, inst, work, i); break;
    case 256: big((const void *)k_debug_factor<256>); hipLaunchKernelGGL(k_debug_factor<256>, dim3(1), dim3(256), dyn_lds, st, ps, inst, work, i); break;
    case 512: big((const void *)k_debug_factor<512>); hipLaunchKernelGGL(k_debug_factor<512>, dim3(1), dim3(512), dyn_lds, st, ps, inst, work, i); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_debug_scalings(int ps, double *inst, double *work, int i, int *ok, int threads, hipStream_t st) {
    switch (threads) {
    case 128: hipLaunchKernelGGL(k_debug_scalings<128>, dim3(1), dim3(128), 0, st, ps, inst, work, i, ok); break;
    case 256: hipLaunchKernelGGL(k_debug_scalings<256>, dim3(1), dim3(256), 0, st, ps, inst, work, i, ok); break;
    case 512: hipLaunchKernelGGL(k_debug_scalings<512>, dim3(1), dim3(512), 0, st, ps, inst, work, i, ok); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#endif // EICOS_MAIN_BUILD
hipError_t solve_occupancy(int threads, int nlds, int idx16, size_t dyn_lds, int *blocks_per_cu) {
    return dispatch_solve(threads, nlds, idx16, [&](const void *fn) {
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, threads, dyn_lds);
    });
}
hipError_t solve_set_max_lds(int threads, int nlds, int idx16, size_t dyn_lds) {
    if (dyn_lds == 0) return hipSuccess;
    // The attribute belongs to the kernel, not to a handle: several live handles (other patterns, the shards of an eicos_multi on one
    // device) launch the same instantiation with different dynamic LDS sizes, so it is always raised to the device's ceiling (160 KB
    // minus the 4 KB budgeted for the static block) and never lowered; the size a launch really uses is its own dyn_lds.
    return dispatch_solve(threads, nlds, idx16, [&](const void *fn) {
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    });
}

hipError_t upload_pattern(int ps, const DevPat &P) {
    if (ps < 0 || ps >= MAX_PATTERNS) return hipErrorInvalidValue;
    return hipMemcpyToSymbol(HIP_SYMBOL(c_pat), &P, sizeof(DevPat), (size_t)ps * sizeof(DevPat), hipMemcpyHostToDevice);
}
#endif // EICOS_ISA_PROBE
#if EICOS_LDSRES
} // namespace ldsres
#elif EICOS_W2
} // namespace w2
#elif EICOS_UBL == 256
} // namespace ubl256
#elif EICOS_UBL == 512
} // namespace ubl512
#elif EICOS_TSPLIT == 128
} // namespace t128
#elif EICOS_TSPLIT == 512
} // namespace t512
#else
int max_patterns() { return MAX_PATTERNS; }
#endif

} // namespace eicos
