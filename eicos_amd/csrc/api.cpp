// C ABI of the batched solver (include/eicos_amd.h): host-side setup, memory, launches.
// There is deliberately NO CPU fallback: without a HIP device every compute entry point fails
// with EICOS_E_NOGPU.
#include "../../include/eicos_amd.h"

#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <emmintrin.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "device_types.hpp"
#include "launch.hpp"
#include "symbolic.hpp"
#include "plans.hpp"
#include "tiles.hpp"
#include "envknob.hpp"

using namespace eicos;
// Workgroups per CU for a batch, at most `max_r`: the cheapest estimate of the launch's duration wins.
//   time of one "round" (every resident workgroup solves one instance) at r per CU, relative to r = 1:  1 + 0.5 (r - 1) up to r = 2; the
//   256-thread kernel beyond two per CU grows in proportion to r (+ 1.5 %): re-measured with the final round-5 library, same box, interleaved
//   (profiles/r06_log_ab_r04_r05_libs.log, r06_log_launch_shapes.log; MPC02): 11.15 ms per round of 512 at two per CU (the 256-VGPR build, dense
//   apex, residual head in LDS) against 17.0 ms per round of 768 at three (168 VGPRs, neither) -- 45.9 against 45.2 instances per ms: the
//   third workgroup buys nothing in the steady state any more, it only rounds a batch differently;
//   a partly filled LAST round costs more than its share (0.55 + 0.45 f) only when it follows a single full round (batch 768: 20.6 ms as
//   1.5 rounds at two per CU, 18.7 ms as one round at three); behind two or more full rounds the longest-first queue evens the tail out
//   and the share is what it costs (measured f = 2/3 behind two rounds: 0.63; f = 1/3 behind five: 0.22).
// MPC02 on 256 CUs: 3 per CU for 513 ... 768 instances only; 1024, 1536, 2048, 3072 and 4096 run at two (measured: +4 % at 1536, +2.5 % at
// 3072, +-1 % at 2048, -2 % at 4096 against three per CU -- inside the +-3 % spread between two processes on one box -- on 1.18 x instead of
// 1.33 x the algorithmic HBM traffic), so the set-up no longer runs twice for the large batches (eicos_batch_create).
static int launch_blocks_per_cu(int batch, int n_cu, int max_r, int threads) {
    double best = 1e300; int best_r = 1;
    for (int r = 1; r <= max_r; r++) {
        const double rounds = (double)batch / ((double)n_cu * r);
        const double full = std::floor(rounds), f = rounds - full;
        const double round_time = (threads == 256 && r > 2) ? 1.5 * 1.015 * (r / 2.0) : 1.0 + 0.5 * (r - 1);
        const double cost = round_time * (full + (f > 0 ? (full >= 2 ? f : 0.55 + 0.45 * f) : 0.0));
        if (cost < best - 1e-12) { best = cost; best_r = r; }
    }
    return best_r;
}
static constexpr int KI_MAX_HOST = 2; // right-hand sides of a dual solve (kernels.hip: KI_MAX)

static thread_local std::string g_err;
// eicos_set_arithmetic_profile: 0 = plans shaped by the launch (default), 1 = plans shaped by the pattern alone (batch-independent bits)
static std::atomic<int> g_arith_profile{0};
static std::mutex g_slot_mu;
static std::map<int, std::vector<char>> g_slot_used; // per device: which constant-memory descriptor slots are taken (under g_slot_mu)
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(EICOS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct eicos_batch {
    ProblemPattern pat;
    Symbolic sym;
    DevPat dp{};
    int batch = 0, device = 0, threads = 256, grid = 0, upd_grid = 0;
    int order_min = 0;        // batches up to this size (one instance per CU) are solved in identity order, larger ones longest-first
    bool last_ordered = false; // how the most recent solve was launched (eicos_debug_trace)
    size_t upd_lds = 0;       // > 0: updateData runs the entry-parallel kernel with this much dynamic LDS (values + maxima)
    int upd_vals_lds = 1;     // 1: its working copy of the values is in LDS too; 0: streamed in place in the instance slab
    int *d_pattern = nullptr;
    int pslot = -1; // slot of this handle's DevPat in the kernels' constant-memory table
    size_t dyn_lds = 0;
    int nlds = 0;
    int w2 = 0;               // 1: solves run the two-waves-per-SIMD build of the 256-thread kernel (w2::launch_solve)
    int ubl = 0;              // 1: solves run the build with the factor operand array U in LDS (ubl256:: / ubl512::launch_solve; one workgroup per CU)
    int ldsres = 0;           // 1: solves run the LDS-resident kernel (ldsres::launch_solve), slabs copied in and out per instance
    size_t pattern_ints = 0;
    double *d_inst = nullptr, *d_work = nullptr, *d_scratch = nullptr;
    int *d_queue = nullptr; // instance queue of the solve kernel (reset per launch)
    double *d_stage = nullptr; size_t stage_doubles = 0; // peer-copy updateData without peer access: persistent staging buffer (one chunk)
    // host-pointer updateData / results (the reference's real signature: updateData(double *...), solution() on the host): two PINNED
    // bounce buffers (hipHostMalloc).  Input chunk k is copied into pin[k & 1] by the host while the GPU's updateData kernel reads chunk
    // k - 1 straight out of the other one over PCIe (the kernel's loads are the transfer: no device-side staging copy, no per-chunk
    // stream synchronisation); results come back through the same buffers, the strided device-to-host copy of chunk k + 1 in flight
    // while the host copies chunk k out.  pin_ev[i]: the last GPU work that touches pin[i].
    double *pin[2] = {nullptr, nullptr}; size_t pin_doubles = 0;
    hipEvent_t pin_ev[2] = {nullptr, nullptr}; bool pin_busy[2] = {false, false};
    int last_update_path = 0; // how the most recent host/peer updateData moved its inputs: 1 pinned bounce, 2 zero-copy (pinned source), 3 peer zero-copy, 4 peer staged copies, 5 fused into the solve launch
    int *d_flag = nullptr;   // debug hooks
    double warm_shift = 0.; // > 0: warm start (eicos_batch_set_warm_start)
    double dyn_delta = 0., dyn_eps = 0.; // > 0: dynamic regularisation (eicos_batch_set_dynamic_regularization)
    hipStream_t own_stream = nullptr, stream = nullptr;
    // HIP events around every solve launch / every updateData call, on the handle's stream.  A RING of pairs: the durations of the last
    // EV_RING launches can be read after the fact (eicos_batch_ms_history), so that a caller timing K back-to-back steps need not
    // synchronise with the GPU inside its loop to learn each launch's duration.  ev_* = the most recent pair of each ring.
    static constexpr int EV_RING = 64;
    hipEvent_t ring_s[EV_RING][2] = {}, ring_u[EV_RING][2] = {};
    hipEvent_t ring_step0[EV_RING] = {}; // per solve slot: start event of the updateData call that preceded it (the start of the caller's "step")
    long n_solve_rec = 0, n_update_rec = 0;
    hipEvent_t ev_s0 = nullptr, ev_s1 = nullptr, ev_u0 = nullptr, ev_u1 = nullptr;
    bool solve_timed = false, update_timed = false;
    bool in_chunked_update = false; // eicos_batch_update records ev_u0/ev_u1 around ALL of its chunks
    int64_t npairs = 0;
    std::vector<int> posB; // CSC entry of L -> slot in the backward value array
    int ub_len = 1;        // length of that array (plan slots + dummy, + the dense apex image)
    int bpc = 1, n_cu = 256; // workgroups per CU of the solve launch; CUs of the device
    int arith_profile = 0;   // eicos_set_arithmetic_profile at creation
    UpdArgs fused{}; bool fused_pending = false; // eicos_batch_update_solve: the arrays the next solve launch pulls in itself
    // ... from pageable host memory: one pinned staging buffer for the whole batch (+ one ready flag per chunk), filled while the kernel runs
    double *stage_pin = nullptr; size_t stage_pin_doubles = 0; unsigned *stage_flags = nullptr; int stage_nflags = 0; unsigned stage_seq = 0;
    int *d_err = nullptr;
    TilePlan tiles;        // tile mode (Symbolic::tile): the dense-front plan
};

namespace {

struct IntPool { // one int32 buffer for every pattern array
    std::vector<int> data;
    size_t add(const std::vector<int> &v) {
        size_t off = data.size();
        data.insert(data.end(), v.begin(), v.end());
        while (data.size() % 4) data.push_back(0); // keep 16-byte alignment of every array
        if (v.empty()) { data.insert(data.end(), 4, 0); }
        return off;
    }
};

struct SlabLayout {
    size_t size = 0;
    int add(size_t count) {
        int off = (int)size;
        size += (count + 7) & ~(size_t)7; // 64-byte granules
        return off;
    }
};

} // namespace

extern "C" {

const char *eicos_last_error(void) { return g_err.c_str(); }

int eicos_set_arithmetic_profile(int profile) {
    if (profile != 0 && profile != 1) return fail(EICOS_E_INVALID, "arithmetic profile must be 0 or 1");
    g_arith_profile.store(profile);
    return EICOS_OK;
}
int eicos_get_arithmetic_profile(void) { return g_arith_profile.load(); }

int eicos_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int batch_create_impl(int n, int m, int p, int l, int ncones, const int *q, const int *Gjc, const int *Gir, const int *Ajc, const int *Air,
                             int batch, int device, bool allow_apex, eicos_batch **out);
constexpr int EICOS_RETRY_NO_APEX = -99; // (internal: never leaves eicos_batch_create)
// The dense apex (symbolic.hpp) keeps an image of its block in LDS.  Whether that costs the launch a resident workgroup per CU is only
// known once the real LDS layout and the runtime's occupancy answer exist -- at the end of the set-up.  So: set up with the apex; if the
// batch is one that would run MORE workgroups per CU than came out, set up once more without it and keep the better launch shape
// (MPC02: batches >= 1536 run three per CU without the apex, two with it: the third workgroup is worth more).
int eicos_batch_create(int n, int m, int p, int l, int ncones, const int *q,
                       const int *Gjc, const int *Gir, const int *Ajc, const int *Air,
                       int batch, int device, eicos_batch **out) {
    int rc = batch_create_impl(n, m, p, l, ncones, q, Gjc, Gir, Ajc, Air, batch, device, g_arith_profile.load() == 0, out);
    if (rc == EICOS_RETRY_NO_APEX) return batch_create_impl(n, m, p, l, ncones, q, Gjc, Gir, Ajc, Air, batch, device, false, out);
    if (rc != EICOS_OK || (*out)->sym.apex0 < 0) return rc;
    eicos_batch *h = *out;
    if (launch_blocks_per_cu(batch, h->n_cu, h->bpc + 1, h->threads) <= h->bpc) return rc; // one more per CU would not be taken anyway
    eicos_batch *h0 = nullptr;
    if (batch_create_impl(n, m, p, l, ncones, q, Gjc, Gir, Ajc, Air, batch, h->device, false, &h0) != EICOS_OK) {
        static std::atomic<bool> told{false}; // (e.g. out of memory with both handles alive: the launch shape WITH the apex is kept -- say so once)
        if (!told.exchange(true)) std::fprintf(stderr, "eicos_amd: the set-up without the dense apex failed (%s); keeping %d workgroup(s) per CU\n", g_err.c_str(), h->bpc);
        return rc;
    }
    if (h0->bpc > h->bpc) { eicos_batch_destroy(h); *out = h0; } else eicos_batch_destroy(h0);
    return EICOS_OK;
}

static int batch_create_impl(int n, int m, int p, int l, int ncones, const int *q,
                       const int *Gjc, const int *Gir, const int *Ajc, const int *Air,
                       int batch, int device, bool allow_apex, eicos_batch **out) {
    if (!out) return fail(EICOS_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n < 0 || m < 0 || p < 0 || ncones < 0 || batch < 1) return fail(EICOS_E_INVALID, "negative dimension or batch < 1");
    if (ncones > 0 && !q) return fail(EICOS_E_INVALID, "ncones > 0 but q is NULL");
    const bool haveG = Gjc && Gir, haveA = Ajc && Air;
    if (!haveG) { m = 0; ncones = 0; } // reference: groups given as NULL are empty (src/eicos.cpp:103-117)
    if (!haveA) p = 0;
    // The reference ignores `l` and derives it as m - sum(q) (src/eicos.cpp:91,155).  l < 0 means "derive"; a caller that
    // does pass l (ECOS convention: l + sum(q) = m) and gets it wrong would silently solve a different cone split.
    if (haveG && l >= 0) {
        long long qs = 0;
        for (int c = 0; c < ncones; c++) qs += q[c];
        if ((long long)l + qs != (long long)m) return fail(EICOS_E_INVALID, "l + sum(q) != m (pass l < 0 to derive l as the reference does)");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(EICOS_E_NOGPU, "no HIP device visible: the solver has no CPU fallback");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev) return fail(EICOS_E_INVALID, "device index out of range");

    eicos_batch *h = new eicos_batch();
    try {
        ProblemPattern &P = h->pat;
        P.n = n; P.m = m; P.p = p; P.nc = ncones;
        P.q.assign(q, q + ncones);
        // compressed CSC as the reference assumes (src/eicos.cpp:2038-2039): pointers start at 0 and do not decrease,
        // row indices in range and strictly increasing inside a column
        auto take = [&](const int *jc, const int *ir, int rows, std::vector<int> &ojc, std::vector<int> &oir, const char *nm) {
            if (jc[0] != 0) throw std::invalid_argument(std::string(nm) + ": column pointers must start at 0");
            for (int j = 0; j < n; j++) if (jc[j + 1] < jc[j]) throw std::invalid_argument(std::string(nm) + ": column pointers decrease");
            ojc.assign(jc, jc + n + 1); oir.assign(ir, ir + jc[n]);
            for (int j = 0; j < n; j++)
                for (int k = jc[j]; k < jc[j + 1]; k++) {
                    if (ir[k] < 0 || ir[k] >= rows) throw std::invalid_argument(std::string(nm) + " row index out of range");
                    if (k > jc[j] && ir[k] <= ir[k - 1]) throw std::invalid_argument(std::string(nm) + ": row indices of a column must be strictly increasing");
                }
        };
        if (haveG) take(Gjc, Gir, m, P.Gjc, P.Gir, "G"); else P.Gjc.assign(n + 1, 0);
        if (haveA) take(Ajc, Air, p, P.Ajc, P.Air, "A"); else P.Ajc.assign(n + 1, 0);
        // (experiment knobs, envknob.hpp: honoured only under EICOS_EXPERIMENT=1, range-checked)
        h->sym = analyze(P, env_knob("EICOS_ORDER", -1, 0, 16), env_knob("EICOS_TILES", -1, 0, 2));
        if (h->sym.tile) h->tiles = build_tile_plan(h->sym);
    } catch (const std::invalid_argument &e) { delete h; return fail(EICOS_E_INVALID, e.what()); }
    catch (const std::runtime_error &e) { delete h; return fail(EICOS_E_UNSUPPORTED, e.what()); }
    catch (const std::exception &e) { delete h; return fail(EICOS_E_INVALID, e.what()); }

    const Symbolic &S = h->sym;
    const ProblemPattern &P = h->pat;
    h->batch = batch; h->device = device;
    DevPat &D = h->dp;
    auto env_int = [](const char *k, int dflt, int lo, int hi) { return env_knob(k, dflt, lo, hi); };
    {
        // workgroup size by problem size (measured, batch 256: dim_K 129 -> 128, 1249 -> 256, >= 3815 -> 512 threads);
        // batches beyond one workgroup per CU are throughput-bound: 256 threads issue a third fewer wavefront-slices
        // per instance than 512 and fit three workgroups per CU (MPC02 pattern: 500 k vs 414 k iterations/s)
        const int dimK = n + p + m + 2 * ncones;
        int n_cu = 256;
        { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device < 0 ? 0 : device) == hipSuccess) n_cu = pr.multiProcessorCount; }
        // (sparse factors only: with ~50 entries per row of L -- the dense-front config -- 512 threads stay ahead)
        // (arithmetic profile 1: every choice that shapes a PLAN -- and with it the order of the floating-point operations -- is made as for a
        // batch beyond one workgroup per CU, whatever the batch really is: workgroup size by pattern size alone, no dense apex, the
        // single-wavefront tree top; the launch shape itself -- grid, LDS residency, dual solves, which are bit-neutral -- follows the real batch)
        h->arith_profile = g_arith_profile.load();
        const bool as_large = h->arith_profile == 1;
        const bool throughput_bound = (batch > n_cu || as_large) && (long long)S.nnzL < 16LL * S.N;
        // one workgroup per CU (batch <= CUs): latency-bound, more wavefronts per instance pay earlier (measured at batch 256 with
        // the 256-VGPR build of the 512-thread kernels: lp_blend / lp_adlittle, dim_K ~ 300: 256 threads +5..8 % over 128;
        // lp_beaconfd / lp_bandm / lp_agg, dim_K 763..1718: 512 threads +7..12 % over 256)
        const int dflt = throughput_bound ? (dimK < 400 ? 128 : 256) : (dimK < 250 ? 128 : (dimK < 700 ? 256 : 512));
        const int t = env_int("EICOS_THREADS", dflt, 128, 512);
        if (t != 128 && t != 256 && t != 512) { delete h; return fail(EICOS_E_INVALID, "EICOS_THREADS must be 128, 256 or 512"); }
        h->threads = t;
        // ---- dense apex: not with 128-thread workgroups (small patterns; kernels.hip: apex_on), not when the caller found that it costs a
        // resident workgroup (eicos_batch_create above) ----
        // (128 threads: only the LDS-resident build carries an apex -- its images then live in the LDS copy of the workspace slab; whether that
        // build is taken is known after the slab layout: if not, the set-up is repeated without the apex, EICOS_RETRY_NO_APEX below)
        if (h->sym.apex0 >= 0 && ((t < 256 && !(t == 128 && env_knob("EICOS_LDSRES", 1, 0, 1) && batch <= n_cu)) || !allow_apex)) h->sym.apex0 = -1;
        // small patterns whose narrow tree top would go to the tile path (hybrid): the level schedule + dense apex does better there -- a
        // handful of 16 x 16 blocks costs two workgroup-wide block levels each, the apex swallows the whole tail in 2 x 64 register steps
        // (lp_adlittle 1.13 -> 1.33 M, lp_blend 0.92 -> 1.08 M iter/s at batch 256; larger tops -- lp_bandm, lp_agg, lp_25fv47 -- stay hybrid:
        // their top blocks are dense and the MFMA factorisation of the block is what pays there)
        if (h->sym.tile == 2 && S.N < APEX_OVER_HYBRID_BELOW && t >= 256 && allow_apex && env_knob("EICOS_TILES", -1, 0, 2) < 0) {
            try {
                Symbolic alt = analyze(P, env_knob("EICOS_ORDER", -1, 0, 16), 0);
                if (alt.apex0 >= 0) { h->sym = std::move(alt); h->tiles = TilePlan(); }
            } catch (const std::exception &) { /* keep the hybrid analysis */ }
        }
        h->n_cu = n_cu;
    }
    h->npairs = S.npairs;
    if (S.npairs >= (int64_t)1 << 31) { delete h; return fail(EICOS_E_UNSUPPORTED, "factor program exceeds 2^31 pairs"); }
    if (S.tile) { // the tile image, the tile arrays of L and their workspace offsets are indexed with 32-bit ints
        const long long img = ((long long)h->tiles.nb + h->tiles.nt) * 256;
        if (img >= IMG_BASE || (long long)S.N + img >= DIAG_POS / 2 || 3 * img * (long long)sizeof(double) > (8LL << 30)) {
            delete h; return fail(EICOS_E_UNSUPPORTED, "dense-front pattern too large: the tile image of L exceeds the per-workgroup workspace budget");
        }
    }
    const bool tile = S.tile != 0;  // some part of L lives in 16 x 16 tiles: all of it (S.tile == 1) or the top block (hybrid, == 2)
    const bool tile1 = S.tile == 1; // pure tile mode: no scalar programs at all
    const TilePlan &TP = h->tiles;
    // NV = length of the KKT-space vectors on the device: dim_K in elimination order, or (tile mode) the blocks padded to 16
    const int NV = tile ? TP.N16 : S.N;
    auto posK = [&](int old) { return tile ? TP.slot[S.iperm[old]] : S.iperm[old]; }; // KKT index -> device slot
    D.n = S.n; D.p = S.p; D.m = S.m; D.l = S.l; D.nc = S.nc; D.N = NV; D.mt = S.mt; D.nV = S.nV;
    D.nnzA = S.nnzA; D.nnzG = S.nnzG; D.nnzL = S.nnzL; D.nlev = S.nlev;

    // ---- slab layouts ----
    SlabLayout L;
    D.i_Av = L.add(S.nnzA); D.i_Gv = L.add(S.nnzG);
    // sliced-ELL plans of the matrix-vector products
    std::vector<int> cag_ptr(S.n + 1, 0), cag_val, cag_k, cag_yz; // stacked columns of [A; G]
    std::vector<int> zexp0(S.m);
    { for (int i = 0; i < S.l; i++) zexp0[i] = i; for (int c = 0; c < S.nc; c++) for (int k = 0; k < S.q[c]; k++) zexp0[S.cone_off[c] + k] = S.cone_off[c] + k + 2 * c; }
    const int gv_rel = D.i_Gv - D.i_Av;
    // ---- G in dense 16 x 16 tiles (dense-front patterns on the tile path: the products are bandwidth-bound) ----
    // Row block = 16 consecutive rows of G; its tiles = the sorted union of the columns of those rows, cut into groups of 16
    // (so a tile's columns need not be consecutive).  One pass over the tiles yields G x (per row block, in registers) and the
    // partial column sums of G' z (per tile, reduced per column in a second, fixed-order pass): G is streamed ONCE per
    // evaluation instead of once in column form and once in row form, with no index bytes.  Taken when the tiles are at
    // least half full; the sliced-ELL plans of the products then hold A only.
    struct GTiles { int on = 0, nrb = 0, nt = 0, W = 0; std::vector<int> rbptr, col, src, cidx; } GT;
    if (S.tile == 1 && S.nnzG > 0 && env_int("EICOS_GTILES", 1, 0, 2)) {
        const int nrb = (S.m + 15) / 16;
        GT.nrb = nrb; GT.rbptr.assign(nrb + 1, 0);
        std::vector<std::vector<int>> rbcols(nrb);
        for (int rb = 0; rb < nrb; rb++) {
            std::vector<int> &cs = rbcols[rb];
            for (int i = rb * 16; i < std::min(S.m, rb * 16 + 16); i++) for (int e = S.Gt_ptr[i]; e < S.Gt_ptr[i + 1]; e++) cs.push_back(S.Gt_col[e]);
            std::sort(cs.begin(), cs.end()); cs.erase(std::unique(cs.begin(), cs.end()), cs.end());
            GT.rbptr[rb + 1] = GT.rbptr[rb] + ((int)cs.size() + 15) / 16;
        }
        GT.nt = GT.rbptr[nrb];
        if (GT.nt > 0 && ((double)S.nnzG >= 0.5 * 256.0 * GT.nt || env_int("EICOS_GTILES", 1, 0, 2) == 2)) { // (2: tests force it on sparse G)
            GT.on = 1;
            GT.col.assign((size_t)GT.nt * 16, -1); GT.src.assign((size_t)GT.nt * 256 + 1, -1);
            std::vector<int> ccount(S.n, 0);
            for (int rb = 0; rb < nrb; rb++) {
                const std::vector<int> &cs = rbcols[rb];
                for (size_t q = 0; q < cs.size(); q++) { GT.col[(size_t)GT.rbptr[rb] * 16 + q] = cs[q]; ccount[cs[q]]++; }
                for (int i = rb * 16; i < std::min(S.m, rb * 16 + 16); i++)
                    for (int e = S.Gt_ptr[i]; e < S.Gt_ptr[i + 1]; e++) {
                        const int q = (int)(std::lower_bound(cs.begin(), cs.end(), S.Gt_col[e]) - cs.begin());
                        GT.src[(size_t)(GT.rbptr[rb] + q / 16) * 256 + tile_op(i - rb * 16, q % 16)] = gv_rel + S.Gt_pos[e];
                    }
            }
            GT.W = 8; // contributions per column the kernel adds (fixed width, padded)
            if (*std::max_element(ccount.begin(), ccount.end()) > GT.W) GT.on = 0; // (a column met by more than 8 tiles: keep the ELL products)
            GT.cidx.assign((size_t)S.n * GT.W, -1); // padding
            std::fill(ccount.begin(), ccount.end(), 0);
            if (GT.on) for (int t = 0; t < GT.nt; t++) for (int k = 0; k < 16; k++) { const int j = GT.col[(size_t)t * 16 + k]; if (j >= 0) GT.cidx[(size_t)j * GT.W + ccount[j]++] = t * 16 + k; }
        }
    }
    const std::vector<int> Gt_ptr_used = GT.on ? std::vector<int>(S.m + 1, 0) : S.Gt_ptr; // (rows without entries: the epilogue still runs)
    for (int j = 0; j < S.n; j++) {
        for (int k = P.Ajc[j]; k < P.Ajc[j + 1]; k++) { cag_val.push_back(k); cag_k.push_back(S.n + P.Air[k]); cag_yz.push_back(P.Air[k]); }
        if (!GT.on) for (int k = P.Gjc[j]; k < P.Gjc[j + 1]; k++) { cag_val.push_back(gv_rel + k); cag_k.push_back(S.n + S.p + zexp0[P.Gir[k]]); cag_yz.push_back(-1 - P.Gir[k]); }
        cag_ptr[j + 1] = (int)cag_val.size();
    }
    EllPlan pcag = build_ell_plan(cag_ptr, S.n, h->threads), prA = build_ell_plan(S.At_ptr, S.p, h->threads),
            prG = build_ell_plan(Gt_ptr_used, S.m, h->threads);
    D.cag_ns = (int)pcag.sl.size(); D.rA_ns = (int)prA.sl.size(); D.rG_ns = (int)prG.sl.size();
    {   auto real = [](const std::vector<SliceMeta> &sl) { int c = (int)sl.size(); while (c > 0 && sl[(size_t)c - 1].cnt == 0) c--; return c; };
        D.cag_ns_r = real(pcag.sl); D.rA_ns_r = real(prA.sl); D.rG_ns_r = real(prG.sl); }
    D.cag_slots = pcag.slots; D.rA_slots = prA.slots; D.rG_slots = prG.slots;
    D.i_cag = L.add((size_t)pcag.slots + 8); D.i_rA = L.add((size_t)prA.slots + 8); D.i_rG = L.add((size_t)prG.slots + 8);
    D.gt_on = GT.on; D.gt_nrb = GT.nrb; D.gt_nt = GT.nt; D.gt_W = GT.W;
    D.i_Gt = GT.on ? (int)L.add((size_t)GT.nt * 256 + 8) : 0;
    D.i_c = L.add((size_t)S.n + S.p + S.m); D.i_b = D.i_c + S.n; D.i_h = D.i_b + S.p; // [c | b | h]: one array in [x | y | z] order (kkt_solve's epilogue)
    D.i_xe = L.add(S.n); D.i_ae = L.add(S.p); D.i_ge = L.add(S.m);
    D.i_Vv = L.add(S.nV); D.i_cst = L.add(4);
    D.i_x = L.add(S.n); D.i_y = L.add(S.p); D.i_z = L.add(S.m); D.i_s = L.add(S.m);
    D.i_info = L.add(DEVINFO_DOUBLES);
    std::vector<int> cag_idx_k(pcag.src.size()), cag_idx_yz(pcag.src.size()), cag_src(pcag.src.size());
    for (size_t sl = 0; sl < pcag.src.size(); sl++) {
        const int e = pcag.src[sl];
        cag_src[sl] = e < 0 ? -1 : cag_val[e];
        cag_idx_k[sl] = e < 0 ? NV : posK(cag_k[e]);                         // elimination order; padding -> zero slot N
        cag_idx_yz[sl] = e < 0 ? 0 : (cag_yz[e] >= 0 ? cag_yz[e] : (D.i_z - D.i_y) + (-1 - cag_yz[e])); // offset from y
    }
    std::vector<int> rA_idx(prA.src.size()), rA_src(prA.src.size()), rG_idx(prG.src.size()), rG_src(prG.src.size());
    std::vector<int> rA_idx_k(prA.src.size()), rG_idx_k(prG.src.size());
    for (size_t sl = 0; sl < prA.src.size(); sl++) { const int e = prA.src[sl]; rA_src[sl] = e < 0 ? -1 : S.At_pos[e]; rA_idx[sl] = e < 0 ? 0 : S.At_col[e]; rA_idx_k[sl] = e < 0 ? NV : posK(S.At_col[e]); }
    for (size_t sl = 0; sl < prG.src.size(); sl++) { const int e = prG.src[sl]; rG_src[sl] = e < 0 ? -1 : gv_rel + S.Gt_pos[e]; rG_idx[sl] = e < 0 ? 0 : S.Gt_col[e]; rG_idx_k[sl] = e < 0 ? NV : posK(S.Gt_col[e]); }
    std::vector<int> ipx(S.n), ipy(S.p), ipz(S.m), ipv(S.nc), ipu(S.nc);
    for (int j = 0; j < S.n; j++) ipx[j] = posK(j);
    for (int r = 0; r < S.p; r++) ipy[r] = posK(S.n + r);
    for (int i = 0; i < S.m; i++) ipz[i] = posK(S.n + S.p + zexp0[i]);
    for (int c = 0; c < S.nc; c++) { const int e0 = S.n + S.p + S.cone_off[c] + 2 * c + S.q[c]; ipv[c] = posK(e0); ipu[c] = posK(e0 + 1); }
    D.inst_stride = L.size;
    SlabLayout Wl;
    // second buffer set of the iterate (ShI::cur / best in kernels.hip): same spacing of y and z as in the instance slab -- the stacked
    // product [A' G'] gathers (y, z) through ONE index array relative to y
    D.w_lam = Wl.add(S.m); D.w_bx = Wl.add(S.n); D.w_by = Wl.add(S.p); D.w_bz = Wl.add(S.m); D.w_bs = Wl.add(S.m);
    if (D.w_bz - D.w_by != D.i_z - D.i_y) { delete h; return fail(EICOS_E_INVALID, "internal: the two buffer sets of the iterate are laid out differently"); }
    D.w_rz = Wl.add(S.m);
    D.w_rhs1k = Wl.add((size_t)S.n + S.p + S.m); D.w_rhs2k = Wl.add((size_t)S.n + S.p + S.m);       // [x | y | z] order
    D.w_dx1 = Wl.add((size_t)S.n + S.p + S.m); D.w_dy1 = D.w_dx1 + S.n; D.w_dz1 = D.w_dy1 + S.p; // [dx | dy | dz]: one array each
    D.w_dx2 = Wl.add((size_t)S.n + S.p + S.m); D.w_dy2 = D.w_dx2 + S.n; D.w_dz2 = D.w_dy2 + S.p;
    D.w_dsw = Wl.add(S.m); D.w_wdz = Wl.add(S.m); D.w_dsa = Wl.add(S.m); D.w_t1 = Wl.add(S.m); D.w_t2 = Wl.add(S.m);
    D.w_lpw = Wl.add(S.l); D.w_lpv = Wl.add(S.l); D.w_csc = Wl.add((size_t)S.nc * CSC_STRIDE); D.w_qv = Wl.add(S.m);
    D.w_trace = Wl.add((size_t)TRACE_ROWS * TRACE_COLS);
    if (GT.on) { // G tile products: partial column sums per tile, G'z per column, G x per row; two right-hand sides
        D.w_gpart = Wl.add((size_t)GT.nt * 16 * 2); D.w_gx = Wl.add((size_t)S.n * 2); D.w_gz = Wl.add((size_t)S.m * 2);
    } else D.w_gpart = D.w_gx = D.w_gz = 0;
    // ---- the arrays of the factorisation / KKT solve ----
    D.w_xk = Wl.add((size_t)NV + 16); D.w_ek = Wl.add((size_t)NV + 16); D.w_dxr = Wl.add(NV);
    D.w_D = Wl.add(NV); D.w_invD = Wl.add((size_t)NV + 8); // (+ the always-zero slot fac_kpad of the deferred-L factorisation) // w_UF / w_UB are added once the slice plans are known

    // ---- pattern arrays ----
    IntPool pool;
    std::vector<int> zexp(S.m), zdsign(S.m, 1), cone_vbase(S.nc), cone_small, cone_big;
    for (int i = 0; i < S.l; i++) zexp[i] = i;
    {
        int vb = S.l;
        for (int c = 0; c < S.nc; c++) {
            const int o = S.cone_off[c], d = S.q[c];
            for (int k = 0; k < d; k++) zexp[o + k] = o + k + 2 * c;
            zdsign[o + d - 1] = -1; // last cone row: -delta in the refinement operator (ref src/eicos.cpp:1552)
            cone_vbase[c] = vb; vb += 3 * d + 1;
            (d >= CONE_BIG ? cone_big : cone_small).push_back(c);
        }
    }
    auto srcoff = [&](int kind, int src) {
        switch (kind) {
        case SRC_A: return D.i_Av + src;
        case SRC_G: return D.i_Gv + src;
        case SRC_V: return D.i_Vv + src;
        case SRC_POSDELTA: return D.i_cst + 0;
        case SRC_NEGDELTA: return D.i_cst + 1;
        default: return D.i_cst + 2;
        }
    };


    // ---- sliced-ELL plans of the two triangular sweeps (device_types.hpp: SliceMeta) ----
    // (tile mode: the sweeps and the factorisation run over the tile plan instead; the scalar plans stay empty)
    TriPlan planF, planB;
    // A handle of at most one workgroup per CU gives NO level to a single wavefront: the idle wavefronts of such a part are issue slots for a
    // neighbour on the CU -- without one, every extra sweep call only adds a cold start (lp_bandm +2.2 %, lp_beaconfd +2.6 %, lp_blend +4 %,
    // lp_adlittle +2 %, lp_agg +0.6 %, lp_25fv47 +-0; MPC02 at three per CU -3 ... -7 %, which keeps its single-wavefront tree top)
    const bool solo_ok = batch > h->n_cu || h->arith_profile == 1;
    if (!tile1) { planF = build_tri_plan(S, h->threads, true, solo_ok); planB = build_tri_plan(S, h->threads, false, solo_ok); }
    else { planF.idx.assign(1, NV); planB.idx.assign(1, NV); planF.pos.assign(S.nnzL, 0); planB.pos.assign(S.nnzL, 0); }
    D.nfs = planF.n_wide; D.nbs = planB.n_wide; D.nfs_solo = planF.n_solo; D.nbs_solo = planB.n_solo; D.nfs_ext = planF.n_ext; D.nUF = planF.slots; D.nUB = planB.slots;
    {   // the real slices of every section: its length without the trailing padding (empty slices)
        auto real = [](const std::vector<SliceMeta> &sl, int first, int count) { while (count > 0 && sl[(size_t)first + count - 1].cnt == 0) count--; return count; };
        // forward plan = [wide | solo | ext], backward plan = [solo | wide] (plans.cpp)
        D.nfs_r = real(planF.sl, 0, planF.n_wide); D.nfs_solo_r = real(planF.sl, planF.n_wide, planF.n_solo); D.nfs_ext_r = real(planF.sl, planF.n_wide + planF.n_solo, planF.n_ext);
        D.nbs_solo_r = real(planB.sl, 0, planB.n_solo); D.nbs_r = real(planB.sl, planB.n_solo, planB.n_wide);
    }
    // every section of a sweep plan is a whole number of queue-depth trips (tri_sweep's remainder loop executes full trips)
    if (D.nfs % TRI_DEPTH || D.nbs % TRI_DEPTH || D.nfs_ext % TRI_DEPTH || D.nfs_solo % TRI_DEPTH_SOLO || D.nbs_solo % TRI_DEPTH_SOLO) {
        delete h; return fail(EICOS_E_INVALID, "internal: a section of a sweep plan is not padded to its queue depth");
    }
    // (value arrays: the plan's slots + the dummy slot, then -- dense apex -- the folded image of the block's own entries (APEX_IMG doubles), zero wherever no
    // entry of L lands: the work slabs are zeroed at creation and the factor program only ever writes entry slots)
    D.w_UF = Wl.add((size_t)planF.ulen + 8); D.w_UB = Wl.add((size_t)planB.ulen + 8);
    const bool apex = !tile && S.apex0 >= 0;
    D.apex_na = apex ? S.N - S.apex0 : 0; D.apex_n0 = apex ? S.apex0 : 0; D.apex_f = planF.apex_base; D.apex_b = planB.apex_base;
    D.apex_split_n = apex ? planF.split_n : 0; D.apex_split_slot = planF.split_slot0; D.apex_split_lane = planF.split_row - D.apex_n0;
    // (the parts' pseudo-rows N + 1 ... use the spare slots of the sweep vector: checked here against the plan's own bound and below, where the
    // vector's stride D.Npad is fixed, against that stride itself)
    if (D.apex_split_n > 0 && (tile || planF.split_slot0 + planF.split_n > scalar_npad(NV))) { delete h; return fail(EICOS_E_INVALID, "internal: the split row of the apex does not fit the spare slots of the sweep vector"); }
    h->posB = planB.pos; h->ub_len = planB.ulen;
    // numeric factorisation program: reads L.*D through the backward (column) slots; slot nUB is the zero dummy
    FactorPlan planX;
    if (!tile1) planX = build_factor_plan(S, h->threads, planB.pos, planB.slots, planF.pos, planF.slots);
    else { planX.pa.assign(1, 0); planX.pb.assign(1, 0); planX.pbU.assign(1, 0); planX.pk.assign(1, 0); }
    D.fac_ns = (int)planX.sl.size(); D.fac_slots = planX.slots; D.fac_nt = (int)planX.target.size();
    if ((long long)planB.ulen + 1 >= IMG_BASE || (long long)S.N >= DIAG_POS / 2) { delete h; return fail(EICOS_E_UNSUPPORTED, "pattern too large for the factor program's destination codes"); }
    {   // level 0 of the factor program: the leaves of the elimination tree have no pairs; the kernel streams over their targets
        // (diagonals first: the per-level task order is stable for equal pair counts) instead of walking their slices
        D.fac_s1 = 0; D.fac_nd0 = 0; D.fac_nt0 = 0;
        if (!tile1 && !planX.sl.empty()) {
            size_t s1 = 1;
            while (s1 < planX.sl.size() && !(planX.sl[s1].newlev & 1)) s1++;
            const int nt0 = s1 < planX.sl.size() ? planX.sl[s1].row0 : (int)planX.target.size();
            bool pairless = true, diag_first = true;
            int nd0 = 0;
            for (int t = 0; t < nt0; t++) {
                const int tgt = planX.target[t];
                if (S.tp[tgt + 1] != S.tp[tgt]) pairless = false;
                if (tgt < S.N) { if (t != nd0) diag_first = false; nd0++; }
            }
            if (pairless && diag_first) { D.fac_s1 = (int)s1; D.fac_nd0 = nd0; D.fac_nt0 = nt0; }
        }
    }
    // KKT entries in target order: the factor's only per-target value stream; tile mode: the dense tile image
    D.w_Kt = Wl.add((tile1 ? (size_t)(TP.nb + TP.nt) * 256 : planX.target.size()) + 8);
    D.w_Kimg = tile1 ? D.w_Kt : (tile ? Wl.add((size_t)(TP.nb + TP.nt) * 256 + 8) : 0); // hybrid: the top block's image beside the scalar stream
    D.tile = S.tile; D.nb = TP.nb; D.nt = TP.nt; D.nblev = TP.nblev; D.tl_base = TP.n0;
    if (tile) { // unit-lower L tiles column-major (LC) and row-major (LR), the strictly lower part of the diagonal tiles (DL)
        D.w_LC = Wl.add((size_t)TP.nt * 256 + 256); D.w_LR = Wl.add((size_t)TP.nt * 256 + 256); // (+ one tile: the dummy loads of padding operations read tile 0)
        D.w_DL = Wl.add((size_t)TP.nb * 256);
    }
    D.w_dual_xk = Wl.add(2 * ((size_t)NV + 16)); D.w_dual_ek = Wl.add(2 * ((size_t)NV + 16)); // dual right-hand-side solves
    D.work_stride = Wl.size;
    std::vector<int> fac_src(planX.target.size()), fac_dst(planX.target.size()), fac_dstF(planX.target.size()), fac_col(planX.target.size(), 0);
    std::vector<int> col_of(S.nnzL);
    for (int j = 0; j < S.N; j++) for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) col_of[e] = j;
    // ---- device form of the slice tables (PackedSlice) and 16-bit gather indices ----
    // One 8-byte entry per lane and slice = its ELL_KMAX gather indices; entry position = off16 of the slice + lane;
    // the entry after the last slice is all padding (read by inactive lanes).  Used when every index fits 16 bits.
    bool idx16_ok = true;
    auto lane_offsets = [](const std::vector<SliceMeta> &sl, int &dummy) {
        std::vector<int> off16(sl.size());
        int pos = 0;
        for (size_t i = 0; i < sl.size(); i++) { off16[i] = pos; pos += sl[i].cnt << sl[i].lg; }
        dummy = pos;
        return off16;
    };
    auto pack16 = [&](const std::vector<SliceMeta> &sl, const std::vector<int> &off16, int dummy, const std::vector<int> &idx, int pad) {
        std::vector<int> words(((size_t)dummy + 1) * 2, 0); // two 32-bit words = four 16-bit indices per lane entry
        auto set = [&](size_t entry, int kk, int v) {
            if (v < 0 || v > 65535) { idx16_ok = false; v = 0; }
            words[entry * 2 + (kk >> 1)] |= v << (16 * (kk & 1));
        };
        for (size_t i = 0; i < sl.size(); i++) {
            const int lanes = sl[i].cnt << sl[i].lg;
            for (int t = 0; t < lanes; t++)
                for (int kk = 0; kk < ELL_KMAX; kk++)
                    set((size_t)off16[i] + t, kk, kk < sl[i].K ? idx[(size_t)sl[i].off + (size_t)kk * lanes + t] : pad);
        }
        for (int kk = 0; kk < ELL_KMAX; kk++) set((size_t)dummy, kk, pad);
        return words;
    };
    auto meta_ints = [](const std::vector<SliceMeta> &v, const std::vector<int> &off16) {
        std::vector<int> o(v.size() * 4);
        for (size_t i = 0; i < v.size(); i++) {
            const PackedSlice ps = pack_slice(v[i], off16.empty() ? 0 : off16[i]);
            std::memcpy(o.data() + 4 * i, &ps, sizeof ps);
        }
        return o;
    };
    const std::vector<int> f_o16 = lane_offsets(planF.sl, D.f_d16), b_o16 = lane_offsets(planB.sl, D.b_d16);
    const std::vector<int> cag_o16 = lane_offsets(pcag.sl, D.cag_d16), rA_o16 = lane_offsets(prA.sl, D.rA_d16), rG_o16 = lane_offsets(prG.sl, D.rG_d16);
    const std::vector<int> f_w16 = pack16(planF.sl, f_o16, D.f_d16, planF.idx, NV), b_w16 = pack16(planB.sl, b_o16, D.b_d16, planB.idx, NV);
    const std::vector<int> cag_k_w16 = pack16(pcag.sl, cag_o16, D.cag_d16, cag_idx_k, NV), cag_yz_w16 = pack16(pcag.sl, cag_o16, D.cag_d16, cag_idx_yz, 0);
    const std::vector<int> rA_w16 = pack16(prA.sl, rA_o16, D.rA_d16, rA_idx, 0), rA_k_w16 = pack16(prA.sl, rA_o16, D.rA_d16, rA_idx_k, NV);
    const std::vector<int> rG_w16 = pack16(prG.sl, rG_o16, D.rG_d16, rG_idx, 0), rG_k_w16 = pack16(prG.sl, rG_o16, D.rG_d16, rG_idx_k, NV);
    // factor program: the (pa, pb) slot pairs of a lane, 16 bytes per lane and slice (pa words then pb words)
    const std::vector<int> x_o16 = lane_offsets(planX.sl, D.fac_d16);
    std::vector<int> fac_w16, fac_w16d, fac_k16;
    {
        const std::vector<int> wa = pack16(planX.sl, x_o16, D.fac_d16, planX.pa, planB.slots), wb = pack16(planX.sl, x_o16, D.fac_d16, planX.pb, planF.slots);
        const std::vector<int> wu = pack16(planX.sl, x_o16, D.fac_d16, planX.pbU, planB.slots); // deferred-L form: both operands are UB slots
        fac_k16 = pack16(planX.sl, x_o16, D.fac_d16, planX.pk, S.N);                             // ... and the pivot column of every pair
        fac_w16.resize(wa.size() * 2); fac_w16d.resize(wa.size() * 2);
        for (size_t e = 0; e * 2 < wa.size(); e++) {
            fac_w16[4 * e] = wa[2 * e]; fac_w16[4 * e + 1] = wa[2 * e + 1]; fac_w16[4 * e + 2] = wb[2 * e]; fac_w16[4 * e + 3] = wb[2 * e + 1];
            fac_w16d[4 * e] = wa[2 * e]; fac_w16d[4 * e + 1] = wa[2 * e + 1]; fac_w16d[4 * e + 2] = wu[2 * e]; fac_w16d[4 * e + 3] = wu[2 * e + 1];
        }
    }
    D.idx16 = (idx16_ok && env_int("EICOS_IDX16", 1, 0, 1)) ? 1 : 0;
    std::vector<int> fsl_i = meta_ints(planF.sl, f_o16), bsl_i = meta_ints(planB.sl, b_o16), fac_sl_i = meta_ints(planX.sl, x_o16);
    std::vector<int> cag_sl_i = meta_ints(pcag.sl, cag_o16), rA_sl_i = meta_ints(prA.sl, rA_o16), rG_sl_i = meta_ints(prG.sl, rG_o16);

    struct Slot { const int **dst; size_t off; };
    std::vector<Slot> slots;
    auto put = [&](const int *&field, const std::vector<int> &v) { slots.push_back({&field, pool.add(v)}); };
    put(D.Ajc, P.Ajc); put(D.Air, P.Air); put(D.At_ptr, S.At_ptr); put(D.At_pos, S.At_pos);
    put(D.Gjc, P.Gjc); put(D.Gir, P.Gir); put(D.Gt_ptr, S.Gt_ptr); put(D.Gt_pos, S.Gt_pos);
    {
        std::vector<int> Acol(S.nnzA), Gcol(S.nnzG);
        for (int j = 0; j < S.n; j++) { for (int k = P.Ajc[j]; k < P.Ajc[j + 1]; k++) Acol[k] = j; for (int k = P.Gjc[j]; k < P.Gjc[j + 1]; k++) Gcol[k] = j; }
        put(D.Acol, Acol); put(D.Gcol, Gcol);
    }
    put(D.cq, S.q); put(D.cone_off, S.cone_off); put(D.cone_vbase, cone_vbase); put(D.cone_small, cone_small); put(D.cone_big, cone_big);
    D.n_small = (int)cone_small.size(); D.n_big = (int)cone_big.size();
    {   // tiny cones (dimension <= TINY_D): descriptor table for the register-resident cone loops; the rest of the small cones
        std::vector<int> tiny_tab, cone_mid;
        for (int c : cone_small) {
            const int d = S.q[c], o = S.cone_off[c];
            if (d > TINY_D) { cone_mid.push_back(c); continue; }
            int rec[TINY_INTS] = {o, d, c, ipv[c], ipu[c], 0, 0, 0, 0, cone_vbase[c], 0, 0};
            for (int k = 0; k < TINY_D; k++) rec[5 + k] = ipz[o + std::min(k, d - 1)]; // (slots past the dimension repeat the last row: loads stay in range)
            tiny_tab.insert(tiny_tab.end(), rec, rec + TINY_INTS);
        }
        D.n_tiny = (int)tiny_tab.size() / TINY_INTS; D.n_mid = (int)cone_mid.size();
        put(D.cone_tiny, tiny_tab); put(D.cone_mid, cone_mid);
        std::vector<int> cone_wave, cone_huge;
        for (int c : cone_big) (S.q[c] <= 64 ? cone_wave : cone_huge).push_back(c);
        D.n_wave = (int)cone_wave.size(); D.n_huge = (int)cone_huge.size();
        put(D.cone_wave, cone_wave); put(D.cone_huge, cone_huge);
    }
    put(D.zdsign, zdsign);
    put(D.f_idx, planF.idx); put(D.b_idx, planB.idx);
    put(D.f_idx16, f_w16); put(D.b_idx16, b_w16); put(D.cag_k16, cag_k_w16); put(D.cag_yz16, cag_yz_w16);
    put(D.rA_16, rA_w16); put(D.rA_k16, rA_k_w16); put(D.rG_16, rG_w16); put(D.rG_k16, rG_k_w16);
    const int *fsl_p = nullptr, *bsl_p = nullptr, *cag_sl_p = nullptr, *rA_sl_p = nullptr, *rG_sl_p = nullptr;
    put(fsl_p, fsl_i); put(bsl_p, bsl_i); put(cag_sl_p, cag_sl_i); put(rA_sl_p, rA_sl_i); put(rG_sl_p, rG_sl_i);
    put(D.cag_idx_k, cag_idx_k); put(D.cag_idx_yz, cag_idx_yz); put(D.cag_src, cag_src);
    put(D.rA_idx, rA_idx); put(D.rA_src, rA_src); put(D.rG_idx, rG_idx); put(D.rG_src, rG_src);
    put(D.rA_idx_k, rA_idx_k); put(D.rG_idx_k, rG_idx_k);
    {   // G tiles: columns of a tile as variable index and as elimination-order slot, slot of every row's z entry
        std::vector<int> gt_colk(GT.col.size()), gt_zslot((size_t)GT.nrb * 16, NV);
        for (size_t q = 0; q < GT.col.size(); q++) gt_colk[q] = GT.col[q] < 0 ? NV : posK(GT.col[q]);
        if (GT.on) for (int i = 0; i < S.m; i++) gt_zslot[i] = posK(S.n + S.p + zexp0[i]);
        put(D.gt_rbptr, GT.rbptr); put(D.gt_col, GT.col); put(D.gt_colk, gt_colk); put(D.gt_zslot, gt_zslot); put(D.gt_cidx, GT.cidx); put(D.gt_src, GT.src);
    }
    put(D.ipx, ipx); put(D.ipy, ipy); put(D.ipz, ipz); put(D.ipv, ipv); put(D.ipu, ipu);
    std::vector<int> ipk(ipx); ipk.insert(ipk.end(), ipy.begin(), ipy.end()); ipk.insert(ipk.end(), ipz.begin(), ipz.end());
    put(D.ipk, ipk);
    const int Npad_v = tile ? NV + 16 : (NV + 1 + 15) & ~15; // (= D.Npad below)
    std::vector<int> zpos; // sweep-vector slots no x / y / z entry lands in: cone expansions, padding (inside the blocks in tile layouts)
    {
        std::vector<char> hit((size_t)Npad_v, 0);
        for (int o : ipk) hit[o] = 1;
        for (int i = 0; i < Npad_v; i++) if (!hit[i]) zpos.push_back(i);
    }
    D.nzpos = (int)zpos.size();
    put(D.zpos, zpos);
    // quasi-definite sign of pivot `pos` (elimination position): + for the x block and the u expansion of every cone
    // (ref setupKKT :1734-1890), - elsewhere; only used by the dynamic-regularisation extension
    auto pivot_positive = [&](int pos) {
        const int orig = S.perm[pos];
        if (orig < S.n) return true;
        int k = S.n + S.p + S.l;
        for (int c = 0; c < S.nc; c++) { if (orig == k + S.q[c] + 1) return true; k += S.q[c] + 2; }
        return false;
    };
    for (size_t t = 0; t < planX.target.size(); t++) {
        const int tgt = planX.target[t];
        if (tgt < S.N) { fac_src[t] = srcoff(S.Dkind[tgt], S.Dsrc[tgt]); fac_dst[t] = -tgt - 1 - (pivot_positive(tgt) ? DIAG_POS : 0); fac_dstF[t] = 0; }
        else { const int e = tgt - S.N; fac_src[t] = srcoff(S.Lkind[e], S.Lsrc[e]); fac_dst[t] = planB.pos[e]; fac_dstF[t] = planF.pos[e]; fac_col[t] = col_of[e]; }
        // hybrid: the targets of the top block (their pairs stop at column n0) are the tile factorisation's input image
        if (S.tile == 2 && tgt < S.N && tgt >= S.n0) { fac_dst[t] = IMG_BASE + TP.D_img[tgt]; fac_dstF[t] = -1; }
        if (S.tile == 2 && tgt >= S.N && col_of[tgt - S.N] >= S.n0) { fac_dst[t] = IMG_BASE + TP.Le_img[tgt - S.N]; fac_dstF[t] = -1; }
    }
    std::vector<int> v2t(std::max(S.nV, 1), D.fac_nt); // entries that are no target (none by construction) -> spare slot
    for (size_t t = 0; t < planX.target.size(); t++)
        if (fac_src[t] >= D.i_Vv && fac_src[t] < D.i_Vv + S.nV) v2t[fac_src[t] - D.i_Vv] = (int)t;
    // ---- tile mode: where every KKT entry lands in the dense tile image, pivot signs, the tile program ----
    std::vector<int> img_dst, img_src, psign(tile ? NV : 0, 1);
    if (tile) {
        for (int j = 0; j < S.N; j++) psign[TP.slot[j]] = pivot_positive(j) ? 1 : -1;
        if (tile1) { // the image is filled straight from the instance slab (hybrid: by the scalar factor program)
            for (int j = 0; j < S.N; j++) {
                img_dst.push_back(TP.D_img[j]); img_src.push_back(srcoff(S.Dkind[j], S.Dsrc[j]));
                if (S.Dkind[j] == SRC_V) v2t[S.Dsrc[j]] = TP.D_img[j];
            }
            for (int e = 0; e < S.nnzL; e++) {
                if (S.Lkind[e] == SRC_ZERO) continue; // fill: stays 0 in the image
                img_dst.push_back(TP.Le_img[e]); img_src.push_back(srcoff(S.Lkind[e], S.Lsrc[e]));
                if (S.Lkind[e] == SRC_V) v2t[S.Lsrc[e]] = TP.Le_img[e];
            }
        }
        for (int d : TP.pad_img) { img_dst.push_back(d); img_src.push_back(D.i_cst + 3); } // padding nodes: identity rows
    }
    D.tl_nimg = (int)img_dst.size();
    put(D.tl_img_dst, img_dst); put(D.tl_img_src, img_src); put(D.tl_psign, psign);
    put(D.tl_blev, TP.blev_ptr); put(D.tl_tgt_lev, TP.tgt_lev_ptr); put(D.tl_tgt, TP.tgt); put(D.tl_tp, TP.tp_ptr);
    put(D.tl_pa, TP.pa); put(D.tl_pb, TP.pb); put(D.tl_pk, TP.pk); put(D.tl_fin_lev, TP.fin_lev_ptr); put(D.tl_fin, TP.fin);
    TileSweeps TSW;
    // (hybrid patterns whose vectors certainly live in LDS -- a serially swept block system relies on the in-order LDS accesses of one wavefront)
    if (tile) TSW = build_tile_sweeps(TP, h->threads / 64, TILE_STRIP, (S.tile == 2 && TP.N16 <= 4096) ? env_int("EICOS_TILE_SERIAL_MAX", 48, 0, 100000) : 0);
    put(D.tl_fops, TSW.fops); put(D.tl_bops, TSW.bops); put(D.tl_fptr, TSW.fptr); put(D.tl_bptr, TSW.bptr);
    put(D.tl_fsplit, TSW.fsplit); put(D.tl_bsplit, TSW.bsplit); put(D.tl_fend, TSW.fend); put(D.tl_bend, TSW.bend);
    TileFactorOps TFO;
    if (tile) {
        // pure tile mode: tiles of the K image no KKT entry lands in (targets that exist through fill only) start from zero without
        // a load; the image tile they would have read stays zero and is never touched (hybrid: the scalar program writes every tile)
        std::vector<char> img_zero((size_t)TP.nb + TP.nt, tile1 ? 1 : 0);
        for (int d : img_dst) img_zero[d / 256] = 0;
        TFO = build_tile_factor_ops(TP, h->threads / 64, TILE_FTRIP, &img_zero);
    }
    put(D.tl_facops, TFO.ops); put(D.tl_facptr, TFO.ptr);
    put(D.tl_ident, TP.ident);
    put(D.tl_trow, TP.t_row); put(D.tl_tcol, TP.t_col); put(D.tl_tc_ptr, TP.tc_ptr); put(D.tl_tr_ptr, TP.tr_ptr); put(D.tl_tr_tile, TP.tr_tile);
    put(D.v2t, v2t);
    const int *fac_sl_p = nullptr;
    put(fac_sl_p, fac_sl_i);
    const int *fac_pb_f = nullptr, *fac_pb_u = nullptr, *fac_p16_f = nullptr, *fac_p16_u = nullptr; // (stored-L / deferred-L forms: chosen with NLDS below)
    put(D.fac_pa, planX.pa); put(fac_pb_f, planX.pb); put(fac_pb_u, planX.pbU); put(fac_p16_f, fac_w16); put(fac_p16_u, fac_w16d);
    put(D.fac_pk, planX.pk); put(D.fac_k16, fac_k16);
    put(D.fac_src, fac_src); put(D.fac_dst, fac_dst); put(D.fac_dstF, fac_dstF); put(D.fac_col, fac_col);

    // ---- device resources ----
    auto bail = [&](int code, const std::string &msg) { eicos_batch_destroy(h); return fail(code, msg); };
#define HIP_TRY_H(expr)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return bail(EICOS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
    // (device set-up -- kernel attributes, occupancy probes, allocations, the constant-memory slot; the shards of an eicos_multi are
    // created on parallel host threads, which overlaps their symbolic analyses above)
    // One lock PER DEVICE: the shards of an eicos_multi that live on different GPUs set their devices up in parallel, two handles on one
    // GPU still take turns (the occupancy probes and hipFuncSetAttribute calls of one device must not interleave).
    static std::mutex g_create_map_mu;
    static std::map<int, std::mutex> g_create_mu;
    std::mutex *dev_mu;
    { std::lock_guard<std::mutex> lk(g_create_map_mu); dev_mu = &g_create_mu[device]; }
    std::lock_guard<std::mutex> create_lock(*dev_mu);
    HIP_TRY_H(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY_H(hipGetDeviceProperties(&prop, device));
    // launch shape: env overrides are for experiments (bench sweeps); defaults chosen from measurements
    // KKT-space vectors (solve vector, current solution, refinement residual) live in LDS when they fit:
    // 160 KiB per CU minus the static block (reductions + scalar state)
    if (D.apex_split_n > 0 && D.apex_split_slot + D.apex_split_n > Npad_v) { delete h; return fail(EICOS_E_INVALID, "internal: the split row of the apex runs past the sweep vector's stride"); }
    D.Npad = Npad_v; // tile ? NV + 16 : (NV + 1 + 15) & ~15 // >= N+1: slot N is the always-zero target of ELL padding (tile mode: a whole zero block)
    {
        D.lm_f = 0; D.lm_b = D.lm_f + D.nfs + D.nfs_solo + D.nfs_ext; D.lm_cag = D.lm_b + D.nbs + D.nbs_solo; D.lm_rA = D.lm_cag + D.cag_ns; D.lm_rG = D.lm_rA + D.rA_ns;
        D.lm_total = D.lm_rG + D.rG_ns;
        const size_t avail = 160 * 1024 - 4096, vec = (size_t)std::max(D.Npad, 16) * sizeof(double);
        const size_t lds_static = 4096; // struct Sh + the per-instance states of kernels.hip (reductions + scalar state), rounded up
        // tile mode: one 16 x 17 fp64 scratch tile per wavefront (dense LDL' of the diagonal tiles), behind the tables
        // ... and the partial-sum slots of split blocks in the tile sweeps (TILE_PARTS x 16 rows x two right-hand sides)
        // dense apex: the packed image of the block's L (same place: the scalar path has no tile scratch)
        const size_t apex_img = (D.apex_na > 0 && h->threads >= 256) ? (size_t)APEX_IMG * sizeof(double) : 0; // (128 threads: the image IS the slab's, in LDS)
        const size_t scratch = tile ? ((size_t)(h->threads / 64) * TILE_SCR + (size_t)TILE_PARTS * 16 * KI_MAX_HOST) * sizeof(double) : apex_img;
        // workgroups per CU that 160 KB of LDS allow with one vector + tables of `slices` entries
        const int wgs_by_regs = (h->threads == 256 ? 3 : (h->threads == 512 ? 2 : 4)) * 4 / (h->threads / 64); // waves_per_eu<T>() of kernels.hip
        auto wgs_per_cu = [&](int slices) {
            return std::min(wgs_by_regs, (int)((160 * 1024) / (vec + (size_t)slices * sizeof(PackedSlice) + scratch + lds_static)));
        };
        // the factor program's table goes to LDS too when it is small and does not cost a resident workgroup
        if (D.fac_ns <= 512 && wgs_per_cu(D.lm_total + D.fac_ns) == wgs_per_cu(D.lm_total)) {
            D.lm_fac = D.lm_total; D.lm_total += D.fac_ns;
        } else D.lm_fac = -1;
        const size_t meta = (size_t)D.lm_total * sizeof(PackedSlice) + scratch;
        // NLDS >= 1 also stages both slice tables in LDS; if they do not fit beside one vector the
        // all-global variant (NLDS = 0, plain __syncthreads between levels) is used
        // KKT-space vectors in LDS: E (rhs / residual / solve vector) and X (current solution), + both slice tables
        int fit = 0;
        if (NV > 0 && meta + vec <= avail) fit = (meta + 2 * vec <= avail) ? 2 : 1;
        // more instances than CUs: keep only E in LDS so that several workgroups share a CU (measured)
        int want = fit;
        if (batch > prop.multiProcessorCount && fit == 2 && 2 * (vec + meta + 4096) <= 160 * 1024) want = 1;
        h->nlds = std::max(0, std::min(fit, env_int("EICOS_NLDS", want, 0, 2)));
        // Dual right-hand-side solves (the two independent systems of the initialisation and of every pass share one
        // sweep over the factor): needs two vectors in LDS.  Pure tile mode (bandwidth-bound on streaming L and G): always.
        // Scalar / hybrid programs: when the batch fits one workgroup per CU -- the sweeps are then a dependent chain of level
        // steps, and a step for two right-hand sides costs far less than two steps
        int dual = (fit == 2 && (tile1 || batch <= prop.multiProcessorCount)) ? 1 : 0;
        dual = env_int("EICOS_DUAL", dual, 0, 1);
        if (fit < 2) dual = 0;
        if (dual) h->nlds = 1;
        D.dual = dual;
        const int nvec = dual ? 2 : h->nlds; // vectors of Npad doubles at the start of the dynamic LDS
        D.meta_lds = h->nlds >= 1 ? 1 : 0;
        // deferred-L factorisation (device_types.hpp: fac_defer): needs the idle LDS solve vector for the mirror of 1/D
        // It trades one LDS read per pair for a write + read of every L entry and one barrier per level: measured +1.5..3.3 % on MPC02 and
        // nine Netlib patterns (pairs/nnzL 1.7..7.5), -1 % on lp_25fv47 (10.1) -- profiles/r03_log_defer.log; the rule below is that fit.
        const int defer_auto = (double)S.npairs <= 8.0 * (double)S.nnzL ? 1 : 0;
        D.fac_defer = (!tile1 && h->nlds >= 1 && env_int("EICOS_FAC_DEFER", defer_auto, 0, 1)) ? 1 : 0;
        D.fac_kpad = S.N;
        h->dyn_lds = h->nlds >= 1 ? (size_t)nvec * vec + meta : (tile ? scratch : 0); // (no LDS vector: the apex sweeps read the global images, no LDS image)
        D.lds_tab = h->nlds >= 1 ? nvec * D.Npad : 0;
        D.tl_scratch = h->nlds >= 1 ? nvec * D.Npad + D.lm_total * 2 : 0; // in doubles from the start of the dynamic LDS
        D.tl_part = D.tl_scratch + (h->threads / 64) * TILE_SCR;
        D.apex_lds = (D.apex_na > 0 && h->nlds >= 1) ? D.tl_scratch : -1; // (no LDS vector: the apex sweeps read the global images)
        D.apex_inplace = 0;
        // LDS-resident variant (small patterns, kernels_ldsres.hip): when the instance slab and the workspace slab fit LDS
        // beside the vectors and tables, k_solve works on LDS copies of both, so the elementwise stages and the products wait
        // for LDS instead of L2 (+12 % on lp_afiro at batch 256; the level-by-level sweeps are issue-bound and do not change:
        // DESIGN.md 5.1).  Only for batches that fit the grid in one round -- beyond that the eight small workgroups per CU
        // of the HBM-slab kernel hide more latency than the <= 3 that LDS holds here (measured, lp_afiro batch 2048).
        h->ldsres = 0; D.lr_inst = D.lr_work = 0;
        if (!tile && h->nlds >= 1 && h->threads == 128 && env_int("EICOS_LDSRES", 1, 0, 1)) {
            const size_t base = (h->dyn_lds + 15) & ~(size_t)15, islab = (D.inst_stride + 1) & ~(size_t)1, wslab = (D.work_stride + 1) & ~(size_t)1;
            const size_t total = base + (islab + wslab) * sizeof(double);
            const size_t per_cu = (160 * 1024) / (total + lds_static); // workgroups per CU that LDS allows
            if (per_cu >= 1 && (size_t)batch <= per_cu * (size_t)prop.multiProcessorCount) {
                h->ldsres = 1; D.lr_inst = (int)(base / sizeof(double)); D.lr_work = D.lr_inst + (int)islab;
                h->dyn_lds = total;
            }
        }
        if (D.apex_na > 0 && h->threads == 128) { // the apex of a 128-thread handle reads the forward image where it is: the LDS copy of the workspace slab
            if (!h->ldsres) return bail(EICOS_RETRY_NO_APEX, "internal: 128-thread handle with an apex but without the LDS-resident build");
            D.apex_lds = D.lr_work + D.w_UF + D.apex_f; D.apex_inplace = 1;
        }
    }
    int bpc = 1;
    {
        const SolveBuild sb = solve_build(h->threads, h->ldsres, false);
        HIP_TRY_H(sb.set_max_lds(h->threads, h->nlds, h->dp.idx16, h->dyn_lds));
        HIP_TRY_H(sb.occupancy(h->threads, h->nlds, h->dp.idx16, h->dyn_lds, &bpc));
    }
    bpc = std::max(1, std::min(bpc, 8));
    HIP_TRY_H(update_set_max_lds()); // (per handle = per device, after hipSetDevice: the entry-parallel updateData kernels use up to 160 KB of dynamic LDS)
    {
        // Workgroups per CU for this batch: the cheapest estimate (launch_blocks_per_cu, measured constants)
        bpc = launch_blocks_per_cu(batch, prop.multiProcessorCount, bpc, h->threads);
    }
    bpc = std::max(1, std::min(bpc, env_int("EICOS_BLOCKS_PER_CU", bpc, 1, 8)));
    // 256 threads at <= 2 workgroups per CU: the build with 256 VGPRs per thread (the default one is held to 168 so that three fit)
    h->w2 = 0;
    if (!h->ldsres && h->threads == 256 && bpc <= 2 && env_int("EICOS_W2", 1, 0, 1)) {
        int got = 0;
        HIP_TRY_H(w2::solve_set_max_lds(h->threads, h->nlds, h->dp.idx16, h->dyn_lds));
        HIP_TRY_H(w2::solve_occupancy(h->threads, h->nlds, h->dp.idx16, h->dyn_lds, &got));
        if (got >= bpc) h->w2 = 1;
    }
    // One workgroup per CU (batch <= CUs) and the factor operand array U = L.*D fits the LDS that the lone workgroup leaves idle: the build that
    // keeps it there (kernels_ubl*.hip).  The numeric factorisation of a deep pattern is a chain of levels that each wait for operand gathers
    // and for their stores to land -- L2 round trips with U in the workspace slab, LDS round trips here; bit-identical results.
    h->ubl = 0; D.ub_lds = -1; D.ub_len = h->ub_len;
    if (!h->ldsres && (h->threads == 256 || h->threads == 512) && bpc == 1 && batch <= prop.multiProcessorCount && h->nlds >= 1 && D.fac_defer && !tile1 &&
        !(D.apex_na > 0 && D.apex_lds < 0) && env_int("EICOS_UBL", 1, 0, 1)) {
        const size_t base = (h->dyn_lds + 15) & ~(size_t)15, need = base + ((size_t)h->ub_len + 8) * sizeof(double);
        if (need + 4096 <= 160 * 1024) { // (4 KB: the static block, as budgeted above)
            const SolveBuild ub = solve_build(h->threads, false, false, true);
            int got = 0;
            HIP_TRY_H(ub.set_max_lds(h->threads, h->nlds, h->dp.idx16, need));
            HIP_TRY_H(ub.occupancy(h->threads, h->nlds, h->dp.idx16, need, &got));
            if (got >= 1) { h->ubl = 1; h->w2 = 0; D.ub_lds = (int)(base / sizeof(double)); h->dyn_lds = need; }
        }
    }
    const SolveBuild sbuild = solve_build(h->threads, h->ldsres, h->w2, h->ubl);
    auto v_set_max_lds = sbuild.set_max_lds;
    auto v_occupancy = sbuild.occupancy;
    // The LDS that `bpc` resident workgroups leave free takes the head of the refinement residual E (device_types.hpp: e_lds): its
    // scattered stores and the read-back stay on chip.  Verified against the runtime's occupancy for the enlarged allocation.
    D.e_lds = 0; D.e_off = 0;
    if (!h->ldsres && h->nlds == 1 && !D.dual && S.tile != 1) {
        const size_t base = (h->dyn_lds + 15) & ~(size_t)15, room = (160 * 1024) / (size_t)bpc;
        size_t xs = room > base + 4096 + 1024 ? std::min<size_t>((size_t)NV, (room - base - 4096 - 1024) / sizeof(double)) & ~(size_t)15 : 0;
        while (xs > 0) {
            int got = 0;
            HIP_TRY_H(v_set_max_lds(h->threads, h->nlds, h->dp.idx16, base + xs * sizeof(double)));
            HIP_TRY_H(v_occupancy(h->threads, h->nlds, h->dp.idx16, base + xs * sizeof(double), &got));
            if (got >= bpc) break;
            xs = (xs * 3 / 4) & ~(size_t)15;
        }
        if (xs > 0) { D.e_lds = (int)xs; D.e_off = (int)(base / sizeof(double)); h->dyn_lds = base + xs * sizeof(double); }
    }
    h->bpc = bpc;
    const int resident = prop.multiProcessorCount * bpc;
    h->grid = std::min(batch, resident);
    h->order_min = prop.multiProcessorCount;
    h->upd_grid = std::min(batch, prop.multiProcessorCount * 4);
    {   // entry-parallel updateData: needs the A / G values and the row / column maxima in LDS and <= 8 vector entries per thread
        const size_t need = ((size_t)S.nnzA + S.nnzG + S.n + S.p + S.m + 8) * sizeof(double);
        const size_t need_max = ((size_t)S.n + S.p + S.m + 8) * sizeof(double); // the row / column maxima alone
        const bool small_vecs = S.n <= 8 * 512 && S.p <= 8 * 512 && S.m <= 16 * 512;
        const int mode = env_int("EICOS_UPDATE_LDS", 1, 0, 2); // 0: thread-per-column kernel, 2: force the streamed-values variant
        if (need <= 156 * 1024 && small_vecs && mode == 1) { h->upd_lds = need; h->upd_vals_lds = 1; h->upd_grid = std::min(batch, prop.multiProcessorCount); }
        else if (need_max <= 156 * 1024 && small_vecs && mode >= 1) { // values streamed in place, maxima in LDS: as many 512-thread workgroups per CU as fit (<= 4)
            h->upd_lds = need_max; h->upd_vals_lds = 0;
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / (need_max + 1024)));
            h->upd_grid = std::min(batch, prop.multiProcessorCount * per_cu);
        }
    }
    h->pattern_ints = pool.data.size();
    HIP_TRY_H(hipMalloc(&h->d_pattern, pool.data.size() * sizeof(int)));
    HIP_TRY_H(hipMemcpy(h->d_pattern, pool.data.data(), pool.data.size() * sizeof(int), hipMemcpyHostToDevice));
    for (auto &s : slots) *s.dst = h->d_pattern + s.off;
    D.fac_pb = D.fac_defer ? fac_pb_u : fac_pb_f; D.fac_p16 = D.fac_defer ? fac_p16_u : fac_p16_f;
    D.fsl = reinterpret_cast<const PackedSlice *>(fsl_p); D.bsl = reinterpret_cast<const PackedSlice *>(bsl_p);
    D.cag_sl = reinterpret_cast<const PackedSlice *>(cag_sl_p); D.rA_sl = reinterpret_cast<const PackedSlice *>(rA_sl_p);
    D.rG_sl = reinterpret_cast<const PackedSlice *>(rG_sl_p);
    D.fac_sl = reinterpret_cast<const PackedSlice *>(fac_sl_p);
    {
        std::lock_guard<std::mutex> lk(g_slot_mu);
        std::vector<char> &used = g_slot_used[device];
        used.resize((size_t)max_patterns(), 0);
        for (int q = 0; q < max_patterns(); q++) if (!used[q]) { h->pslot = q; used[q] = 1; break; }
    }
    if (h->pslot < 0) return bail(EICOS_E_INVALID, "too many live handles on this device (64)");
    HIP_TRY_H(upload_pattern(h->pslot, h->dp)); // (the default namespace always: updateData and the debug kernels live there)
    if (sbuild.upload != upload_pattern) HIP_TRY_H(sbuild.upload(h->pslot, h->dp));
    HIP_TRY_H(hipMalloc(&h->d_inst, (size_t)batch * D.inst_stride * sizeof(double)));
    HIP_TRY_H(hipMemset(h->d_inst, 0, (size_t)batch * D.inst_stride * sizeof(double)));
    HIP_TRY_H(hipMalloc(&h->d_work, (size_t)h->grid * D.work_stride * sizeof(double)));
    HIP_TRY_H(hipMemset(h->d_work, 0, (size_t)h->grid * D.work_stride * sizeof(double)));
    HIP_TRY_H(hipMalloc(&h->d_queue, (16 + (size_t)batch) * sizeof(int))); // [0] queue head, [16..] longest-first order
    HIP_TRY_H(hipMalloc(&h->d_scratch, (size_t)h->upd_grid * (size_t)(S.n + S.p + S.m + 8) * sizeof(double)));
    HIP_TRY_H(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    // (the event pairs of the timing rings are created on first use: next_events)
    HIP_TRY_H(hipDeviceSynchronize());
    *out = h;
    return EICOS_OK;
}

// ---- single-instance surface: a batch of one (reference include/eicos.hpp:151-163, test/ecos.h:11-34)
int eicos_create(int n, int m, int p, int l, int ncones, const int *q, const double *Gpr, const int *Gjc, const int *Gir,
                 const double *Apr, const int *Ajc, const int *Air, const double *c, const double *h, const double *b,
                 int device, eicos_batch **out) {
    if (!out) return fail(EICOS_E_INVALID, "out is NULL");
    const bool haveG = Gpr && Gjc && Gir, haveA = Apr && Ajc && Air; // NULL groups as in src/eicos.cpp:103-117
    if (n > 0 && !c) return fail(EICOS_E_INVALID, "c is NULL");
    eicos_batch *hd = nullptr;
    (void)l; // ignored exactly as by the reference's constructor (src/eicos.cpp:91): derived as m - sum(q)
    int rc = eicos_batch_create(n, m, p, -1, ncones, q, haveG ? Gjc : nullptr, haveG ? Gir : nullptr,
                                haveA ? Ajc : nullptr, haveA ? Air : nullptr, 1, device, &hd);
    if (rc != EICOS_OK) return rc;
    rc = eicos_batch_update(hd, 0, 1, haveG ? Gpr : nullptr, haveA ? Apr : nullptr, c, haveG ? h : nullptr, haveA ? b : nullptr);
    if (rc != EICOS_OK) { eicos_batch_destroy(hd); return rc; }
    *out = hd;
    return EICOS_OK;
}
int eicos_update(eicos_batch *hd, const double *Gpr, const double *Apr, const double *c, const double *h, const double *b) {
    return eicos_batch_update(hd, 0, 1, Gpr, Apr, c, h, b);
}
int eicos_solve(eicos_batch *hd, int *exitcode) {
    if (hd && hd->batch != 1) return fail(EICOS_E_INVALID, "eicos_solve needs a handle made by eicos_create (batch of one)");
    return eicos_batch_solve(hd, exitcode);
}
int eicos_solution(eicos_batch *hd, double *x) { return eicos_batch_solution(hd, x); }
int eicos_info_get(eicos_batch *hd, eicos_info *info) { return eicos_batch_info(hd, info); }
int eicos_destroy(eicos_batch *hd) { return eicos_batch_destroy(hd); }

int eicos_batch_destroy(eicos_batch *h) {
    if (!h) return EICOS_OK;
    (void)hipSetDevice(h->device);
    // in-flight work may sit on a caller stream (eicos_batch_set_stream): wait for it before the slabs go away
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->own_stream) { (void)hipStreamSynchronize(h->own_stream); (void)hipStreamDestroy(h->own_stream); }
    for (int i = 0; i < eicos_batch::EV_RING; i++)
        for (hipEvent_t e : {h->ring_s[i][0], h->ring_s[i][1], h->ring_u[i][0], h->ring_u[i][1]}) if (e) (void)hipEventDestroy(e);
    for (void *ptr : {(void *)h->d_pattern, (void *)h->d_inst, (void *)h->d_work, (void *)h->d_queue, (void *)h->d_scratch,
                      (void *)h->d_stage, (void *)h->d_flag})
        if (ptr) (void)hipFree(ptr);
    for (int i = 0; i < 2; i++) { if (h->pin[i]) (void)hipHostFree(h->pin[i]); if (h->pin_ev[i]) (void)hipEventDestroy(h->pin_ev[i]); }
    if (h->stage_pin) (void)hipHostFree(h->stage_pin);
    if (h->stage_flags) (void)hipHostFree(h->stage_flags);
    if (h->d_err) (void)hipFree(h->d_err);
    // the constant-memory descriptor slot is handed out again only after nothing can read it any more
    if (h->pslot >= 0) { std::lock_guard<std::mutex> lk(g_slot_mu); g_slot_used[h->device][h->pslot] = 0; }
    delete h;
    return EICOS_OK;
}

int eicos_batch_set_warm_start(eicos_batch *h, double shift) {
    if (!h || !(shift >= 0.)) return fail(EICOS_E_INVALID, "bad argument");
    h->warm_shift = shift;
    return EICOS_OK;
}

int eicos_batch_set_dynamic_regularization(eicos_batch *h, double delta, double eps) {
    if (!h || !(delta >= 0.) || !(eps >= 0.)) return fail(EICOS_E_INVALID, "bad argument");
    h->dyn_delta = delta; h->dyn_eps = eps;
    return EICOS_OK;
}

int eicos_batch_set_stream(eicos_batch *h, void *hip_stream) {
    if (!h) return fail(EICOS_E_INVALID, "NULL handle");
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return EICOS_OK;
}

// the next event pair of a timing ring (created on first use) becomes the handle's current pair
static int next_events(hipEvent_t (*ring)[2], long &count, hipEvent_t &e0, hipEvent_t &e1) {
    hipEvent_t *slot = ring[count % eicos_batch::EV_RING];
    for (int i = 0; i < 2; i++) if (!slot[i]) HIP_TRY(hipEventCreate(&slot[i]));
    e0 = slot[0]; e1 = slot[1]; count++;
    return EICOS_OK;
}
static int begin_update_timing(eicos_batch *h) {
    int rc = next_events(h->ring_u, h->n_update_rec, h->ev_u0, h->ev_u1);
    if (rc != EICOS_OK) return rc;
    HIP_TRY(hipEventRecord(h->ev_u0, h->stream));
    return EICOS_OK;
}

int eicos_batch_update_device(eicos_batch *h, int first, int count, const double *dG, const double *dA,
                              const double *dc, const double *dh, const double *db) {
    if (!h) return fail(EICOS_E_INVALID, "NULL handle");
    if (first < 0 || count < 0 || first + count > h->batch) return fail(EICOS_E_INVALID, "instance range out of bounds");
    if (dG && !dh && h->dp.m > 0) return fail(EICOS_E_INVALID, "Gpr given without h");
    if (dA && !db && h->dp.p > 0) return fail(EICOS_E_INVALID, "Apr given without b");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->in_chunked_update) { const int rc = begin_update_timing(h); if (rc != EICOS_OK) return rc; }
    HIP_TRY(launch_update(h->pslot, h->d_inst, first, count, dG, dA, dc, dh, db, h->d_scratch, std::min(count, h->upd_grid), h->upd_lds, h->upd_vals_lds, h->stream));
    if (!h->in_chunked_update) { HIP_TRY(hipEventRecord(h->ev_u1, h->stream)); h->update_timed = true; }
    return EICOS_OK;
}

// ---- host memory helpers -------------------------------------------------------------------------------------------------
// A few persistent host threads that split large memcpy calls between pageable and pinned memory (one core copies ~10 GB/s, a PCIe 5
// x16 link moves ~50 GB/s: a single-threaded bounce copy would be the slowest stage of a host-pointer updateData).  Shared by every
// handle of the process, started on first use, joined at exit.
namespace {
// memcpy with non-temporal stores: the destination of a bounce copy is read next by the GPU over PCIe (or is the caller's result array), never
// by this core -- regular stores would first read every destination line into the cache (read-for-ownership: a third more memory traffic)
// and evict the caller's working set.  glibc switches to such stores only above a few MB per call; the pool's pieces are ~1.5 MB.
static void stream_copy(void *dst, const void *src, size_t bytes) {
    static const bool nt = env_knob("EICOS_COPY_NT", 1, 0, 1) != 0;
    char *d = (char *)dst; const char *s = (const char *)src;
    if (!nt || bytes < 4096) { std::memcpy(d, s, bytes); return; }
    const size_t head = std::min(bytes, (size_t)((16 - ((uintptr_t)d & 15)) & 15));
    std::memcpy(d, s, head); d += head; s += head; bytes -= head;
    const size_t blocks = bytes / 64;
    for (size_t i = 0; i < blocks; i++, d += 64, s += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i *)s), b = _mm_loadu_si128((const __m128i *)(s + 16));
        const __m128i c = _mm_loadu_si128((const __m128i *)(s + 32)), e = _mm_loadu_si128((const __m128i *)(s + 48));
        _mm_stream_si128((__m128i *)d, a); _mm_stream_si128((__m128i *)(d + 16), b);
        _mm_stream_si128((__m128i *)(d + 32), c); _mm_stream_si128((__m128i *)(d + 48), e);
    }
    _mm_sfence();
    std::memcpy(d, s, bytes - blocks * 64);
}
class CopyPool {
  public:
    static CopyPool &get() { static CopyPool p; return p; }
    // dst[0, bytes) = src[0, bytes), cut into pieces of >= 1 MB over the pool's threads and the caller
    void copy(void *dst, const void *src, size_t bytes) {
        const size_t piece = 1u << 20;
        const int parts = (int)std::min<size_t>((size_t)nthreads_ + 1, (bytes + piece - 1) / piece);
        if (parts <= 1) { stream_copy(dst, src, bytes); return; }
        const size_t per = ((bytes + parts - 1) / parts + 63) & ~(size_t)63;
        // completion state on the caller's stack: the count is changed and the caller notified INSIDE the lock, so a helper's last access to
        // these objects is its unlock, which the caller's wait cannot overtake
        int left = parts - 1;
        std::mutex done_mu; std::condition_variable done_cv;
        for (int k = 1; k < parts; k++) {
            const size_t a = std::min(bytes, (size_t)k * per), b = std::min(bytes, a + per);
            push([=, &left, &done_mu, &done_cv] {
                if (b > a) stream_copy((char *)dst + a, (const char *)src + a, b - a);
                std::lock_guard<std::mutex> lk(done_mu);
                if (--left == 0) done_cv.notify_one();
            });
        }
        stream_copy(dst, src, std::min(bytes, per));
        std::unique_lock<std::mutex> lk(done_mu);
        done_cv.wait(lk, [&] { return left == 0; });
    }
  private:
    // cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a container's hardware_concurrency() is the host's)
    static int usable_cores() {
        int n = (int)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0}; long period = 0;
            if (std::fscanf(f, "%31s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) n = std::max(1, std::min(n, (int)(std::atol(q) / period)));
            std::fclose(f);
        }
        return n;
    }
    CopyPool() {
        // the bounce copy of a host-pointer updateData must keep up with the PCIe link (~50 GB/s; one core copies ~10 GB/s): up to ten
        // helper threads, two cores left to the caller and the runtime; EICOS_COPY_THREADS overrides (experiments)
        const int cores = usable_cores();
        nthreads_ = env_knob("EICOS_COPY_THREADS", std::max(0, std::min(10, cores - 2)), 0, 64);
        for (int i = 0; i < nthreads_; i++) th_.emplace_back([this] { run(); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void push(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(f)); } cv_.notify_one(); }
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                f = std::move(q_.front()); q_.erase(q_.begin());
            }
            f();
        }
    }
    int nthreads_ = 0; bool stop_ = false;
    std::vector<std::thread> th_;
    std::vector<std::function<void()>> q_;
    std::mutex mu_; std::condition_variable cv_;
};

// Is `p` host memory the GPU can address directly (hipHostMalloc / hipHostRegister / eicos_host_alloc)?  Then kernels read or write it
// in place over PCIe and no bounce copy is needed.
// 0 = pageable (or managed) host-addressable memory, 1 = pinned / registered host memory, 2 = device memory
int pointer_kind(const void *p) {
    if (!p) return 0;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return 0; } // (plain malloc memory: "invalid value")
    if (a.type == hipMemoryTypeHost) return 1;
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeArray) return 2;
    return 0; // (managed memory is host-addressable: it takes the bounce path like pageable memory)
}
// Is the WHOLE extent [p, p + bytes) pinned / registered host memory?  The kernels read (updateData) or the copy engine writes (results) every byte of
// it in place, so the first byte alone does not decide: a pointer into a registered buffer with a count that runs past its end, or a buffer
// registered a second time with a larger size (hipHostRegister reports "already registered" and maps nothing new), would be a GPU page
// fault instead of an error code.  First byte, last byte and one probe per 2 MB in between (a lookup costs about a microsecond).
bool is_pinned_host(const void *p, size_t bytes) {
    if (pointer_kind(p) != 1) return false;
    if (bytes <= 1) return true;
    const char *b = (const char *)p;
    if (pointer_kind(b + bytes - 1) != 1) return false;
    for (size_t o = 2u << 20; o < bytes - 1; o += 2u << 20) if (pointer_kind(b + o) != 1) return false;
    return true;
}
// the handle's two pinned bounce buffers hold at least `doubles` each
int ensure_pin(eicos_batch *h, size_t doubles) {
    if (doubles <= h->pin_doubles) return EICOS_OK;
    for (int i = 0; i < 2; i++) {
        if (h->pin_busy[i]) { HIP_TRY(hipEventSynchronize(h->pin_ev[i])); h->pin_busy[i] = false; }
        if (h->pin[i]) { (void)hipHostFree(h->pin[i]); h->pin[i] = nullptr; }
        if (!h->pin_ev[i]) HIP_TRY(hipEventCreateWithFlags(&h->pin_ev[i], hipEventDisableTiming));
    }
    h->pin_doubles = 0;
    for (int i = 0; i < 2; i++) HIP_TRY(hipHostMalloc((void **)&h->pin[i], doubles * sizeof(double), hipHostMallocDefault));
    h->pin_doubles = doubles;
    return EICOS_OK;
}
constexpr size_t PIN_CHUNK_BYTES = 16u << 20; // bounce buffer size aimed at (per buffer)
} // namespace

void *eicos_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); g_err = "hipHostMalloc failed"; return nullptr; }
    return p;
}
int eicos_host_free(void *p) {
    if (!p) return EICOS_OK;
    HIP_TRY(hipHostFree(p));
    return EICOS_OK;
}
// Pin arrays the caller already owns (std::vector storage, numpy arrays, ...) IN PLACE: hipHostRegister.  From then on updateData / solution
// treat them like eicos_host_alloc memory (no bounce copy).  Registering costs about as much as a few bounce copies of the same
// bytes, so it pays for arrays that are reused across calls -- the sample-by-sample rewrite of an MPC loop; the caller unregisters
// before freeing them.
int eicos_host_register(void *p, size_t bytes) {
    if (!p || bytes == 0) return fail(EICOS_E_INVALID, "bad argument");
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e == hipErrorHostMemoryAlreadyRegistered) { // fine only if the existing registration covers the whole range asked for
        (void)hipGetLastError();
        if (is_pinned_host(p, bytes)) return EICOS_OK;
        return fail(EICOS_E_INVALID, "eicos_host_register: the pointer is already registered with a SMALLER extent (unregister it first)");
    }
    if (e != hipSuccess) return fail(EICOS_E_HIP, std::string("hipHostRegister: ") + hipGetErrorString(e));
    return EICOS_OK;
}
int eicos_host_unregister(void *p) {
    if (!p) return EICOS_OK;
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(EICOS_E_HIP, std::string("hipHostUnregister: ") + hipGetErrorString(e)); }
    return EICOS_OK;
}
int eicos_batch_last_update_path(eicos_batch *h) { return h ? h->last_update_path : fail(EICOS_E_INVALID, "NULL handle"); }

// updateData from buffers that are not in the handle's HBM: host memory (src_dev < 0) or the HBM of another GPU (src_dev = that device,
// eicos_multi_update_device).
//   host, pageable : rows go through the two pinned bounce buffers in chunks -- the host copies chunk k + 1 in (CopyPool) while the
//                    updateData kernel of chunk k reads its inputs straight from the other buffer over PCIe; returns when the last chunk
//                    has been COPIED (the caller's arrays are free again), the kernels are still in flight on the handle's stream
//   host, pinned   : (every given array addressable by the GPU) ONE kernel launch reads the caller's arrays in place; the call
//                    waits for it, so that the caller may overwrite them on return -- the reference's updateData is synchronous too
//   peer, access on: the kernel reads the other GPU's HBM in place over xGMI (asynchronous, like eicos_batch_update_device)
//   peer, no access: hipMemcpyPeerAsync into a device staging buffer, chunk by chunk
int eicos_internal_update_staged(eicos_batch *h, int first, int count, const double *G, const double *A,
                                 const double *c, const double *hh, const double *b, int src_dev) {
    if (!h) return fail(EICOS_E_INVALID, "NULL handle");
    if (first < 0 || count < 0 || first + count > h->batch) return fail(EICOS_E_INVALID, "instance range out of bounds");
    const DevPat &D = h->dp;
    if (G && !hh && D.m > 0) return fail(EICOS_E_INVALID, "Gpr given without h");
    if (A && !b && D.p > 0) return fail(EICOS_E_INVALID, "Apr given without b");
    HIP_TRY(hipSetDevice(h->device));
    if (count == 0) return EICOS_OK;
    const double *hv = G ? hh : nullptr, *bv = A ? b : nullptr; // (h is read only with Gpr, b only with Apr: reference src/eicos.cpp:2053-2074)
    struct Arr { const double *src; size_t w; };
    const Arr arr[5] = {{G, (size_t)D.nnzG}, {A, (size_t)D.nnzA}, {c, (size_t)D.n}, {hv, (size_t)D.m}, {bv, (size_t)D.p}};
    size_t per = 0; // doubles per instance that are actually given (+ 8 of padding per array keeps every row 64-byte aligned)
    for (const Arr &a : arr) if (a.src) per += a.w;
    if (per == 0) { // nothing given: everything is kept -- still a valid updateData (re-equilibrates what is there)
        return eicos_batch_update_device(h, first, count, nullptr, nullptr, nullptr, nullptr, nullptr);
    }
    auto whole_range = [&](int path) { // one launch on the caller's pointers
        h->last_update_path = path;
        return eicos_batch_update_device(h, first, count, G, A, c, hv, bv);
    };
    if (src_dev >= 0) {
        int can = 0;
        if (src_dev == h->device) can = 1;
        else if (hipDeviceCanAccessPeer(&can, h->device, src_dev) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        if (can && src_dev != h->device) {
            const hipError_t e = hipDeviceEnablePeerAccess(src_dev, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
            (void)hipGetLastError();
        }
        if (can && !env_knob("EICOS_PEER_STAGED", 0, 0, 1)) return whole_range(3);
        // no peer access: staged peer copies, one chunk at a time through the device staging buffer
        h->last_update_path = 4;
        const int chunk = 256;
        const size_t need = (size_t)std::min(count, chunk) * (per + 5 * 8);
        if (need > h->stage_doubles) {
            if (h->d_stage) { HIP_TRY(hipStreamSynchronize(h->stream)); (void)hipFree(h->d_stage); h->d_stage = nullptr; h->stage_doubles = 0; }
            HIP_TRY(hipMalloc(&h->d_stage, need * sizeof(double)));
            h->stage_doubles = need;
        }
        int rc = EICOS_OK;
        { const int rc0 = begin_update_timing(h); if (rc0 != EICOS_OK) return rc0; }
        h->in_chunked_update = true;
        for (int o = 0; o < count && rc == EICOS_OK; o += chunk) {
            const int cnt = std::min(chunk, count - o);
            const double *dptr[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
            double *at = h->d_stage;
            for (int k = 0; k < 5 && rc == EICOS_OK; k++) {
                if (!arr[k].src) continue;
                dptr[k] = at;
                if (arr[k].w && hipMemcpyPeerAsync(at, h->device, arr[k].src + (size_t)o * arr[k].w, src_dev, (size_t)cnt * arr[k].w * sizeof(double), h->stream) != hipSuccess)
                    rc = fail(EICOS_E_HIP, "hipMemcpyPeerAsync failed");
                at += (size_t)cnt * arr[k].w + 8;
            }
            if (rc == EICOS_OK) rc = eicos_batch_update_device(h, first + o, cnt, dptr[0], dptr[1], dptr[2], dptr[3], dptr[4]);
            // the staging buffer is reused by the next chunk
            if (rc == EICOS_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(EICOS_E_HIP, "stream sync failed in update");
        }
        h->in_chunked_update = false;
        if (rc == EICOS_OK) { HIP_TRY(hipEventRecord(h->ev_u1, h->stream)); h->update_timed = true; }
        return rc;
    }
    // ---- host pointers ----
    bool all_pinned = true, any_device = false;
    for (const Arr &a : arr) if (a.src && a.w) {
        const int kind = pointer_kind(a.src);
        if (kind == 2) any_device = true;
        // pinned in place only when EVERY byte the kernel will read is mapped (else the bounce path, which reads with the host's own loads)
        if (kind != 1 || !is_pinned_host(a.src, (size_t)count * a.w * sizeof(double))) all_pinned = false;
    }
    // a device pointer handed to the HOST-pointer entry point must not reach the bounce copy (a host memcpy from it would fault)
    if (any_device) return fail(EICOS_E_INVALID, "eicos_batch_update takes host pointers: an array lives in device memory (use eicos_batch_update_device)");
    if (all_pinned && !env_knob("EICOS_HOST_BOUNCE", 0, 0, 1)) {
        const int rc = whole_range(2);
        if (rc != EICOS_OK) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream)); // the caller may overwrite its arrays on return
        return EICOS_OK;
    }
    h->last_update_path = 1;
    const size_t row = per + 5 * 8;
    int chunk = (int)std::max<size_t>(16, PIN_CHUNK_BYTES / (row * sizeof(double)));
    chunk = std::min(chunk, count);
    if (count > chunk && count < 2 * chunk) chunk = (count + 1) / 2; // two even chunks rather than a long one and a stub
    int rc = ensure_pin(h, (size_t)chunk * row);
    if (rc != EICOS_OK) return rc;
    rc = begin_update_timing(h);
    if (rc != EICOS_OK) return rc;
    h->in_chunked_update = true;
    CopyPool &pool = CopyPool::get();
    int k = 0;
    for (int o = 0; o < count && rc == EICOS_OK; o += chunk, k++) {
        const int cnt = std::min(chunk, count - o), bi = k & 1;
        if (h->pin_busy[bi]) { // the kernel that read this buffer two chunks ago (or an earlier call's) must have finished
            if (hipEventSynchronize(h->pin_ev[bi]) != hipSuccess) { rc = fail(EICOS_E_HIP, "event sync failed in update"); break; }
            h->pin_busy[bi] = false;
        }
        const double *dptr[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        double *at = h->pin[bi];
        for (int q = 0; q < 5; q++) {
            if (!arr[q].src) continue;
            dptr[q] = at;
            if (arr[q].w) pool.copy(at, arr[q].src + (size_t)o * arr[q].w, (size_t)cnt * arr[q].w * sizeof(double));
            at += (size_t)cnt * arr[q].w + 8;
        }
        rc = eicos_batch_update_device(h, first + o, cnt, dptr[0], dptr[1], dptr[2], dptr[3], dptr[4]); // reads the pinned buffer in place
        if (rc == EICOS_OK) {
            if (hipEventRecord(h->pin_ev[bi], h->stream) != hipSuccess) rc = fail(EICOS_E_HIP, "event record failed in update");
            else h->pin_busy[bi] = true;
        }
    }
    h->in_chunked_update = false;
    if (rc == EICOS_OK) { HIP_TRY(hipEventRecord(h->ev_u1, h->stream)); h->update_timed = true; }
    return rc;
}

int eicos_batch_update(eicos_batch *h, int first, int count, const double *G, const double *A,
                       const double *c, const double *hh, const double *b) {
    return eicos_internal_update_staged(h, first, count, G, A, c, hh, b, -1);
}

int eicos_batch_solve_async(eicos_batch *h) {
    if (!h) return fail(EICOS_E_INVALID, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    { const int rc = next_events(h->ring_s, h->n_solve_rec, h->ev_s0, h->ev_s1); if (rc != EICOS_OK) return rc; }
    h->ring_step0[(h->n_solve_rec - 1) % eicos_batch::EV_RING] = h->update_timed ? h->ev_u0 : h->ev_s0;
    HIP_TRY(hipEventRecord(h->ev_s0, h->stream));
    HIP_TRY(solve_build(h->threads, h->ldsres, h->w2, h->ubl).launch(h->pslot, h->d_inst, h->d_work, h->batch, h->d_queue, h->d_queue + 16, h->grid, h->threads, h->nlds,
                                                             h->dp.idx16, h->order_min, h->warm_shift, h->dyn_delta, h->dyn_eps, h->dyn_lds, h->stream,
                                                             h->fused_pending ? &h->fused : nullptr));
    HIP_TRY(hipEventRecord(h->ev_s1, h->stream));
    h->solve_timed = true;
    h->last_ordered = h->batch > h->order_min;
    return EICOS_OK;
}

int eicos_batch_sync(eicos_batch *h) {
    if (!h) return fail(EICOS_E_INVALID, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return EICOS_OK;
}

// rows [off, off + width) of every instance slab -> dst[batch][width] on the host.  Pinned destination: one strided device-to-host copy
// straight into it.  Pageable destination: chunks through the two pinned bounce buffers, the copy of chunk k + 1 in flight while the
// host copies chunk k out (a strided hipMemcpy2D into pageable memory is staged by the runtime row by row).
static int fetch_rows(eicos_batch *h, double *dst, int off, int width) {
    if (!dst || width == 0) return EICOS_OK;
    const size_t wb = (size_t)width * sizeof(double), pitch = h->dp.inst_stride * sizeof(double);
    if (is_pinned_host(dst, (size_t)h->batch * wb) || (pointer_kind(dst) != 1 && (size_t)h->batch * wb < (256u << 10))) {
        HIP_TRY(hipMemcpy2DAsync(dst, wb, h->d_inst + off, pitch, wb, (size_t)h->batch, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return EICOS_OK;
    }
    // (chunks of ~4 MB: small enough that the device-to-host copy of one chunk and the host copy of the previous one overlap on a result of a few MB)
    int chunk = (int)std::max<size_t>(16, (PIN_CHUNK_BYTES / 4) / wb);
    chunk = std::min(chunk, h->batch);
    int rc = ensure_pin(h, (size_t)chunk * width);
    if (rc != EICOS_OK) return rc;
    for (int i = 0; i < 2; i++) if (h->pin_busy[i]) { HIP_TRY(hipEventSynchronize(h->pin_ev[i])); h->pin_busy[i] = false; }
    CopyPool &pool = CopyPool::get();
    auto issue = [&](int o, int bi) -> int {
        const int cnt = std::min(chunk, h->batch - o);
        HIP_TRY(hipMemcpy2DAsync(h->pin[bi], wb, h->d_inst + (size_t)o * h->dp.inst_stride + off, pitch, wb, (size_t)cnt, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipEventRecord(h->pin_ev[bi], h->stream));
        return EICOS_OK;
    };
    rc = issue(0, 0);
    int k = 0;
    for (int o = 0; o < h->batch && rc == EICOS_OK; o += chunk, k++) {
        const int cnt = std::min(chunk, h->batch - o), bi = k & 1;
        if (o + chunk < h->batch) rc = issue(o + chunk, bi ^ 1);
        if (rc != EICOS_OK) break;
        HIP_TRY(hipEventSynchronize(h->pin_ev[bi]));
        pool.copy(dst + (size_t)o * width, h->pin[bi], (size_t)cnt * wb);
    }
    return rc;
}

int eicos_batch_info(eicos_batch *h, eicos_info *info) {
    if (!h || !info) return fail(EICOS_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<DevInfo> tmp(h->batch);
    HIP_TRY(hipMemcpy2D(tmp.data(), sizeof(DevInfo), h->d_inst + h->dp.i_info, h->dp.inst_stride * sizeof(double),
                        sizeof(DevInfo), (size_t)h->batch, hipMemcpyDeviceToHost));
    for (int i = 0; i < h->batch; i++) {
        const DevInfo &d = tmp[i];
        eicos_info &o = info[i];
        o.pcost = d.pcost; o.dcost = d.dcost; o.pres = d.pres; o.dres = d.dres; o.gap = d.gap; o.relgap = d.relgap;
        o.sigma = d.sigma; o.mu = d.mu; o.step = d.step; o.step_aff = d.step_aff; o.kapovert = d.kapovert;
        o.pinfres = d.pinfres; o.dinfres = d.dinfres; o.tau = d.tau; o.kap = d.kap;
        o.has_relgap = d.has_relgap; o.has_pinfres = d.has_pinfres; o.has_dinfres = d.has_dinfres;
        o.pinf = d.pinf; o.dinf = d.dinf; o.iter = d.iter; o.nitref1 = d.nitref1; o.nitref2 = d.nitref2;
        o.nitref3 = d.nitref3; o.exitcode = d.exitcode; o.n_factor = d.n_factor; o.n_ldlsolve = d.n_ldlsolve;
        o.n_sweep = d.n_sweep; o.reserved_ = 0; o.solve_us = d.solve_us;
    }
    return EICOS_OK;
}

int eicos_batch_solve(eicos_batch *h, int *exitcodes) {
    int rc = eicos_batch_solve_async(h);
    if (rc != EICOS_OK) return rc;
    rc = eicos_batch_sync(h);
    if (rc != EICOS_OK) return rc;
    if (exitcodes) {
        std::vector<eicos_info> info(h->batch);
        rc = eicos_batch_info(h, info.data());
        if (rc != EICOS_OK) return rc;
        for (int i = 0; i < h->batch; i++) exitcodes[i] = info[i].exitcode;
    }
    return EICOS_OK;
}

// updateData + solve in ONE call: the reference's updateData(double *...) followed by solve() (include/eicos.hpp:155-158) for the whole batch.
// When every given array is memory the GPU addresses directly (pinned / registered host memory, or device memory) the updateData of an
// instance is run by the solve kernel's own workgroup right before it solves that instance (kernels.hip: update_instance): the transfer over
// PCIe is the workgroups' loads, spread over the launch and hidden behind the other workgroups' compute -- a separate updateData kernel can
// only run BEFORE the solve (the registers and the LDS of a CU are fully owned by its resident solve workgroups), so its transfer time adds to
// every step.  x_out (optional, [batch][n]): pinned host / device memory is written by the kernel as each instance finishes; pageable memory
// is filled by eicos_batch_solution afterwards.  Anything else (pageable inputs -- unless staging is switched on, below --, a handle without an
// LDS vector, vectors beyond the in-register scaling accumulators) takes eicos_batch_update + eicos_batch_solve: same results, bit for bit, on
// every path.
// Synchronous; exitcodes optional.
int eicos_batch_update_solve(eicos_batch *h, const double *G, const double *A, const double *c, const double *hh, const double *b,
                             double *x_out, int *exitcodes) {
    if (!h) return fail(EICOS_E_INVALID, "NULL handle");
    const DevPat &D = h->dp;
    if (G && !hh && D.m > 0) return fail(EICOS_E_INVALID, "Gpr given without h");
    if (A && !b && D.p > 0) return fail(EICOS_E_INVALID, "Apr given without b");
    HIP_TRY(hipSetDevice(h->device));
    const double *hv = G ? hh : nullptr, *bv = A ? b : nullptr; // (h is read only with Gpr, b only with Apr: reference src/eicos.cpp:2053-2074)
    struct Arr { const double *src; size_t w; };
    const Arr arr[5] = {{G, (size_t)D.nnzG}, {A, (size_t)D.nnzA}, {c, (size_t)D.n}, {hv, (size_t)D.m}, {bv, (size_t)D.p}};
    auto gpu_addressable = [&](const void *ptr, size_t bytes) { const int k = pointer_kind(ptr); return k == 2 || (k == 1 && is_pinned_host(ptr, bytes)); };
    const bool fused = h->nlds >= 1 && D.n <= 8 * h->threads && D.p <= 8 * h->threads && D.m <= 16 * h->threads && env_knob("EICOS_FUSED_UPDATE", 1, 0, 1);
    // arrays the GPU cannot address (pageable memory) are STAGED: copied into the handle's pinned staging buffer while the kernel runs
    bool staged[5] = {false, false, false, false, false};
    size_t stage_need = 0;
    for (int k = 0; k < 5; k++) if (arr[k].src && arr[k].w && !gpu_addressable(arr[k].src, (size_t)h->batch * arr[k].w * sizeof(double))) {
        if (pointer_kind(arr[k].src) == 2) return fail(EICOS_E_INVALID, "an array straddles device memory");
        staged[k] = true; stage_need += (size_t)h->batch * arr[k].w + 8;
    }
    const bool any_staged = stage_need > 0;
    const bool x_direct = x_out && D.n > 0 && gpu_addressable(x_out, (size_t)h->batch * D.n * sizeof(double));
    int rc;
    // (staging pageable arrays while the kernel runs is OFF by default: measured on five boxes against the bounce pipeline + solve it is
    // +5.7 ... -6.2 % -- the host's copy is the pace either way, and on a box with slow host cores the kernel's own PCIe pulls and flag polls
    // slow that copy further; EICOS_FUSED_STAGED=1 under EICOS_EXPERIMENT=1 turns it on: docs/HISTORY.md A.11 item 9)
    if (!fused || (any_staged && !env_knob("EICOS_FUSED_STAGED", 0, 0, 1))) {
        bool all_device = true; // (device arrays on a handle without the fused path: the device-pointer updateData)
        for (int k = 0; k < 5; k++) if (arr[k].src && arr[k].w && pointer_kind(arr[k].src) != 2) all_device = false;
        rc = all_device ? eicos_batch_update_device(h, 0, h->batch, G, A, c, hh, b) : eicos_batch_update(h, 0, h->batch, G, A, c, hh, b);
        if (rc == EICOS_OK) rc = eicos_batch_solve_async(h);
        if (rc != EICOS_OK) return rc;
        rc = eicos_batch_sync(h);
        if (rc != EICOS_OK) return rc;
        if (x_out && D.n > 0) { rc = fetch_rows(h, x_out, D.i_x, D.n); if (rc != EICOS_OK) return rc; }
    } else {
        const double *ptr[5] = {G, A, c, hv, bv};
        // chunks of ~12 MB over the staged arrays: the copy pool splits an array's share of a chunk into pieces of >= 1 MB over its threads, so
        // a chunk must be large enough to keep them busy and small enough that the first workgroups start after a fraction of the whole copy
        size_t per = 0;
        for (int k = 0; k < 5; k++) if (staged[k]) per += arr[k].w;
        const int chunk = any_staged ? (int)std::max<size_t>(8, std::min<size_t>((size_t)h->batch, (12u << 20) / std::max<size_t>(per * sizeof(double), 1))) : h->batch;
        const int nchunks = (h->batch + chunk - 1) / chunk;
        if (any_staged) {
            if (stage_need > h->stage_pin_doubles) {
                HIP_TRY(hipStreamSynchronize(h->stream));
                if (h->stage_pin) { (void)hipHostFree(h->stage_pin); h->stage_pin = nullptr; h->stage_pin_doubles = 0; }
                HIP_TRY(hipHostMalloc((void **)&h->stage_pin, stage_need * sizeof(double), hipHostMallocDefault));
                h->stage_pin_doubles = stage_need;
            }
            if (nchunks > h->stage_nflags) {
                HIP_TRY(hipStreamSynchronize(h->stream));
                if (h->stage_flags) { (void)hipHostFree(h->stage_flags); h->stage_flags = nullptr; h->stage_nflags = 0; }
                HIP_TRY(hipHostMalloc((void **)&h->stage_flags, (size_t)nchunks * sizeof(unsigned), hipHostMallocDefault));
                std::memset(h->stage_flags, 0, (size_t)nchunks * sizeof(unsigned));
                h->stage_nflags = nchunks; h->stage_seq = 0;
            }
            if (!h->d_err) { HIP_TRY(hipMalloc((void **)&h->d_err, sizeof(int))); HIP_TRY(hipMemset(h->d_err, 0, sizeof(int))); }
            double *at = h->stage_pin;
            for (int k = 0; k < 5; k++) if (staged[k]) { ptr[k] = at; at += (size_t)h->batch * arr[k].w + 8; }
            h->stage_seq++;
            if (h->stage_seq == 0) { std::memset(h->stage_flags, 0, (size_t)h->stage_nflags * sizeof(unsigned)); h->stage_seq = 1; } // (wrapped)
        }
        h->last_update_path = any_staged ? 6 : 5;
        rc = begin_update_timing(h); // (an empty updateData interval in the timing ring: the work is inside the solve launch)
        if (rc != EICOS_OK) return rc;
        HIP_TRY(hipEventRecord(h->ev_u1, h->stream)); h->update_timed = true;
        h->fused = UpdArgs{ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], x_direct ? x_out : nullptr, 1,
                           any_staged ? h->stage_flags : nullptr, chunk, h->stage_seq, h->d_err};
        h->fused_pending = true;
        rc = eicos_batch_solve_async(h);
        h->fused_pending = false;
        if (rc != EICOS_OK) return rc; // (nothing was launched: no workgroup waits for a flag)
        if (any_staged) { // the kernel is running: copy chunk by chunk and release each chunk's flag behind its rows
            CopyPool &pool = CopyPool::get();
            for (int q = 0; q < nchunks; q++) {
                const size_t r0 = (size_t)q * chunk, rows = std::min<size_t>((size_t)chunk, (size_t)h->batch - r0);
                for (int k = 0; k < 5; k++) if (staged[k]) pool.copy(const_cast<double *>(ptr[k]) + r0 * arr[k].w, arr[k].src + r0 * arr[k].w, rows * arr[k].w * sizeof(double));
                __atomic_store_n(&h->stage_flags[q], h->stage_seq, __ATOMIC_RELEASE); // (stream_copy ends with an sfence: the rows are visible before the flag)
            }
        }
        rc = eicos_batch_sync(h);
        if (rc != EICOS_OK) return rc;
        if (any_staged) {
            int err = 0;
            HIP_TRY(hipMemcpy(&err, h->d_err, sizeof(int), hipMemcpyDeviceToHost));
            if (err) { HIP_TRY(hipMemset(h->d_err, 0, sizeof(int))); return fail(EICOS_E_HIP, "fused updateData: a workgroup timed out waiting for its staged rows"); }
        }
        if (x_out && D.n > 0 && !x_direct) { rc = fetch_rows(h, x_out, D.i_x, D.n); if (rc != EICOS_OK) return rc; }
    }
    if (exitcodes) {
        std::vector<eicos_info> info(h->batch);
        rc = eicos_batch_info(h, info.data());
        if (rc != EICOS_OK) return rc;
        for (int i = 0; i < h->batch; i++) exitcodes[i] = info[i].exitcode;
    }
    return EICOS_OK;
}

int eicos_batch_solution(eicos_batch *h, double *x) {
    if (!h || !x) return fail(EICOS_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return fetch_rows(h, x, h->dp.i_x, h->dp.n);
}

int eicos_batch_duals(eicos_batch *h, double *y, double *z, double *s) {
    if (!h) return fail(EICOS_E_INVALID, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int rc = fetch_rows(h, y, h->dp.i_y, h->dp.p);
    if (rc == EICOS_OK) rc = fetch_rows(h, z, h->dp.i_z, h->dp.m);
    if (rc == EICOS_OK) rc = fetch_rows(h, s, h->dp.i_s, h->dp.m);
    return rc;
}

int eicos_batch_solution_device(eicos_batch *h, const double **dx, size_t *stride) {
    if (!h || !dx || !stride) return fail(EICOS_E_INVALID, "NULL argument");
    *dx = h->d_inst + h->dp.i_x; *stride = h->dp.inst_stride;
    return EICOS_OK;
}

int eicos_batch_kernel_build(eicos_batch *h) {
    if (!h) return fail(EICOS_E_INVALID, "NULL handle");
    return h->ldsres ? 1 : (h->w2 ? 2 : (h->ubl ? 3 : 0));
}

int eicos_internal_device(const eicos_batch *h) { return h ? h->device : -1; }
// ms from the start of `from`'s most recent solve to the end of `to`'s (two handles on ONE device: the span of a device that holds
// several shards of an eicos_multi); both solves must have completed
int eicos_internal_solve_span_ms(eicos_batch *from, eicos_batch *to, float *ms) {
    if (!from || !to || !ms || from->device != to->device || !from->solve_timed || !to->solve_timed) return fail(EICOS_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(from->device));
    HIP_TRY(hipEventSynchronize(from->ev_s1));
    HIP_TRY(hipEventSynchronize(to->ev_s1));
    HIP_TRY(hipEventElapsedTime(ms, from->ev_s0, to->ev_s1));
    return EICOS_OK;
}

int eicos_batch_dims(eicos_batch *h, eicos_dims *o) {
    if (!h || !o) return fail(EICOS_E_INVALID, "NULL argument");
    const Symbolic &S = h->sym;
    o->n = S.n; o->m = S.m; o->p = S.p; o->l = S.l; o->ncones = S.nc; o->dim_K = S.N; o->nnzA = S.nnzA; o->nnzG = S.nnzG;
    o->nnzK = S.nnzK; o->nnzL = S.nnzL; o->nlevels = S.nlev; o->order_mode = S.order_mode; o->batch = h->batch; o->device = h->device;
    o->factor_pairs = S.npairs;
    o->inst_bytes = h->dp.inst_stride * sizeof(double); o->work_bytes = h->dp.work_stride * sizeof(double);
    o->pattern_bytes = h->pattern_ints * sizeof(int);
    o->threads_per_block = h->threads; o->resident_blocks = h->grid; o->lds_bytes = (int)h->dyn_lds; o->instances_per_block = 1;
    o->lds_resident = h->ldsres; o->factor_path = h->sym.tile; o->cone_order = h->sym.cone_order; o->dual_rhs = h->dp.dual;
    o->arithmetic_profile = h->arith_profile; o->apex_nodes = h->dp.apex_na; o->solo_slices = h->dp.nfs_solo + h->dp.nbs_solo;
    return EICOS_OK;
}

static int elapsed(eicos_batch *h, bool ok, hipEvent_t a, hipEvent_t b, float *ms) {
    if (!h || !ms) return fail(EICOS_E_INVALID, "NULL argument");
    if (!ok) return fail(EICOS_E_INVALID, "nothing timed yet");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(b));
    HIP_TRY(hipEventElapsedTime(ms, a, b));
    return EICOS_OK;
}
int eicos_batch_last_solve_ms(eicos_batch *h, float *ms) { return elapsed(h, h && h->solve_timed, h ? h->ev_s0 : nullptr, h ? h->ev_s1 : nullptr, ms); }
int eicos_batch_last_update_ms(eicos_batch *h, float *ms) { return elapsed(h, h && h->update_timed, h ? h->ev_u0 : nullptr, h ? h->ev_u1 : nullptr, ms); }
// Durations (ms, HIP events on the handle's stream) of the most recent launches, oldest first: which = 0 the solve launches, 1 the
// updateData calls, 2 the span from the start of the updateData call that preceded a solve launch to the end of that solve (one "step").  Returns how many were written (<= cap, <= 64: the ring's depth); waits for the most recent one to finish.
int eicos_batch_ms_history(eicos_batch *h, int which, float *ms, int cap) {
    if (!h || !ms || cap < 0 || which < 0 || which > 2) return fail(EICOS_E_INVALID, "bad argument");
    hipEvent_t (*ring)[2] = which != 1 ? h->ring_s : h->ring_u;
    const long total = which != 1 ? h->n_solve_rec : h->n_update_rec;
    if (which != 1 ? !h->solve_timed : !h->update_timed) return 0;
    const int cnt = (int)std::min<long>(std::min<long>(cap, eicos_batch::EV_RING), total);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(which != 1 ? h->ev_s1 : h->ev_u1));
    for (int i = 0; i < cnt; i++) {
        const long at = (total - cnt + i) % eicos_batch::EV_RING;
        hipEvent_t *slot = ring[at];
        // (which = 2: the step's span.  The update ring turns as fast as the solve ring when updateData and solve alternate -- the caller's step --
        // so the start event a solve slot remembers is still that step's; with several updates per solve the span is NaN or too short: documented)
        if (hipEventElapsedTime(&ms[i], which == 2 ? h->ring_step0[at] : slot[0], slot[1]) != hipSuccess) { (void)hipGetLastError(); ms[i] = NAN; } // (a call that failed half way)
    }
    return cnt;
}

int eicos_debug_factor(eicos_batch *h, int inst, double *Dout, double *Uout) {
    if (!h || inst < 0 || inst >= h->batch) return fail(EICOS_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(launch_debug_factor(h->pslot, h->d_inst, h->d_work, inst, h->threads, h->sym.tile ? h->dyn_lds : 0, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const DevPat &P = h->dp;
    if (h->sym.tile) { // tiles -> the scalar view (D per elimination position, U = L D per CSC entry of L)
        const Symbolic &S = h->sym;
        const TilePlan &TP = h->tiles;
        std::vector<double> Dv((size_t)TP.N16), LR((size_t)TP.nt * 256 + 1);
        HIP_TRY(hipMemcpy(Dv.data(), h->d_work + P.w_D, Dv.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (TP.nt) HIP_TRY(hipMemcpy(LR.data(), h->d_work + P.w_LR, (size_t)TP.nt * 256 * sizeof(double), hipMemcpyDeviceToHost));
        if (Dout) for (int j = 0; j < S.N; j++) Dout[j] = Dv[TP.slot[j]];
        if (Uout) {
            // the diagonal tiles themselves (strictly lower part, row-major) as the factorisation's triangular solves read them
            std::vector<double> Ld((size_t)TP.nb * 256, 0.0);
            HIP_TRY(hipMemcpy(Ld.data(), h->d_work + P.w_DL, Ld.size() * sizeof(double), hipMemcpyDeviceToHost));
            std::vector<int> colj(S.nnzL);
            for (int j = 0; j < S.N; j++) for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) colj[e] = j;
            std::vector<double> ub;
            if (S.tile == 2) { // hybrid: the columns below the top block are in the scalar backward slots
                ub.resize((size_t)P.nUB + 1);
                HIP_TRY(hipMemcpy(ub.data(), h->d_work + P.w_UB, (size_t)P.nUB * sizeof(double), hipMemcpyDeviceToHost));
            }
            for (int e = 0; e < S.nnzL; e++) {
                if (S.tile == 2 && colj[e] < S.n0) { Uout[e] = ub[h->posB[e]]; continue; }
                const int rr = TP.Le_rc[e] >> 4, cc = TP.Le_rc[e] & 15;
                const double lv = TP.Le_tile[e] >= 0 ? LR[(size_t)TP.Le_tile[e] * 256 + tile_res(rr, cc)] : Ld[(size_t)(-1 - TP.Le_tile[e]) * 256 + rr * 16 + cc];
                Uout[e] = lv * Dv[TP.slot[colj[e]]];
            }
        }
        return EICOS_OK;
    }
    if (Dout) HIP_TRY(hipMemcpy(Dout, h->d_work + h->dp.w_D, (size_t)h->sym.N * sizeof(double), hipMemcpyDeviceToHost));
    if (Uout) {
        std::vector<double> ub((size_t)h->ub_len + 1);
        HIP_TRY(hipMemcpy(ub.data(), h->d_work + h->dp.w_UB, (size_t)h->ub_len * sizeof(double), hipMemcpyDeviceToHost));
        for (int e = 0; e < h->dp.nnzL; e++) Uout[e] = ub[h->posB[e]];
    }
    return EICOS_OK;
}

int eicos_debug_kkt(eicos_batch *h, int inst, int *rows, int *cols, double *vals) {
    if (!h || inst < 0 || inst >= h->batch) return fail(EICOS_E_INVALID, "bad argument");
    const Symbolic &S = h->sym;
    const DevPat &D = h->dp;
    if (rows) std::copy(S.K_row.begin(), S.K_row.end(), rows);
    if (cols) std::copy(S.K_col.begin(), S.K_col.end(), cols);
    if (vals) {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipStreamSynchronize(h->stream));
        std::vector<double> slab(D.inst_stride);
        HIP_TRY(hipMemcpy(slab.data(), h->d_inst + (size_t)inst * D.inst_stride, D.inst_stride * sizeof(double), hipMemcpyDeviceToHost));
        for (int e = 0; e < S.nnzK; e++) {
            int off;
            switch (S.K_kind[e]) { // same sources as the factor's value stream (srcoff in eicos_batch_create)
            case SRC_A: off = D.i_Av + S.K_src[e]; break;
            case SRC_G: off = D.i_Gv + S.K_src[e]; break;
            case SRC_V: off = D.i_Vv + S.K_src[e]; break;
            case SRC_POSDELTA: off = D.i_cst + 0; break;
            case SRC_NEGDELTA: off = D.i_cst + 1; break;
            default: off = D.i_cst + 2; break;
            }
            vals[e] = slab[off];
        }
    }
    return EICOS_OK;
}

int eicos_debug_scalings(eicos_batch *h, int inst, const double *s, const double *z, double *V, int *ran) {
    if (!h || inst < 0 || inst >= h->batch || !s || !z) return fail(EICOS_E_INVALID, "bad argument");
    const DevPat &D = h->dp;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!h->d_flag) HIP_TRY(hipMalloc(&h->d_flag, 16 * sizeof(int)));
    double *I = h->d_inst + (size_t)inst * D.inst_stride;
    if (D.m > 0) {
        HIP_TRY(hipMemcpy(I + D.i_s, s, (size_t)D.m * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(I + D.i_z, z, (size_t)D.m * sizeof(double), hipMemcpyHostToDevice));
    }
    HIP_TRY(launch_debug_scalings(h->pslot, h->d_inst, h->d_work, inst, h->d_flag, h->threads, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int ok = 0;
    HIP_TRY(hipMemcpy(&ok, h->d_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (ran) *ran = ok;
    if (V && D.nV > 0) HIP_TRY(hipMemcpy(V, I + D.i_Vv, (size_t)D.nV * sizeof(double), hipMemcpyDeviceToHost));
    return EICOS_OK;
}

int eicos_debug_trace(eicos_batch *h, int inst, double *out) {
    if (!h || !out || inst < 0 || inst >= h->batch) return fail(EICOS_E_INVALID, "bad argument");
    if (h->batch > h->grid) return fail(EICOS_E_INVALID, "trace is per workspace slot: needs batch <= resident instances");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int slot = inst;
    if (h->last_ordered) { // the launch took the instances in longest-first order: slot q holds instance order[q]
        std::vector<int> ord(h->batch);
        HIP_TRY(hipMemcpy(ord.data(), h->d_queue + 16, (size_t)h->batch * sizeof(int), hipMemcpyDeviceToHost));
        slot = (int)(std::find(ord.begin(), ord.end(), inst) - ord.begin());
        if (slot >= h->batch) return fail(EICOS_E_INVALID, "instance not found in the launch order");
    }
    HIP_TRY(hipMemcpy(out, h->d_work + (size_t)slot * h->dp.work_stride + h->dp.w_trace,
                      (size_t)TRACE_ROWS * TRACE_COLS * sizeof(double), hipMemcpyDeviceToHost));
    return EICOS_OK;
}

int eicos_debug_pattern(eicos_batch *h, int *perm, int *Lp, int *Li) {
    if (!h) return fail(EICOS_E_INVALID, "NULL handle");
    const Symbolic &S = h->sym;
    if (perm) std::copy(S.perm.begin(), S.perm.end(), perm);
    if (Lp) std::copy(S.Lp.begin(), S.Lp.end(), Lp);
    if (Li) std::copy(S.Li.begin(), S.Li.end(), Li);
    return EICOS_OK;
}


// Host-only self check of the symbolic analysis (no GPU needed): random quasi-definite values
// on the KKT pattern, the factor program and the level-scheduled gather solves executed
// sequentially exactly as the kernels index them, then || K x - b ||_inf / || b ||_inf.
double eicos_debug_host_check(int n, int m, int p, int ncones, const int *q, const int *Gjc, const int *Gir,
                              const int *Ajc, const int *Air, unsigned seed, int order_mode, int *stats /*[8] or NULL*/) {
    try {
        ProblemPattern P;
        P.n = n; P.m = m; P.p = p; P.nc = ncones; P.q.assign(q, q + ncones);
        if (Gjc && Gir) { P.Gjc.assign(Gjc, Gjc + n + 1); P.Gir.assign(Gir, Gir + Gjc[n]); } else { P.Gjc.assign(n + 1, 0); P.m = 0; P.nc = 0; P.q.clear(); }
        if (Ajc && Air) { P.Ajc.assign(Ajc, Ajc + n + 1); P.Air.assign(Air, Air + Ajc[n]); } else { P.Ajc.assign(n + 1, 0); P.p = 0; }
        Symbolic S = analyze(P, order_mode, 0); // the scalar programs (the tile and hybrid paths have their own checks)
        const int N = S.N;
        if (stats) { stats[0] = N; stats[1] = S.nnzK; stats[2] = S.nnzL; stats[3] = S.nlev; stats[4] = (int)std::min<int64_t>(S.npairs, 2147483647); stats[5] = S.order_mode; stats[6] = S.max_row_len; stats[7] = S.max_col_len; }
        // permutation sanity
        std::vector<char> seen(N, 0);
        for (int k = 0; k < N; k++) { if (S.perm[k] < 0 || S.perm[k] >= N || seen[S.perm[k]]) return -1.0; seen[S.perm[k]] = 1; }
        unsigned long long st = seed * 2654435761ull + 12345;
        auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return ((st >> 11) & 0xFFFFFFFFFFFFull) / (double)(1ull << 48); };
        std::vector<double> Kv(S.nnzK);
        for (int e = 0; e < S.nnzK; e++) {
            const int r = S.K_row[e], c = S.K_col[e];
            if (r == c) Kv[e] = (r < S.n ? 1.0 : -1.0) * (4.0 + rnd());
            else Kv[e] = 0.2 * (rnd() - 0.5);
        }
        // values per L entry / diagonal in permuted order
        std::vector<double> U(S.nnzL, 0.0), Ur(S.nnzL, 0.0), D(N, 0.0), invD(N, 0.0), Lv(S.nnzL, 0.0), Dv(N, 0.0);
        for (int e = 0; e < S.nnzK; e++) {
            const int a = S.iperm[S.K_row[e]], b = S.iperm[S.K_col[e]];
            if (a == b) Dv[a] = Kv[e];
            else {
                const int i = std::max(a, b), j = std::min(a, b);
                auto it = std::lower_bound(S.Li.begin() + S.Lp[j], S.Li.begin() + S.Lp[j + 1], i);
                Lv[it - S.Li.begin()] = Kv[e];
            }
        }
        for (int v = 0; v < S.nlev; v++)
            for (int t = S.ftask_ptr[v]; t < S.ftask_ptr[v + 1]; t++) {
                const int tgt = S.ftask[t];
                double s = 0;
                for (int64_t k = S.tp[tgt]; k < S.tp[tgt + 1]; k++) s += U[S.pa[k]] * U[S.pb[k]] * invD[S.pk[k]];
                if (tgt < N) { D[tgt] = Dv[tgt] - s; invD[tgt] = 1.0 / D[tgt]; }
                else { const int e = tgt - N; U[e] = Lv[e] - s; Ur[S.Cpos[e]] = U[e]; }
            }
        std::vector<double> rhs(N), x(N);
        for (int i = 0; i < N; i++) rhs[i] = rnd() - 0.5;
        // the two sweeps exactly as the kernel walks its sliced-ELL plans (lane by lane)
        double plan_err = 0;
        for (int T : {128, 256, 512}) {
            TriPlan pf = build_tri_plan(S, T, true), pb = build_tri_plan(S, T, false);
            std::vector<double> UF(pf.ulen, 0.0), UB(pb.ulen, 0.0), ws(scalar_npad(N), 0.0);
            { // numeric factorisation through the sliced-ELL factor plan, lane by lane as the kernel does it
                FactorPlan px = build_factor_plan(S, T, pb.pos, pb.slots, pf.pos, pf.slots);
                if (getenv("EICOS_PLAN_STATS")) { // developer aid: shape of the three programs for this workgroup size
                    auto stat = [&](const char *nm, const std::vector<SliceMeta> &sl, int slots) {
                        int lev = 0, kmax = 0; long lanes = 0, kl = 0;
                        for (const SliceMeta &m : sl) { lev += m.newlev & 1; kmax = std::max(kmax, m.K); lanes += (long)m.cnt << m.lg; kl += (long)m.K; }
                        fprintf(stderr, "[plan T=%d] %-8s slices %zu levels %d slots %d sum(K) %ld maxK %d active-lane slices %.2f\n", T, nm,
                                sl.size(), lev, slots, kl, kmax, (double)lanes / T);
                    };
                    stat("forward", pf.sl, pf.slots); stat("backward", pb.sl, pb.slots); stat("factor", px.sl, px.slots);
                    fprintf(stderr, "[plan T=%d] factor targets %zu pairs %lld\n", T, px.target.size(), (long long)S.tp.back());
                    if (T == 512 || (T == 256 && getenv("EICOS_PLAN_STATS")[0] == '2')) {
                        fprintf(stderr, "[plan] level sizes:");
                        for (int v = 0; v < S.nlev; v++) fprintf(stderr, " %d", S.lev_ptr[v + 1] - S.lev_ptr[v]);
                        fprintf(stderr, "\n[plan T=%d] backward slices (lanes x K):", T);
                        for (const SliceMeta &m : pb.sl) fprintf(stderr, " %d%sx%d", m.cnt << m.lg, (m.newlev & 1) ? "*" : "", m.K);
                        fprintf(stderr, "\n[plan T=%d] forward slices (lanes x K):", T);
                        for (const SliceMeta &m : pf.sl) fprintf(stderr, " %d%sx%d", m.cnt << m.lg, (m.newlev & 1) ? "*" : "", m.K);
                        fprintf(stderr, "\n");
                    }
                }
                std::vector<double> D2(N, 0.0), iD2(N, 0.0);
                std::vector<int> colof(S.nnzL);
                for (int j = 0; j < N; j++) for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) colof[e] = j;
                size_t s0 = 0;
                std::vector<double> carry;
                while (s0 < px.sl.size()) { // one level at a time: phase A (U, D), then phase B (L = U / D[col])
                    size_t s1 = s0 + 1;
                    while (s1 < px.sl.size() && !(px.sl[s1].newlev & 1)) s1++;
                    for (size_t si = s0; si < s1; si++) {
                        const SliceMeta &m = px.sl[si];
                        const int g = 1 << m.lg, lanes = m.cnt * g;
                        if (lanes > T) throw std::logic_error("factor slice wider than the workgroup");
                        if (m.K > ELL_KMAX) throw std::logic_error("factor slice deeper than the prefetch depth");
                        if (!m.cont) carry.assign(m.cnt, 0.0);
                        for (int r = 0; r < m.cnt; r++) {
                            double acc = carry[r];
                            for (int q = 0; q < g; q++)
                                for (int kk = 0; kk < m.K; kk++) { const int slot = m.off + kk * lanes + r * g + q; acc += UB[px.pa[slot]] * UF[px.pb[slot]]; }
                            if (m.more) { carry[r] = acc; continue; } // sub-slices of one set of targets accumulate
                            const int tgt = px.target[m.row0 + r];
                            if (tgt < N) { D2[tgt] = Dv[tgt] - acc; iD2[tgt] = 1.0 / D2[tgt]; }
                            else { const int e = tgt - N; UB[pb.pos[e]] = Lv[e] - acc; }
                        }
                    }
                    for (size_t si = s0; si < s1; si++)
                        for (int r = 0; r < px.sl[si].cnt && !px.sl[si].more; r++) {
                            const int tgt = px.target[px.sl[si].row0 + r];
                            if (tgt >= N) { const int e = tgt - N; UF[pf.pos[e]] = UB[pb.pos[e]] * iD2[colof[e]]; }
                        }
                    s0 = s1;
                }
                for (int e = 0; e < S.nnzL; e++) plan_err = std::max(plan_err, std::fabs(UB[pb.pos[e]] - U[e]) / (1.0 + std::fabs(U[e])));
                for (int jn = 0; jn < N; jn++) plan_err = std::max(plan_err, std::fabs(D2[jn] - D[jn]) / (1.0 + std::fabs(D[jn])));
            }
            for (int i = 0; i < N; i++) ws[i] = rhs[S.perm[i]];
            auto sweep = [&](const TriPlan &pl, const std::vector<double> &val, bool fwd) {
                for (const SliceMeta &m : pl.sl) {
                    const int g = 1 << m.lg, lanes = m.cnt * g;
                    if (lanes > T) throw std::logic_error("slice wider than the workgroup");
                    std::vector<double> acc(m.cnt, 0.0);
                    for (int t = 0; t < lanes; t++)
                        for (int kk = 0; kk < m.K; kk++) { const int slot = m.off + kk * lanes + t; acc[t / g] += val[slot] * ws[pl.idx[slot]]; }
                    if (m.K > ELL_KMAX) throw std::logic_error("slice deeper than the prefetch depth");
                    for (int r = 0; r < m.cnt; r++) {
                        const int i = m.row0 + r;
                        ws[i] = (fwd || m.more) ? ws[i] - acc[r] : (ws[i] - acc[r]) * invD[i]; // L y = b ; x = (y - U' x) / D
                    }
                }
            };
            sweep(pf, UF, true);
            if (S.apex0 >= 0) { // the dense apex as apex_solve walks it: a column of the block per forward step, a row per backward step
                const int n0 = S.apex0, na = N - n0;
                for (int q_ = 0; q_ < pf.split_n; q_++) { ws[pf.split_row] += ws[pf.split_slot0 + q_]; ws[pf.split_slot0 + q_] = 0.; } // the parts of the split row
                if (na > APEX_MAX || pf.n_ext % TRI_DEPTH) throw std::logic_error("apex: bad shape");
                for (int k = 0; k < na; k++) for (int i = k + 1; i < na; i++) ws[n0 + i] -= UF[pf.apex_base + apex_img_at(i, k)] * ws[n0 + k];
                for (int i = na - 1; i >= 0; i--) {
                    ws[n0 + i] *= invD[n0 + i];
                    for (int k = 0; k < i; k++) ws[n0 + k] -= UB[pb.apex_base + apex_img_at(i, k)] * ws[n0 + i];
                }
            }
            sweep(pb, UB, false);
            std::vector<double> xt(N);
            for (int j = 0; j < N; j++) xt[S.perm[j]] = ws[j];
            if (T == 128) x = xt;
            for (int j = 0; j < N; j++) plan_err = std::max(plan_err, std::fabs(xt[j] - x[j]));
        }
        if (plan_err > 1e-9) return -3.0;
        std::vector<double> r(rhs);
        for (int e = 0; e < S.nnzK; e++) {
            const int a = S.K_row[e], b = S.K_col[e];
            r[a] -= Kv[e] * x[b];
            if (a != b) r[b] -= Kv[e] * x[a];
        }
        double nr = 0, nb = 0;
        for (int i = 0; i < N; i++) { nr = std::max(nr, std::fabs(r[i])); nb = std::max(nb, std::fabs(rhs[i])); }
        return N ? nr / nb : 0.0;
    } catch (const std::exception &e) { g_err = e.what(); return -2.0; }
}


// Host-only self check of the TILE path (no GPU needed): random quasi-definite values on the KKT pattern, the block
// factorisation and the two tile sweeps executed exactly as the kernels index them (tile image, pair lists, finalise
// lists, CSR / CSC tile views, column- / row-major tile copies, inverse diagonal tiles), then ||K x - b|| / ||b||.
// stats[8] = {dim_K, nnzK, nnzL, block levels, tile pairs, order_mode, blocks, off-diagonal tiles}.
static double host_check_tiles_impl(int n, int m, int p, int ncones, const int *q, const int *Gjc, const int *Gir,
                                    const int *Ajc, const int *Air, unsigned seed, int order_mode, int *stats, int mode) {
    try {
        ProblemPattern P;
        P.n = n; P.m = m; P.p = p; P.nc = ncones; P.q.assign(q, q + ncones);
        if (Gjc && Gir) { P.Gjc.assign(Gjc, Gjc + n + 1); P.Gir.assign(Gir, Gir + Gjc[n]); } else { P.Gjc.assign(n + 1, 0); P.m = 0; P.nc = 0; P.q.clear(); }
        if (Ajc && Air) { P.Ajc.assign(Ajc, Ajc + n + 1); P.Air.assign(Air, Air + Ajc[n]); } else { P.Ajc.assign(n + 1, 0); P.p = 0; }
        Symbolic S = analyze(P, order_mode, mode);
        if (S.tile != mode) return -10.0; // (hybrid requested, but the schedule has no tail worth handing to the tile path)
        TilePlan TP = build_tile_plan(S);
        const int n0 = TP.n0;
        if (getenv("EICOS_PLAN_STATS")) (void)build_tile_sweeps(TP, 8, TILE_PF);
        const int N = S.N, nb = TP.nb, nt = TP.nt, N16 = TP.N16;
        if (stats) { stats[0] = N; stats[1] = S.nnzK; stats[2] = S.nnzL; stats[3] = TP.nblev; stats[4] = (int)std::min<int64_t>(TP.npairs, 2147483647); stats[5] = S.order_mode; stats[6] = nb; stats[7] = nt; }
        unsigned long long st = seed * 2654435761ull + 12345;
        auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return ((st >> 11) & 0xFFFFFFFFFFFFull) / (double)(1ull << 48); };
        std::vector<double> Kv(S.nnzK);
        for (int e = 0; e < S.nnzK; e++) {
            const int r = S.K_row[e], c = S.K_col[e];
            if (r == c) Kv[e] = (r < S.n ? 1.0 : -1.0) * (4.0 + rnd());
            else Kv[e] = 0.2 * (rnd() - 0.5);
        }
        // values of P K P' per entry of L / per diagonal
        std::vector<double> Lv(S.nnzL, 0.0), Dv(N, 0.0);
        for (int e = 0; e < S.nnzK; e++) {
            const int a = S.iperm[S.K_row[e]], b = S.iperm[S.K_col[e]];
            if (a == b) Dv[a] = Kv[e];
            else {
                const int i = std::max(a, b), j = std::min(a, b);
                auto it = std::lower_bound(S.Li.begin() + S.Lp[j], S.Li.begin() + S.Lp[j + 1], i);
                if (it == S.Li.begin() + S.Lp[j + 1] || *it != i) return -4.0;
                Lv[it - S.Li.begin()] = Kv[e];
            }
        }
        // the tile image: pure tile mode -- P K P' itself (what the solve prologue scatters from the instance slab);
        // hybrid -- written by the scalar factor program below (K minus the updates from the columns under the top block)
        std::vector<double> img((size_t)(nb + nt) * 256, 0.0);
        for (int d : TP.pad_img) img[d] = 1.0;
        std::vector<double> LC((size_t)nt * 256 + 1, 0.0), LR((size_t)nt * 256 + 1, 0.0), DC((size_t)nb * 256, 0.0), DR((size_t)nb * 256, 0.0), Dall(N16 + 1, 0.0), invDall(N16 + 1, 0.0);
        double *D = Dall.data() + n0, *invD = invDall.data() + n0; // block-relative views (DevPat::tl_base)
        constexpr int TW = 256; // workgroup size the scalar plans of the hybrid are laid out for in this check
        TriPlan pf, pb;
        std::vector<double> UF, UB;
        if (mode == 1) {
            for (int j = 0; j < N; j++) img[TP.D_img[j]] = Dv[j];
            for (int e = 0; e < S.nnzL; e++) img[TP.Le_img[e]] = Lv[e];
        } else {
            // ---- hybrid: the scalar factor program, lane by lane as stage_factor walks it (api.cpp's destination codes) ----
            pf = build_tri_plan(S, TW, true); pb = build_tri_plan(S, TW, false);
            FactorPlan px = build_factor_plan(S, TW, pb.pos, pb.slots, pf.pos, pf.slots);
            UF.assign(pf.slots + 1, 0.0); UB.assign(pb.slots + 1, 0.0);
            std::vector<int> colof(S.nnzL);
            for (int j = 0; j < N; j++) for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) colof[e] = j;
            size_t s0 = 0;
            std::vector<double> carry;
            while (s0 < px.sl.size()) {
                size_t s1 = s0 + 1;
                while (s1 < px.sl.size() && !(px.sl[s1].newlev & 1)) s1++;
                for (size_t si = s0; si < s1; si++) {
                    const SliceMeta &sm = px.sl[si];
                    const int g = 1 << sm.lg, lanes = sm.cnt * g;
                    if (lanes > TW || sm.K > ELL_KMAX) return -6.0;
                    if (!sm.cont) carry.assign(sm.cnt, 0.0);
                    for (int r = 0; r < sm.cnt; r++) {
                        double acc = carry[r];
                        for (int qq = 0; qq < g; qq++)
                            for (int kk = 0; kk < sm.K; kk++) { const int slot = sm.off + kk * lanes + r * g + qq; acc += UB[px.pa[slot]] * UF[px.pb[slot]]; }
                        if (sm.more) { carry[r] = acc; continue; }
                        const int tgt = px.target[sm.row0 + r];
                        if (tgt < N) {
                            if (tgt >= n0) img[TP.D_img[tgt]] = Dv[tgt] - acc;                         // top block: image
                            else { Dall[tgt] = Dv[tgt] - acc; invDall[tgt] = 1.0 / Dall[tgt]; }
                        } else {
                            const int e = tgt - N;
                            if (colof[e] >= n0) img[TP.Le_img[e]] = Lv[e] - acc;                       // top block: image
                            else UB[pb.pos[e]] = Lv[e] - acc;
                        }
                    }
                }
                for (size_t si = s0; si < s1; si++)
                    for (int r = 0; r < px.sl[si].cnt && !px.sl[si].more; r++) {
                        const int tgt = px.target[px.sl[si].row0 + r];
                        if (tgt >= N && colof[tgt - N] < n0) { const int e = tgt - N; UF[pf.pos[e]] = UB[pb.pos[e]] * invDall[colof[e]]; }
                    }
                s0 = s1;
            }
        }
        for (int v = 0; v < TP.nblev; v++) {
            for (int qi = TP.tgt_lev_ptr[v]; qi < TP.tgt_lev_ptr[v + 1]; qi++) { // phase 1
                const int tg = TP.tgt[qi];
                double Tt[16][16];
                for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) Tt[r][c] = img[(size_t)tg * 256 + tile_res(r, c)];
                for (int e = TP.tp_ptr[qi]; e < TP.tp_ptr[qi + 1]; e++) {
                    const double *A = LC.data() + (size_t)TP.pa[e] * 256, *B = LC.data() + (size_t)TP.pb[e] * 256, *d = D + TP.pk[e] * 16;
                    for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) { double sacc = 0; for (int k = 0; k < 16; k++) sacc += A[tile_op(r, k)] * (B[tile_op(c, k)] * d[k]); Tt[r][c] -= sacc; }
                }
                if (tg >= nb) { double *o = LC.data() + (size_t)(tg - nb) * 256; for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) o[tile_op(r, c)] = Tt[r][c]; continue; }
                const int J = tg;
                for (int j = 0; j < 16; j++) {
                    const double dj = Tt[j][j];
                    for (int c = j + 1; c < 16; c++) for (int r = c; r < 16; r++) Tt[r][c] -= (Tt[r][j] / dj) * Tt[c][j];
                    for (int r = j + 1; r < 16; r++) Tt[r][j] /= dj;
                }
                for (int c = 0; c < 16; c++) {
                    D[J * 16 + c] = Tt[c][c]; invD[J * 16 + c] = 1.0 / Tt[c][c];
                    double mc[16];
                    for (int r = 0; r < 16; r++) { double sacc = (r == c) ? 1.0 : 0.0; for (int k = 0; k < r; k++) sacc -= Tt[r][k] * mc[k]; mc[r] = (r < c) ? 0.0 : sacc; }
                    for (int r = 0; r < 16; r++) { DC[(size_t)J * 256 + tile_op(r, c)] = mc[r]; DR[(size_t)J * 256 + tile_res(r, c)] = mc[r]; }
                }
            }
            for (int qi = TP.fin_lev_ptr[v]; qi < TP.fin_lev_ptr[v + 1]; qi++) { // phase 2
                const int t = TP.fin[qi], J = TP.t_col[t];
                double Tt[16][16];
                for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) {
                    double sacc = 0;
                    for (int k = 0; k < 16; k++) sacc += LC[(size_t)t * 256 + tile_op(r, k)] * DC[(size_t)J * 256 + tile_op(c, k)]; // T[r][k] * Linv[c][k]
                    Tt[r][c] = sacc * invD[J * 16 + c];
                }
                for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) { LR[(size_t)t * 256 + tile_res(r, c)] = Tt[r][c]; LC[(size_t)t * 256 + tile_op(r, c)] = Tt[r][c]; }
            }
        }
        std::vector<double> rhs(N), wsall(N16 + 17, 0.0);
        for (int i = 0; i < N; i++) rhs[i] = rnd() - 0.5;
        for (int i = 0; i < N; i++) wsall[TP.slot[i]] = rhs[S.perm[i]];
        double *ws = wsall.data() + n0; // block-relative view
        auto ell_sweep = [&](const TriPlan &pl, const std::vector<double> &val, bool fwd) { // the scalar sweeps, slice by slice
            for (const SliceMeta &sm : pl.sl) {
                const int g = 1 << sm.lg, lanes = sm.cnt * g;
                std::vector<double> acc(sm.cnt, 0.0);
                for (int t = 0; t < lanes; t++)
                    for (int kk = 0; kk < sm.K; kk++) { const int slot = sm.off + kk * lanes + t; acc[t / g] += val[slot] * wsall[pl.idx[slot]]; }
                for (int r = 0; r < sm.cnt; r++) {
                    const int i = sm.row0 + r;
                    wsall[i] = (fwd || sm.more) ? wsall[i] - acc[r] : (wsall[i] - acc[r]) * invDall[i];
                }
            }
        };
        if (mode == 2) ell_sweep(pf, UF, true); // levels under the top block, then its rows against them (the plan's extra level)
        auto block = [&](int B, const std::vector<int> &tiles_of, int e0, int e1, const std::vector<double> &val, const std::vector<double> &dia, bool fwd) {
            double acc[16] = {0};
            for (int e = e0; e < e1; e++) {
                const int t = fwd ? tiles_of[e] : e, vb = fwd ? TP.t_col[t] : TP.t_row[t];
                // forward: LC in operand order, out[r] += L[r][k] y[k]; backward: LR in result order read as the transposed operand, out[c] += L[r][c] x[r]
                for (int c = 0; c < 16; c++) for (int k = 0; k < 16; k++) acc[c] += val[(size_t)t * 256 + tile_op(c, k)] * ws[vb * 16 + k];
            }
            double r[16], o[16] = {0};
            for (int c = 0; c < 16; c++) r[c] = (fwd ? ws[B * 16 + c] : ws[B * 16 + c] * invD[B * 16 + c]) - acc[c];
            if (TP.ident[B]) { for (int c = 0; c < 16; c++) ws[B * 16 + c] = r[c]; return; } // identity diagonal tile: skipped by the kernel
            for (int c = 0; c < 16; c++) for (int k = 0; k < 16; k++) o[c] += dia[(size_t)B * 256 + tile_op(c, k)] * r[k];
            for (int c = 0; c < 16; c++) ws[B * 16 + c] = o[c];
        };
        for (int v = 0; v < TP.nblev; v++) for (int B = TP.blev_ptr[v]; B < TP.blev_ptr[v + 1]; B++) block(B, TP.tr_tile, TP.tr_ptr[B], TP.tr_ptr[B + 1], LC, DC, true);
        for (int v = TP.nblev - 1; v >= 0; v--) for (int B = TP.blev_ptr[v]; B < TP.blev_ptr[v + 1]; B++) block(B, TP.tr_tile, TP.tc_ptr[B], TP.tc_ptr[B + 1], LR, DR, false);
        if (mode == 2) ell_sweep(pb, UB, false);
        std::vector<double> x(N);
        for (int j = 0; j < N; j++) x[S.perm[j]] = wsall[TP.slot[j]];
        for (int s_ = 0; s_ < N16; s_++) { bool real = false; for (int j = 0; j < N && !real; j++) real = TP.slot[j] == s_; if (!real && wsall[s_] != 0.0) return -5.0; if (N > 4000) break; } // padding slots stay 0
        std::vector<double> r(rhs);
        for (int e = 0; e < S.nnzK; e++) {
            const int a = S.K_row[e], b = S.K_col[e];
            r[a] -= Kv[e] * x[b];
            if (a != b) r[b] -= Kv[e] * x[a];
        }
        double nr = 0, nbn = 0;
        for (int i = 0; i < N; i++) { nr = std::max(nr, std::fabs(r[i])); nbn = std::max(nbn, std::fabs(rhs[i])); }
        return N ? nr / nbn : 0.0;
    } catch (const std::exception &e) { g_err = e.what(); return -2.0; }
}


// Host-only: the elimination order and block partition the tile path would use (perm[new] = KKT index, blk_ptr[nblk + 1]); returns the
// number of blocks, < 0 on error.  Development aid for ordering studies (tools/dev), no GPU needed.
int eicos_debug_host_tile_order(int n, int m, int p, int ncones, const int *q, const int *Gjc, const int *Gir,
                                const int *Ajc, const int *Air, int order_mode, int *perm, int *blk_ptr, int *stats, int *Lp, int *Li) {
    try {
        ProblemPattern P;
        P.n = n; P.m = m; P.p = p; P.nc = ncones; P.q.assign(q, q + ncones);
        if (Gjc && Gir) { P.Gjc.assign(Gjc, Gjc + n + 1); P.Gir.assign(Gir, Gir + Gjc[n]); } else { P.Gjc.assign(n + 1, 0); P.m = 0; P.nc = 0; P.q.clear(); }
        if (Ajc && Air) { P.Ajc.assign(Ajc, Ajc + n + 1); P.Air.assign(Air, Air + Ajc[n]); } else { P.Ajc.assign(n + 1, 0); P.p = 0; }
        Symbolic S = analyze(P, order_mode, 1);
        TilePlan TP = build_tile_plan(S);
        if (perm) std::copy(S.perm.begin(), S.perm.end(), perm);
        if (blk_ptr) std::copy(S.blk_ptr.begin(), S.blk_ptr.end(), blk_ptr);
        if (Lp) std::copy(S.Lp.begin(), S.Lp.end(), Lp);
        if (Li) std::copy(S.Li.begin(), S.Li.end(), Li);
        if (stats) { stats[0] = S.N; stats[1] = S.nnzL; stats[2] = TP.nb; stats[3] = TP.nt; stats[4] = TP.nblev; stats[5] = (int)std::min<int64_t>(TP.npairs, 2147483647); stats[6] = S.order_mode; stats[7] = S.cone_order; }
        return S.nblk;
    } catch (const std::exception &e) { g_err = e.what(); return -2; }
}

double eicos_debug_host_check_tiles(int n, int m, int p, int ncones, const int *q, const int *Gjc, const int *Gir,
                                    const int *Ajc, const int *Air, unsigned seed, int order_mode, int *stats) {
    return host_check_tiles_impl(n, m, p, ncones, q, Gjc, Gir, Ajc, Air, seed, order_mode, stats, 1);
}
// hybrid: scalar programs under the cut + tile path on the top block; returns -10 when the pattern does not qualify
double eicos_debug_host_check_hybrid(int n, int m, int p, int ncones, const int *q, const int *Gjc, const int *Gir,
                                     const int *Ajc, const int *Air, unsigned seed, int order_mode, int *stats) {
    return host_check_tiles_impl(n, m, p, ncones, q, Gjc, Gir, Ajc, Air, seed, order_mode, stats, 2);
}

} // extern "C"
