// Host-callable launchers of the kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include "device_types.hpp"

namespace eicos {
// Fused updateData (eicos_batch_update_solve): when `on`, every workgroup of the solve kernel first runs updateData for the instance it is
// about to solve, reading row `instance` of these [batch][...] arrays (NULL = keep the group; device or pinned host memory), and -- x != NULL
// -- writes the instance's solution to row `instance` of x [batch][n] when it is done.
// STAGED host arrays (pageable memory): the arrays are the handle's pinned staging buffer, which the host fills chunk by chunk WHILE the
// kernel runs -- flags[instance / chunk] == seq once the chunk holding an instance has been copied (pinned, host-written); the workgroup polls
// it before it touches the instance's rows (bounded: after ~5 s it gives up and raises *err).
struct UpdArgs { const double *G, *A, *c, *h, *b; double *x; int on; const unsigned *flags; int chunk; unsigned seq; int *err; };
hipError_t launch_solve(int ps, double *inst, double *work, int B, int *queue, int *order, int grid, int threads, int nlds, int idx16,
                        int order_min, double warm, double dyn_delta, double dyn_eps, size_t dyn_lds, hipStream_t st, const UpdArgs *upd = nullptr);
hipError_t launch_update(int ps, double *inst, int first, int count, const double *Gpr, const double *Apr,
                         const double *c, const double *h, const double *b, double *scratch, int grid, size_t lds_bytes, int vals_in_lds, hipStream_t st);
hipError_t update_set_max_lds();
hipError_t launch_debug_factor(int ps, double *inst, double *work, int i, int threads, size_t dyn_lds, hipStream_t st);
hipError_t launch_debug_scalings(int ps, double *inst, double *work, int i, int *ok, int threads, hipStream_t st);
hipError_t solve_occupancy(int threads, int nlds, int idx16, size_t dyn_lds, int *blocks_per_cu);
hipError_t solve_set_max_lds(int threads, int nlds, int idx16, size_t dyn_lds);
hipError_t upload_pattern(int ps, const DevPat &P);
int max_patterns();
// LDS-resident variant of k_solve (kernels_ldsres.hip = kernels.hip compiled with EICOS_LDSRES): same arguments
namespace ldsres {
hipError_t launch_solve(int ps, double *inst, double *work, int B, int *queue, int *order, int grid, int threads, int nlds, int idx16,
                        int order_min, double warm, double dyn_delta, double dyn_eps, size_t dyn_lds, hipStream_t st, const UpdArgs *upd = nullptr);
hipError_t solve_occupancy(int threads, int nlds, int idx16, size_t dyn_lds, int *blocks_per_cu);
hipError_t solve_set_max_lds(int threads, int nlds, int idx16, size_t dyn_lds);
hipError_t upload_pattern(int ps, const DevPat &P);
} // namespace ldsres
// 256-thread k_solve with the register budget of two waves per SIMD (kernels_w2.hip = kernels.hip compiled with EICOS_W2)
namespace w2 {
hipError_t launch_solve(int ps, double *inst, double *work, int B, int *queue, int *order, int grid, int threads, int nlds, int idx16,
                        int order_min, double warm, double dyn_delta, double dyn_eps, size_t dyn_lds, hipStream_t st, const UpdArgs *upd = nullptr);
hipError_t solve_occupancy(int threads, int nlds, int idx16, size_t dyn_lds, int *blocks_per_cu);
hipError_t solve_set_max_lds(int threads, int nlds, int idx16, size_t dyn_lds);
hipError_t upload_pattern(int ps, const DevPat &P);
} // namespace w2
// the 128- and 512-thread k_solve of the default build, in their own translation units (kernels_t128.hip / kernels_t512.hip = kernels.hip
// compiled with EICOS_TSPLIT): the default namespace keeps the 256-thread one
namespace t128 {
hipError_t launch_solve(int ps, double *inst, double *work, int B, int *queue, int *order, int grid, int threads, int nlds, int idx16,
                        int order_min, double warm, double dyn_delta, double dyn_eps, size_t dyn_lds, hipStream_t st, const UpdArgs *upd = nullptr);
hipError_t solve_occupancy(int threads, int nlds, int idx16, size_t dyn_lds, int *blocks_per_cu);
hipError_t solve_set_max_lds(int threads, int nlds, int idx16, size_t dyn_lds);
hipError_t upload_pattern(int ps, const DevPat &P);
} // namespace t128
namespace t512 {
hipError_t launch_solve(int ps, double *inst, double *work, int B, int *queue, int *order, int grid, int threads, int nlds, int idx16,
                        int order_min, double warm, double dyn_delta, double dyn_eps, size_t dyn_lds, hipStream_t st, const UpdArgs *upd = nullptr);
hipError_t solve_occupancy(int threads, int nlds, int idx16, size_t dyn_lds, int *blocks_per_cu);
hipError_t solve_set_max_lds(int threads, int nlds, int idx16, size_t dyn_lds);
hipError_t upload_pattern(int ps, const DevPat &P);
} // namespace t512
// the 256- / 512-thread k_solve with the factor operand array U resident in LDS (kernels_ubl256.hip / kernels_ubl512.hip = kernels.hip compiled
// with EICOS_UBL): launches of one workgroup per CU whose U fits the idle LDS
namespace ubl256 {
hipError_t launch_solve(int ps, double *inst, double *work, int B, int *queue, int *order, int grid, int threads, int nlds, int idx16,
                        int order_min, double warm, double dyn_delta, double dyn_eps, size_t dyn_lds, hipStream_t st, const UpdArgs *upd = nullptr);
hipError_t solve_occupancy(int threads, int nlds, int idx16, size_t dyn_lds, int *blocks_per_cu);
hipError_t solve_set_max_lds(int threads, int nlds, int idx16, size_t dyn_lds);
hipError_t upload_pattern(int ps, const DevPat &P);
} // namespace ubl256
namespace ubl512 {
hipError_t launch_solve(int ps, double *inst, double *work, int B, int *queue, int *order, int grid, int threads, int nlds, int idx16,
                        int order_min, double warm, double dyn_delta, double dyn_eps, size_t dyn_lds, hipStream_t st, const UpdArgs *upd = nullptr);
hipError_t solve_occupancy(int threads, int nlds, int idx16, size_t dyn_lds, int *blocks_per_cu);
hipError_t solve_set_max_lds(int threads, int nlds, int idx16, size_t dyn_lds);
hipError_t upload_pattern(int ps, const DevPat &P);
} // namespace ubl512
// one k_solve build = these four entry points
struct SolveBuild {
    decltype(&launch_solve) launch;
    decltype(&solve_occupancy) occupancy;
    decltype(&solve_set_max_lds) set_max_lds;
    decltype(&upload_pattern) upload;
};
// the build a handle's solves run: U in LDS (one workgroup per CU, 256 / 512 threads), LDS-resident (128 threads), two-waves-per-SIMD (256 threads),
// or the default one of its workgroup size
inline SolveBuild solve_build(int threads, bool ldsres, bool w2, bool ubl = false) {
    if (ubl && threads == 256) return {ubl256::launch_solve, ubl256::solve_occupancy, ubl256::solve_set_max_lds, ubl256::upload_pattern};
    if (ubl && threads == 512) return {ubl512::launch_solve, ubl512::solve_occupancy, ubl512::solve_set_max_lds, ubl512::upload_pattern};
    if (ldsres) return {ldsres::launch_solve, ldsres::solve_occupancy, ldsres::solve_set_max_lds, ldsres::upload_pattern};
    if (w2) return {w2::launch_solve, w2::solve_occupancy, w2::solve_set_max_lds, w2::upload_pattern};
    if (threads == 128) return {t128::launch_solve, t128::solve_occupancy, t128::solve_set_max_lds, t128::upload_pattern};
    if (threads == 512) return {t512::launch_solve, t512::solve_occupancy, t512::solve_set_max_lds, t512::upload_pattern};
    return {launch_solve, solve_occupancy, solve_set_max_lds, upload_pattern};
}
} // namespace eicos
