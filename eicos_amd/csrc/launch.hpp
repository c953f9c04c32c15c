// Host-callable launchers of the kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include "device_types.hpp"

namespace eicos {
hipError_t launch_solve(int ps, double *inst, double *work, int B, int *queue, int *order, int grid, int threads, int nlds, int idx16,
                        int order_min, double warm, double dyn_delta, double dyn_eps, size_t dyn_lds, hipStream_t st);
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
                        int order_min, double warm, double dyn_delta, double dyn_eps, size_t dyn_lds, hipStream_t st);
hipError_t solve_occupancy(int threads, int nlds, int idx16, size_t dyn_lds, int *blocks_per_cu);
hipError_t solve_set_max_lds(int threads, int nlds, int idx16, size_t dyn_lds);
hipError_t upload_pattern(int ps, const DevPat &P);
} // namespace ldsres
// 256-thread k_solve with the register budget of two waves per SIMD (kernels_w2.hip = kernels.hip compiled with EICOS_W2)
namespace w2 {
hipError_t launch_solve(int ps, double *inst, double *work, int B, int *queue, int *order, int grid, int threads, int nlds, int idx16,
                        int order_min, double warm, double dyn_delta, double dyn_eps, size_t dyn_lds, hipStream_t st);
hipError_t solve_occupancy(int threads, int nlds, int idx16, size_t dyn_lds, int *blocks_per_cu);
hipError_t solve_set_max_lds(int threads, int nlds, int idx16, size_t dyn_lds);
hipError_t upload_pattern(int ps, const DevPat &P);
} // namespace w2
} // namespace eicos
