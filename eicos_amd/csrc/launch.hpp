// Host-callable launchers of the kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace eicos {
hipError_t launch_solve(const DevPat *dP, double *inst, double *work, int B, int grid, int threads, size_t dyn_lds,
                        hipStream_t st);
hipError_t launch_update(const DevPat *dP, double *inst, int first, int count, const double *Gpr, const double *Apr,
                         const double *c, const double *h, const double *b, double *scratch, int grid, hipStream_t st);
hipError_t launch_debug_factor(const DevPat *dP, double *inst, double *work, int i, hipStream_t st);
hipError_t solve_occupancy(int threads, size_t dyn_lds, int *blocks_per_cu);
hipError_t solve_set_max_lds(int threads, size_t dyn_lds);
} // namespace eicos
