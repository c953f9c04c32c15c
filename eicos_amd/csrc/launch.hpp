// Host-callable launchers of the kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace eicos {
hipError_t launch_solve(const DevPat &P, double *inst, double *work, int B, int grid, int threads, hipStream_t st);
hipError_t launch_update(const DevPat &P, double *inst, int first, int count, const double *Gpr, const double *Apr,
                         const double *c, const double *h, const double *b, double *scratch, int grid, hipStream_t st);
hipError_t launch_debug_factor(const DevPat &P, double *inst, double *work, int i, hipStream_t st);
hipError_t solve_occupancy(int threads, int *blocks_per_cu);
} // namespace eicos
