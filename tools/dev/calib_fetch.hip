// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths k_solve uses (MI355X_MICROARCH.md: only
// 16 B/lane streaming is calibrated).  Streams a 1 GiB buffer once per kernel with 4, 8 and 16 bytes per lane,
// temporal and nontemporal, plus an 8 B/lane streaming store; compare the counters with 2^30 bytes per kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/dev/calib_fetch.hip -o build_exp/calib_fetch
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib/f -- build_exp/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
template <class V, bool NT> __global__ void k_read(const V *p, size_t n, double *out) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        V v;
        if constexpr (NT) v = __builtin_nontemporal_load(&p[i]); else v = p[i];
        const unsigned *w = reinterpret_cast<const unsigned *>(&v);
        for (unsigned k = 0; k < sizeof(V) / 4; k++) acc += w[k];
    }
    if (acc == 1.2345) out[0] = acc;
}
// Random 8-byte gather (the access shape of the global-vector fallback's ELL gathers): n loads spread by a
// multiplicative hash over the whole buffer, so nearly every load touches its own HBM granule.
__global__ void k_gather8(const double *p, size_t n, size_t mask, double *out) {
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += p[(i * 0x9E3779B97F4A7C15ull >> 20) & mask];
    if (acc == 1.2345) out[0] = acc;
}
__global__ void k_write8(double *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
int main() {
    const size_t bytes = 1ull << 30;
    void *buf; double *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess || hipMemset(buf, 0, bytes) != hipSuccess) return 1;
    hipLaunchKernelGGL((k_read<unsigned, false>), dim3(4096), dim3(256), 0, 0, (const unsigned *)buf, bytes / 4, out);
    hipLaunchKernelGGL((k_read<double, false>), dim3(4096), dim3(256), 0, 0, (const double *)buf, bytes / 8, out);
    hipLaunchKernelGGL((k_read<double, true>), dim3(4096), dim3(256), 0, 0, (const double *)buf, bytes / 8, out);
    hipLaunchKernelGGL((k_read<double2, false>), dim3(4096), dim3(256), 0, 0, (const double2 *)buf, bytes / 16, out);
    // 2^24 gathered loads = 2^27 useful bytes; FETCH_SIZE x 1024 / 2^24 = bytes the counter charges per gathered load
    hipLaunchKernelGGL(k_gather8, dim3(4096), dim3(256), 0, 0, (const double *)buf, (size_t)1 << 24, (bytes / 8) - 1, out);
    hipLaunchKernelGGL(k_write8, dim3(4096), dim3(256), 0, 0, (double *)buf, bytes / 8);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("done: each kernel touches %zu bytes\n", bytes);
    return 0;
}
