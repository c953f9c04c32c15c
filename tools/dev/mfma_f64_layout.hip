// Lane <-> element maps of v_mfma_f64_16x16x4_f64 on gfx950, checked against a scalar product (developer aid).
//   hipcc --offload-arch=gfx950 -O2 tools/dev/mfma_f64_layout.hip -o /tmp/mfma_layout && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
// A[16][16] row-major (r, k), B[16][16] row-major (k, c): C = A * B with four K=4 steps
__global__ void k(const double *A, const double *B, double *C) {
    const int l = threadIdx.x;
    d4 acc = {0., 0., 0., 0.};
    for (int s = 0; s < 4; s++) {
        const int kk = 4 * s + (l >> 4);
        const double a = A[(l & 15) * 16 + kk]; // assumed: lane holds A[row = l&15][k = l>>4]
        const double b = B[kk * 16 + (l & 15)]; // assumed: lane holds B[k = l>>4][col = l&15]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r]; // assumed: C[row = (l>>4) + 4 reg][col = l&15]
}
int main() {
    std::vector<double> A(256), B(256), C(256), R(256, 0.);
    for (int i = 0; i < 256; i++) { A[i] = (i * 7 % 13) - 6 + 0.25 * (i % 5); B[i] = (i * 11 % 17) - 8 + 0.5 * (i % 3); }
    for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) for (int q = 0; q < 16; q++) R[r * 16 + c] += A[r * 16 + q] * B[q * 16 + c];
    double *dA, *dB, *dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 2048);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 256; i++) err = fmax(err, fabs(C[i] - R[i]));
    printf("mfma_f64_16x16x4 layout check: max |C - ref| = %g  %s\n", err, err == 0. ? "OK" : "MISMATCH");
    return err != 0.;
}
