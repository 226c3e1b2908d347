// Issue-rate microbenchmark for v_mfma_f64_16x16x4_f64 on gfx950: W waves per SIMD, 8 independent accumulators each.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip ; prints TFLOP/s for 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double *out, int iters, double a, double b) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    double x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double *d;
    hipMalloc(&d, 8 * 1024 * 1024 * 8);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps, blocks = 256 * 4;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.0, 2.0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0, 2.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * (threads / 64) * iters * 8.0 * 2048.0;
        printf("waves/SIMD %d: %.2f ms, %.1f TFLOP/s\n", wps, ms, flop / ms * 1e-9);
    }
    return 0;
}
