// HBM read-rate microbenchmark for the access pattern of LS-fast step 2 (fft_rows_power_kernel) on gfx950.
//   linear : every workgroup reads one contiguous 192 KB block (3 x 64 KB)
//   tiled  : workgroup (x, target) reads, for each of 3 grids and 32 column tiles, RT = 8 rows x 256 B = 2 KB contiguous,
//            tiles 256 KB apart — exactly what step 2 reads from the [c / 16][k1][c % 16] intermediate
// 128 of 256 threads issue 32 x 16-byte loads each (as phase 1 of the kernel does), values are summed and one double per
// thread is written.  Build: hipcc --offload-arch=gfx950 -O3 -o hbm_read_patterns hbm_read_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int N1 = 1024, N2 = 512, RT = 8, TW = 16;
__global__ __launch_bounds__(256) void rd(const double2 *__restrict__ g, double *__restrict__ out, int tiled, int all_threads) {
    const int tid = threadIdx.x, lb = blockIdx.y, r0 = blockIdx.x * RT;
    double acc = 0.0;
    const int nload_threads = all_threads ? 256 : 128;
    if (tid < nload_threads) {
        for (int gi = 0; gi < 3; ++gi) {
            const double2 *G = g + ((size_t)(lb * 3 + gi) << 19);
            double2 v[32];
            if (all_threads) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {  // 256 threads x 16 loads cover the same 64 KB
                    const int e = i * 256 + tid;             // element within the WG's 4096 per grid
                    const int tile = e >> 7, w = e & 127;    // 128 elements (8 rows x 16) per tile
                    const size_t a = tiled ? ((size_t)tile * (N1 * TW) + (size_t)r0 * TW + w) : ((size_t)blockIdx.x * 4096 + e);
                    v[i] = G[a];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc += v[i].x + v[i].y;
            } else {
                const int jl = tid & 15, f = (tid >> 4) & 7;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const size_t a = tiled ? ((size_t)i * (N1 * TW) + (size_t)(r0 + f) * TW + jl)
                                           : ((size_t)blockIdx.x * 4096 + (size_t)i * 128 + f * 16 + jl);
                    v[i] = G[a];
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) acc += v[i].x + v[i].y;
            }
        }
    }
    out[((size_t)lb * gridDim.x + blockIdx.x) * 256 + tid] = acc;
}
int main() {
    const int B = 85;
    double2 *g;
    double *out;
    hipMalloc(&g, (size_t)B * 3 * N1 * N2 * 16);
    hipMalloc(&out, (size_t)B * (N1 / RT) * 256 * 8);
    hipMemset(g, 0, (size_t)B * 3 * N1 * N2 * 16);
    const double bytes = (double)B * 3 * N1 * N2 * 16;
    for (int all = 0; all < 2; ++all)
        for (int tiled = 0; tiled < 2; ++tiled) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(rd, dim3(N1 / RT, B), dim3(256), 0, 0, g, out, tiled, all);
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(rd, dim3(N1 / RT, B), dim3(256), 0, 0, g, out, tiled, all);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s loads, %s: %.1f us per pass, %.2f TB/s\n", all ? "256-thread x16" : "128-thread x32", tiled ? "tiled " : "linear",
                   ms * 100.0, bytes / (ms / 10.0 * 1e-3) * 1e-12);
        }
    return 0;
}
