// Tile shape against occupancy for an MFMA kernel that generates its operands (the moment-form Gram's step): a wave owns TW x TW accumulator
// tiles (v_mfma_f64_16x16x4_f64), per step TW + TW operands of three LDS factors each, reads of step s + 1 issued before the MFMAs of step s,
// products after them.  4 x 4 tiles need 128 accumulator VGPRs (2 waves per SIMD), 3 x 3 need 72 (3 waves), 2 x 2 need 32 (4+ waves).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_tile_shape mfma_f64_tile_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int TW, int WPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void k(double *out, int iters, const int *__restrict__ idx) {
    __shared__ double us[64 * 17];
    for (int e = threadIdx.x; e < 64 * 17; e += 256) {
        unsigned h = (unsigned)e * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 13;
        h *= 2246822519u;
        us[e] = ((double)(h & 0xffffff) / 8388608.0 - 1.0) * 1.5 + 1e-17 * (double)(h >> 8);
    }
    __syncthreads();
    d4 acc[TW][TW];
    for (int i = 0; i < TW; ++i)
        for (int j = 0; j < TW; ++j) acc[i][j] = d4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63, lq = lane >> 4;
    int oa[TW][3], ob[TW][3];
    for (int i = 0; i < TW; ++i)
        for (int p = 0; p < 3; ++p) {
            oa[i][p] = idx[(lane * 7 + i * 3 + p) & 255] & 15;
            ob[i][p] = idx[(lane * 5 + i * 3 + p + 100) & 255] & 15;
        }
    double av[TW], bv[TW], ra[TW][3], rb[TW][3];
    for (int i = 0; i < TW; ++i) {
        av[i] = 1.0 + lane * 1e-3 + i;
        bv[i] = 2.0 + lane * 1e-3 - i;
        for (int p = 0; p < 3; ++p) ra[i][p] = rb[i][p] = 1.0 + 1e-6 * (i + p);
    }
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        const double *row = us + (((it + 1) * 4 + lq) & 63) * 17;
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                ra[i][p] = row[oa[i][p]];
                rb[i][p] = row[ob[i][p]];
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int j = 0; j < TW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TW; ++i) {
            av[i] = ra[i][0] * ra[i][1] * ra[i][2];
            bv[i] = rb[i][0] * rb[i][1] * rb[i][2];
        }
    }
    double s = 0;
    for (int i = 0; i < TW; ++i)
        for (int j = 0; j < TW; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int TW, int WPS>
void run(double *d, const int *idx) {
    const int iters = 40000, blocks = 256 * WPS, threads = 256;  // WPS workgroups of 4 waves per CU = WPS waves per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<TW, WPS>), dim3(blocks), dim3(threads), 0, 0, d, 100, idx);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<TW, WPS>), dim3(blocks), dim3(threads), 0, 0, d, iters, idx);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * (double)(TW * TW) * 2048.0;
    printf("%d x %d tiles per wave, %d waves per SIMD: %d MFMAs, %d products, %d LDS reads per step   %.2f ms  %.1f TFLOP/s  (%.2f of 78.6)\n", TW, TW, WPS,
           TW * TW, 4 * TW, 6 * TW, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 78.6);
}
int main() {
    double *d;
    int *idx, h[256];
    for (int i = 0; i < 256; ++i) h[i] = (i * 37 + 11) % 16;
    (void)hipMalloc(&d, 1 << 24);
    (void)hipMalloc(&idx, sizeof(h));
    (void)hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice);
    run<4, 2>(d, idx);
    run<4, 1>(d, idx);
    run<3, 3>(d, idx);
    run<3, 2>(d, idx);
    run<2, 4>(d, idx);
    run<2, 8>(d, idx);
    run<4, 2>(d, idx);
    return 0;
}
