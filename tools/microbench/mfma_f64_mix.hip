// What v_mfma_f64_16x16x4_f64 reaches on gfx950 when its operands are GENERATED next to it, as the moment-form Gram of pld.hip does:
// a wave owns 4 x 4 accumulator tiles; per step 4 + 4 operands, each the product of three factors read from LDS.
//   mode 0: 16 MFMAs per step, constant operands                     (the issue peak)
//   mode 1: + the 16 fp64 multiplications of the 8 operands          (factors in registers)
//   mode 2: + the 24 ds_read_b64 of the factors                      (the Gram kernel's step)
//   mode 3: mode 2 with the reads of step s + 1 issued before the MFMAs of step s and multiplied after them (its software pipeline)
// 2 waves per SIMD (256 VGPRs each, like the kernel).  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_mix mfma_f64_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(double *out, int iters, const int *__restrict__ idx, int rnd) {
    __shared__ double us[64 * 17];
    for (int e = threadIdx.x; e < 64 * 17; e += 256) {
        unsigned h = (unsigned)e * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 13;
        h *= 2246822519u;
        // rnd: factors with random mantissas in [-1.5, 1.5] (every operand bit toggles), else 1 + 1e-9 e
        us[e] = rnd ? ((double)(h & 0xffffff) / 8388608.0 - 1.0) * 1.5 + 1e-17 * (double)(h >> 8) : 1.0 + 1e-9 * e;
    }
    __syncthreads();
    d4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63, lq = lane >> 4;
    int oa[4][3], ob[4][3];
    for (int i = 0; i < 4; ++i)
        for (int p = 0; p < 3; ++p) {
            oa[i][p] = idx[(lane * 7 + i * 3 + p) & 255] & 15;
            ob[i][p] = idx[(lane * 5 + i * 3 + p + 100) & 255] & 15;
        }
    double av[4], bv[4], ra[4][3], rb[4][3];
    for (int i = 0; i < 4; ++i) {
        av[i] = 1.0 + lane * 1e-3 + i;
        bv[i] = 2.0 + lane * 1e-3 - i;
        for (int p = 0; p < 3; ++p) ra[i][p] = rb[i][p] = 1.0 + 1e-6 * (i + p);
    }
    auto issue = [&](int s) {
        const double *row = us + ((s * 4 + lq) & 63) * 17;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                ra[i][p] = row[oa[i][p]];
                rb[i][p] = row[ob[i][p]];
            }
    };
    auto multiply = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = ra[i][0] * ra[i][1] * ra[i][2];
            bv[i] = rb[i][0] * rb[i][1] * rb[i][2];
        }
    };
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) issue(it);
        if (MODE == 1 || MODE == 2) multiply();
        if (MODE == 3) {
            __builtin_amdgcn_sched_barrier(0);
            issue(it + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
        if (MODE == 3) {
            __builtin_amdgcn_sched_barrier(0);
            multiply();
        }
        if (MODE == 1)  // keep the factors changing so the products are not hoisted
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i][0] = av[i] * 0.999;
                rb[i][0] = bv[i] * 1.001;
            }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(double *d, const int *idx, const char *what, int rnd = 0) {
    const int iters = 60000, blocks = 256 * 2, threads = 256;  // 2 workgroups of 4 waves per CU = 2 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100, idx, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, idx, rnd);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 16.0 * 2048.0;
    printf("%-70s %.2f ms  %.1f TFLOP/s  (%.2f of 78.6)\n", what, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 78.6);
}
int main() {
    double *d;
    int *idx, h[256];
    for (int i = 0; i < 256; ++i) h[i] = (i * 37 + 11) % 16;
    hipMalloc(&d, 1 << 24);
    hipMalloc(&idx, sizeof(h));
    hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(d, idx, "16 MFMAs per step, constant operands");
    run<1>(d, idx, "+ 16 fp64 multiplications per step (8 three-factor operands)");
    run<2>(d, idx, "+ 24 ds_read_b64 per step, read -> multiply -> MFMA in order");
    run<3>(d, idx, "same, reads of step s + 1 before the MFMAs of step s, products after");
    run<3>(d, idx, "same, RANDOM factors (every operand bit toggles)", 1);
    run<3>(d, idx, "same, RANDOM factors, again", 1);
    run<0>(d, idx, "16 MFMAs per step, constant operands, again");
    return 0;
}
