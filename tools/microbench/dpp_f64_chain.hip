// Microbenchmark behind the round-4 BLS prefix chain (bls.hip) on gfx950.
// The chain acc = bins[i] + acc is sequential by contract (the reference's rounding).  A lone wave issues an LDS
// instruction only every ~20 cycles, so a lane that feeds its own chain (two bins per ds_read2_b64) runs at ~11-14 cycles
// per bin although a dependent v_add_f64 takes 7.  gfx90a+ has DPP for 64-bit VOP2 with the row_newbcast controls:
//     v_fmac_f64_dpp acc, x, one row_newbcast:k        acc = x[lane k of this row] * 1.0 + acc  (= x + acc, one rounding)
// so 16 lanes of a row can each LOAD one bin (16 bins per LDS instruction) and all of them run the same chain, the
// operand of step k broadcast from lane k.  This program checks that the result is bit-identical to the sequential sum and
// measures cycles per bin for
//   lane  : the round-3 scheme (one lane per chain, ring of four 8-bin register sets)
//   bcast : the row_newbcast chain (rows 0 / 1 = the two components, ring of RING one-bin sets)
//   mov   : v_mov_b64_dpp row_newbcast + v_add_f64 (what the compiler makes of __builtin_amdgcn_update_dpp)
// with `busy` other waves of the workgroup hammering LDS meanwhile (the pipelined scan's situation).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o dpp_f64_chain dpp_f64_chain.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr int CAP = 6656;  // bins per component (multiple of 128)

#define FMAC_BCAST(k)                                                                                     \
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(one))

__device__ __forceinline__ void chain16(double &acc, double x, double one) {
    asm volatile("s_nop 1");
    FMAC_BCAST(0);
    FMAC_BCAST(1);
    FMAC_BCAST(2);
    FMAC_BCAST(3);
    FMAC_BCAST(4);
    FMAC_BCAST(5);
    FMAC_BCAST(6);
    FMAC_BCAST(7);
    FMAC_BCAST(8);
    FMAC_BCAST(9);
    FMAC_BCAST(10);
    FMAC_BCAST(11);
    FMAC_BCAST(12);
    FMAC_BCAST(13);
    FMAC_BCAST(14);
    FMAC_BCAST(15);
}

template <int K>
__device__ __forceinline__ double bcast_mov(double x) {
    // row_newbcast:K = 0x150 + K
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x150 + K, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x150 + K, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// mode 0: lane chain, 1: bcast fmac, 2: bcast mov + add.  carries[c][blk + 1] = running sum after every 32nd bin.
__global__ __launch_bounds__(1024) void chain_kernel(const double *__restrict__ in, double *__restrict__ carries, int n_bins,
                                                     int mode, int busy, long long *__restrict__ cyc, double *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *ya = reinterpret_cast<double *>(smem), *wa = ya + CAP;
    double *car = wa + CAP;  // [2][CAP / 32 + 1]
    double *scratch = car + 2 * (CAP / 32 + 1);
    volatile int *done = reinterpret_cast<volatile int *>(scratch + 1024);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < CAP; i += blockDim.x) {
        ya[i] = i < n_bins ? in[i] : 0.0;
        wa[i] = i < n_bins ? in[CAP + i] : 0.0;
    }
    if (tid == 0) *done = 0;
    if (tid < 1024) scratch[tid] = 1.0;
    __syncthreads();
    const int nblk = n_bins >> 5;  // n_bins is a multiple of 128 here
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        const long long t0 = clock64();
        const unsigned long long w0 = wall_clock64();
        if (mode == 0) {
            if (lane < 2) {
                const double *comp = lane ? wa : ya;
                double *cr = car + lane * (CAP / 32 + 1);
                double acc = 0.0;
                cr[0] = 0.0;
                double x0[8], x1[8], x2[8], x3[8];
                const int lastq = nblk * 4 - 1;
                auto load8 = [&](double(&x)[8], int q) {
                    const double *cp = comp + (min(q, lastq) << 3);
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = cp[u];
                };
                auto add8 = [&](double(&x)[8]) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = x[u] + acc;
                };
                load8(x0, 0);
                load8(x1, 1);
                load8(x2, 2);
                load8(x3, 3);
                for (int blk = 0; blk < nblk; ++blk) {
                    const int q = blk << 2;
                    add8(x0);
                    load8(x0, q + 4);
                    __builtin_amdgcn_sched_barrier(0);
                    add8(x1);
                    load8(x1, q + 5);
                    __builtin_amdgcn_sched_barrier(0);
                    add8(x2);
                    load8(x2, q + 6);
                    __builtin_amdgcn_sched_barrier(0);
                    add8(x3);
                    load8(x3, q + 7);
                    cr[blk + 1] = acc;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // rows 0 / 1: components y / ivar (rows 2, 3 repeat them: no divergence, nothing stored)
            const int row = (lane >> 4) & 1, l = lane & 15;
            const double *comp = row ? wa : ya;
            double *cr = car + row * (CAP / 32 + 1);
            double acc = 0.0;
            const double one = 1.0;
            if (lane < 32 && l == 0) cr[0] = 0.0;
            constexpr int RING = 8;  // 16-bin sets in flight
            double xs[RING];
            const int nset = n_bins >> 4, lasts = nset - 1;
#pragma unroll
            for (int s = 0; s < RING; ++s) xs[s] = comp[(min(s, lasts) << 4) + l];
            for (int s0 = 0; s0 < nset; s0 += RING) {
#pragma unroll
                for (int s = 0; s < RING; ++s) {
                    const double x = xs[s];
                    xs[s] = comp[(min(s0 + s + RING, lasts) << 4) + l];
                    if (mode == 1) {
                        chain16(acc, x, one);
                    } else {
                        acc = bcast_mov<0>(x) + acc;
                        acc = bcast_mov<1>(x) + acc;
                        acc = bcast_mov<2>(x) + acc;
                        acc = bcast_mov<3>(x) + acc;
                        acc = bcast_mov<4>(x) + acc;
                        acc = bcast_mov<5>(x) + acc;
                        acc = bcast_mov<6>(x) + acc;
                        acc = bcast_mov<7>(x) + acc;
                        acc = bcast_mov<8>(x) + acc;
                        acc = bcast_mov<9>(x) + acc;
                        acc = bcast_mov<10>(x) + acc;
                        acc = bcast_mov<11>(x) + acc;
                        acc = bcast_mov<12>(x) + acc;
                        acc = bcast_mov<13>(x) + acc;
                        acc = bcast_mov<14>(x) + acc;
                        acc = bcast_mov<15>(x) + acc;
                    }
                    if (s & 1)
                        if (lane < 32 && l == 0) cr[((s0 + s) >> 1) + 1] = acc;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const long long t1 = clock64();
        const unsigned long long w1 = wall_clock64();
        if (lane == 0) {
            cyc[0] = t1 - t0;
            cyc[1] = (long long)(w1 - w0);
            *done = 1;
        }
        __builtin_amdgcn_s_setprio(0);
    } else if (wave <= busy) {
        // other waves keep the LDS pipe and their SIMDs busy (reads + fp64 arithmetic, like the scan)
        double a = 0.0;
        int i = tid & 1023;
        while (!*done) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a = fma(scratch[i], 1.0000001, a);
                i = (i * 5 + 1) & 1023;
            }
        }
        sink[blockIdx.x * 1024 + tid] = a;
    }
    __syncthreads();
    for (int i = tid; i < 2 * (CAP / 32 + 1); i += blockDim.x) carries[(size_t)blockIdx.x * 2 * (CAP / 32 + 1) + i] = car[i];
}

int main() {
    const int n_bins = 6528;  // 51 x 128
    std::mt19937_64 rng(7);
    std::vector<double> in(2 * CAP, 0.0);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    std::uniform_int_distribution<int> e(-30, 30);
    for (int c = 0; c < 2; ++c)
        for (int i = 0; i < n_bins; ++i) in[c * CAP + i] = std::ldexp(u(rng), e(rng)) + (c ? 1.0 : 0.0);
    std::vector<double> want(2 * (CAP / 32 + 1), 0.0);
    for (int c = 0; c < 2; ++c) {
        double acc = 0.0;
        for (int i = 0; i < n_bins; ++i) {
            volatile double s = in[c * CAP + i] + acc;
            acc = s;
            if ((i & 31) == 31) want[c * (CAP / 32 + 1) + (i >> 5) + 1] = acc;
        }
    }
    double *d_in, *d_car, *d_sink;
    long long *d_cyc;
    const int NBLK = 256;
    CK(hipMalloc(&d_in, in.size() * 8));
    CK(hipMalloc(&d_car, (size_t)NBLK * want.size() * 8));
    CK(hipMalloc(&d_sink, (size_t)NBLK * 1024 * 8));
    CK(hipMalloc(&d_cyc, 16));
    CK(hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice));
    const size_t lds = (size_t)(2 * CAP + 2 * (CAP / 32 + 1) + 1024 + 2) * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const char *names[3] = {"lane ", "bcast", "mov  "};
    for (int mode = 0; mode < 3; ++mode)
        for (int busy : {0, 3, 15}) {
            CK(hipMemset(d_car, 0, (size_t)NBLK * want.size() * 8));
            long long cyc[2] = {0, 0};
            for (int rep = 0; rep < 3; ++rep) {
                chain_kernel<<<NBLK, 1024, lds>>>(d_in, d_car, n_bins, mode, busy, d_cyc, d_sink);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(cyc, d_cyc, 16, hipMemcpyDeviceToHost));
            std::vector<double> got((size_t)NBLK * want.size());
            CK(hipMemcpy(got.data(), d_car, got.size() * 8, hipMemcpyDeviceToHost));
            long bad = 0, checked = 0;
            const int per = CAP / 32 + 1, nblk = n_bins >> 5;  // entries 0 .. nblk of each component are written
            for (int b = 0; b < NBLK; ++b)
                for (int c = 0; c < 2; ++c)
                    for (int i = 0; i <= nblk; ++i, ++checked)
                        if (std::memcmp(&got[b * want.size() + c * per + i], &want[c * per + i], 8) != 0) ++bad;
            printf("%s busy waves %2d: %6.2f clock64 ticks per bin, %6.2f us per %d-bin chain pair (wall clock), carries differing from the "
                   "sequential sum: %ld of %zu\n",
                   names[mode], busy, (double)cyc[0] / n_bins, (double)cyc[1] / 100.0, n_bins, bad, (size_t)checked);
        }
    return 0;
}
