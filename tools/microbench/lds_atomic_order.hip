// Microbenchmarks behind the round-3 BLS histogram (bls.hip) on gfx950.
//   order   : does ONE ds_add_f64 wave instruction whose lanes hit the same LDS address apply the additions in
//             increasing lane order?  (floating-point sums: the order decides the bits.)  Random index patterns with
//             adjacent and non-adjacent duplicates, values spread over 20 binades, many chunks in program order;
//             compared bit for bit with the host's sequential lane-order sum (and with the reverse order).
//   chain   : cycles per dependent v_add_f64 (the BLS prefix chain), 2 active lanes and 64.
//   atomrate: cycles per ds_add_f64 pair (16-B-strided addresses, ~1.4 lanes per address) for 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_order lds_atomic_order.hip
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

constexpr int NB = 256;  // bins

__global__ __launch_bounds__(64) void order_kernel(const int *__restrict__ idx, const double2 *__restrict__ val, int nchunk,
                                                   double2 *__restrict__ out) {
    __shared__ double2 bins[NB];
    const int lane = threadIdx.x;
    for (int i = lane; i < NB; i += 64) bins[i] = make_double2(0.0, 0.0);
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * nchunk * 64;
    for (int c = 0; c < nchunk; ++c) {
        const int j = idx[base + c * 64 + lane];
        const double2 v = val[base + c * 64 + lane];
        if (j >= 0) {
            atomicAdd(&bins[j].x, v.x);
            atomicAdd(&bins[j].y, v.y);
        }
    }
    __syncthreads();
    for (int i = lane; i < NB; i += 64) out[(size_t)blockIdx.x * NB + i] = bins[i];
}

__global__ __launch_bounds__(64) void chain_kernel(double *__restrict__ out, long long *__restrict__ cyc, int n, int lanes) {
    const int lane = threadIdx.x;
    double acc = (double)lane, x = 1.0 + 1e-9 * lane;
    if (lane < lanes) {
        const long long t0 = clock64();
#pragma unroll 1
        for (int i = 0; i < n; i += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = acc + x;
            asm volatile("" : "+v"(acc));
        }
        const long long t1 = clock64();
        out[lane] = acc;
        if (lane == 0) cyc[0] = t1 - t0;
    }
}

// mode 0: double2 bins, ~1.4 lanes per bin; 1: split y[] / w[] arrays, ~1.4 lanes per bin; 2: double2, one lane per bin;
// 3: split arrays, one lane per bin; 4: plain RMW (ds_read_b128, two adds, ds_write_b128) on double2, one lane per bin
__global__ __launch_bounds__(1024) void atomrate_kernel(double *__restrict__ out, long long *__restrict__ cyc, int iters, int mode) {
    __shared__ double2 bins[4096];
    double *ya = reinterpret_cast<double *>(bins), *wa = ya + 4096;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) bins[i] = make_double2(0.0, 0.0);
    __syncthreads();
    const long long t0 = clock64();
    const bool dense = mode >= 2 && mode != 5;
    int j = wave * 128 + (dense ? lane : (lane * 45) / 64);
    const int step = dense ? 64 : 45;
    if (mode == 0 || mode == 2) {
        for (int it = 0; it < iters; ++it) {
            atomicAdd(&bins[j & 4095].x, 1.0);
            atomicAdd(&bins[j & 4095].y, 2.0);
            j += step;
        }
    } else if (mode == 1 || mode == 3) {
        for (int it = 0; it < iters; ++it) {
            atomicAdd(&ya[j & 4095], 1.0);
            atomicAdd(&wa[j & 4095], 2.0);
            j += step;
        }
    } else if (mode == 5) {  // split arrays, ~1.4 lanes per bin, run leaders first and second members in a second instruction
        const int jp = wave * 128 + ((lane - 1) * 45) / 64;
        const bool leader = lane == 0 || jp != j;
        for (int it = 0; it < iters; ++it) {
            if (leader) {
                atomicAdd(&ya[j & 4095], 1.0);
                atomicAdd(&wa[j & 4095], 2.0);
            }
            if (!leader) {
                atomicAdd(&ya[j & 4095], 1.0);
                atomicAdd(&wa[j & 4095], 2.0);
            }
            j += step;
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            double2 v = bins[j & 4095];
            v.x += 1.0;
            v.y += 2.0;
            bins[j & 4095] = v;
            j += step;
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cyc[0] = t1 - t0;
    double s = 0;
    for (int i = tid; i < 4096; i += blockDim.x) s += bins[i].x + bins[i].y;
    out[tid] = s;
}

int main() {
    // ---- order
    const int blocks = 512, nchunk = 40;
    std::mt19937_64 rng(12345);
    std::vector<int> idx((size_t)blocks * nchunk * 64);
    std::vector<double2> val(idx.size());
    for (int b = 0; b < blocks; ++b)
        for (int c = 0; c < nchunk; ++c) {
            const int mode = (b + c) % 4;
            int cur = (int)(rng() % NB);
            for (int l = 0; l < 64; ++l) {
                int j;
                if (mode == 0) {  // monotone runs of 1-3 (the BLS pattern)
                    if (l && (rng() % 10) < 6) cur = (cur + 1) % NB;
                    j = cur;
                } else if (mode == 1) {  // fully random: non-adjacent duplicates
                    j = (int)(rng() % 24);
                } else if (mode == 2) {  // everyone on one or two addresses
                    j = (int)(rng() % 2) * 7;
                } else {  // random with holes (inactive lanes)
                    j = (rng() % 5) ? (int)(rng() % NB) : -1;
                }
                const size_t p = ((size_t)b * nchunk + c) * 64 + l;
                idx[p] = j;
                const double m = 1.0 + (double)(rng() % 1000000) * 1e-6;
                val[p] = make_double2(ldexp(m, (int)(rng() % 20) - 10) * ((rng() & 1) ? 1 : -1), ldexp(m, (int)(rng() % 20)));
            }
        }
    int *d_idx;
    double2 *d_val, *d_out;
    CK(hipMalloc(&d_idx, idx.size() * 4));
    CK(hipMalloc(&d_val, val.size() * 16));
    CK(hipMalloc(&d_out, (size_t)blocks * NB * 16));
    CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_val, val.data(), val.size() * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(order_kernel, dim3(blocks), dim3(64), 0, 0, d_idx, d_val, nchunk, d_out);
    CK(hipDeviceSynchronize());
    std::vector<double2> out((size_t)blocks * NB);
    CK(hipMemcpy(out.data(), d_out, out.size() * 16, hipMemcpyDeviceToHost));
    long long bad_fwd = 0, bad_rev = 0, tot = 0;
    for (int b = 0; b < blocks; ++b) {
        std::vector<double2> f(NB, make_double2(0, 0)), r(NB, make_double2(0, 0));
        for (int c = 0; c < nchunk; ++c) {
            for (int l = 0; l < 64; ++l) {
                const size_t p = ((size_t)b * nchunk + c) * 64 + l;
                if (idx[p] >= 0) {
                    f[idx[p]].x += val[p].x;
                    f[idx[p]].y += val[p].y;
                }
            }
            for (int l = 63; l >= 0; --l) {
                const size_t p = ((size_t)b * nchunk + c) * 64 + l;
                if (idx[p] >= 0) {
                    r[idx[p]].x += val[p].x;
                    r[idx[p]].y += val[p].y;
                }
            }
        }
        for (int i = 0; i < NB; ++i) {
            const double2 o = out[(size_t)b * NB + i];
            ++tot;
            if (memcmp(&o, &f[i], 16)) ++bad_fwd;
            if (memcmp(&o, &r[i], 16)) ++bad_rev;
        }
    }
    printf("order: %lld bins checked; mismatches vs lane-order sum: %lld; vs reverse-lane-order sum: %lld  => %s\n", tot,
           bad_fwd, bad_rev, bad_fwd == 0 ? "LANE-ORDERED" : "NOT lane-ordered");

    // ---- chain
    double *d_o;
    long long *d_c, hc;
    CK(hipMalloc(&d_o, 8192 * 8));
    CK(hipMalloc(&d_c, 8));
    for (int lanes : {2, 64}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, 0, d_o, d_c, 65536, lanes);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(&hc, d_c, 8, hipMemcpyDeviceToHost));
        printf("chain: %d active lanes: %.2f clock64 ticks per dependent v_add_f64\n", lanes, (double)hc / 65536.0);
    }
    // ---- atomrate
    for (int mode = 0; mode < 6; ++mode)
        for (int nt : {64, 256, 1024}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(atomrate_kernel, dim3(1), dim3(nt), 0, 0, d_o, d_c, 4096, mode);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(&hc, d_c, 8, hipMemcpyDeviceToHost));
            printf("atomrate mode %d: %4d threads (1 CU): %.1f ticks per 64-cadence update per wave, %.2f per update overall\n", mode, nt,
                   (double)hc / 4096.0, (double)hc / 4096.0 / (nt / 64));
        }
    return 0;
}
