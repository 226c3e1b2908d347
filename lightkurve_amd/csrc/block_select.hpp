// block_select.hpp — workgroup-wide order statistics and moments over a strided view of global memory.
// Used by the sigma-clip loop of the regression corrector and by flatten (nanmedian / nanstd).
// All routines must be called by every thread of the workgroup (they contain barriers) and return the same
// value in every thread.  `sh` is a scratch area of at least 264 64-bit words in LDS.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lk {

__device__ __forceinline__ unsigned long long f64_sortable(double x) {
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_from_sortable(unsigned long long u) {
    u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
    return __longlong_as_double((long long)u);
}

// k-th smallest (0-based) of { val(i) : 0 <= i < n, keep(i) }, MSB-first radix select, 8 bits per pass.
// Precondition: 0 <= k < #kept.  val / keep are functors evaluated on the fly (no staging copy).
template <class Val, class Keep>
__device__ double block_select_kth(int n, long long k, Val val, Keep keep, unsigned long long *sh) {
    unsigned int *hist = reinterpret_cast<unsigned int *>(sh);          // 256 counters
    unsigned long long *ctl = sh + 128;                                 // [0] prefix, [1] remaining k
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) {
        ctl[0] = 0ull;
        ctl[1] = (unsigned long long)k;
    }
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        for (int i = tid; i < 256; i += nt) hist[i] = 0u;
        __syncthreads();
        const unsigned long long prefix = ctl[0];
        const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
        for (int i = tid; i < n; i += nt) {
            if (!keep(i)) continue;
            const unsigned long long u = f64_sortable(val(i));
            if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 0xffu], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long kk = ctl[1];
            int b = 0;
            for (; b < 255; ++b) {
                const unsigned int c = hist[b];
                if (kk < c) break;
                kk -= c;
            }
            ctl[0] = prefix | ((unsigned long long)b << shift);
            ctl[1] = kk;
        }
        __syncthreads();
    }
    const double r = f64_from_sortable(ctl[0]);
    __syncthreads();
    return r;
}

// deterministic workgroup sum (fixed tree order)
__device__ __forceinline__ double block_sum_dyn(double x, double *shd) {
    const int tid = threadIdx.x, nt = blockDim.x;
    shd[tid] = x;
    __syncthreads();
    for (int s = nt >> 1; s > 0; s >>= 1) {
        if (tid < s) shd[tid] += shd[tid + s];
        __syncthreads();
    }
    const double r = shd[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ long long block_count_dyn(long long x, long long *shl) {
    const int tid = threadIdx.x, nt = blockDim.x;
    shl[tid] = x;
    __syncthreads();
    for (int s = nt >> 1; s > 0; s >>= 1) {
        if (tid < s) shl[tid] += shl[tid + s];
        __syncthreads();
    }
    const long long r = shl[0];
    __syncthreads();
    return r;
}

// numpy.median of the kept values (mean of the two middle ones for an even count); NaN if none kept.
template <class Val, class Keep>
__device__ double block_median(int n, long long count, Val val, Keep keep, unsigned long long *sh) {
    if (count <= 0) return __longlong_as_double(0x7ff8000000000000ll);
    const double a = block_select_kth(n, (count - 1) / 2, val, keep, sh);
    if (count & 1) return a;
    const double b = block_select_kth(n, count / 2, val, keep, sh);
    return (a + b) * 0.5;
}

}  // namespace lk
