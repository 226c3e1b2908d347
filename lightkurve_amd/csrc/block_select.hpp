// block_select.hpp — workgroup-wide order statistics and moments over a strided view of global memory.
// Used by the sigma-clip loop of the regression corrector and by flatten (nanmedian / nanstd).
// All routines must be called by every thread of the workgroup (they contain barriers) and return the same
// value in every thread.  `sh` is a scratch area of at least max(264, blockDim.x) 64-bit words in LDS.
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
        // Real light curves put most keys in one or two digits (flux ~ 1.0), which would serialise 64 same-address
        // LDS atomics per wave instruction: lanes with the leader's digit are counted by ballot and added once;
        // after three such rounds whatever is left (few, scattered digits) goes through plain atomics.
        for (int i0 = 0; i0 < n; i0 += nt) {
            const int i = i0 + tid;
            bool m = i < n && keep(i);
            unsigned int d = 0u;
            if (m) {
                const unsigned long long u = f64_sortable(val(i));
                m = (u & himask) == prefix;
                d = (unsigned int)(u >> shift) & 0xffu;
            }
            unsigned long long todo = __ballot(m);
            for (int round = 0; round < 3 && todo; ++round) {
                const int leader = __ffsll((long long)todo) - 1;
                const unsigned int d0 = (unsigned int)__shfl((int)d, leader);
                const bool mine = m && d == d0;
                const unsigned long long same = __ballot(mine);
                if ((tid & 63) == leader) atomicAdd(&hist[d0], (unsigned int)__popcll(same));
                if (mine) m = false;
                todo &= ~same;
            }
            if (m) atomicAdd(&hist[d], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            // first bin b with k < cumulative count: lane l owns bins 4l..4l+3, wave-wide inclusive scan of the
            // lane totals, the lane whose range straddles k resolves its four bins
            const unsigned long long kk = ctl[1];
            const unsigned int c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const unsigned int tot = c0 + c1 + c2 + c3;
            unsigned int inc = tot;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned int v = __shfl_up(inc, o);
                if (tid >= o) inc += v;
            }
            const unsigned long long before = inc - tot;
            // the select precondition (k < #kept with this prefix) guarantees exactly one lane qualifies
            if (kk >= before && kk < (unsigned long long)inc) {
                unsigned long long r = kk - before;
                int b = 4 * tid;
                if (r >= c0) {
                    r -= c0;
                    ++b;
                    if (r >= c1) {
                        r -= c1;
                        ++b;
                        if (r >= c2) {
                            r -= c2;
                            ++b;
                        }
                    }
                }
                ctl[0] = prefix | ((unsigned long long)b << shift);
                ctl[1] = r;
            }
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
    const long long k = (count - 1) / 2;
    const double a = block_select_kth(n, k, val, keep, sh);
    if (count & 1) return a;
    // the (k+1)-th smallest without a second select: it is `a` again if at least k + 2 kept values are <= a,
    // otherwise the smallest kept value above a — one pass with a count and a min
    const int tid = threadIdx.x, nt = blockDim.x;
    long long le = 0;
    double up = __longlong_as_double(0x7ff0000000000000ll);
    for (int i = tid; i < n; i += nt)
        if (keep(i)) {
            const double v = val(i);
            if (v <= a)
                ++le;
            else
                up = fmin(up, v);
        }
    const long long le_all = block_count_dyn(le, reinterpret_cast<long long *>(sh));
    double *shd = reinterpret_cast<double *>(sh);
    shd[tid] = up;
    __syncthreads();
    for (int s2 = nt >> 1; s2 > 0; s2 >>= 1) {
        if (tid < s2) shd[tid] = fmin(shd[tid], shd[tid + s2]);
        __syncthreads();
    }
    const double upper = shd[0];
    __syncthreads();
    const double b = le_all >= k + 2 ? a : upper;
    return (a + b) * 0.5;
}

}  // namespace lk
