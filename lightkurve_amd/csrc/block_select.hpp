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

// Out-of-line copy for the rarely taken fallback of block_select_sampled: kept out of the caller's register allocation
// (inlined, its eight passes cost the hot path ~0.15 ms per call in spills although they never ran).
template <class Val, class Keep>
__device__ __noinline__ double block_select_kth_cold(int n, long long k, Val val, Keep keep, unsigned long long *sh) {
    return block_select_kth(n, k, val, keep, sh);
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

// ------------------------------------------------------------------------------------------------ strided passes
// for i = tid, tid + nt, ... < n: proc(i, load(i)) with U loads in flight per thread.  A plain `for (i = tid; i < n;
// i += nt)` loop over global memory waits one full memory latency per element (the compiler cannot hoist loads over
// the byte stores / LDS atomics in the bodies here); issuing U loads before the first use turns n / nt latencies into
// n / (nt U).
template <int U, class L, class P>
__device__ __forceinline__ void strided_pass(int n, L load, P proc) {
    const int tid = threadIdx.x, nt = blockDim.x;
    int i = tid;
#ifndef LK_SP_PLAIN
    for (; i + (U - 1) * nt < n; i += U * nt) {
        decltype(load(0)) v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = load(i + u * nt);
#pragma unroll
        for (int u = 0; u < U; ++u) proc(i + u * nt, v[u]);
    }
#endif
    for (; i < n; i += nt) proc(i, load(i));
}

// ------------------------------------------------------------------------------------------------ wave-level helpers
// Workgroup sum with two barriers: 64-lane butterfly (DPP/shuffles), one LDS word per wave, lane-serial combine.
// Fixed order, so deterministic; `sh` needs blockDim.x / 64 (+1) 64-bit words.
__device__ __forceinline__ double block_sum_fast(double x, double *shd) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    const int tid = threadIdx.x, nw = (blockDim.x + 63) >> 6;
    __syncthreads();  // shd may still be read by the previous user
    if ((tid & 63) == 0) shd[tid >> 6] = x;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < nw; ++w) r += shd[w];
    return r;
}

__device__ __forceinline__ long long block_count_fast(long long x, long long *shl) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    const int tid = threadIdx.x, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((tid & 63) == 0) shl[tid >> 6] = x;
    __syncthreads();
    long long r = 0;
    for (int w = 0; w < nw; ++w) r += shl[w];
    return r;
}

// Exclusive prefix sum of one int per thread across the workgroup (two barriers).  total (same in all threads) via *tot.
__device__ __forceinline__ int block_exscan_int(int x, int *shi, int *tot) {
    const int tid = threadIdx.x, lane = tid & 63, nw = (blockDim.x + 63) >> 6;
    int inc = x;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    __syncthreads();
    if (lane == 63) shi[tid >> 6] = inc;
    __syncthreads();
    int base = 0, all = 0;
    for (int w = 0; w < nw; ++w) {
        const int c = shi[w];
        if (w < (tid >> 6)) base += c;
        all += c;
    }
    *tot = all;
    return base + inc - x;
}

// ascending bitonic sort of S = 2^m 64-bit keys in LDS.  Every thread keeps E = S / blockDim.x consecutive keys in
// registers: compare-exchanges at distance < E are register moves, at distance < 64 E lane shuffles inside the wave
// (no barrier), and only the few stages at distance >= 64 E go through LDS (two barriers each) — 6 LDS stages instead
// of 66 barriers for 2048 keys on 512 threads.  Falls back to the plain one-barrier-per-stage LDS network when S is not
// E x blockDim.x with E in {1, 2, 4, 8}.
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int lane_delta) {
    const int lo = __shfl_xor((int)(unsigned int)(v & 0xffffffffull), lane_delta);
    const int hi = __shfl_xor((int)(unsigned int)(v >> 32), lane_delta);
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo;
}

template <int E>
__device__ __forceinline__ void lds_bitonic_sort_regs(unsigned long long *keys, int S) {
    const int tid = threadIdx.x;
    unsigned long long r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = keys[tid * E + e];
    for (int size = 2; size <= S; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64 * E) {  // partner in another wave: through LDS
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) keys[tid * E + e] = r[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid * E + e;
                    const unsigned long long p = keys[i ^ stride];
                    const bool keep_min = ((i & stride) == 0) == ((i & size) == 0);
                    r[e] = keep_min ? (r[e] < p ? r[e] : p) : (r[e] > p ? r[e] : p);
                }
            } else if (stride >= E) {  // partner in another lane of this wave
                const int ld = stride / E;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = tid * E + e;
                    const unsigned long long p = shfl_xor_u64(r[e], ld);
                    const bool keep_min = ((i & stride) == 0) == ((i & size) == 0);
                    r[e] = keep_min ? (r[e] < p ? r[e] : p) : (r[e] > p ? r[e] : p);
                }
            } else {  // partner in this thread's registers (stride in {1, 2, 4})
                unsigned long long q[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    unsigned long long p = r[e];
                    if (stride == 1) p = r[(e ^ 1) & (E - 1)];
                    if (stride == 2) p = r[(e ^ 2) & (E - 1)];
                    if (stride == 4) p = r[(e ^ 4) & (E - 1)];
                    const int i = tid * E + e;
                    const bool keep_min = ((i & stride) == 0) == ((i & size) == 0);
                    q[e] = keep_min ? (r[e] < p ? r[e] : p) : (r[e] > p ? r[e] : p);
                }
#pragma unroll
                for (int e = 0; e < E; ++e) r[e] = q[e];
            }
        }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) keys[tid * E + e] = r[e];
    __syncthreads();
}

__device__ __forceinline__ void lds_bitonic_sort(unsigned long long *keys, int S) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if ((nt & 63) == 0 && S == nt) return lds_bitonic_sort_regs<1>(keys, S);
    if ((nt & 63) == 0 && S == 2 * nt) return lds_bitonic_sort_regs<2>(keys, S);
    if ((nt & 63) == 0 && S == 4 * nt) return lds_bitonic_sort_regs<4>(keys, S);
    // (S == 8 nt would keep 8 keys = 32 VGPRs of sort state per thread: the plain network below serves it — since the
    // histogram select took over the large candidate sets this is a rare fall-back, and the register budget of every
    // caller is set by its hot paths)
    for (int size = 2; size <= S; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int p = tid; p < S / 2; p += nt) {
                const int lo_i = ((p / stride) * 2 * stride) + (p % stride), hi_i = lo_i + stride;
                const bool up = (lo_i & size) == 0;
                const unsigned long long a = keys[lo_i], b = keys[hi_i];
                if ((a > b) == up) {
                    keys[lo_i] = b;
                    keys[hi_i] = a;
                }
            }
            __syncthreads();
        }
}

// ------------------------------------------------------------------------------------------------ histogram select in LDS
// Ranks qa and qb (either may be -1 = not wanted) among the nc values cand[0..nc) that all lie strictly inside (lo, hi):
// a SEL_NB-bin histogram over the linear map of (lo, hi) (LDS atomics), a workgroup scan of the bins, the members of the one
// or two bins that hold the wanted ranks gathered into a short list and ranked by counting.  The map v -> bin is monotone, so
// every value of a lower bin is smaller than every value of a higher one and the result is the exact order statistic.
// Two sweeps over the candidates and a 1024-bin scan (~5 us) instead of a 2048- or 4096-key bitonic sort (30 - 100 us with
// every workgroup of a launch sorting at once: round 5's stop-point profile of the phase-split flatten pipeline).
// Returns false (workgroup-uniform, nothing modified) when it does not apply: non-finite or empty bracket, no room behind
// the candidates for the histogram, or more than SEL_LIST values in the wanted bins (heavy ties) — the caller sorts then.
constexpr int SEL_NB = 1024;
constexpr int SEL_LIST = 64;

__device__ __forceinline__ bool lds_hist_select(const double *cand, int nc, int cap, int qa, int qb, double lo, double hi,
                                                unsigned long long *sh, double *va, double *vb) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nc_pad = (nc + 1) & ~1;
    if (!(hi > lo) || !isfinite(lo) || !isfinite(hi) || nc_pad + SEL_NB / 2 + SEL_LIST + 2 > cap || nc <= 0) return false;
    int *hist = reinterpret_cast<int *>(const_cast<double *>(cand) + nc_pad);
    double *list = const_cast<double *>(cand) + nc_pad + SEL_NB / 2;
    int *ctl = reinterpret_cast<int *>(sh + 210);       // [0] bin of qa, [1] its exclusive prefix, [2], [3] same for qb, [4] list length
    double *outv = reinterpret_cast<double *>(sh + 220);
    __syncthreads();
    for (int i = tid; i < SEL_NB; i += nt) hist[i] = 0;
    if (tid == 0) {
        ctl[0] = ctl[2] = -1;
        ctl[1] = ctl[3] = ctl[4] = 0;
    }
    __syncthreads();
    const double scale = (double)SEL_NB / (hi - lo);
    auto bin = [&](double v) { return min(max((int)((v - lo) * scale), 0), SEL_NB - 1); };
    for (int i = tid; i < nc; i += nt) atomicAdd(&hist[bin(cand[i])], 1);
    __syncthreads();
    const int BPT = (SEL_NB + nt - 1) / nt, b0 = tid * BPT;
    int local = 0;
    for (int u = 0; u < BPT; ++u)
        if (b0 + u < SEL_NB) local += hist[b0 + u];
    int tot;
    int run = block_exscan_int(local, reinterpret_cast<int *>(sh), &tot);
    for (int u = 0; u < BPT; ++u)
        if (b0 + u < SEL_NB) {
            const int c = hist[b0 + u];
            if (qa >= run && qa < run + c) {
                ctl[0] = b0 + u;
                ctl[1] = run;
            }
            if (qb >= run && qb < run + c) {
                ctl[2] = b0 + u;
                ctl[3] = run;
            }
            run += c;
        }
    __syncthreads();
    const int ba = ctl[0], bb = ctl[2];
    for (int i = tid; i < nc; i += nt) {
        const double v = cand[i];
        const int b = bin(v);
        if (b == ba || b == bb) {
            const int slot = atomicAdd(&ctl[4], 1);
            if (slot < SEL_LIST) list[slot] = v;
        }
    }
    __syncthreads();
    const int m = ctl[4];
    if (m > SEL_LIST || (qa >= 0 && ba < 0) || (qb >= 0 && bb < 0)) {
        __syncthreads();
        return false;
    }
    if (tid < m) {
        const double v = list[tid];
        const int b = bin(v);
        int r = (b == ba) ? ctl[1] : ctl[3];
        for (int u = 0; u < m; ++u) {
            const double w = list[u];
            if (bin(w) == b && (w < v || (w == v && u < tid))) ++r;
        }
        if (r == qa) outv[0] = v;
        if (r == qb) outv[1] = v;
    }
    __syncthreads();
    if (qa >= 0) *va = outv[0];
    if (qb >= 0) *vb = outv[1];
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------------------------------------ sampled selection
// k-th smallest of the kept values in about ONE pass over the data instead of the eight of block_select_kth
// (Floyd-Rivest style): a strided sample of <= SEL_SAMPLE kept values is sorted in LDS, two pivots lo <= hi bracket
// the wanted rank, one pass counts the values below lo / equal to lo / equal to hi and collects the ones strictly
// between into `cand` (LDS, `cap` doubles); the answer is then found inside the candidates.  If the bracket misses
// or overflows — heavy-tailed sample luck, adversarial order — the full radix select runs instead, so the result is
// ALWAYS the exact order statistic.  count = number of kept values (> k).  Also serves rank k + 1 (for the median of
// an even count) from the same pass: *next receives it when want_next.
constexpr int SEL_SAMPLE = 1024;

// A side computation that rides on the ONE pass over all values (the bracket's collect pass): side(i, v, lo) sees every kept
// value v = val(i) together with `lo`, a lower bound of the order statistic being selected (the bracket's lower pivot).
// *side_ran tells the caller whether that pass happened (the all-in-LDS and fallback routes do not run it).
struct NoSide {
    __device__ __forceinline__ void operator()(int, double, double) const {}
};

template <class Val, class Keep, class Side = NoSide>
__device__ double block_select_sampled(int n, long long count, long long k, Val val, Keep keep, unsigned long long *sh,
                                       double *cand, int cap, bool want_next, double *next, int dbg = -1,
                                       double *spacing = nullptr,  // *spacing: mean gap between values around rank k (0 = unknown)
                                       Side side = Side(), bool *side_ran = nullptr) {
    if (spacing) *spacing = 0.0;
    if (side_ran) *side_ran = false;
    const int tid = threadIdx.x, nt = blockDim.x;
    int *ictl = reinterpret_cast<int *>(sh + 200);  // [0] ncand, [1] sample size   (sh[0..199] are used by the callees)
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(cand);
    auto fallback = [&]() -> double {
        const double a = block_select_kth_cold(n, k, val, keep, sh);
        if (want_next) *next = (k + 1 < count) ? block_select_kth_cold(n, k + 1, val, keep, sh) : a;
        return a;
    };
    auto lds_val = [&](int i) { return cand[i]; };
    auto lds_all = [&](int) { return true; };
    // ranks q (and q + 1 if want_next) among the nc candidates sitting in cand[]: sort them in place (as sortable keys,
    // padded with +max keys to a power of two) when the padded size fits, else two radix selects
    auto sorted_ranks = [&](int nc, long long q) -> double {
        int S2 = 2;
        while (S2 < nc) S2 <<= 1;
        if (S2 > cap) {
            const double a = block_select_kth(nc, q, lds_val, lds_all, sh);
            if (want_next) *next = (q + 1 < nc) ? block_select_kth(nc, q + 1, lds_val, lds_all, sh) : a;
            return a;
        }
        for (int i = tid; i < S2; i += nt) keys[i] = i < nc ? f64_sortable(cand[i]) : ~0ull;
        __syncthreads();
        lds_bitonic_sort(keys, S2);
        const double a = f64_from_sortable(keys[q]);
        if (want_next) *next = (q + 1 < nc) ? f64_from_sortable(keys[q + 1]) : a;
        __syncthreads();
        return a;
    };
    if (cap < 2 * SEL_SAMPLE) return fallback();
    if (count <= (long long)min(cap, 2 * SEL_SAMPLE)) {
        // everything fits a 2048-key sort: collect once, select in LDS (between that and `cap` values the sampled route
        // below — 1024-key sample sort, ~15 % of the values collected, histogram — beats collecting and sorting them all)
        if (tid == 0) ictl[0] = 0;
        __syncthreads();
        strided_pass<8>(n, [&](int i) { return val(i); }, [&](int i, double v) {
            if (keep(i)) cand[atomicAdd(&ictl[0], 1)] = v;
        });
        __syncthreads();
        const int nc = ictl[0];
        // ranks k (and k + 1) by the histogram over [min, max] of the values when there is room for it behind them
        {
            double mn = INFINITY, mx = -INFINITY;
            for (int i = tid; i < nc; i += nt) {
                mn = fmin(mn, cand[i]);
                mx = fmax(mx, cand[i]);
            }
            for (int o = 32; o > 0; o >>= 1) {
                mn = fmin(mn, __shfl_xor(mn, o));
                mx = fmax(mx, __shfl_xor(mx, o));
            }
            double *red = reinterpret_cast<double *>(sh);
            __syncthreads();
            if ((tid & 63) == 0) {
                red[2 * (tid >> 6)] = mn;
                red[2 * (tid >> 6) + 1] = mx;
            }
            __syncthreads();
            for (int w = 0; w < ((nt + 63) >> 6); ++w) {
                mn = fmin(mn, red[2 * w]);
                mx = fmax(mx, red[2 * w + 1]);
            }
            __syncthreads();
            const bool nb = want_next && k + 1 < (long long)nc;
            double hva = 0.0, hvb = 0.0;
            if (k < (long long)nc && lds_hist_select(cand, nc, cap, (int)k, nb ? (int)k + 1 : -1, mn, mx, sh, &hva, &hvb)) {
                if (want_next) *next = nb ? hvb : hva;
                return hva;
            }
        }
        return sorted_ranks(nc, k);
    }
    // ---- strided sample -> keys[0..S), padded with +inf keys to a power of two, bitonic sort
    const int S = SEL_SAMPLE;
    for (int j = tid; j < S; j += nt) {
        const int i = (int)(((long long)j * n) / S);
        keys[j] = keep(i) ? f64_sortable(val(i)) : ~0ull;
    }
    __syncthreads();
    if (dbg == 0) return 0.0;  // (profiling aid: stop after the sample gather / sample sort / collect pass / counts)
    lds_bitonic_sort(keys, S);
    if (dbg == 1) return 0.0;
    // sample size = number of non-padding keys (the padding sorts to the end)
    {
        int c = 0;
        for (int j = tid; j < S; j += nt) c += keys[j] != ~0ull ? 1 : 0;
        const long long s_all = block_count_fast(c, reinterpret_cast<long long *>(sh));
        if (s_all < 64) return fallback();  // (uniform: every thread sees the same s_all)
        const double pos = ((double)k + 0.5) * (double)s_all / (double)count;
        // half-width of the bracket in sample ranks: the rank of the wanted value among a random sample scatters by
        // sigma = sqrt(s_all) / 2 around pos.  4 sigma (a miss — the full radix select, ~300 us — once in ~16 000 selects: with
        // one light curve per workgroup and all of them in one wave of workgroups, ONE miss is what the whole launch waits
        // for; at the 2.9 sigma this used before every launch of 1000 selects had a few), never below 2.65 sigma, and not
        // more than ~2600 expected candidates: they are ranked by a histogram, not sorted (lds_hist_select).
        const double sq = sqrt((double)s_all);
        const int delta = (int)fmin(2.0 * sq + 6.0, fmax(1.2 * sq + 4.0, 2600.0 * (double)s_all / (2.0 * (double)count)));
        const int r_lo = (int)pos - delta, r_hi = (int)pos + delta;
        // pivots (registers, same in every thread): -inf / +inf when the bracket runs off the sample
        const double lo = r_lo < 0 ? -INFINITY : f64_from_sortable(keys[r_lo]);
        const double hi = r_hi >= (int)s_all ? INFINITY : f64_from_sortable(keys[r_hi]);
        __syncthreads();  // the sample (aliasing cand) is dead from here on
        if (tid == 0) ictl[0] = 0;
        __syncthreads();
        long long c_less = 0, c_eqlo = 0, c_eqhi = 0;
        strided_pass<8>(n, [&](int i) { return val(i); }, [&](int i, double v) {
            if (keep(i)) {
                side(i, v, lo);
                if (v < lo)
                    ++c_less;
                else if (v == lo)
                    ++c_eqlo;
                else if (v < hi) {
                    const int slot = atomicAdd(&ictl[0], 1);
                    if (slot < cap) cand[slot] = v;
                } else if (v == hi)
                    ++c_eqhi;
            }
        });
        if (side_ran) *side_ran = true;
        if (dbg == 2) return 0.0;
        long long n_less = block_count_fast(c_less, reinterpret_cast<long long *>(sh));
        long long n_eqlo = block_count_fast(c_eqlo, reinterpret_cast<long long *>(sh));
        long long n_eqhi = block_count_fast(c_eqhi, reinterpret_cast<long long *>(sh));
        __syncthreads();
        int nc = ictl[0];
        if (spacing && nc > 0 && isfinite(lo) && isfinite(hi)) *spacing = (hi - lo) / (double)nc;
        double slo = lo, shi_v = hi;  // the bracket the candidates in cand[] lie strictly inside
        if (nc > cap) {
            // More values inside the bracket than the LDS list holds: its expected content grows as n / sqrt(sample), past
            // `cap` beyond ~50 000 values (a stitched multi-sector light curve).  One more pass bins the bracket's values
            // (SEL_NB counters over the linear map of (lo, hi), which is monotone: a lower bin holds smaller values), the bins
            // that hold the wanted ranks become the new bracket, and a last pass collects just those — three passes in all
            // where the radix select behind fallback() takes eight per rank.
            const long long q0 = k - n_less - n_eqlo;
            const bool wa = q0 >= 0 && q0 < (long long)nc, wb = want_next && k + 1 < count && q0 + 1 >= 0 && q0 + 1 < (long long)nc;
            if (!(wa || wb) || !isfinite(lo) || !isfinite(hi) || !(hi > lo) || cap < SEL_NB) return fallback();
            const long long r_first = wa ? q0 : q0 + 1, r_last = wb ? q0 + 1 : q0;
            int *hist = reinterpret_cast<int *>(cand);
            int *ctl = reinterpret_cast<int *>(sh + 210);  // [0] first bin, [1] values below it, [2] last bin, [3] values through it
            __syncthreads();
            for (int i = tid; i < SEL_NB; i += nt) hist[i] = 0;
            if (tid == 0) ctl[0] = ctl[2] = -1;
            __syncthreads();
            const double scale = (double)SEL_NB / (hi - lo);
            auto bin = [&](double v) { return min(max((int)((v - lo) * scale), 0), SEL_NB - 1); };
            strided_pass<8>(n, [&](int i) { return val(i); }, [&](int i, double v) {
                if (keep(i) && v > lo && v < hi) atomicAdd(&hist[bin(v)], 1);
            });
            __syncthreads();
            const int BPT = (SEL_NB + nt - 1) / nt, b0 = tid * BPT;
            int local = 0;
            for (int u = 0; u < BPT; ++u)
                if (b0 + u < SEL_NB) local += hist[b0 + u];
            int tot;
            int run = block_exscan_int(local, reinterpret_cast<int *>(sh), &tot);
            for (int u = 0; u < BPT; ++u)
                if (b0 + u < SEL_NB) {
                    const int c = hist[b0 + u];
                    if (r_first >= run && r_first < run + c) {
                        ctl[0] = b0 + u;
                        ctl[1] = run;
                    }
                    if (r_last >= run && r_last < run + c) {
                        ctl[2] = b0 + u;
                        ctl[3] = run + c;
                    }
                    run += c;
                }
            __syncthreads();
            const int ba = ctl[0], bb = ctl[2], below = ctl[1], through = ctl[3];
            __syncthreads();  // (hist aliases cand: every thread has its bins before the list is rebuilt)
            if (ba < 0 || bb < ba || tot != nc || through - below > cap) return fallback();
            if (tid == 0) ictl[0] = 0;
            __syncthreads();
            strided_pass<8>(n, [&](int i) { return val(i); }, [&](int i, double v) {
                if (keep(i) && v > lo && v < hi) {
                    const int b = bin(v);
                    if (b >= ba && b <= bb) {
                        const int slot = atomicAdd(&ictl[0], 1);
                        if (slot < cap) cand[slot] = v;
                    }
                }
            });
            __syncthreads();
            if (ictl[0] != through - below) return fallback();
            // the values below the sub-bracket now count as "less"; its edges only steer the next histogram (clamped bins)
            n_less += n_eqlo + below;
            n_eqlo = 0;
            n_eqhi = 0;
            nc = through - below;
            slo = lo + (double)ba / scale;
            shi_v = lo + (double)(bb + 1) / scale;
        }
        if (dbg == 3) return 0.0;
        // ranks k and k + 1 relative to the candidates; if they fall among them: histogram select, else (and when that does
        // not apply) the candidates are sorted once (keys[] aliases cand[]) and rank lookups are plain LDS reads
        const long long qa = k - n_less - n_eqlo, qb = qa + 1;
        const bool need_a = qa >= 0 && qa < (long long)nc;
        const bool need_b = want_next && k + 1 < count && qb >= 0 && qb < (long long)nc;
        double hva = 0.0, hvb = 0.0;
        const bool hist_ok = (need_a || need_b) &&
                             lds_hist_select(cand, nc, cap, need_a ? (int)qa : -1, need_b ? (int)qb : -1, slo, shi_v, sh, &hva, &hvb);
        int S2 = 2;
        while (S2 < nc) S2 <<= 1;
        const bool sorted = !hist_ok && (need_a || need_b) && S2 <= cap;
        if (sorted) {
            for (int i = tid; i < S2; i += nt) keys[i] = i < nc ? f64_sortable(cand[i]) : ~0ull;
            __syncthreads();
            lds_bitonic_sort(keys, S2);
        }
        if (dbg == 4) return 0.0;
        // rank r (0-based among all kept) -> value, or "miss"
        bool miss = false;
        auto at_rank = [&](long long r) -> double {
            long long q = r - n_less;
            if (q < 0) {
                miss = true;
                return 0.0;
            }
            if (q < n_eqlo) return lo;
            q -= n_eqlo;
            if (q < nc) {
                if (hist_ok) return q == qa ? hva : hvb;  // (q is relative to the candidates here: qa or qa + 1)
                return sorted ? f64_from_sortable(keys[q]) : block_select_kth(nc, q, lds_val, lds_all, sh);
            }
            q -= nc;
            if (q < n_eqhi) return hi;
            miss = true;
            return 0.0;
        };
        // (every branch above depends only on workgroup-uniform values, so the barriers inside block_select_kth are safe)
        const double a = at_rank(k);
        if (dbg == 5) return 0.0;
        double b = a;
        if (!miss && want_next && k + 1 < count) b = at_rank(k + 1);
        if (dbg == 6) return 0.0;
        __syncthreads();  // every thread has read its ranks before cand[] / keys[] are reused by the caller
        if (dbg == 7) return 0.0;
#ifdef LK_SEL_DEBUG
        if (tid == 0 && (blockIdx.x < 2 || miss))
            printf("[sel] blk %d n %d count %lld k %lld s_all %lld delta %d lo %.17g hi %.17g less %lld eqlo %lld nc %d eqhi %lld miss %d a %.17g\n",
                   blockIdx.x, n, count, k, s_all, delta, lo, hi, n_less, n_eqlo, nc, n_eqhi, (int)miss, a);
#endif
        if (miss) return fallback();
        if (want_next) *next = b;
        return a;
    }
}

// numpy.median of the kept values through block_select_sampled; NaN if none kept.
template <class Val, class Keep, class Side = NoSide>
__device__ double block_median_sampled(int n, long long count, Val val, Keep keep, unsigned long long *sh, double *cand,
                                       int cap, int dbg = -1, double *spacing = nullptr, Side side = Side(),
                                       bool *side_ran = nullptr) {
    if (spacing) *spacing = 0.0;
    if (side_ran) *side_ran = false;
    if (count <= 0) return __longlong_as_double(0x7ff8000000000000ll);
    const long long k = (count - 1) / 2;
    double nxt = 0.0;
    const double a = block_select_sampled(n, count, k, val, keep, sh, cand, cap, (count & 1) == 0, &nxt, dbg, spacing, side,
                                          side_ran);
    return (count & 1) ? a : (a + nxt) * 0.5;
}

// numpy.median of the kept values when a good GUESS is at hand (the median of nearly the same set, e.g. the previous
// clipping iteration's): ONE pass counts the values below guess - width and collects those inside [guess - width, guess +
// width]; if the two middle ranks fall among the collected values they are sorted (a few hundred keys) and the answer is
// exact.  Otherwise *ok = false (every thread) and the caller runs block_median_sampled.  No sample, no 2048-key sort.
template <class Val, class Keep, class Side = NoSide>
__device__ double block_median_near(int n, long long count, Val val, Keep keep, double guess, double width,
                                    unsigned long long *sh, double *cand, int cap, bool *ok, Side side = Side()) {
    const int tid = threadIdx.x, nt = blockDim.x;
    *ok = false;
    if (count <= 0 || !(width > 0.0) || !isfinite(guess)) return 0.0;
    int *ictl = reinterpret_cast<int *>(sh + 200);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(cand);
    const long long k = (count - 1) / 2;
    const bool want_next = (count & 1) == 0;
    const double lo = guess - width, hi = guess + width;
    __syncthreads();
    if (tid == 0) ictl[0] = 0;
    __syncthreads();
    long long c_less = 0;
    strided_pass<8>(n, [&](int i) { return val(i); }, [&](int i, double v) {
        if (keep(i)) {
            side(i, v, lo);   // (valid only if *ok comes back true: then the median is >= lo)
            if (v < lo)
                ++c_less;
            else if (v <= hi) {
                const int slot = atomicAdd(&ictl[0], 1);
                if (slot < cap) cand[slot] = v;
            }
        }
    });
    const long long n_less = block_count_fast(c_less, reinterpret_cast<long long *>(sh));
    __syncthreads();
    const int nc = ictl[0];
    const long long q = k - n_less;
    int S2 = 2;
    while (S2 < nc) S2 <<= 1;
    if (S2 < nt && (nt & (nt - 1)) == 0) S2 = nt;  // the one-key-per-thread register sort
    const bool good = nc <= cap && S2 <= cap && q >= 0 && q + (want_next ? 1 : 0) < (long long)nc;
    __syncthreads();
    if (!good) return 0.0;
    for (int i = tid; i < S2; i += nt) keys[i] = i < nc ? f64_sortable(cand[i]) : ~0ull;
    __syncthreads();
    lds_bitonic_sort(keys, S2);
    const double a = f64_from_sortable(keys[q]);
    const double b = want_next ? f64_from_sortable(keys[q + 1]) : a;
    __syncthreads();
    *ok = true;
    return want_next ? (a + b) * 0.5 : a;
}

}  // namespace lk
