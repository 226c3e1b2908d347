// lsfast.hip — lightkurve's DEFAULT Lomb-Scargle method (ls_method="fast") on gfx950: the Press & Rybicki
// extirpolation + FFT evaluation of the trig sums, then the same closed-form GLS as the exact kernels.
//
// Reference arithmetic (followed step by step so results agree with the reference's 'fast' output to ~1e-10,
// where 'fast' itself is ~1e-3 of the peak away from the exact methods):
//   astropy lombscargle/implementations/fast_impl.py:74-131   (weights, centring, three trig_sum calls, closed form)
//   astropy lombscargle/implementations/utils.py:81-158        (trig_sum: Nfft = bitceil(5 Nf), phase factor for f0>0,
//                                                               tnorm, ifft, Nfft scaling)
//   astropy lombscargle/implementations/utils.py:14-78         (extirpolate: 4-point Lagrange spreading)
// called by lightkurve at src/lightkurve/periodogram.py:961-964 with method='fast' (the default, :650).
//
// This is the HBM-bound formulation of the path: per target 3 complex grids of Nfft = 2^19 points (8 MB each) are
// zeroed, filled by atomics, transformed by a hand-written four-step FFT (Nfft = N1 x N2: column FFTs of length N1
// with the inter-step twiddles, then row FFTs of length N2, both in LDS, in-place radix-2 on bit-reversed loads)
// and reduced to M powers.  Algorithmic HBM traffic per target: grids 3 x (zero 8 MB + 2 passes x (read + write)
// 8 MB) = 120 MB, + 16 B/cadence in and 8 B/frequency out.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "lk_common.hpp"
#include "ls_epilogue.hpp"

namespace lk {

constexpr int SPREAD_W_C = 1024;  // cells owned by one workgroup of lsf_spread_owner_kernel

struct FastStats {
    double wsum, ybar, YY, t0;
    double yws, pad;  // sum w (y - ybar) (the bias entry of X^T y in the multi-term solve)
};

// per target: weights, mean about y[0], YY, t0 = min t; w[i] (normalised) and wy[i] = w (y - ybar)
__global__ __launch_bounds__(256) void lsf_prep_kernel(const double *__restrict__ t, const double *__restrict__ y,
                                                        const double *__restrict__ dy,
                                                        const int64_t *__restrict__ n_off, int center,
                                                        double *__restrict__ w_out, double *__restrict__ wy_out,
                                                        FastStats *__restrict__ stats, double df, int nfft, int m2,
                                                        int *__restrict__ rows_used, int *__restrict__ spread_tab,
                                                        int ntab, int *__restrict__ tab16 = nullptr, int ntab16 = 0) {
    __shared__ double sh[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = n_off[b], n = n_off[b + 1] - lo;
    auto bsum = [&](double x) {
        sh[tid] = x;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) sh[tid] += sh[tid + s];
            __syncthreads();
        }
        const double r = sh[0];
        __syncthreads();
        return r;
    };
    double acc = 0.0, tmin = INFINITY, tmax = -INFINITY;
    for (int64_t i = tid; i < n; i += 256) {
        if (dy) {
            const double d = dy[lo + i];
            acc += 1.0 / (d * d);
        }
        tmin = fmin(tmin, t[lo + i]);
        tmax = fmax(tmax, t[lo + i]);
    }
    const double wsum = dy ? bsum(acc) : (double)n;
    sh[tid] = tmin;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = fmin(sh[tid], sh[tid + s]);
        __syncthreads();
    }
    const double t0 = sh[0];
    __syncthreads();
    sh[tid] = tmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = fmax(sh[tid], sh[tid + s]);
        __syncthreads();
    }
    const double t1 = sh[0];
    __syncthreads();
    if (tid < 3 && rows_used) {
        // grid rows (of N2 cells) that can receive a sample: cells <= tnorm_max + 1; everything wraps if the span
        // reaches Nfft.  Grids 0, 1 use df, grid 2 uses 2 df.
        const double span = (t1 - t0) * (double)nfft * df * (tid == 2 ? 2.0 : 1.0);
        const int nrows = nfft >> m2;
        rows_used[b * 4 + tid] = (span >= (double)nfft - 8.0) ? nrows : min(nrows, (int)((span + 4.0) / (double)(1 << m2)) + 1);
    }
    if (rows_used) {
        // "ordered" targets (time sorted, no wrap of the 2 df grid): grid positions are monotone in the cadence
        // index, so the spreading kernel can own cell ranges and use plain stores instead of global atomics
        int unsorted = 0;
        for (int64_t i = tid; i + 1 < n; i += 256) unsorted |= (t[lo + i + 1] < t[lo + i]) ? 1 : 0;
        const int any_unsorted = __syncthreads_or(unsorted);
        const bool nowrap = (t1 - t0) * (double)nfft * df * 2.0 < (double)nfft - 8.0;
        if (tid == 0) rows_used[b * 4 + 3] = (!any_unsorted && nowrap) ? 1 : 0;
    }
    if (spread_tab) {
        // Search tables for the owner-computes spreader (ordered targets: grid positions grow with the cadence index).
        // Per grid g and 1024-cell block k:  lo_tab[k] = first cadence with position >= k W - 4,  hi_tab[k] = first
        // cadence with position >= k W + 3 — what the spreader used to find with two block-wide binary searches per
        // workgroup.  Every cadence fills the thresholds that fall between its predecessor's position and its own.
        int *tab = spread_tab + (size_t)b * 6 * ntab;
        const double W = (double)SPREAD_W_C;
        for (int g = 0; g < 3; ++g) {
            const double dff = df * (g == 2 ? 2.0 : 1.0);
            int *lo_tab = tab + (size_t)(2 * g) * ntab, *hi_tab = lo_tab + ntab;
            for (int64_t i = tid; i < n; i += 256) {
                const double p = fmod((t[lo + i] - t0) * (double)nfft * dff, (double)nfft);
                const double pp = i > 0 ? fmod((t[lo + i - 1] - t0) * (double)nfft * dff, (double)nfft) : -1e300;
                // thresholds x_k = k W - 4 with pp < x_k <= p
                long long k0 = i > 0 ? (long long)floor((pp + 4.0) / W) + 1 : 0, k1 = (long long)floor((p + 4.0) / W);
                for (long long k = max(k0, 0ll); k <= min(k1, (long long)ntab - 1); ++k) lo_tab[k] = (int)i;
                k0 = i > 0 ? (long long)floor((pp - 3.0) / W) + 1 : 0;
                k1 = (long long)floor((p - 3.0) / W);
                for (long long k = max(k0, 0ll); k <= min(k1, (long long)ntab - 1); ++k) hi_tab[k] = (int)i;
            }
            // thresholds beyond the last cadence
            const double pl = fmod((t[lo + n - 1] - t0) * (double)nfft * dff, (double)nfft);
            for (int k = tid; k < ntab; k += 256) {
                if ((double)k * W - 4.0 > pl) lo_tab[k] = (int)n;
                if ((double)k * W + 3.0 > pl) hi_tab[k] = (int)n;
            }
        }
    }
    if (tab16) {
        // Fine table for the extirpolation fused into the pruned column kernel: per grid g and 16-cell group G,
        // tab16[G] = first cadence with position >= 16 G - 4 (ordered targets).  A tile row then knows its cadences without
        // any search: [tab16[G], tab16[G + 2]) covers every stencil that reaches cells [16 G, 16 G + 16).
        int *tb = tab16 + (size_t)b * 3 * ntab16;
        for (int g = 0; g < 3; ++g) {
            const double dff = df * (g == 2 ? 2.0 : 1.0);
            int *tg = tb + (size_t)g * ntab16;
            for (int64_t i = tid; i < n; i += 256) {
                const double p = fmod((t[lo + i] - t0) * (double)nfft * dff, (double)nfft);
                const double pp = i > 0 ? fmod((t[lo + i - 1] - t0) * (double)nfft * dff, (double)nfft) : -1e300;
                const long long k0 = i > 0 ? (long long)floor((pp + 4.0) / 16.0) + 1 : 0, k1 = (long long)floor((p + 4.0) / 16.0);
                for (long long k = max(k0, 0ll); k <= min(k1, (long long)ntab16 - 1); ++k) tg[k] = (int)i;
            }
            const double pl = fmod((t[lo + n - 1] - t0) * (double)nfft * dff, (double)nfft);
            for (int k = tid; k < ntab16; k += 256)
                if ((double)k * 16.0 - 4.0 > pl) tg[k] = (int)n;
        }
    }
    const double y0 = y[lo];
    double ybar = 0.0;
    if (center) {
        acc = 0.0;
        for (int64_t i = tid; i < n; i += 256) {
            const double d = dy ? dy[lo + i] : 1.0;
            acc = fma((1.0 / (d * d)) / wsum, y[lo + i] - y0, acc);
        }
        ybar = bsum(acc) + y0;
    }
    acc = 0.0;
    for (int64_t i = tid; i < n; i += 256) {
        const double d = dy ? dy[lo + i] : 1.0;
        const double w = (1.0 / (d * d)) / wsum;
        const double yc = y[lo + i] - ybar;
        acc = fma(w * yc, yc, acc);
        w_out[lo + i] = w;
        wy_out[lo + i] = w * yc;
    }
    const double YY = bsum(acc);
    acc = 0.0;
    for (int64_t i = tid; i < n; i += 256) acc += wy_out[lo + i];  // each thread re-reads what it wrote
    const double yws = bsum(acc);
    if (tid == 0) stats[b] = FastStats{wsum, ybar, YY, t0, yws, 0.0};
}

// astropy extirpolate (M = 4) of one complex sample h at position x into grid[0..nfft)
__device__ __forceinline__ void extirpolate4(double2 *__restrict__ grid, int nfft, double x, double hr, double hi) {
    if (fmod(x, 1.0) == 0.0) {
        const int i = (int)x;
        unsafeAtomicAdd(&grid[i].x, hr);
        unsafeAtomicAdd(&grid[i].y, hi);
        return;
    }
    int ilo = (int)(x - 2.0);  // astype(int): truncation toward zero
    ilo = min(max(ilo, 0), nfft - 4);
    const double d0 = x - (double)ilo, d1 = d0 - 1.0, d2 = d0 - 2.0, d3 = d0 - 3.0;
    // d1..d3 as the reference forms them: x - ilo - k (same value: ilo + k is exact in double)
    const double prod = ((d0 * d1) * d2) * d3;
    const double nr = hr * prod, ni = hi * prod;
    // j = 0..3: ind = ilo + 3 - j, denominators 6, -2, 2, -6
    const double q3 = 6.0 * d3, q2 = -2.0 * d2, q1 = 2.0 * d1, q0 = -6.0 * d0;
    unsafeAtomicAdd(&grid[ilo + 3].x, nr / q3);
    unsafeAtomicAdd(&grid[ilo + 3].y, ni / q3);
    unsafeAtomicAdd(&grid[ilo + 2].x, nr / q2);
    unsafeAtomicAdd(&grid[ilo + 2].y, ni / q2);
    unsafeAtomicAdd(&grid[ilo + 1].x, nr / q1);
    unsafeAtomicAdd(&grid[ilo + 1].y, ni / q1);
    unsafeAtomicAdd(&grid[ilo].x, nr / q0);
    unsafeAtomicAdd(&grid[ilo].y, ni / q0);
}

// spread every cadence of targets [b0, b0 + nb) into its three grids: 0: w*y at f, 1: w at f, 2: w at 2f
__global__ __launch_bounds__(256) void lsf_scatter_kernel(const double *__restrict__ t, const double *__restrict__ w,
                                                           const double *__restrict__ wy,
                                                           const int64_t *__restrict__ n_off,
                                                           const FastStats *__restrict__ stats, int b0, double f0,
                                                           double df, int nfft, int fit_mean,
                                                           double2 *__restrict__ grids,
                                                           const int *__restrict__ rows_used) {
    if (rows_used && rows_used[blockIdx.y * 4 + 3]) return;  // ordered target: handled by lsf_spread_owner_kernel
    const int b = b0 + blockIdx.y;
    const int64_t lo = n_off[b];
    const int n = (int)(n_off[b + 1] - lo);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double tt = t[lo + i] - stats[b].t0;
    double2 *g0 = grids + (size_t)blockIdx.y * 3 * nfft, *g1 = g0 + nfft, *g2 = g1 + nfft;
    const double wi = w[lo + i], wyi = wy[lo + i];
    const double twopi = 6.283185307179586;
    for (int fac = 1; fac <= 2; ++fac) {
        const double dff = df * (double)fac, f0f = f0 * (double)fac;
        double c = 1.0, s = 0.0;
        if (f0f > 0.0) sincos(twopi * f0f * tt, &s, &c);
        const double tn = fmod(tt * (double)nfft * dff, (double)nfft);
        if (fac == 1) {
            extirpolate4(g0, nfft, tn, wyi * c, wyi * s);
            if (fit_mean) extirpolate4(g1, nfft, tn, wi * c, wi * s);
        } else {
            extirpolate4(g2, nfft, tn, wi * c, wi * s);
        }
    }
}

// Owner-computes spreading for ordered targets (sorted time, no wrap): grid positions grow with the cadence index,
// so workgroup (x, target, g) owns cells [x W, (x+1) W) of grid g, finds the cadences whose 4-point stencils reach
// them by two block-wide probes, accumulates in LDS (ds_add_f64) and writes its cells once with plain, coalesced
// stores — zeros included, so no memset and no global atomics.  g: 0 = w*y at f, 1 = w at f, 2 = w at 2 f.
constexpr int SPREAD_W = SPREAD_W_C;

__global__ __launch_bounds__(256) void lsf_spread_owner_kernel(const double *__restrict__ t, const double *__restrict__ w,
                                                                const double *__restrict__ wy,
                                                                const int64_t *__restrict__ n_off,
                                                                const FastStats *__restrict__ stats, int b0, double f0,
                                                                double df, int nfft, int m2, int fit_mean,
                                                                double2 *__restrict__ grids,
                                                                const int *__restrict__ rows_used,
                                                                const int *__restrict__ spread_tab, int ntab) {
    __shared__ double2 acc[SPREAD_W];
    const int lb = blockIdx.y, g = blockIdx.z, tid = threadIdx.x;
    if (!rows_used[lb * 4 + 3]) return;
    const int ncell = rows_used[lb * 4 + g] << m2;  // cells the column transform will read
    const int c_lo = blockIdx.x * SPREAD_W, c_hi = min(c_lo + SPREAD_W, ncell);
    if (c_lo >= ncell) return;
    double2 *G = grids + ((size_t)lb * 3 + g) * (size_t)nfft;
    if (g == 1 && !fit_mean) {  // unused grid: keep it defined
        for (int c = c_lo + tid; c < c_hi; c += 256) G[c] = make_double2(0.0, 0.0);
        return;
    }
    const int b = b0 + lb;
    const int64_t lo = n_off[b];
    const int n = (int)(n_off[b + 1] - lo);
    t += lo;
    const double t0 = stats[b].t0;
    const double fac = g == 2 ? 2.0 : 1.0;
    const double dff = df * fac, f0f = f0 * fac;
    const double *amp = (g == 0 ? wy : w) + lo;
    auto pos = [&](int i) { return fmod((t[i] - t0) * (double)nfft * dff, (double)nfft); };
    // first cadence with pos >= x (positions are non-decreasing): 256-way probes until the bracket fits one pass
    auto lower = [&](double x) -> int {
        int lo_i = 0, hi_i = n;  // answer in [lo_i, hi_i]
        while (hi_i - lo_i > 256) {
            const int stride = (hi_i - lo_i + 255) / 256;
            const int ip = lo_i + tid * stride;
            const int cnt = __syncthreads_count(ip < hi_i && pos(ip) < x);  // probes below x form a prefix
            if (cnt == 0) {
                hi_i = lo_i;  // even the first element is >= x
                break;
            }
            const int nlo = lo_i + (cnt - 1) * stride + 1;  // just after the last probe below x
            hi_i = min(lo_i + cnt * stride, hi_i);
            lo_i = nlo;
        }
        const int i2 = lo_i + tid;
        const int cnt2 = __syncthreads_count(i2 < hi_i && pos(i2) < x);
        return lo_i + cnt2;
    };
    int i_lo, i_hi;
    if (spread_tab) {  // precomputed by lsf_prep_kernel: no searching
        const int *tab = spread_tab + ((size_t)b * 6 + 2 * g) * ntab;
        i_lo = tab[blockIdx.x];
        i_hi = tab[ntab + min((int)blockIdx.x + 1, ntab - 1)];
    } else {
        i_lo = lower((double)c_lo - 4.0);
        i_hi = lower((double)c_hi + 3.0);
    }
    for (int c = tid; c < SPREAD_W; c += 256) acc[c] = make_double2(0.0, 0.0);
    __syncthreads();
    const double twopi = 6.283185307179586;
    auto add = [&](int cell, double vr, double vi) {
        if (cell >= c_lo && cell < c_hi) {
            unsafeAtomicAdd(&acc[cell - c_lo].x, vr);
            unsafeAtomicAdd(&acc[cell - c_lo].y, vi);
        }
    };
    for (int i = i_lo + tid; i < i_hi; i += 256) {
        const double tt = t[i] - t0;
        double c = 1.0, s = 0.0;
        if (f0f > 0.0) sincos(twopi * f0f * tt, &s, &c);
        const double x = fmod(tt * (double)nfft * dff, (double)nfft);
        const double hr = amp[i] * c, hi = amp[i] * s;
        if (fmod(x, 1.0) == 0.0) {
            add((int)x, hr, hi);
        } else {
            int ilo = (int)(x - 2.0);
            ilo = min(max(ilo, 0), nfft - 4);
            const double d0 = x - (double)ilo, d1 = d0 - 1.0, d2 = d0 - 2.0, d3 = d0 - 3.0;
            const double prod = ((d0 * d1) * d2) * d3;
            const double nr = hr * prod, ni = hi * prod;
            const double q3 = 6.0 * d3, q2 = -2.0 * d2, q1 = 2.0 * d1, q0 = -6.0 * d0;
            add(ilo + 3, nr / q3, ni / q3);
            add(ilo + 2, nr / q2, ni / q2);
            add(ilo + 1, nr / q1, ni / q1);
            add(ilo, nr / q0, ni / q0);
        }
    }
    __syncthreads();
    for (int c = c_lo + tid; c < c_hi; c += 256) G[c] = acc[c - c_lo];
}

// ------------------------------------------------------------------------------------------------ four-step FFT
// In-place radix-2 DIT over nf transforms of length n = 2^m stored back to back in LDS (inputs already in
// bit-reversed order), e^{+2 pi i ...} kernel (numpy ifft without the 1/n).  tw[k] = e^{+2 pi i k / n}, k < n/2.
__device__ __forceinline__ void lds_fft_dit(double2 *x, const double2 *tw, int n, int m, int nf) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int halfn = n >> 1;
    for (int s = 1; s <= m; ++s) {
        const int half = 1 << (s - 1);
        const int tstep = n >> s;
        for (int p = tid; p < nf * halfn; p += nt) {
            const int f = p >> (m - 1), j = p & (halfn - 1);
            const int grp = j >> (s - 1), k = j & (half - 1);
            const int i0 = f * n + (grp << s) + k, i1 = i0 + half;
            const double2 wv = tw[k * tstep];
            const double2 a = x[i0], bq = x[i1];
            const double tr = wv.x * bq.x - wv.y * bq.y, ti = wv.x * bq.y + wv.y * bq.x;
            x[i0] = make_double2(a.x + tr, a.y + ti);
            x[i1] = make_double2(a.x - tr, a.y - ti);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void lds_twiddles(double2 *tw, int n) {
    for (int k = threadIdx.x; k < (n >> 1); k += blockDim.x) {
        double s, c;
        sincospi(2.0 * (double)k / (double)n, &s, &c);
        tw[k] = make_double2(c, s);
    }
}

// step 1: for CT columns, length-N1 transforms along r of x[r N2 + c], times the twiddle e^{2 pi i c k1 / N}, in place
__global__ __launch_bounds__(256) void fft_cols_kernel(double2 *__restrict__ grids, int m1, int m2, int CT) {
    extern __shared__ __attribute__((aligned(16))) double2 lds2[];
    const int N1 = 1 << m1, N2 = 1 << m2;
    double2 *x = lds2, *tw = lds2 + (size_t)CT * N1;
    double2 *G = grids + ((size_t)blockIdx.y << (m1 + m2));
    const int c0 = blockIdx.x * CT;
    lds_twiddles(tw, N1);
    for (int e = threadIdx.x; e < CT * N1; e += 256) {
        const int r = e / CT, cc = e - r * CT;
        const int rr = (int)(__brev((unsigned)r) >> (32 - m1));
        x[cc * N1 + rr] = G[(size_t)r * N2 + c0 + cc];
    }
    __syncthreads();
    lds_fft_dit(x, tw, N1, m1, CT);
    const double invN = 1.0 / (double)((size_t)1 << (m1 + m2));
    for (int e = threadIdx.x; e < CT * N1; e += 256) {
        const int k1 = e / CT, cc = e - k1 * CT;
        const long long ck = (long long)(c0 + cc) * k1;  // < N
        double s, c;
        sincospi(2.0 * (double)ck * invN, &s, &c);
        const double2 v = x[cc * N1 + k1];
        G[(size_t)k1 * N2 + c0 + cc] = make_double2(v.x * c - v.y * s, v.x * s + v.y * c);
    }
}

// step 2: for RT rows k1, length-N2 transforms along c; X[k1 + N1 k2] kept for k < nkeep into spec[g][k]
__global__ __launch_bounds__(256) void fft_rows_kernel(const double2 *__restrict__ grids, int m1, int m2, int RT,
                                                        int nkeep, double2 *__restrict__ spec) {
    extern __shared__ __attribute__((aligned(16))) double2 lds2[];
    const int N2 = 1 << m2;
    double2 *x = lds2, *tw = lds2 + (size_t)RT * N2;
    const double2 *G = grids + ((size_t)blockIdx.y << (m1 + m2));
    double2 *S = spec + (size_t)blockIdx.y * nkeep;
    const int r0 = blockIdx.x * RT;
    lds_twiddles(tw, N2);
    for (int e = threadIdx.x; e < RT * N2; e += 256) {
        const int rr = e / N2, c = e - rr * N2;
        const int cr = m2 ? (int)(__brev((unsigned)c) >> (32 - m2)) : 0;
        x[rr * N2 + cr] = G[(size_t)(r0 + rr) * N2 + c];
    }
    __syncthreads();
    lds_fft_dit(x, tw, N2, m2, RT);
    for (int e = threadIdx.x; e < RT * N2; e += 256) {
        const int k2 = e / RT, rr = e - k2 * RT;
        const long long k = (long long)(r0 + rr) + ((long long)k2 << m1);
        if (k < nkeep) S[k] = x[rr * N2 + k2];
    }
}

// ------------------------------------------------------------------------------------------------ register FFTs
// Faster four-step kernels for 2^4 <= N1, N2 <= 2^10: a length-n = A*Bq block transform is itself split in two:
// phase 1: Bq threads each run an A-point FFT entirely in registers (A <= 32), apply e^{2 pi i j ka / n} and park the
// result in LDS; phase 2: A threads each run a Bq-point register FFT over the transposed data.  One LDS write +
// one LDS read per point instead of log2(n) read-modify-write passes, two barriers instead of log2(n).
__device__ constexpr double R32C[16] = {1.0, 0.98078528040323043, 0.92387953251128674, 0.83146961230254524,
                                        0.70710678118654752, 0.55557023301960218, 0.38268343236508978,
                                        0.19509032201612825, 0.0, -0.19509032201612825, -0.38268343236508978,
                                        -0.55557023301960218, -0.70710678118654752, -0.83146961230254524,
                                        -0.92387953251128674, -0.98078528040323043};
__device__ constexpr double R32S[16] = {0.0, 0.19509032201612825, 0.38268343236508978, 0.55557023301960218,
                                        0.70710678118654752, 0.83146961230254524, 0.92387953251128674,
                                        0.98078528040323043, 1.0, 0.98078528040323043, 0.92387953251128674,
                                        0.83146961230254524, 0.70710678118654752, 0.55557023301960218,
                                        0.38268343236508978, 0.19509032201612825};

__host__ __device__ constexpr int brev_c(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// in-place radix-2 decimation-in-frequency FFT of 2^L points in registers, kernel e^{+2 pi i nk/2^L};
// the result is left in bit-reversed order: v[p] = X[brev(p)]
template <int L>
__device__ __forceinline__ void reg_fft(double2 (&v)[1 << L]) {
#pragma unroll
    for (int s = L; s >= 1; --s) {
        const int half = 1 << (s - 1);
#pragma unroll
        for (int g = 0; g < (1 << L); g += (1 << s)) {
#pragma unroll
            for (int k = 0; k < half; ++k) {
                const double2 a = v[g + k], b = v[g + k + half];
                v[g + k] = make_double2(a.x + b.x, a.y + b.y);
                const double dx = a.x - b.x, dy = a.y - b.y;
                const int ti = k * (32 >> s);  // e^{2 pi i k / 2^s} as a 32nd root
                if (ti == 0)
                    v[g + k + half] = make_double2(dx, dy);
                else if (ti == 8)
                    v[g + k + half] = make_double2(-dy, dx);
                else
                    v[g + k + half] = make_double2(dx * R32C[ti] - dy * R32S[ti], dx * R32S[ti] + dy * R32C[ti]);
            }
        }
    }
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// Block transform of NF sequences of length n = 2^(LA+LB).  load(f, e) returns element e of sequence f;
// store(f, k, value) receives output k.  tile: NF * A * (Bq + 1) double2 of LDS.  Threads needed: NF * max(A, Bq).
template <int LA, int LB, int MODE, class Load, class Store>
__device__ __forceinline__ void block_fft(int NF, double2 *tile, Load load, Store store, int tw = 1) {
    constexpr int A = 1 << LA, Bq = 1 << LB, n = A * Bq, LDT = Bq + 1, FST = A * LDT + 1;
    const int tid = threadIdx.x;
    // phase 1: thread (f, j) transforms x[i*Bq + j], i < A.  Thread order, chosen for coalescing of load():
    // MODE 0: j fastest (sequence elements are contiguous);  MODE 1: f fastest (sequences are adjacent in memory);
    // MODE 2: (j % tw) fastest, then f, then j / tw (tiled intermediate layout: tw columns x all sequences)
    if (tid < NF * Bq) {
        int f, j;
        if (MODE == 1) {
            f = tid % NF;
            j = tid / NF;
        } else if (MODE == 2) {
            const int jl = tid % tw, rest = tid / tw;
            f = rest % NF;
            j = jl + tw * (rest / NF);
        } else {
            f = tid / Bq;
            j = tid % Bq;
        }
        double2 v[A];
#pragma unroll
        for (int i = 0; i < A; ++i) v[i] = load(f, i * Bq + j);
        reg_fft<LA>(v);
        double s1, c1;
        sincospi(2.0 * (double)j / (double)n, &s1, &c1);
        const double2 step = make_double2(c1, s1);
        // outputs ka = 0..A-1 in order, twiddle e^{2 pi i j ka / n} by running product
        double2 w = make_double2(1.0, 0.0);
        double2 *row = tile + (size_t)f * FST + j;
#pragma unroll
        for (int ka = 0; ka < A; ++ka) {
            row[ka * LDT] = cmul(v[brev_c(ka, LA)], w);
            w = cmul(w, step);
        }
    }
    __syncthreads();
    // phase 2: thread (f, ka) transforms T[ka][j], j < Bq -> X[ka + A kb]
    if (tid < NF * A) {
        const int ka = tid / NF, f = tid - ka * NF;  // f fastest: neighbouring threads write neighbouring sequences
        const double2 *row = tile + (size_t)f * FST + (size_t)ka * LDT;
        double2 u[Bq];
#pragma unroll
        for (int j = 0; j < Bq; ++j) u[j] = row[j];
        reg_fft<LB>(u);
#pragma unroll
        for (int kb = 0; kb < Bq; ++kb) store(f, ka + A * kb, u[brev_c(kb, LB)], kb);
    }
}

// Same transform with the LDS exchange done in two halves (real parts, then imaginary parts): the tile is
// NF * (A * (Bq + 1) + pad) doubles — half the footprint, so twice the workgroups fit a CU — at the price of two more
// barriers.  A thread holds im(v) (A doubles) and re(u) (Bq doubles) across the exchange, no more than v or u alone.
// pad = 32 / NF doubles keeps the f-fastest phase-2 reads and phase-1 writes on distinct banks.
template <int LA, int LB, int MODE, class Load, class Store>
__device__ __forceinline__ void block_fft_split(int NF, double *tile, Load load, Store store, int tw = 1) {
    constexpr int A = 1 << LA, Bq = 1 << LB, n = A * Bq, LDT = Bq + 1;
    const int FST = A * LDT + (NF >= 32 ? 1 : 32 / NF);
    const int tid = threadIdx.x;
    const bool p1 = tid < NF * Bq, p2 = tid < NF * A;
    int f = 0, j = 0;
    if (MODE == 1) {
        f = tid % NF;
        j = tid / NF;
    } else if (MODE == 2) {
        const int jl = tid % tw, rest = tid / tw;
        f = rest % NF;
        j = jl + tw * (rest / NF);
    } else {
        f = tid / Bq;
        j = tid % Bq;
    }
    double2 v[A];
    if (p1) {
#pragma unroll
        for (int i = 0; i < A; ++i) v[i] = load(f, i * Bq + j);
        reg_fft<LA>(v);
        double s1, c1;
        sincospi(2.0 * (double)j / (double)n, &s1, &c1);
        const double2 step = make_double2(c1, s1);
        double2 w = make_double2(1.0, 0.0);
#pragma unroll
        for (int ka = 0; ka < A; ++ka) {
            v[brev_c(ka, LA)] = cmul(v[brev_c(ka, LA)], w);
            w = cmul(w, step);
        }
    }
    double *row1 = tile + (size_t)f * FST + j;
    const int ka2 = tid / NF, f2 = tid - ka2 * NF;
    const double *row2 = tile + (size_t)f2 * FST + (size_t)ka2 * LDT;
    double2 u[Bq];
    if (p1) {
#pragma unroll
        for (int ka = 0; ka < A; ++ka) row1[ka * LDT] = v[brev_c(ka, LA)].x;
    }
    __syncthreads();
    if (p2) {
#pragma unroll
        for (int q = 0; q < Bq; ++q) u[q].x = row2[q];
    }
    __syncthreads();
    if (p1) {
#pragma unroll
        for (int ka = 0; ka < A; ++ka) row1[ka * LDT] = v[brev_c(ka, LA)].y;
    }
    __syncthreads();
    if (p2) {
#pragma unroll
        for (int q = 0; q < Bq; ++q) u[q].y = row2[q];
        reg_fft<LB>(u);
#pragma unroll
        for (int kb = 0; kb < Bq; ++kb) store(f2, ka2 + A * kb, u[brev_c(kb, LB)], kb);
    }
}

// step 1 (register version): CT columns c0..c0+CT-1; rows >= rows_used[g] are known zeros and are not loaded
template <int LA, int LB, bool SPLIT>
__global__ __launch_bounds__(256, SPLIT ? 2 : 1) void fft_cols_reg_kernel(double2 *__restrict__ grids, int m2, int CT,
                                                            const int *__restrict__ rows_used,
                                                            double2 *__restrict__ gout, int tw) {
    extern __shared__ __attribute__((aligned(16))) double2 lds2[];
    constexpr int m1 = LA + LB, A = 1 << LA;
    const int N2 = 1 << m2;
    double2 *G = grids + ((size_t)blockIdx.y << (m1 + m2));
    const int c0 = blockIdx.x * CT;
    const int ru = rows_used ? rows_used[(blockIdx.y / 3) * 4 + (blockIdx.y % 3)] : (1 << m1);
    const double invN = 1.0 / (double)((size_t)1 << (m1 + m2));
    auto load = [&](int f, int r) -> double2 {
        return r < ru ? G[(size_t)r * N2 + c0 + f] : make_double2(0.0, 0.0);
    };
    // outputs of one thread (fixed ka) come in order kb = 0, 1, ...: the inter-step twiddle e^{2 pi i c k1 / N},
    // k1 = ka + A kb, advances by e^{2 pi i c A / N} each time
    int last_f = -1;
    double2 w = make_double2(1.0, 0.0), step = w;
    const int twl = tw > 0 ? 31 - __clz(tw) : 0;  // tw is a power of two (negative: row-tiled layout, see below)
    auto store = [&](int f, int k1, double2 v, int) {
        const int c = c0 + f;
        if (f != last_f) {  // first output of this thread: k1 = ka
            double s, cc;
            sincospi(2.0 * (double)((long long)c * k1) * invN, &s, &cc);
            w = make_double2(cc, s);
            sincospi(2.0 * (double)((long long)c * A) * invN, &s, &cc);
            step = make_double2(cc, s);
            last_f = f;
        }
        // gout: separate buffer in the tiled layout [c / tw][k1][c % tw] (a workgroup's output is one contiguous
        // run and step 2 reads 16 B x tw x RT runs); otherwise in place in the natural layout
        if (gout && tw < 0) {
            // row-tiled layout [k1 / H][c][k1 % H], H = -tw: the H rows one step-2 workgroup transforms are one
            // contiguous N2 * H * 16-B chunk; this kernel's stores come in runs of CT * H * 16 B
            const int hl = 31 - __clz(-tw);
            gout[((size_t)blockIdx.y << (m1 + m2)) + (((((size_t)(k1 >> hl)) << m2) + c) << hl) + (k1 & (-tw - 1))] = cmul(v, w);
        } else if (gout)
            gout[((size_t)blockIdx.y << (m1 + m2)) + ((((size_t)(c >> twl) << m1) + k1) << twl) + (c & (tw - 1))] = cmul(v, w);
        else
            G[(size_t)k1 * N2 + c] = cmul(v, w);
        w = cmul(w, step);
    };
    if (SPLIT)
        block_fft_split<LA, LB, 1>(CT, reinterpret_cast<double *>(lds2), load, store);
    else
        block_fft<LA, LB, 1>(CT, lds2, load, store);
}

// step 1 for grids whose samples sit in the first rows only (time span x df << 1, the normal case: lightkurve's
// default grid has span x df = 1/5, 2/5 on the 2f grid).  With only P = 2^LP non-zero inputs an N1-point column
// transform is Q = N1 / P transforms of length P of the pre-twiddled input,
//     X[Q q + s] = sum_{n < P} (x[n] W_N1^{n s}) W_P^{n q},          s < Q, q < P,
// so the LDS exchange tile is P points per column instead of N1: PRUNED_CT = 16 columns fit where the full transform
// holds 4.  That is what this kernel is for — 16-column tiles make the intermediate's [c / 16][k1][c % 16] layout
// deliver 256-B runs per row to this kernel's loads and RT x 256-B runs to the row kernel (4 x the run length of the
// full-length kernel above), and the workgroups are small enough for 2 waves per SIMD.  The input column is loaded
// once and kept in registers over the Q passes.  perm != 0 stores pass s as one contiguous block (row index
// k1' = s P + q; the row kernel undoes the permutation), perm == 0 keeps the natural row order k1 = Q q + s.
constexpr int PRUNED_CT = 16;

// Arguments of the extirpolation when it is fused into the pruned column kernel (t == nullptr: not fused, the kernel
// reads the grids lsf_spread_owner_kernel wrote).
struct SpreadArgs {
    const double *t, *w, *wy;
    const int64_t *n_off;
    const FastStats *stats;
    const int *tab16;  // lsf_prep_kernel's per-16-cell table: first cadence with position >= 16 G - 4
    int b0, ntab16, nfft, fit_mean;
    double f0, df;
};

template <int LP>
__global__ __launch_bounds__(PRUNED_CT * (1 << ((LP + 1) / 2)), 2) void fft_cols_pruned_kernel(
    const double2 *__restrict__ grids, int m1, int m2, const int *__restrict__ rows_used, double2 *__restrict__ gout,
    int perm, int tpad, SpreadArgs sa) {
    extern __shared__ __attribute__((aligned(16))) double2 lds2[];
    constexpr int LA = (LP + 1) / 2, LB = LP / 2, A = 1 << LA, Bq = 1 << LB, P = 1 << LP, LDT = Bq + 1, FST = A * LDT + 1;
    constexpr int CT = PRUNED_CT;
    const int N1 = 1 << m1, N2 = 1 << m2, Q = N1 >> LP;
    const double2 *G = grids + ((size_t)blockIdx.y << (m1 + m2));
    // tpad elements of padding after every column tile of the intermediate: without it the 32 tiles of a row start
    // exactly N1 * 256 B = 256 KB apart and step 2's reads of one row all fall on the same HBM channel
    const size_t tstride = ((size_t)CT << m1) + (size_t)tpad, gstride = (size_t)(N2 / CT) * tstride;
    double2 *O = gout + (size_t)blockIdx.y * gstride + (size_t)blockIdx.x * tstride;  // this column tile
    const int c0 = blockIdx.x * CT;
    const int ru = rows_used[(blockIdx.y / 3) * 4 + (blockIdx.y % 3)];
    const int tid = threadIdx.x;
    const int f = tid % CT, jk = tid / CT;  // column of the tile; j (phase 1) or ka (phase 2)
    const bool p1 = jk < Bq, p2 = jk < A;
    const double invN1 = 1.0 / (double)N1, invN = 1.0 / (double)((size_t)1 << (m1 + m2));
    double2 xin[A];
    if (sa.t != nullptr) {
        // ---- fused extirpolation (ordered targets: grid position grows with the cadence index).  The tile's inputs are
        // 16 cells of each of the P live rows: thread r builds row r's 16 cells in LDS from the few cadences whose 4-point
        // stencils reach them — found through the prep kernel's per-1024-cell tables and a short binary search — adding
        // in cadence order (deterministic, unlike the LDS atomics of lsf_spread_owner_kernel).  The spread grid, 81 %
        // zeros, is never written to or read from HBM: that was 13 % of the step's bytes and two launches per chunk.
        const int lb = blockIdx.y / 3, g = blockIdx.y - 3 * lb, b = sa.b0 + lb;
        const int64_t lo = sa.n_off[b];
        const double *tt = sa.t + lo, *amp = (g == 0 ? sa.wy : sa.w) + lo;
        const double t0 = sa.stats[b].t0, fac = g == 2 ? 2.0 : 1.0, dff = sa.df * fac, f0f = sa.f0 * fac;
        const double dn = (double)sa.nfft;
        const int *tg = sa.tab16 + ((size_t)b * 3 + g) * sa.ntab16;
        const bool unused = g == 1 && !sa.fit_mean;
        for (int r = tid; r < P; r += (int)blockDim.x) {
            double2 *cells = lds2 + (size_t)r * 17;
#pragma unroll
            for (int j = 0; j < CT; ++j) cells[j] = make_double2(0.0, 0.0);
            if (r < ru && !unused) {
                const int nA = r * N2 + c0, G16 = nA >> 4;
                const double xhi = (double)nA + 19.0;
                // candidates: positions in [nA - 4, nA + 28): no search, a handful of cadences, loads issued four at a time
                const int i_lo = tg[G16], i_hi = tg[min(G16 + 2, sa.ntab16 - 1)];
                for (int i0 = i_lo; i0 < i_hi; i0 += 4) {
                    double tdv[4], av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = min(i0 + u, i_hi - 1);
                        tdv[u] = tt[i] - t0;
                        av[u] = amp[i];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (i0 + u >= i_hi) break;
                        const double td = tdv[u];
                        double x = td * dn * dff;
                        if (!(x < dn)) x = fmod(x, dn);  // ordered targets never wrap: the fmod of the reference is the identity
                        if (x >= xhi) break;
                        double c = 1.0, sn = 0.0;
                        if (f0f > 0.0) sincos(6.283185307179586 * f0f * td, &sn, &c);
                        const double hr = av[u] * c, hi = av[u] * sn;
                        auto add = [&](int cell, double vr, double vi) {
                            const int j = cell - nA;
                            if (j >= 0 && j < CT) {
                                double2 v = cells[j];
                                v.x += vr;
                                v.y += vi;
                                cells[j] = v;
                            }
                        };
                        if (x == floor(x)) {  // fmod(x, 1) == 0
                            add((int)x, hr, hi);
                        } else {  // astropy extirpolate, M = 4 (same arithmetic as extirpolate4 above)
                            int ilo = (int)(x - 2.0);
                            ilo = min(max(ilo, 0), sa.nfft - 4);
                            const double d0 = x - (double)ilo, d1 = d0 - 1.0, d2 = d0 - 2.0, d3 = d0 - 3.0;
                            const double prod = ((d0 * d1) * d2) * d3;
                            const double nr = hr * prod, ni = hi * prod;
                            const double q3 = 6.0 * d3, q2 = -2.0 * d2, q1 = 2.0 * d1, q0 = -6.0 * d0;
                            add(ilo + 3, nr / q3, ni / q3);
                            add(ilo + 2, nr / q2, ni / q2);
                            add(ilo + 1, nr / q1, ni / q1);
                            add(ilo, nr / q0, ni / q0);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (p1) {
#pragma unroll
            for (int i = 0; i < A; ++i) xin[i] = lds2[(size_t)(i * Bq + jk) * 17 + f];
        }
        __syncthreads();  // the exchange tile of the passes below reuses this LDS
    } else if (p1) {
#pragma unroll
        for (int i = 0; i < A; ++i) {
            const int r = i * Bq + jk;
            xin[i] = r < ru ? G[(size_t)r * N2 + c0 + f] : make_double2(0.0, 0.0);
        }
    }
    for (int s = 0; s < Q; ++s) {
        if (p1) {
            double2 v[A];
            if (s == 0) {
#pragma unroll
                for (int i = 0; i < A; ++i) v[i] = xin[i];
            } else {
                // W_N1^{(i Bq + j) s} = e^{2 pi i j s / N1} (e^{2 pi i Bq s / N1})^i
                double sn, cs;
                sincospi(2.0 * (double)(jk * s) * invN1, &sn, &cs);
                double2 w = make_double2(cs, sn);
                sincospi(2.0 * (double)(Bq * s) * invN1, &sn, &cs);
                const double2 st = make_double2(cs, sn);
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    v[i] = cmul(xin[i], w);
                    w = cmul(w, st);
                }
            }
            reg_fft<LA>(v);
            double sn, cs;
            sincospi(2.0 * (double)jk / (double)P, &sn, &cs);  // intra-transform twiddle e^{2 pi i j ka / P}, by running product
            const double2 tw_j = make_double2(cs, sn);
            double2 w = make_double2(1.0, 0.0);
            double2 *row = lds2 + (size_t)f * FST + jk;
#pragma unroll
            for (int ka = 0; ka < A; ++ka) {
                row[ka * LDT] = cmul(v[brev_c(ka, LA)], w);
                w = cmul(w, tw_j);
            }
        }
        __syncthreads();
        if (p2) {
            const double2 *row = lds2 + (size_t)f * FST + (size_t)jk * LDT;
            double2 u[Bq];
#pragma unroll
            for (int j = 0; j < Bq; ++j) u[j] = row[j];
            reg_fft<LB>(u);
            // inter-step twiddle e^{2 pi i c k1 / N}, k1 = Q (ka + A kb) + s: advances by e^{2 pi i c Q A / N} per kb
            double sn, cs;
            sincospi(2.0 * (double)((long long)(c0 + f) * Q * A) * invN, &sn, &cs);
            const double2 stepc = make_double2(cs, sn);
            sincospi(2.0 * (double)((long long)(c0 + f) * (Q * jk + s)) * invN, &sn, &cs);
            double2 w = make_double2(cs, sn);
#pragma unroll
            for (int kb = 0; kb < Bq; ++kb) {
                const int q = jk + A * kb;
                const int k1p = perm ? s * P + q : Q * q + s;
                O[(size_t)k1p * CT + f] = cmul(u[brev_c(kb, LB)], w);
                w = cmul(w, stepc);
            }
        }
        __syncthreads();
    }
}

// step 2 (register version): RT rows r0..r0+RT-1, outputs k = k1 + N1 k2 < nkeep kept
template <int LA, int LB>
__global__ __launch_bounds__(256) void fft_rows_reg_kernel(const double2 *__restrict__ grids, int m1, int RT, int nkeep,
                                                            double2 *__restrict__ spec) {
    extern __shared__ __attribute__((aligned(16))) double2 lds2[];
    constexpr int m2 = LA + LB;
    const int N2 = 1 << m2;
    const double2 *G = grids + ((size_t)blockIdx.y << (m1 + m2));
    double2 *S = spec + (size_t)blockIdx.y * nkeep;
    const int r0 = blockIdx.x * RT;
    auto load = [&](int f, int c) -> double2 { return G[(size_t)(r0 + f) * N2 + c]; };
    auto store = [&](int f, int k2, double2 v, int) {
        const long long k = (long long)(r0 + f) + ((long long)k2 << m1);
        if (k < nkeep) S[k] = v;
    };
    block_fft<LA, LB, 0>(RT, lds2, load, store);
}

// ------------------------------------------------------------------------------------------------ three-phase FFT
// Block transform of NF sequences of length n = A*B*C (A = 2^LA, ...) in THREE register phases with two LDS exchanges.
// A thread never holds more than max(A, B, C) <= 16 points, so the kernels built on it need ~100 VGPRs instead of
// ~220-290 and 4-5 waves per SIMD stay resident: loads of one wave overlap the butterflies of the others (the
// two-phase kernels above run 1-2 waves per SIMD and alternate between waiting on HBM and computing).
//   input index  e = a*B*C + b*C + c,   output index  k = ka + A*kb + A*B*kc
//   phase 1  thread t = b*C + c        A-point FFT over a, times W_n^{t ka}            -> LDS [ka][t]       (stride S1)
//   phase 2  thread (ka, c)            B-point FFT over b, times W_{BC}^{c kb}         -> LDS [c][ka + A kb] (stride S3)
//   phase 3  thread t3 = ka + A*kb     C-point FFT over c                               -> store(f, t3 + A*B*kc, ., kc)
// load(f, e) / store(f, k, value, kc) as in block_fft.  Threads per sequence T = max(BC, AC, AB); tile: NF * FFT3_TILE
// double2.  MODE as in block_fft (thread order of the phase-1 loads).
template <int LA, int LB, int LC>
struct Fft3 {
    static constexpr int A = 1 << LA, B = 1 << LB, C = 1 << LC, n = A * B * C;
    static constexpr int T = (B * C > A * C ? (B * C > A * B ? B * C : A * B) : (A * C > A * B ? A * C : A * B));
    static constexpr int S1 = B * C + 8, S3 = A * B + 2;  // strides chosen so 16-lane groups of 16-B accesses tile the banks
    static constexpr int TILE = (A * S1 > C * S3 ? A * S1 : C * S3);
};

// KOUT: only the outputs kc < KOUT of phase 3 are wanted (C = 8, KOUT <= 2: two direct sums instead of the 8-point FFT).
// steps != nullptr: the two per-thread twiddle steps (W_n^t, W_{BC}^{c2}), which depend on the thread only — a caller
// that transforms several arrays with the same thread mapping computes them once (fft3_twiddle_steps) instead of two
// sincospi per array.
template <int LA, int LB, int LC, int MODE, int KOUT = (1 << LC), class Load, class Store>
__device__ __forceinline__ void block_fft3(int NF, double2 *tile, Load load, Store store, int tw = 1,
                                           const double2 *steps = nullptr) {
    using F = Fft3<LA, LB, LC>;
    constexpr int A = F::A, B = F::B, C = F::C, n = F::n, T = F::T, S1 = F::S1, S3 = F::S3;
    const int tid = threadIdx.x;
    int f, t;
    if (MODE == 1) {
        f = tid % NF;
        t = tid / NF;
    } else if (MODE == 2) {
        const int jl = tid % tw, rest = tid / tw;
        f = rest % NF;
        t = jl + tw * (rest / NF);
    } else {
        f = tid / T;
        t = tid % T;
    }
    const bool live = tid < NF * T;
    double2 *my = tile + (size_t)f * F::TILE;
    // ---- phase 1
    if (live && t < B * C) {
        double2 v[A];
#pragma unroll
        for (int a = 0; a < A; ++a) v[a] = load(f, a * (B * C) + t);
        reg_fft<LA>(v);
        double2 step;
        if (steps) {
            step = steps[0];
        } else {
            double s1, c1;
            sincospi(2.0 * (double)t / (double)n, &s1, &c1);
            step = make_double2(c1, s1);
        }
        double2 w = make_double2(1.0, 0.0);
#pragma unroll
        for (int ka = 0; ka < A; ++ka) {
            my[ka * S1 + t] = cmul(v[brev_c(ka, LA)], w);
            w = cmul(w, step);
        }
    }
    __syncthreads();
    // ---- phase 2
    double2 u[B];
    const int ka2 = t / C, c2 = t % C;
    if (live && t < A * C) {
#pragma unroll
        for (int b = 0; b < B; ++b) u[b] = my[ka2 * S1 + b * C + c2];
        reg_fft<LB>(u);
    }
    __syncthreads();  // every read of the [ka][t] layout is done before the tile is overwritten
    if (live && t < A * C) {
        double2 step;
        if (steps) {
            step = steps[1];
        } else {
            double s1, c1;
            sincospi(2.0 * (double)c2 / (double)(B * C), &s1, &c1);
            step = make_double2(c1, s1);
        }
        double2 w = make_double2(1.0, 0.0);
#pragma unroll
        for (int kb = 0; kb < B; ++kb) {
            my[c2 * S3 + ka2 + A * kb] = cmul(u[brev_c(kb, LB)], w);
            w = cmul(w, step);
        }
    }
    __syncthreads();
    // ---- phase 3
    if (live && t < A * B) {
        double2 z[C];
#pragma unroll
        for (int c = 0; c < C; ++c) z[c] = my[c * S3 + t];
        if (C == 8 && KOUT <= 2) {
            // X[0] = sum z_c;  X[1] = sum z_c W^c, W = e^{+2 pi i / 8}: (z0 - z4) + i (z2 - z6) + W (z1 - z5) + W^3 (z3 - z7)
            const double2 x0 = make_double2(((z[0].x + z[4].x) + (z[2].x + z[6].x)) + ((z[1].x + z[5].x) + (z[3].x + z[7].x)),
                                            ((z[0].y + z[4].y) + (z[2].y + z[6].y)) + ((z[1].y + z[5].y) + (z[3].y + z[7].y)));
            store(f, t, x0, 0);
            if (KOUT > 1) {
                const double r = 0.70710678118654752440;
                const double2 d04 = make_double2(z[0].x - z[4].x, z[0].y - z[4].y), d26 = make_double2(z[2].x - z[6].x, z[2].y - z[6].y);
                const double2 d15 = make_double2(z[1].x - z[5].x, z[1].y - z[5].y), d37 = make_double2(z[3].x - z[7].x, z[3].y - z[7].y);
                // W d15 = r ((x - y) + i (x + y));  W^3 d37 = r ((-x - y) + i (x - y))
                const double2 x1 = make_double2((d04.x - d26.y) + r * ((d15.x - d15.y) - (d37.x + d37.y)),
                                                (d04.y + d26.x) + r * ((d15.x + d15.y) + (d37.x - d37.y)));
                store(f, t + A * B, x1, 1);
            }
        } else {
            reg_fft<LC>(z);
#pragma unroll
            for (int kc = 0; kc < C; ++kc) store(f, t + A * B * kc, z[brev_c(kc, LC)], kc);
        }
    }
}

// the two twiddle steps of block_fft3 for this thread (MODE 1 / 2 thread mapping as in block_fft3)
template <int LA, int LB, int LC, int MODE>
__device__ __forceinline__ void fft3_twiddle_steps(int NF, int tw, double2 *steps) {
    using F = Fft3<LA, LB, LC>;
    const int tid = threadIdx.x;
    int t;
    if (MODE == 1) {
        t = tid / NF;
    } else if (MODE == 2) {
        const int jl = tid % tw, rest = tid / tw;
        t = jl + tw * (rest / NF);
    } else {
        t = tid % F::T;
    }
    double s1, c1;
    sincospi(2.0 * (double)t / (double)F::n, &s1, &c1);
    steps[0] = make_double2(c1, s1);
    sincospi(2.0 * (double)(t % F::C) / (double)(F::B * F::C), &s1, &c1);
    steps[1] = make_double2(c1, s1);
}

// step 2 fused with the closed form, three-phase version: thread (f, t3) ends with the outputs k2 = t3 + A*B*kc of row
// r0 + f and keeps those below M / N1 (kc < KC) for the three grids
template <int LA, int LB, int LC, int KC>
__global__ __launch_bounds__(512) void fft_rows_power3_kernel(const double2 *__restrict__ grids, int m1, int RT,
                                                               const int64_t *__restrict__ n_off,
                                                               const FastStats *__restrict__ stats, int b0, double f0,
                                                               double df, int64_t M, int fit_mean, int norm,
                                                               const double *__restrict__ scale,
                                                               double *__restrict__ power, int tw) {
    extern __shared__ __attribute__((aligned(16))) double2 lds2[];
    using F = Fft3<LA, LB, LC>;
    constexpr int m2 = LA + LB + LC;
    const int twl = tw > 0 ? 31 - __clz(tw) : 0;
    const int lb = blockIdx.y, r0 = blockIdx.x * RT;
    double2 keep[3][KC];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int q = 0; q < KC; ++q) keep[g][q] = make_double2(0.0, 0.0);
    double2 steps[2];  // the thread's twiddle steps are the same for the three grids
    if (tw < 0)
        fft3_twiddle_steps<LA, LB, LC, 1>(RT, tw, steps);
    else
        fft3_twiddle_steps<LA, LB, LC, 2>(RT, tw, steps);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        if (g == 1 && !fit_mean) continue;
        const double2 *G = grids + ((size_t)(lb * 3 + g) << (m1 + m2));
        auto store = [&](int, int, double2 v, int kc) {
            if (kc < KC) keep[g][kc < KC ? kc : 0] = v;
        };
        if (tw < 0) {  // row-tiled layout (RT == -tw): this workgroup's rows are one contiguous chunk, f fastest
            const double2 *Gt = G + (((size_t)blockIdx.x << m2) * (size_t)RT);
            auto load = [&](int f, int c) -> double2 { return Gt[(size_t)c * RT + f]; };
            block_fft3<LA, LB, LC, 1, KC>(RT, lds2, load, store, 1, steps);
        } else {
            auto load = [&](int f, int c) -> double2 {
                return G[((((size_t)(c >> twl) << m1) + (r0 + f)) << twl) + (c & (tw - 1))];
            };
            block_fft3<LA, LB, LC, 2, KC>(RT, lds2, load, store, tw, steps);
        }
        __syncthreads();
    }
    // the (f, t) mapping of block_fft3 (MODE 1 for the row-tiled layout, MODE 2 otherwise)
    const int tid = threadIdx.x;
    if (tid >= RT * F::T) return;
    int f, t3;
    if (tw < 0) {
        f = tid % RT;
        t3 = tid / RT;
    } else {
        const int jl = tid % tw, rest = tid / tw;
        f = rest % RT;
        t3 = jl + tw * (rest / RT);
    }
    if (t3 >= F::A * F::B) return;
    const int b = b0 + lb;
    const FastStats st = stats[b];
    const double nn = (double)(n_off[b + 1] - n_off[b]);
    const double sc = scale ? scale[b] : 1.0;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const long long k = (long long)(r0 + f) + ((long long)(t3 + F::A * F::B * kc) << m1);
        if (k >= M) continue;
        double2 a = keep[0][kc], bq = keep[1][kc], c2 = keep[2][kc];
        if (st.t0 != 0.0) {
            const double twopi = 6.283185307179586;
            double s, c;
            sincos(twopi * st.t0 * (f0 + df * (double)k), &s, &c);
            a = make_double2(a.x * c - a.y * s, a.x * s + a.y * c);
            bq = make_double2(bq.x * c - bq.y * s, bq.x * s + bq.y * c);
            sincos(twopi * st.t0 * (2.0 * f0 + 2.0 * df * (double)k), &s, &c);
            c2 = make_double2(c2.x * c - c2.y * s, c2.x * s + c2.y * c);
        }
        power[(size_t)b * (size_t)M + k] = gls_power_sums(a.y, a.x, bq.y, bq.x, c2.y, c2.x, fit_mean, norm, st.YY,
                                                          0.5 * st.wsum, nn, sc);
    }
}

// step 2 fused with the closed form: one workgroup transforms rows r0..r0+RT-1 of the THREE grids of a target in
// turn (the LDS tile is reused), each phase-2 thread keeps the <= KB outputs it owns that fall below M, and the
// power is computed in registers: the three spectra never go to memory.
template <int LA, int LB, int KB>
__global__ __launch_bounds__(512) void fft_rows_power_kernel(const double2 *__restrict__ grids, int m1, int RT,
                                                              const int64_t *__restrict__ n_off,
                                                              const FastStats *__restrict__ stats, int b0, double f0,
                                                              double df, int64_t M, int fit_mean, int norm,
                                                              const double *__restrict__ scale,
                                                              double *__restrict__ power, int tw, int lp, int tpad) {
    extern __shared__ __attribute__((aligned(16))) double2 lds2[];
    constexpr int m2 = LA + LB, A = 1 << LA;
    const int N2 = 1 << m2;
    const int twl = tw ? 31 - __clz(tw) : 0;
    // padded column-tile stride of the intermediate (see fft_cols_pruned_kernel); tpad = 0: the plain power-of-two layout
    const size_t tstride = tw ? (((size_t)tw << m1) + (size_t)tpad) : 0, gstride = tw ? (size_t)(N2 >> twl) * tstride : ((size_t)1 << (m1 + m2));
    const int lb = blockIdx.y, r0 = blockIdx.x * RT;
    double2 keep0[KB], keep1[KB], keep2[KB];
#pragma unroll
    for (int q = 0; q < KB; ++q) keep0[q] = keep1[q] = keep2[q] = make_double2(0.0, 0.0);
    {
        const double2 *G = grids + (size_t)(lb * 3 + 0) * gstride;
        auto load = [&](int f, int c) -> double2 {
            return tw ? G[(size_t)(c >> twl) * tstride + ((size_t)(r0 + f) << twl) + (c & (tw - 1))] : G[(size_t)(r0 + f) * N2 + c];
        };
        auto store = [&](int, int, double2 v, int kb) {
            if (kb < KB) keep0[kb < KB ? kb : 0] = v;
        };
        if (tw)
            block_fft<LA, LB, 2>(RT, lds2, load, store, tw);
        else
            block_fft<LA, LB, 0>(RT, lds2, load, store);
    }
    __syncthreads();
    if (fit_mean) {
        const double2 *G = grids + (size_t)(lb * 3 + 1) * gstride;
        auto load = [&](int f, int c) -> double2 {
            return tw ? G[(size_t)(c >> twl) * tstride + ((size_t)(r0 + f) << twl) + (c & (tw - 1))] : G[(size_t)(r0 + f) * N2 + c];
        };
        auto store = [&](int, int, double2 v, int kb) {
            if (kb < KB) keep1[kb < KB ? kb : 0] = v;
        };
        if (tw)
            block_fft<LA, LB, 2>(RT, lds2, load, store, tw);
        else
            block_fft<LA, LB, 0>(RT, lds2, load, store);
    }
    __syncthreads();
    {
        const double2 *G = grids + (size_t)(lb * 3 + 2) * gstride;
        auto load = [&](int f, int c) -> double2 {
            return tw ? G[(size_t)(c >> twl) * tstride + ((size_t)(r0 + f) << twl) + (c & (tw - 1))] : G[(size_t)(r0 + f) * N2 + c];
        };
        auto store = [&](int, int, double2 v, int kb) {
            if (kb < KB) keep2[kb < KB ? kb : 0] = v;
        };
        if (tw)
            block_fft<LA, LB, 2>(RT, lds2, load, store, tw);
        else
            block_fft<LA, LB, 0>(RT, lds2, load, store);
    }
    const int tid = threadIdx.x;
    if (tid >= RT * A) return;
    const int ka = tid / RT, f = tid - ka * RT;  // the phase-2 mapping of block_fft
    const int b = b0 + lb;
    const FastStats st = stats[b];
    const double nn = (double)(n_off[b + 1] - n_off[b]);
    const double sc = scale ? scale[b] : 1.0;
    // row r0 + f of the intermediate is k1 itself, or (lp > 0: fft_cols_pruned_kernel with perm) row s P + q of k1 = Q q + s
    const int rp = r0 + f;
    const int k1 = lp ? (((rp & ((1 << lp) - 1)) << (m1 - lp)) + (rp >> lp)) : rp;
    // e^{2 pi i t0 f} for this thread's outputs k = k1 + N1 (ka + A kb): one sincos for kb = 0 and one for the step between
    // consecutive kb (a rotation by 2 pi t0 df N1 A), the 2f phase by the double-angle formulas (its argument is exactly
    // twice the 1f one) — 2 sincos per thread instead of 2 per output; the products drift by < 1e-15 over KB <= 8 steps.
    const double twopi = 6.283185307179586;
    double ph_c = 1.0, ph_s = 0.0, st_c = 1.0, st_s = 0.0;
    if (st.t0 != 0.0) {
        const long long kfirst = (long long)k1 + ((long long)ka << m1);
        sincos(twopi * st.t0 * (f0 + df * (double)kfirst), &ph_s, &ph_c);
        sincos(twopi * st.t0 * (df * (double)((long long)A << m1)), &st_s, &st_c);
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const long long k = (long long)k1 + ((long long)(ka + A * kb) << m1);
        if (k < M) {
            double2 a = keep0[kb], bq = keep1[kb], c2 = keep2[kb];
            if (st.t0 != 0.0) {
                const double c = ph_c, s = ph_s;
                a = make_double2(a.x * c - a.y * s, a.x * s + a.y * c);
                bq = make_double2(bq.x * c - bq.y * s, bq.x * s + bq.y * c);
                const double cc = c * c - s * s, ss = 2.0 * s * c;
                c2 = make_double2(c2.x * cc - c2.y * ss, c2.x * ss + c2.y * cc);
            }
            power[(size_t)b * (size_t)M + k] = gls_power_sums(a.y, a.x, bq.y, bq.x, c2.y, c2.x, fit_mean, norm, st.YY,
                                                              0.5 * st.wsum, nn, sc);
        }
        const double nc = ph_c * st_c - ph_s * st_s, ns = ph_s * st_c + ph_c * st_s;
        ph_c = nc;
        ph_s = ns;
    }
}

// ------------------------------------------------------------------------------------------------ fastchi2
// astropy lombscargle_fastchi2 (fastchi2_impl.py:60-137): the multi-term fit of chi2_impl with every trig sum taken
// from the extirpolated FFT grids (trig_sum with freq_factor = m).  3 nterms grids per target:
//   g <  nterms : w (y - ybar) at harmonic g + 1              g >= nterms : w at harmonic g - nterms + 1 (up to 2 nterms)
__global__ __launch_bounds__(256) void lsf_scatter_multi_kernel(const double *__restrict__ t, const double *__restrict__ w,
                                                                 const double *__restrict__ wy,
                                                                 const int64_t *__restrict__ n_off,
                                                                 const FastStats *__restrict__ stats, int b0, double f0,
                                                                 double df, int nfft, int nterms,
                                                                 double2 *__restrict__ grids) {
    const int b = b0 + blockIdx.y;
    const int64_t lo = n_off[b];
    const int n = (int)(n_off[b + 1] - lo);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double tt = t[lo + i] - stats[b].t0;
    double2 *g0 = grids + (size_t)blockIdx.y * 3 * nterms * nfft;
    const double wi = w[lo + i], wyi = wy[lo + i];
    const double twopi = 6.283185307179586;
    for (int fac = 1; fac <= 2 * nterms; ++fac) {
        const double dff = df * (double)fac, f0f = f0 * (double)fac;
        double c = 1.0, s = 0.0;
        if (f0f > 0.0) sincos(twopi * f0f * tt, &s, &c);
        const double tn = fmod(tt * (double)nfft * dff, (double)nfft);
        extirpolate4(g0 + (size_t)(nterms + fac - 1) * nfft, nfft, tn, wi * c, wi * s);
        if (fac <= nterms) extirpolate4(g0 + (size_t)(fac - 1) * nfft, nfft, tn, wyi * c, wyi * s);
    }
}

template <int NT>
__global__ __launch_bounds__(256) void lsf_chi2_power_kernel(const double2 *__restrict__ spec,
                                                              const int64_t *__restrict__ n_off,
                                                              const FastStats *__restrict__ stats, int b0, double f0,
                                                              double df, int64_t M, int fit_mean, int norm,
                                                              const double *__restrict__ scale,
                                                              double *__restrict__ power) {
    const int b = b0 + blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const FastStats st = stats[b];
    const double2 *s0 = spec + (size_t)blockIdx.y * 3 * NT * M;
    Chi2Sums<NT> sums;
    const double twopi = 6.283185307179586;
#pragma unroll
    for (int m = 1; m <= 2 * NT; ++m) {
        double cph = 1.0, sph = 0.0;
        if (st.t0 != 0.0) sincos(twopi * st.t0 * (double)m * (f0 + df * (double)j), &sph, &cph);  // utils.py:151-153
        const double2 a = s0[(size_t)(NT + m - 1) * M + j];
        sums.Cw[m - 1] = a.x * cph - a.y * sph;
        sums.Sw[m - 1] = a.x * sph + a.y * cph;
        if (m <= NT) {
            const double2 y = s0[(size_t)(m - 1) * M + j];
            sums.Cy[m - 1] = y.x * cph - y.y * sph;
            sums.Sy[m - 1] = y.x * sph + y.y * cph;
        }
    }
    const double n = (double)(n_off[b + 1] - n_off[b]);
    power[(size_t)b * (size_t)M + j] =
        chi2_normalise(sums.solve_lu(st.yws, fit_mean), norm, st.YY, 0.5 * st.wsum, n, scale ? scale[b] : 1.0);
}

// closed form from the three spectra (C = real, S = imag of the unnormalised inverse transform)
__global__ __launch_bounds__(256) void lsf_power_kernel(const double2 *__restrict__ spec,
                                                         const int64_t *__restrict__ n_off,
                                                         const FastStats *__restrict__ stats, int b0, double f0,
                                                         double df, int64_t M, int fit_mean, int norm,
                                                         const double *__restrict__ scale,
                                                         double *__restrict__ power) {
    const int b = b0 + blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const FastStats st = stats[b];
    const double2 *s0 = spec + (size_t)blockIdx.y * 3 * M;
    double2 a = s0[j], bq = s0[M + j], c2 = s0[2 * M + j];
    if (st.t0 != 0.0) {  // utils.py:151-153: fftgrid *= exp(2 pi i t0 f), f on the (factor-scaled) grid
        const double twopi = 6.283185307179586;
        double s, c;
        sincos(twopi * st.t0 * (f0 + df * (double)j), &s, &c);
        a = make_double2(a.x * c - a.y * s, a.x * s + a.y * c);
        bq = make_double2(bq.x * c - bq.y * s, bq.x * s + bq.y * c);
        sincos(twopi * st.t0 * (2.0 * f0 + 2.0 * df * (double)j), &s, &c);
        c2 = make_double2(c2.x * c - c2.y * s, c2.x * s + c2.y * c);
    }
    const double n = (double)(n_off[b + 1] - n_off[b]);
    power[(size_t)b * (size_t)M + j] = gls_power_sums(a.y, a.x, bq.y, bq.x, c2.y, c2.x, fit_mean, norm, st.YY,
                                                      0.5 * st.wsum, n, scale ? scale[b] : 1.0);
}

// zero only the grid rows that can receive samples (the column transform treats the others as zeros)
__global__ __launch_bounds__(256) void lsf_zero_kernel(double2 *__restrict__ grids, int m1, int m2,
                                                        const int *__restrict__ rows_used) {
    if (rows_used[(blockIdx.y / 3) * 4 + 3]) return;  // ordered target: the owner-computes spreader writes every cell
    const int ru = rows_used[(blockIdx.y / 3) * 4 + (blockIdx.y % 3)];
    const int N2 = 1 << m2;
    double2 *G = grids + ((size_t)blockIdx.y << (m1 + m2));
    for (int r = blockIdx.x * 8; r < min(blockIdx.x * 8 + 8, ru); ++r)
        for (int c = threadIdx.x; c < N2; c += 256) G[(size_t)r * N2 + c] = make_double2(0.0, 0.0);
}

template <int LA, int LB>
static void launch_cols_t(int m2, int ngrids, double2 *grids, const int *rows_used, double2 *gout, int tw,
                          hipStream_t stream) {
    constexpr int A = 1 << LA, Bq = 1 << LB, n = A * Bq, LDT = Bq + 1, FST = A * LDT + 1;
    const int N2 = 1 << m2;
    static const int ct_pts = getenv("LK_FFT_CT_PTS") ? atoi(getenv("LK_FFT_CT_PTS")) : 4096;  // points per column tile
    const int CT = std::max(1, std::min(N2, std::min(ct_pts / n, 256 / std::max(A, Bq))));
    const int nt = ((CT * std::max(A, Bq) + 63) / 64) * 64;
    static const bool split = getenv("LK_FFT_SPLIT") ? atoi(getenv("LK_FFT_SPLIT")) != 0 : false;  // measured 4 % slower (spills at 2 waves/SIMD)
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fft_cols_reg_kernel<LA, LB, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fft_cols_reg_kernel<LA, LB, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    if (split) {
        const int fsts = A * LDT + (CT >= 32 ? 1 : 32 / CT);
        hipLaunchKernelGGL((fft_cols_reg_kernel<LA, LB, true>), dim3(N2 / CT, ngrids), dim3(nt), (size_t)CT * fsts * 8,
                           stream, grids, m2, CT, rows_used, gout, tw);
    } else {
        hipLaunchKernelGGL((fft_cols_reg_kernel<LA, LB, false>), dim3(N2 / CT, ngrids), dim3(nt), (size_t)CT * FST * 16,
                           stream, grids, m2, CT, rows_used, gout, tw);
    }
}

// width of the tiled intermediate layout: the column kernel's CT, capped by the row kernel's Bq (both powers of 2)
static int tile_width(int m1, int m2) {
    const int la1 = (m1 + 1) / 2, lb1 = m1 / 2, n1 = 1 << m1;
    static const int ct_pts = getenv("LK_FFT_CT_PTS") ? atoi(getenv("LK_FFT_CT_PTS")) : 4096;
    const int ct = std::max(1, std::min(1 << m2, std::min(ct_pts / n1, 256 / std::max(1 << la1, 1 << lb1))));
    const int bq2 = 1 << (m2 / 2);
    return std::min(ct, bq2);
}

template <int LA, int LB>
static void launch_rows_t(int m1, int ngrids, const double2 *grids, int nkeep, double2 *spec, hipStream_t stream) {
    constexpr int A = 1 << LA, Bq = 1 << LB, n = A * Bq, LDT = Bq + 1, FST = A * LDT + 1;
    const int N1 = 1 << m1;
    const int RT = std::max(1, std::min(N1, std::min(4096 / n, 256 / std::max(A, Bq))));
    const int nt = ((RT * std::max(A, Bq) + 63) / 64) * 64;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fft_rows_reg_kernel<LA, LB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((fft_rows_reg_kernel<LA, LB>), dim3(N1 / RT, ngrids), dim3(nt), (size_t)RT * FST * 16, stream,
                       grids, m1, RT, nkeep, spec);
}

struct FusedArgs {
    const int64_t *n_off;
    const FastStats *stats;
    int b0;
    double f0, df;
    int64_t M;
    int fit_mean, norm;
    const double *scale;
    double *power;
};

template <int LA, int LB, int KB>
static void launch_rows_power_t(int m1, int ntargets, const double2 *grids, const FusedArgs &a, int tw,
                                hipStream_t stream, int lp = 0, int tpad = 0) {
    constexpr int A = 1 << LA, Bq = 1 << LB, n = A * Bq, LDT = Bq + 1, FST = A * LDT + 1;
    const int N1 = 1 << m1;
    int RT = std::max(1, std::min(N1, std::min(4096 / n, 256 / std::max(A, Bq))));
    // LK_FFT_RT=16: twice the rows per workgroup (twice the contiguous run per column tile the loads see) at the price
    // of one workgroup per CU (140 KB of LDS at N2 = 512)
    if (const char *e = getenv("LK_FFT_RT")) {
        const int want = atoi(e);
        if (want >= 1 && want <= N1 && (N1 % want) == 0 && want * std::max(A, Bq) <= 512 &&
            (size_t)want * FST * 16 <= 160 * 1024)
            RT = want;
    }
    const int nt = ((RT * std::max(A, Bq) + 63) / 64) * 64;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fft_rows_power_kernel<LA, LB, KB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((fft_rows_power_kernel<LA, LB, KB>), dim3(N1 / RT, ntargets), dim3(nt), (size_t)RT * FST * 16,
                       stream, grids, m1, RT, a.n_off, a.stats, a.b0, a.f0, a.df, a.M, a.fit_mean, a.norm, a.scale,
                       a.power, tw, lp, tpad);
}

// returns false if the (m2, outputs-per-thread) combination has no fused instantiation
static bool rows_power_available(int m1, int m2, int64_t M) {
    const int LA = (m2 + 1) / 2, Aa = 1 << LA;
    const long long k2need = (M + ((long long)1 << m1) - 1) >> m1;
    return m2 >= 4 && m2 <= 10 && (k2need + Aa - 1) / Aa <= 8;
}

template <int LA, int LB, int LC, int KC>
static void launch_rows_power3_t(int m1, int ntargets, const double2 *grids, const FusedArgs &a, int tw, hipStream_t stream) {
    using F = Fft3<LA, LB, LC>;
    const int N1 = 1 << m1;
    static const int rt_env = getenv("LK_FFT3_RT") ? atoi(getenv("LK_FFT3_RT")) : 4;  // 4 rows: 37 KB LDS, 4 workgroups per CU
    int RT = std::max(1, std::min(N1, 512 / F::T));
    if (rt_env > 0) RT = std::max(1, std::min(RT, rt_env));
    while (RT > 1 && (N1 % RT)) --RT;
    if (tw < 0) RT = -tw;  // the row-tiled intermediate fixes the rows per workgroup
    const int nt = ((RT * F::T + 63) / 64) * 64;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fft_rows_power3_kernel<LA, LB, LC, KC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((fft_rows_power3_kernel<LA, LB, LC, KC>), dim3(N1 / RT, ntargets), dim3(nt),
                       (size_t)RT * F::TILE * 16, stream, grids, m1, RT, a.n_off, a.stats, a.b0, a.f0, a.df, a.M,
                       a.fit_mean, a.norm, a.scale, a.power, tw);
}

// three-phase step 2 where an instantiation exists (needs the tiled intermediate layout, tw >= 1, T % tw == 0)
static bool launch_rows_power3(int m1, int m2, int ntargets, const double2 *grids, const FusedArgs &a, int tw,
                               hipStream_t stream) {
    // measured on configs[1]: +2 % on one box, -1.3 % on another vs the two-phase kernel at 2 waves/SIMD — step 2 is
    // limited by how fast HBM serves its 64-B x RT runs, not by occupancy — so the three-phase kernel is opt-in
    if (!(getenv("LK_FFT3") && atoi(getenv("LK_FFT3")) == 1)) return false;
    const long long k2need = (a.M + ((long long)1 << m1) - 1) >> m1;
#define LK_RP3(la, lb, lc)                                                              \
    {                                                                                   \
        const int ab = 1 << ((la) + (lb));                                              \
        const int kc = (int)((k2need + ab - 1) / ab);                                   \
        if (kc <= 2) {                                                                  \
            launch_rows_power3_t<la, lb, lc, 2>(m1, ntargets, grids, a, tw, stream);    \
            return true;                                                                \
        }                                                                               \
        if (kc <= 4 && (1 << (lc)) >= 4) {                                              \
            launch_rows_power3_t<la, lb, lc, 4>(m1, ntargets, grids, a, tw, stream);    \
            return true;                                                                \
        }                                                                               \
        return false;                                                                   \
    }
    switch (m2) {
        case 8: LK_RP3(3, 3, 2)
        case 9: LK_RP3(3, 3, 3)
        case 10: LK_RP3(4, 3, 3)
        default: return false;
    }
#undef LK_RP3
}

static bool launch_rows_power(int m1, int m2, int ntargets, const double2 *grids, const FusedArgs &a, int tw,
                              hipStream_t stream, int lp = 0, int tpad = 0) {
    if (lp == 0 && launch_rows_power3(m1, m2, ntargets, grids, a, tw, stream)) return true;
    if (tw < 0) return false;  // the two-phase kernel below reads the column-tiled layout only
    const int LA = (m2 + 1) / 2, Aa = 1 << LA;
    const long long k2need = (a.M + ((long long)1 << m1) - 1) >> m1;
    const int kb = (int)((k2need + Aa - 1) / Aa);
    if (kb > 8) return false;
#define LK_RP(la, lb)                                                                  \
    if (kb <= 4)                                                                       \
        launch_rows_power_t<la, lb, 4>(m1, ntargets, grids, a, tw, stream, lp, tpad);  \
    else                                                                               \
        launch_rows_power_t<la, lb, 8>(m1, ntargets, grids, a, tw, stream, lp, tpad);  \
    return true;
    switch (m2) {
        case 4: LK_RP(2, 2)
        case 5: LK_RP(3, 2)
        case 6: LK_RP(3, 3)
        case 7: LK_RP(4, 3)
        case 8: LK_RP(4, 4)
        case 9: LK_RP(5, 4)
        case 10: LK_RP(5, 5)
        default: return false;
    }
#undef LK_RP
}

static void launch_cols_reg(int m1, int m2, int ngrids, double2 *grids, const int *rows_used, double2 *gout, int tw,
                            hipStream_t stream) {
    switch (m1) {
        case 4: launch_cols_t<2, 2>(m2, ngrids, grids, rows_used, gout, tw, stream); break;
        case 5: launch_cols_t<3, 2>(m2, ngrids, grids, rows_used, gout, tw, stream); break;
        case 6: launch_cols_t<3, 3>(m2, ngrids, grids, rows_used, gout, tw, stream); break;
        case 7: launch_cols_t<4, 3>(m2, ngrids, grids, rows_used, gout, tw, stream); break;
        case 8: launch_cols_t<4, 4>(m2, ngrids, grids, rows_used, gout, tw, stream); break;
        case 9: launch_cols_t<5, 4>(m2, ngrids, grids, rows_used, gout, tw, stream); break;
        default: launch_cols_t<5, 5>(m2, ngrids, grids, rows_used, gout, tw, stream); break;
    }
}

static void launch_rows_reg(int m1, int m2, int ngrids, const double2 *grids, int nkeep, double2 *spec,
                            hipStream_t stream) {
    switch (m2) {
        case 4: launch_rows_t<2, 2>(m1, ngrids, grids, nkeep, spec, stream); break;
        case 5: launch_rows_t<3, 2>(m1, ngrids, grids, nkeep, spec, stream); break;
        case 6: launch_rows_t<3, 3>(m1, ngrids, grids, nkeep, spec, stream); break;
        case 7: launch_rows_t<4, 3>(m1, ngrids, grids, nkeep, spec, stream); break;
        case 8: launch_rows_t<4, 4>(m1, ngrids, grids, nkeep, spec, stream); break;
        case 9: launch_rows_t<5, 4>(m1, ngrids, grids, nkeep, spec, stream); break;
        default: launch_rows_t<5, 5>(m1, ngrids, grids, nkeep, spec, stream); break;
    }
}

// per call: the largest rows_used over all targets and grids, and the number of targets that are not "ordered"
__global__ __launch_bounds__(256) void lsf_plan_kernel(const int *__restrict__ rows_used, int B, int *__restrict__ plan) {
    int mx = 0, unordered = 0;
    for (int b = threadIdx.x; b < B; b += 256) {
        mx = max(mx, max(rows_used[b * 4], max(rows_used[b * 4 + 1], rows_used[b * 4 + 2])));
        unordered += rows_used[b * 4 + 3] ? 0 : 1;
    }
    __shared__ int smx[256], sun[256];
    smx[threadIdx.x] = mx;
    sun[threadIdx.x] = unordered;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smx[threadIdx.x] = max(smx[threadIdx.x], smx[threadIdx.x + s]);
            sun[threadIdx.x] += sun[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        plan[0] = smx[0];
        plan[1] = sun[0];
    }
}

template <int LP>
static void launch_cols_pruned_t(int m1, int m2, int ngrids, const double2 *grids, const int *rows_used, double2 *gout,
                                 int perm, int tpad, const SpreadArgs &sa, hipStream_t stream) {
    constexpr int LA = (LP + 1) / 2, LB = LP / 2, A = 1 << LA, Bq = 1 << LB, LDT = Bq + 1, FST = A * LDT + 1;
    static_assert(PRUNED_CT * FST >= (1 << LP) * 17, "the exchange tile must hold the fused spreader's P x 17 input cells");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fft_cols_pruned_kernel<LP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((fft_cols_pruned_kernel<LP>), dim3((1 << m2) / PRUNED_CT, ngrids), dim3(PRUNED_CT * A),
                       (size_t)PRUNED_CT * FST * 16, stream, grids, m1, m2, rows_used, gout, perm, tpad, sa);
}

static bool launch_cols_pruned(int lp, int m1, int m2, int ngrids, const double2 *grids, const int *rows_used,
                               double2 *gout, int perm, int tpad, const SpreadArgs &sa, hipStream_t stream) {
    switch (lp) {
        case 5: launch_cols_pruned_t<5>(m1, m2, ngrids, grids, rows_used, gout, perm, tpad, sa, stream); return true;
        case 6: launch_cols_pruned_t<6>(m1, m2, ngrids, grids, rows_used, gout, perm, tpad, sa, stream); return true;
        case 7: launch_cols_pruned_t<7>(m1, m2, ngrids, grids, rows_used, gout, perm, tpad, sa, stream); return true;
        case 8: launch_cols_pruned_t<8>(m1, m2, ngrids, grids, rows_used, gout, perm, tpad, sa, stream); return true;
        default: return false;
    }
}

static int ilog2_ceil(long long v) {
    int m = 0;
    while (((long long)1 << m) < v) ++m;
    return m;
}

int lsfast_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *dy,
                  double f0, double df, int64_t M, int fit_mean, int center_data, int normalization,
                  const double *scale, int oversampling, double *power, hipStream_t stream) {
    LK_REQUIRE(B >= 0 && n_off_host != nullptr, "bad batch description");
    LK_REQUIRE(M >= 0, "M must be >= 0");
    if (B == 0 || M == 0) return LK_OK;
    LK_REQUIRE(t && y && power, "t, y, power must be non-NULL");
    LK_REQUIRE(normalization >= LK_NORM_STANDARD && normalization <= LK_NORM_LK_PSD, "unknown normalization %d",
               normalization);
    LK_REQUIRE(f0 >= 0.0, "Frequencies must be positive");
    LK_REQUIRE(df > 0.0, "Frequency steps must be positive");
    LK_REQUIRE(oversampling >= 1 && oversampling <= 64, "oversampling outside 1..64");
    LK_REQUIRE(n_off_host[0] == 0, "n_off[0] must be 0");
    int64_t nmax = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t n = n_off_host[b + 1] - n_off_host[b];
        LK_REQUIRE(n >= 1 && n < ((int64_t)1 << 30), "target %d has %lld cadences", b, (long long)n);
        nmax = std::max(nmax, n);
    }
    const int m = std::max(3, ilog2_ceil((long long)M * oversampling));  // Nfft = bitceil(Nf * oversampling)
    LK_REQUIRE(m <= 24, "FFT grid 2^%d too large (M = %lld)", m, (long long)M);
    const int nfft = 1 << m, m1 = (m + 1) / 2, m2 = m / 2;
    const int N1 = 1 << m1, N2 = 1 << m2;
    const int CT = std::max(1, std::min(N2, 4096 / N1)), RT = std::max(1, std::min(N1, 4096 / N2));
    const size_t ntot = (size_t)n_off_host[B];
    // targets per chunk: keep the grids (3 x 16 B x Nfft per target) within ~2 GiB
    size_t chunk_bytes = (size_t)2 << 30;
    if (const char *e = getenv("LK_FAST_CHUNK_MB")) chunk_bytes = (size_t)std::max(1, atoi(e)) << 20;  // tuning knob
    const int Bc = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, chunk_bytes / ((size_t)48 * nfft)));
    h->ws.reset();
    // padding after each 16-column tile of the FFT intermediate (pruned path only), in 16-byte elements
    static const int tpad_env = getenv("LK_LSF_TILE_PAD") ? std::max(0, atoi(getenv("LK_LSF_TILE_PAD"))) : 0;
    const size_t pad_elems = ((size_t)(nfft >> m1) / PRUNED_CT + 1) * (size_t)tpad_env;
    int rc = h->ws.reserve((size_t)(B + 1) * 8 + (size_t)B * sizeof(FastStats) + 2 * (ntot * 8 + 256) +
                           (size_t)Bc * 3 * nfft * 16 * 4 + (size_t)Bc * 3 * pad_elems * 16 * 2 + (size_t)Bc * 3 * M * 16 +
                           (size_t)B * 16 + (size_t)B * 6 * ((nfft + 1023) / 1024 + 2) * 4 + (size_t)B * 3 * (nfft / 16 + 2) * 4 + 16384);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    FastStats *d_stats = (FastStats *)h->ws.alloc((size_t)B * sizeof(FastStats));
    double *d_w = (double *)h->ws.alloc(ntot * 8), *d_wy = (double *)h->ws.alloc(ntot * 8);
    double2 *d_grids = (double2 *)h->ws.alloc((size_t)Bc * 3 * nfft * 16);
    double2 *d_spec = (double2 *)h->ws.alloc((size_t)Bc * 3 * M * 16);
    rc = h->stage.copy(d_off, n_off_host, (size_t)(B + 1) * 8, stream);
    if (rc) return rc;
    static bool attr_set = false;
    if (!attr_set) {
        LK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fft_cols_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        LK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fft_rows_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        attr_set = true;
    }
    const bool reg_path = m1 >= 4 && m1 <= 10 && m2 >= 4 && m2 <= 10 && !getenv("LK_FFT_RADIX2");
    const bool fused = reg_path && !getenv("LK_FFT_UNFUSED") && rows_power_available(m1, m2, M);
    int tw = tile_width(m1, m2);
    {
        // row-tiled intermediate (negative tw = -rows per tile): step-2 workgroups would stream contiguous chunks, but
        // step 1's stores shrink to CT * H * 16-B runs — measured slower (18.5 ms at H = 8, 15.5 at H = 4 vs 15.0), so
        // it stays an experiment behind LK_FFT_LAYOUT=1 (LK_FFT_RTILE sets H).
        const long long k2need = (M + ((long long)1 << m1) - 1) >> m1;
        const int la3 = m2 == 10 ? 4 : 3, lb3 = 3, lc3 = m2 - la3 - lb3;
        const bool three = fused && m2 >= 8 && m2 <= 10 && getenv("LK_FFT3") && atoi(getenv("LK_FFT3")) == 1 &&
                           (k2need + (1 << (la3 + lb3)) - 1) / (1 << (la3 + lb3)) <= (lc3 >= 2 ? 4 : 2);
        const int h8 = getenv("LK_FFT_RTILE") ? atoi(getenv("LK_FFT_RTILE")) : 8;
        const int T3 = 1 << std::max(std::max(lb3 + lc3, la3 + lc3), la3 + lb3);
        if (three && getenv("LK_FFT_LAYOUT") && atoi(getenv("LK_FFT_LAYOUT")) == 1 && (h8 == 2 || h8 == 4 || h8 == 8) &&
            h8 * T3 <= 512 && (1 << m1) % h8 == 0)
            tw = -h8;
    }
    double2 *d_grids2 = fused ? (double2 *)h->ws.alloc((size_t)Bc * 3 * (nfft + pad_elems) * 16) : nullptr;
    LK_REQUIRE(!fused || d_grids2 != nullptr, "workspace exhausted");
    int *d_rows = (int *)h->ws.alloc((size_t)B * 4 * 4);
    int *d_plan = (int *)h->ws.alloc(64);
    const bool tab_env = getenv("LK_LSF_TABLES") ? atoi(getenv("LK_LSF_TABLES")) != 0 : true;
    const int ntab = (nfft + SPREAD_W - 1) / SPREAD_W + 2;
    int *d_tab = (reg_path && tab_env) ? (int *)h->ws.alloc((size_t)B * 6 * ntab * 4) : nullptr;
    // Fused extirpolation (LK_LSF_FUSED_SPREAD=1, off by default): measured 13.55 ms against 13.49 ms per 1000 targets —
    // the column kernel grows by 93 us per launch (sincos and four divisions per cadence visit next to the FFT's own fp64
    // work), the separate spreader's 100 us were mostly hidden behind its neighbours, and the prep kernel pays 0.3 ms for
    // the fine table.  It would start to pay with the phases precomputed per cadence and the Lagrange weights formed
    // without divisions; until then the separate spreader stays.
    static const bool fuse_env = getenv("LK_LSF_FUSED_SPREAD") && atoi(getenv("LK_LSF_FUSED_SPREAD")) == 1;
    const int ntab16 = nfft / 16 + 2;
    int *d_tab16 = (d_tab && fuse_env && nfft >= 4096) ? (int *)h->ws.alloc((size_t)B * 3 * ntab16 * 4) : nullptr;
    hipLaunchKernelGGL(lsf_prep_kernel, dim3(B), dim3(256), 0, stream, t, y, dy, d_off, (fit_mean || center_data) ? 1 : 0,
                       d_w, d_wy, d_stats, df, nfft, m2, d_rows, d_tab, ntab, d_tab16, ntab16);
    // ---- plan: the pruned column kernel applies when every grid of every target keeps its samples in the first
    // P <= 256 rows (P < N1) and the row kernel can read 16-column tiles.  The decision needs one device word, so
    // the call synchronises `stream` once here (20-30 us against a >= 1 ms step).
    const bool pruned_env = getenv("LK_LSF_PRUNED") ? atoi(getenv("LK_LSF_PRUNED")) != 0 : true;
    const int perm_env = getenv("LK_LSF_PERM") ? atoi(getenv("LK_LSF_PERM")) : 0;
    const bool streams_env = getenv("LK_LSF_STREAMS") ? atoi(getenv("LK_LSF_STREAMS")) != 0 : true;
    int lp = 0;
    if (fused && pruned_env && tw > 0 && m2 >= 8 && m2 <= 10 && N2 >= PRUNED_CT) {
        if (!h->h_plan) LK_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h->h_plan), 64, hipHostMallocDefault));
        hipLaunchKernelGGL(lsf_plan_kernel, dim3(1), dim3(256), 0, stream, d_rows, B, d_plan);
        LK_HIP_CHECK(hipMemcpyAsync(h->h_plan, d_plan, 8, hipMemcpyDeviceToHost, stream));
        LK_HIP_CHECK(hipStreamSynchronize(stream));
        const int max_rows = h->h_plan[0];
        const int want = std::max(5, ilog2_ceil(std::max(1, max_rows)));
        if (want <= 8 && want < m1) lp = want;
    }
    // ---- fused extirpolation (opt-in, see above): every target ordered (h_plan[1] = number of unordered ones), the pruned
    // column kernel in use and the fine table built -> no spread grid at all
    const bool fused_spread = lp != 0 && d_tab16 != nullptr && h->h_plan[1] == 0 && PRUNED_CT == 16 && N2 % PRUNED_CT == 0;
    // ---- two streams: the spreader of chunk k+1 (LDS atomics, latency bound) runs on h->s_aux under the FFT kernels
    // of chunk k (HBM / VALU bound) on the caller's stream; the spread grids are double buffered, events order the
    // hand-overs.  All s_aux work is consumed through events by `stream`, so the caller still sees one stream.
    const bool two_streams = fused && streams_env && B > Bc && !fused_spread;
    double2 *d_gridsB = nullptr;
    if (two_streams) {
        d_gridsB = (double2 *)h->ws.alloc((size_t)Bc * 3 * nfft * 16);
        LK_REQUIRE(d_gridsB != nullptr, "workspace exhausted");
        if (!h->s_aux) {
            LK_HIP_CHECK(hipStreamCreateWithFlags(&h->s_aux, hipStreamNonBlocking));
            for (int i = 0; i < 4; ++i) LK_HIP_CHECK(hipEventCreateWithFlags(&h->ev_aux[i], hipEventDisableTiming));
        }
        // s_aux may start once the prep kernel's outputs exist
        LK_HIP_CHECK(hipEventRecord(h->ev_aux[0], stream));
        LK_HIP_CHECK(hipStreamWaitEvent(h->s_aux, h->ev_aux[0], 0));
    }
    // ---- rows on their own stream: step 2 of chunk k (reads, with compute gaps: 4.4 TB/s) runs on h->s_rows while step 1
    // of chunk k + 1 (mostly writes, 5.7 TB/s) runs on the caller's stream — a CU holds one workgroup of each (LDS 70 + 70
    // KB) — the intermediate is double buffered.  Measured 13.29 against 13.39 ms per 1000 targets (-0.8 %): the step moves
    // 64 GB at an average 4.8 TB/s whichever way the kernels are interleaved, so it is opt-in (LK_LSF_ROWS_STREAM=1).
    static const bool rows_env = getenv("LK_LSF_ROWS_STREAM") && atoi(getenv("LK_LSF_ROWS_STREAM")) == 1;
    const bool rows_stream = fused && rows_env && B > Bc;
    double2 *d_grids2B = nullptr;
    if (rows_stream) {
        d_grids2B = (double2 *)h->ws.alloc((size_t)Bc * 3 * (nfft + pad_elems) * 16);
        LK_REQUIRE(d_grids2B != nullptr, "workspace exhausted");
        if (!h->s_rows) {
            LK_HIP_CHECK(hipStreamCreateWithFlags(&h->s_rows, hipStreamNonBlocking));
            for (int i = 0; i < 4; ++i) LK_HIP_CHECK(hipEventCreateWithFlags(&h->ev_rows[i], hipEventDisableTiming));
        }
    }
    hipEvent_t *ev_spread = &h->ev_aux[0], *ev_cols = &h->ev_aux[2];  // [2] each, indexed by the grid buffer
    const size_t ldsA = ((size_t)CT * N1 + N1 / 2 + 1) * 16, ldsB = ((size_t)RT * N2 + N2 / 2 + 1) * 16;
    int chunk = 0;
    bool rows_pending[2] = {false, false};
    for (int b0 = 0; b0 < B; b0 += Bc, ++chunk) {
        const int nb = std::min(Bc, B - b0);
        const int buf = two_streams ? (chunk & 1) : 0;
        double2 *gr = buf ? d_gridsB : d_grids;
        hipStream_t ss = two_streams ? h->s_aux : stream;  // the spreader's stream
        if (two_streams && chunk >= 2) LK_HIP_CHECK(hipStreamWaitEvent(ss, ev_cols[buf], 0));  // chunk-2's step 1 has read gr
        if (fused_spread) {
            const FusedArgs fa{d_off, d_stats, b0, f0, df, M, fit_mean, normalization, scale, power};
            const SpreadArgs sa{t, d_w, d_wy, d_off, d_stats, d_tab16, b0, ntab16, nfft, fit_mean, f0, df};
            LK_REQUIRE(launch_cols_pruned(lp, m1, m2, nb * 3, gr, d_rows + (size_t)b0 * 4, d_grids2, perm_env, tpad_env, sa, stream),
                       "no pruned column kernel for 2^%d rows", lp);
            LK_REQUIRE(launch_rows_power(m1, m2, nb, d_grids2, fa, PRUNED_CT, stream, perm_env ? lp : 0, tpad_env),
                       "no step-2 kernel for this layout");
            continue;
        }
        if (reg_path) {
            hipLaunchKernelGGL(lsf_zero_kernel, dim3((N1 + 7) / 8, nb * 3), dim3(256), 0, ss, gr, m1, m2,
                               d_rows + (size_t)b0 * 4);
        } else {
            LK_HIP_CHECK(hipMemsetAsync(gr, 0, (size_t)nb * 3 * nfft * 16, ss));
        }
        hipLaunchKernelGGL(lsf_scatter_kernel, dim3((unsigned)((nmax + 255) / 256), nb), dim3(256), 0, ss, t, d_w,
                           d_wy, d_off, d_stats, b0, f0, df, nfft, fit_mean, gr,
                           reg_path ? d_rows + (size_t)b0 * 4 : (const int *)nullptr);
        if (reg_path)
            hipLaunchKernelGGL(lsf_spread_owner_kernel, dim3((unsigned)((nfft + SPREAD_W - 1) / SPREAD_W), nb, 3),
                               dim3(256), 0, ss, t, d_w, d_wy, d_off, d_stats, b0, f0, df, nfft, m2, fit_mean,
                               gr, d_rows + (size_t)b0 * 4, d_tab, ntab);
        if (two_streams) {
            LK_HIP_CHECK(hipEventRecord(ev_spread[buf], ss));
            LK_HIP_CHECK(hipStreamWaitEvent(stream, ev_spread[buf], 0));
        }
        if (reg_path) {
            const FusedArgs fa{d_off, d_stats, b0, f0, df, M, fit_mean, normalization, scale, power};
            if (fused) {
                // step 1 into the second buffer in the tiled layout, step 2 + closed form straight to `power`
                const int ib = rows_stream ? (chunk & 1) : 0;
                double2 *g2 = ib ? d_grids2B : d_grids2;
                if (rows_stream && rows_pending[ib]) LK_HIP_CHECK(hipStreamWaitEvent(stream, h->ev_rows[2 + ib], 0));  // chunk - 2's rows have read g2
                if (lp) {
                    LK_REQUIRE(launch_cols_pruned(lp, m1, m2, nb * 3, gr, d_rows + (size_t)b0 * 4, g2, perm_env, tpad_env, SpreadArgs{}, stream),
                               "no pruned column kernel for 2^%d rows", lp);
                } else {
                    launch_cols_reg(m1, m2, nb * 3, gr, d_rows + (size_t)b0 * 4, g2, tw, stream);
                }
                if (two_streams) LK_HIP_CHECK(hipEventRecord(ev_cols[buf], stream));
                if (rows_stream) {
                    LK_HIP_CHECK(hipEventRecord(h->ev_rows[ib], stream));            // columns of this chunk are in g2
                    LK_HIP_CHECK(hipStreamWaitEvent(h->s_rows, h->ev_rows[ib], 0));
                    LK_REQUIRE(launch_rows_power(m1, m2, nb, g2, fa, lp ? PRUNED_CT : tw, h->s_rows, (lp && perm_env) ? lp : 0, lp ? tpad_env : 0),
                               "no step-2 kernel for this layout");
                    LK_HIP_CHECK(hipEventRecord(h->ev_rows[2 + ib], h->s_rows));    // g2 may be overwritten after this
                    rows_pending[ib] = true;
                    continue;
                }
                LK_REQUIRE(launch_rows_power(m1, m2, nb, g2, fa, lp ? PRUNED_CT : tw, stream, (lp && perm_env) ? lp : 0, lp ? tpad_env : 0),
                           "no step-2 kernel for this layout");
                continue;
            }
            launch_cols_reg(m1, m2, nb * 3, gr, d_rows + (size_t)b0 * 4, nullptr, 1, stream);
            launch_rows_reg(m1, m2, nb * 3, gr, (int)M, d_spec, stream);
        } else {
            hipLaunchKernelGGL(fft_cols_kernel, dim3(N2 / CT, nb * 3), dim3(256), ldsA, stream, gr, m1, m2, CT);
            hipLaunchKernelGGL(fft_rows_kernel, dim3(N1 / RT, nb * 3), dim3(256), ldsB, stream, gr, m1, m2, RT,
                               (int)M, d_spec);
        }
        hipLaunchKernelGGL(lsf_power_kernel, dim3((unsigned)((M + 255) / 256), nb), dim3(256), 0, stream, d_spec,
                           d_off, d_stats, b0, f0, df, M, fit_mean, normalization, scale, power);
    }
    for (int ib = 0; ib < 2; ++ib)  // the caller's stream sees every row transform finished
        if (rows_pending[ib]) LK_HIP_CHECK(hipStreamWaitEvent(stream, h->ev_rows[2 + ib], 0));
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ fastchi2 launcher
int lsfastchi2_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *dy,
                      double f0, double df, int64_t M, int nterms, int fit_mean, int center_data, int normalization,
                      const double *scale, int oversampling, double *power, hipStream_t stream) {
    LK_REQUIRE(nterms >= 1 && nterms <= LK_MAX_NTERMS, "nterms must be between 1 and %d (got %d)", LK_MAX_NTERMS, nterms);
    if (nterms == 1)
        return lsfast_launch(h, B, n_off_host, t, y, dy, f0, df, M, fit_mean, center_data, normalization, scale,
                             oversampling, power, stream);
    LK_REQUIRE(B >= 0 && n_off_host != nullptr, "bad batch description");
    LK_REQUIRE(M >= 0, "M must be >= 0");
    if (B == 0 || M == 0) return LK_OK;
    LK_REQUIRE(t && y && power, "t, y, power must be non-NULL");
    LK_REQUIRE(f0 >= 0.0, "Frequencies must be positive");
    LK_REQUIRE(df > 0.0, "Frequency steps must be positive");
    LK_REQUIRE(oversampling >= 1, "oversampling must be >= 1");
    LK_REQUIRE(normalization >= LK_NORM_STANDARD && normalization <= LK_NORM_LK_PSD, "unknown normalization %d",
               normalization);
    LK_REQUIRE(n_off_host[0] == 0, "n_off[0] must be 0");
    int64_t nmax = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t n = n_off_host[b + 1] - n_off_host[b];
        LK_REQUIRE(n >= 1 && n < ((int64_t)1 << 30), "target %d has %lld cadences", b, (long long)n);
        nmax = std::max(nmax, n);
    }
    const int m = ilog2_ceil((long long)oversampling * (long long)M);
    LK_REQUIRE(m >= 2 && m <= 24, "FFT grid of 2^%d points is outside the supported range (2^2 .. 2^24: the in-LDS column\n"
               "transform of longer grids does not fit 160 KB)", m);
    const int nfft = 1 << m, m1 = (m + 1) / 2, m2 = m / 2;
    const int N1 = 1 << m1, N2 = 1 << m2;
    const int NG = 3 * nterms;
    const size_t ntot = (size_t)n_off_host[B];
    const size_t per_target = (size_t)NG * nfft * 16;
    const int Bc = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, ((size_t)2 << 30) / per_target));
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 8 + (size_t)B * sizeof(FastStats) + 2 * (ntot * 8 + 256) +
                           (size_t)Bc * per_target + (size_t)Bc * NG * M * 16 + 16384);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    FastStats *d_stats = (FastStats *)h->ws.alloc((size_t)B * sizeof(FastStats));
    double *d_w = (double *)h->ws.alloc(ntot * 8), *d_wy = (double *)h->ws.alloc(ntot * 8);
    double2 *d_grids = (double2 *)h->ws.alloc((size_t)Bc * per_target);
    double2 *d_spec = (double2 *)h->ws.alloc((size_t)Bc * NG * M * 16);
    rc = h->stage.copy(d_off, n_off_host, (size_t)(B + 1) * 8, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(lsf_prep_kernel, dim3(B), dim3(256), 0, stream, t, y, dy, d_off, (fit_mean || center_data) ? 1 : 0,
                       d_w, d_wy, d_stats, df, nfft, m2, (int *)nullptr, (int *)nullptr, 0);
    const bool reg_path = m1 >= 4 && m1 <= 10 && m2 >= 4 && m2 <= 10;
    static bool attr_set = false;
    if (!attr_set) {
        LK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fft_cols_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        LK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fft_rows_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        attr_set = true;
    }
    const int CT = std::max(1, std::min(N2, 4096 / N1)), RT = std::max(1, std::min(N1, 4096 / N2));
    const size_t ldsA = ((size_t)CT * N1 + N1 / 2 + 1) * 16, ldsB = ((size_t)RT * N2 + N2 / 2 + 1) * 16;
    for (int b0 = 0; b0 < B; b0 += Bc) {
        const int nb = std::min(Bc, B - b0);
        LK_HIP_CHECK(hipMemsetAsync(d_grids, 0, (size_t)nb * per_target, stream));
        hipLaunchKernelGGL(lsf_scatter_multi_kernel, dim3((unsigned)((nmax + 255) / 256), nb), dim3(256), 0, stream, t, d_w,
                           d_wy, d_off, d_stats, b0, f0, df, nfft, nterms, d_grids);
        if (reg_path) {
            launch_cols_reg(m1, m2, nb * NG, d_grids, nullptr, nullptr, 1, stream);
            launch_rows_reg(m1, m2, nb * NG, d_grids, (int)M, d_spec, stream);
        } else {
            hipLaunchKernelGGL(fft_cols_kernel, dim3(N2 / CT, nb * NG), dim3(256), ldsA, stream, d_grids, m1, m2, CT);
            hipLaunchKernelGGL(fft_rows_kernel, dim3(N1 / RT, nb * NG), dim3(256), ldsB, stream, d_grids, m1, m2, RT,
                               (int)M, d_spec);
        }
        const dim3 pg((unsigned)((M + 255) / 256), nb);
#define LK_FC2(NT)                                                                                                    \
    hipLaunchKernelGGL(lsf_chi2_power_kernel<NT>, pg, dim3(256), 0, stream, d_spec, d_off, d_stats, b0, f0, df, M,    \
                       fit_mean, normalization, scale, power)
        switch (nterms) {
            case 2: LK_FC2(2); break;
            case 3: LK_FC2(3); break;
            default: LK_FC2(4); break;
        }
#undef LK_FC2
    }
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
