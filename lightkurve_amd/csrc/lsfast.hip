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
// This is the HBM-bound formulation of the path.  Per target, three complex grids of Nfft = bitceil(5 M) points:
//   lsf_prep_kernel           weights, centring, per-target sums, the rows of each grid that can hold samples, and the
//                             per-1024-cell cadence tables of the spreader
//   lsf_spread_owner_kernel   extirpolation without global atomics (time-sorted targets whose 2f grid does not wrap): a
//                             workgroup owns 1024 cells, each of its waves 256 of them, accumulated in LDS in a fixed
//                             order (lane-ordered ds_add_f64: bitwise reproducible) and written once, zeros included.
//                             Other targets: lsf_zero_kernel + lsf_scatter_kernel (global atomics).
//   fft_cols_pruned_kernel    step 1 of a hand-written four-step FFT (Nfft = N1 x N2): only P << N1 rows hold samples,
//                             so the N1-point column transform is N1 / P interleaved P-point transforms in registers;
//                             output x inter-step twiddle into a column-tiled intermediate [c / 16][k1][c % 16]
//   fft_rows_power_kernel     step 2 fused with the closed form: the three spectra never reach HBM; the M powers are
//                             written and (peaks entry) a per-workgroup (max, argmax) partial, reduced by lsf_peaks_kernel
// Register FFTs cover 2^4 <= N1, N2 <= 2^10; other sizes run the in-LDS radix-2 kernels (fft_cols_kernel,
// fft_rows_kernel + lsf_power_kernel), which the multi-term fastchi2 path (3 nterms grids per target) shares with
// fft_cols_reg_kernel / fft_rows_reg_kernel.  Algorithmic HBM traffic per target at configs[1]: the intermediate out and
// in (2 x 3 x 16 B x Nfft = 50 MB), the sample-bearing rows out and in (~6 MB), 40 B per cadence, 8 B per frequency.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "lk_common.hpp"
#include "ls_epilogue.hpp"

namespace lk {

constexpr int SPREAD_W_C = 1024;  // cells owned by one workgroup of lsf_spread_owner_kernel

struct FastStats {
    double wsum, ybar, YY, t0;
    double yws;   // sum w (y - ybar) (the bias entry of X^T y in the multi-term solve)
    double vmax;  // max over the cadences of max(w, |w (y - ybar)|): the scale of the scatter kernels' quantum
};

// per target: weights, mean about y[0], YY, t0 = min t; w[i] (normalised) and wy[i] = w (y - ybar)
constexpr int PREP_NT = 1024;
__global__ __launch_bounds__(PREP_NT) void lsf_prep_kernel(const double *__restrict__ t, const double *__restrict__ y,
                                                        const double *__restrict__ dy,
                                                        const int64_t *__restrict__ n_off, int center,
                                                        double *__restrict__ w_out, double *__restrict__ wy_out,
                                                        FastStats *__restrict__ stats, double df, int nfft, int m2,
                                                        int *__restrict__ rows_used, int *__restrict__ spread_tab,
                                                        int ntab) {
    constexpr int NT = PREP_NT;
    __shared__ double sh[NT];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = n_off[b], n = n_off[b + 1] - lo;
    auto bsum = [&](double x) {
        sh[tid] = x;
        __syncthreads();
        for (int s = NT / 2; s > 0; s >>= 1) {
            if (tid < s) sh[tid] += sh[tid + s];
            __syncthreads();
        }
        const double r = sh[0];
        __syncthreads();
        return r;
    };
    // (the kernel is bound by its sweeps over the cadences: one for the time statistics, one for the tables, two over y)
    double acc = 0.0, tmin = INFINITY, tmax = -INFINITY;
    int unsorted = 0;
    for (int64_t i = tid; i < n; i += NT) {
        if (dy) {
            const double d = dy[lo + i];
            acc += 1.0 / (d * d);
        }
        const double ti = t[lo + i];
        tmin = fmin(tmin, ti);
        tmax = fmax(tmax, ti);
        if (i + 1 < n) unsorted |= (t[lo + i + 1] < ti) ? 1 : 0;
    }
    const double wsum = dy ? bsum(acc) : (double)n;
    sh[tid] = tmin;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = fmin(sh[tid], sh[tid + s]);
        __syncthreads();
    }
    const double t0 = sh[0];
    __syncthreads();
    sh[tid] = tmax;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
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
        const double W = (double)SPREAD_W_C, dn = (double)nfft;
        auto posn = [&](double tt, double dff) {
            const double x = (tt - t0) * dn * dff;
            return x < dn ? x : fmod(x, dn);  // the reference's fmod; the identity for every target the spreader takes
        };
        // grids 0 and 1 share df (identical tables, both written: the spreader indexes them by grid), grid 2 uses 2 df
        for (int64_t i = tid; i < n; i += NT) {
            const double ti = t[lo + i], tp = i > 0 ? t[lo + i - 1] : 0.0;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const double dff = df * (gg ? 2.0 : 1.0);
                const double p = posn(ti, dff), pp = i > 0 ? posn(tp, dff) : -1e300;
                int *lo_a = tab + (size_t)(gg ? 4 : 0) * ntab, *hi_a = lo_a + ntab;
                int *lo_b = gg ? nullptr : tab + (size_t)2 * ntab, *hi_b = gg ? nullptr : lo_b + ntab;
                // thresholds x_k = k W - 4 with pp < x_k <= p
                long long k0 = i > 0 ? (long long)floor((pp + 4.0) / W) + 1 : 0, k1 = (long long)floor((p + 4.0) / W);
                for (long long k = max(k0, 0ll); k <= min(k1, (long long)ntab - 1); ++k) {
                    lo_a[k] = (int)i;
                    if (lo_b) lo_b[k] = (int)i;
                }
                k0 = i > 0 ? (long long)floor((pp - 3.0) / W) + 1 : 0;
                k1 = (long long)floor((p - 3.0) / W);
                for (long long k = max(k0, 0ll); k <= min(k1, (long long)ntab - 1); ++k) {
                    hi_a[k] = (int)i;
                    if (hi_b) hi_b[k] = (int)i;
                }
            }
        }
        // thresholds beyond the last cadence
        for (int g = 0; g < 3; ++g) {
            int *lo_tab = tab + (size_t)(2 * g) * ntab, *hi_tab = lo_tab + ntab;
            const double pl = posn(t[lo + n - 1], df * (g == 2 ? 2.0 : 1.0));
            for (int k = tid; k < ntab; k += NT) {
                if ((double)k * W - 4.0 > pl) lo_tab[k] = (int)n;
                if ((double)k * W + 3.0 > pl) hi_tab[k] = (int)n;
            }
        }
    }
    const double y0 = y[lo];
    double ybar = 0.0;
    if (center) {
        acc = 0.0;
        for (int64_t i = tid; i < n; i += NT) {
            const double d = dy ? dy[lo + i] : 1.0;
            acc = fma((1.0 / (d * d)) / wsum, y[lo + i] - y0, acc);
        }
        ybar = bsum(acc) + y0;
    }
    acc = 0.0;
    double acc2 = 0.0, vmx = 0.0;
    for (int64_t i = tid; i < n; i += NT) {
        const double d = dy ? dy[lo + i] : 1.0;
        const double w = (1.0 / (d * d)) / wsum;
        const double yc = y[lo + i] - ybar;
        acc = fma(w * yc, yc, acc);
        acc2 += w * yc;
        vmx = fmax(vmx, fmax(w, fabs(w * yc)));
        w_out[lo + i] = w;
        wy_out[lo + i] = w * yc;
    }
    const double YY = bsum(acc);
    const double yws = bsum(acc2);
    sh[tid] = vmx;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = fmax(sh[tid], sh[tid + s]);
        __syncthreads();
    }
    if (tid == 0) stats[b] = FastStats{wsum, ybar, YY, t0, yws, sh[0]};
}

// astropy extirpolate (M = 4) of one complex sample h at position x into grid[0..nfft)
// The scatter kernels add with GLOBAL atomics, whose order is not fixed.  To be reproducible bit for bit all the same,
// every addend is first rounded to a multiple of a quantum q = 2^e chosen per target (2^-50 of the largest possible
// addend, i.e. about the addend's own last bit): sums of such multiples are exact in double as long as they stay below
// 2^53 q = 4 x the largest possible addend, and exact additions commute.  That holds wherever the grid is sparsely
// filled — the 5-fold oversampled grids of real periodograms receive ~0.15 cadences per cell; a cell that piles up more
// (tiny grids, many coinciding times) is rounded like any double sum: as accurate as before, merely no longer
// order-independent.  (A coarser quantum with more headroom was tried first: 2^-46 cost 1e-9 of the power where the
// five-point fit is ill-conditioned.)
struct Quantum {
    double q, iq;
    __device__ __forceinline__ double operator()(double v) const { return q > 0.0 ? rint(v * iq) * q : v; }
};
__device__ __forceinline__ Quantum make_quantum(double vmax) {
    const double q = (vmax > 0.0 && isfinite(vmax)) ? ldexp(1.0, ilogb(1.25 * vmax) - 49) : 0.0;
    return Quantum{q, q > 0.0 ? 1.0 / q : 0.0};
}

__device__ __forceinline__ void extirpolate4(double2 *__restrict__ grid, int nfft, double x, double hr, double hi,
                                             const Quantum &Q) {
    if (fmod(x, 1.0) == 0.0) {
        const int i = (int)x;
        unsafeAtomicAdd(&grid[i].x, Q(hr));
        unsafeAtomicAdd(&grid[i].y, Q(hi));
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
    unsafeAtomicAdd(&grid[ilo + 3].x, Q(nr / q3));
    unsafeAtomicAdd(&grid[ilo + 3].y, Q(ni / q3));
    unsafeAtomicAdd(&grid[ilo + 2].x, Q(nr / q2));
    unsafeAtomicAdd(&grid[ilo + 2].y, Q(ni / q2));
    unsafeAtomicAdd(&grid[ilo + 1].x, Q(nr / q1));
    unsafeAtomicAdd(&grid[ilo + 1].y, Q(ni / q1));
    unsafeAtomicAdd(&grid[ilo].x, Q(nr / q0));
    unsafeAtomicAdd(&grid[ilo].y, Q(ni / q0));
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
    const Quantum Q = make_quantum(stats[b].vmax);
    for (int fac = 1; fac <= 2; ++fac) {
        const double dff = df * (double)fac, f0f = f0 * (double)fac;
        double c = 1.0, s = 0.0;
        if (f0f > 0.0) sincos(twopi * f0f * tt, &s, &c);
        const double tn = fmod(tt * (double)nfft * dff, (double)nfft);
        if (fac == 1) {
            extirpolate4(g0, nfft, tn, wyi * c, wyi * s, Q);
            if (fit_mean) extirpolate4(g1, nfft, tn, wi * c, wi * s, Q);
        } else {
            extirpolate4(g2, nfft, tn, wi * c, wi * s, Q);
        }
    }
}

// Owner-computes spreading for ordered targets (sorted time, no wrap): grid positions grow with the cadence index,
// so workgroup (x, target, g) owns cells [x W, (x+1) W) of grid g and knows from the prep kernel's tables which cadences
// reach them.  Each of its four waves owns 256 of the cells: it walks ALL the workgroup's cadences and adds the stencil
// points that fall in its own cells with LDS atomics.  Same-address lanes of one ds_add_f64 are applied in lane order and
// a wave's LDS instructions execute in program order (tools/microbench/lds_atomic_order.hip), and no cell is touched by
// two waves, so the accumulation order is fixed: two runs give bit-identical grids.  The cells are written once with
// plain, coalesced stores — zeros included, so no memset and no global atomics.  g: 0 = w*y at f, 1 = w at f, 2 = w at 2f.
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
    t += lo;
    const double t0 = stats[b].t0;
    const double fac = g == 2 ? 2.0 : 1.0;
    const double dff = df * fac, f0f = f0 * fac;
    const double *amp = (g == 0 ? wy : w) + lo;
    // cadences whose 4-point stencils can reach this workgroup's cells (lsf_prep_kernel's tables)
    const int *tab = spread_tab + ((size_t)b * 6 + 2 * g) * ntab;
    const int i_lo = tab[blockIdx.x], i_hi = tab[ntab + min((int)blockIdx.x + 1, ntab - 1)];
    for (int c = tid; c < SPREAD_W; c += 256) acc[c] = make_double2(0.0, 0.0);
    __syncthreads();
    const double twopi = 6.283185307179586;
    const int lane = tid & 63, w_lo = c_lo + (tid >> 6) * (SPREAD_W / 4), w_hi = min(w_lo + SPREAD_W / 4, c_hi);
    auto add = [&](int cell, double vr, double vi) {
        if (cell >= w_lo && cell < w_hi) {  // this wave's cells only
            unsafeAtomicAdd(&acc[cell - c_lo].x, vr);
            unsafeAtomicAdd(&acc[cell - c_lo].y, vi);
        }
    };
    // the wave's own cadences: positions grow with the cadence index, so those whose stencils reach [w_lo, w_hi) are one
    // run inside [i_lo, i_hi), found by 64-way probes (wave ballots)
    auto pos = [&](int i) { return fmod((t[i] - t0) * (double)nfft * dff, (double)nfft); };
    auto lower = [&](double x) -> int {  // first cadence in [i_lo, i_hi) with position >= x
        int lo_i = i_lo, hi_i = i_hi;
        while (hi_i - lo_i > 64) {
            const int stride = (hi_i - lo_i + 63) / 64;
            const int ip = lo_i + lane * stride;
            const int cnt = __popcll(__ballot(ip < hi_i && pos(ip) < x));  // probes below x form a prefix
            if (cnt == 0) {
                hi_i = lo_i;
                break;
            }
            const int nlo = lo_i + (cnt - 1) * stride + 1;
            hi_i = min(lo_i + cnt * stride, hi_i);
            lo_i = nlo;
        }
        const int i2 = lo_i + lane;
        return lo_i + __popcll(__ballot(i2 < hi_i && pos(i2) < x));
    };
    const int wi_lo = lower((double)w_lo - 4.0), wi_hi = lower((double)w_hi + 3.0);
    for (int i = wi_lo + lane; i < wi_hi; i += 64) {
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

// step 1 (register version): CT columns c0..c0+CT-1; rows >= rows_used[g] are known zeros and are not loaded
template <int LA, int LB>
__global__ __launch_bounds__(256) void fft_cols_reg_kernel(double2 *__restrict__ grids, int m2, int CT,
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
    const int twl = 31 - __clz(max(tw, 1));  // tw is a power of two
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
        if (gout)
            gout[((size_t)blockIdx.y << (m1 + m2)) + ((((size_t)(c >> twl) << m1) + k1) << twl) + (c & (tw - 1))] = cmul(v, w);
        else
            G[(size_t)k1 * N2 + c] = cmul(v, w);
        w = cmul(w, step);
    };
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
// once and kept in registers over the Q passes; rows keep their natural order k1 = Q q + s.
constexpr int PRUNED_CT = 16;

template <int LP>
__global__ __launch_bounds__(PRUNED_CT * (1 << ((LP + 1) / 2)), 2) void fft_cols_pruned_kernel(
    const double2 *__restrict__ grids, int m1, int m2, const int *__restrict__ rows_used, double2 *__restrict__ gout) {
    extern __shared__ __attribute__((aligned(16))) double2 lds2[];
    constexpr int LA = (LP + 1) / 2, LB = LP / 2, A = 1 << LA, Bq = 1 << LB, P = 1 << LP, LDT = Bq + 1, FST = A * LDT + 1;
    constexpr int CT = PRUNED_CT;
    const int N1 = 1 << m1, N2 = 1 << m2, Q = N1 >> LP;
    const double2 *G = grids + ((size_t)blockIdx.y << (m1 + m2));
    const size_t tstride = (size_t)CT << m1, gstride = (size_t)(N2 / CT) * tstride;
    double2 *O = gout + (size_t)blockIdx.y * gstride + (size_t)blockIdx.x * tstride;  // this column tile
    const int c0 = blockIdx.x * CT;
    const int ru = rows_used[(blockIdx.y / 3) * 4 + (blockIdx.y % 3)];
    const int tid = threadIdx.x;
    const int f = tid % CT, jk = tid / CT;  // column of the tile; j (phase 1) or ka (phase 2)
    const bool p1 = jk < Bq, p2 = jk < A;
    const double invN1 = 1.0 / (double)N1, invN = 1.0 / (double)((size_t)1 << (m1 + m2));
    double2 xin[A];
    if (p1) {
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
                O[(size_t)(Q * q + s) * CT + f] = cmul(u[brev_c(kb, LB)], w);
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

// step 2 fused with the closed form: one workgroup transforms rows r0..r0+RT-1 of the THREE grids of a target in
// turn (the LDS tile is reused), each phase-2 thread keeps the <= KB outputs it owns that fall below M, and the
// power is computed in registers: the three spectra never go to memory.  With `peaks` every wave also leaves its
// (largest power, lowest index attaining it; NaN skipped) in peaks[target][workgroup][wave] for lsf_peaks_kernel — the
// periodogram's max_power / argmax without a second pass over the B x M spectra.
struct PeakPart {
    double v;
    long long k;  // -1: no finite power in this workgroup's share
};

__device__ __forceinline__ bool peak_better(double v2, long long k2, double v, long long k) {
    return k2 >= 0 && (k < 0 || v2 > v || (v2 == v && k2 < k));  // np.nanargmax: the first maximum wins
}

template <int LA, int LB, int KB>
__global__ __launch_bounds__(512) void fft_rows_power_kernel(const double2 *__restrict__ grids, int m1, int RT,
                                                              const int64_t *__restrict__ n_off,
                                                              const FastStats *__restrict__ stats, int b0, double f0,
                                                              double df, int64_t M, int fit_mean, int norm,
                                                              const double *__restrict__ scale,
                                                              double *__restrict__ power, int tw,
                                                              PeakPart *__restrict__ peaks) {
    extern __shared__ __attribute__((aligned(16))) double2 lds2[];
    constexpr int m2 = LA + LB, A = 1 << LA;
    const int N2 = 1 << m2;
    const int twl = tw ? 31 - __clz(tw) : 0;
    // column-tiled intermediate [c / tw][k1][c % tw] (tw = 0: the natural row-major layout)
    const size_t tstride = tw ? ((size_t)tw << m1) : 0, gstride = (size_t)1 << (m1 + m2);
    const int lb = blockIdx.y, r0 = blockIdx.x * RT;
    double2 keep0[KB], keep1[KB], keep2[KB];
#pragma unroll
    for (int q = 0; q < KB; ++q) keep0[q] = keep1[q] = keep2[q] = make_double2(0.0, 0.0);
    auto transform = [&](int g, double2(&keep)[KB]) {
        const double2 *G = grids + (size_t)(lb * 3 + g) * gstride;
        auto load = [&](int f, int c) -> double2 {
            return tw ? G[(size_t)(c >> twl) * tstride + ((size_t)(r0 + f) << twl) + (c & (tw - 1))] : G[(size_t)(r0 + f) * N2 + c];
        };
        auto store = [&](int, int, double2 v, int kb) {
            if (kb < KB) keep[kb < KB ? kb : 0] = v;
        };
        if (tw)
            block_fft<LA, LB, 2>(RT, lds2, load, store, tw);
        else
            block_fft<LA, LB, 0>(RT, lds2, load, store);
    };
    transform(0, keep0);
    __syncthreads();
    if (fit_mean) transform(1, keep1);
    __syncthreads();
    transform(2, keep2);
    const int tid = threadIdx.x;
    double best_v = 0.0;
    long long best_k = -1;
    if (tid < RT * A) {
        const int ka = tid / RT, f = tid - ka * RT;  // the phase-2 mapping of block_fft
        const int b = b0 + lb;
        const FastStats st = stats[b];
        const double nn = (double)(n_off[b + 1] - n_off[b]);
        const double sc = scale ? scale[b] : 1.0;
        const int k1 = r0 + f;
        // e^{2 pi i t0 f} for this thread's outputs k = k1 + N1 (ka + A kb): one sincos for kb = 0 and one for the step
        // between consecutive kb (a rotation by 2 pi t0 df N1 A), the 2f phase by the double-angle formulas (its argument is
        // exactly twice the 1f one) — 2 sincos per thread instead of 2 per output; the products drift by < 1e-15 over KB <= 8
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
                const double pw = gls_power_sums(a.y, a.x, bq.y, bq.x, c2.y, c2.x, fit_mean, norm, st.YY, 0.5 * st.wsum, nn, sc);
                power[(size_t)b * (size_t)M + k] = pw;
                if (pw == pw && peak_better(pw, k, best_v, best_k)) {  // ascending k within the thread: strict > would do
                    best_v = pw;
                    best_k = k;
                }
            }
            const double nc = ph_c * st_c - ph_s * st_s, ns = ph_s * st_c + ph_c * st_s;
            ph_c = nc;
            ph_s = ns;
        }
    }
    if (peaks == nullptr) return;
    // one partial per WAVE (shuffles only: no barrier, no LDS): lsf_peaks_kernel reduces them
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_xor(best_v, o);
        const long long k2 = __shfl_xor(best_k, o);
        if (peak_better(v2, k2, best_v, best_k)) {
            best_v = v2;
            best_k = k2;
        }
    }
    if ((tid & 63) == 0)
        peaks[((size_t)lb * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (tid >> 6)] = PeakPart{best_v, best_k};
}

// per target: the best of its workgroups' partials -> max_power (NaN if no finite power), argmax (-1 then)
__global__ __launch_bounds__(64) void lsf_peaks_kernel(const PeakPart *__restrict__ peaks, int nparts, int b0,
                                                        double *__restrict__ max_out, int64_t *__restrict__ arg_out) {
    const PeakPart *pp = peaks + (size_t)blockIdx.x * nparts;
    double v = 0.0;
    long long k = -1;
    for (int i = threadIdx.x; i < nparts; i += 64)
        if (peak_better(pp[i].v, pp[i].k, v, k)) {
            v = pp[i].v;
            k = pp[i].k;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_xor(v, o);
        const long long k2 = __shfl_xor(k, o);
        if (peak_better(v2, k2, v, k)) {
            v = v2;
            k = k2;
        }
    }
    if (threadIdx.x == 0) {
        max_out[b0 + blockIdx.x] = k >= 0 ? v : NAN;
        arg_out[b0 + blockIdx.x] = (int64_t)k;
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
    const Quantum Q = make_quantum(stats[b].vmax);
    for (int fac = 1; fac <= 2 * nterms; ++fac) {
        const double dff = df * (double)fac, f0f = f0 * (double)fac;
        double c = 1.0, s = 0.0;
        if (f0f > 0.0) sincos(twopi * f0f * tt, &s, &c);
        const double tn = fmod(tt * (double)nfft * dff, (double)nfft);
        extirpolate4(g0 + (size_t)(nterms + fac - 1) * nfft, nfft, tn, wi * c, wi * s, Q);
        if (fac <= nterms) extirpolate4(g0 + (size_t)(fac - 1) * nfft, nfft, tn, wyi * c, wyi * s, Q);
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

constexpr int COLS_TILE_PTS = 4096;  // points per column tile of the full-length register kernel

template <int LA, int LB>
static void launch_cols_t(lk_handle *h, int m2, int ngrids, double2 *grids, const int *rows_used, double2 *gout, int tw,
                          hipStream_t stream) {
    constexpr int A = 1 << LA, Bq = 1 << LB, n = A * Bq, LDT = Bq + 1, FST = A * LDT + 1;
    const int N2 = 1 << m2;
    const int CT = std::max(1, std::min(N2, std::min(COLS_TILE_PTS / n, 256 / std::max(A, Bq))));
    const int nt = ((CT * std::max(A, Bq) + 63) / 64) * 64;
    (void)want_lds(h, reinterpret_cast<const void *>(fft_cols_reg_kernel<LA, LB>), 160 * 1024);
    hipLaunchKernelGGL((fft_cols_reg_kernel<LA, LB>), dim3(N2 / CT, ngrids), dim3(nt), (size_t)CT * FST * 16, stream, grids,
                       m2, CT, rows_used, gout, tw);
}

// width of the tiled intermediate layout: the column kernel's CT, capped by the row kernel's Bq (both powers of 2)
static int tile_width(int m1, int m2) {
    const int la1 = (m1 + 1) / 2, lb1 = m1 / 2, n1 = 1 << m1;
    const int ct = std::max(1, std::min(1 << m2, std::min(COLS_TILE_PTS / n1, 256 / std::max(1 << la1, 1 << lb1))));
    const int bq2 = 1 << (m2 / 2);
    return std::min(ct, bq2);
}

template <int LA, int LB>
static void launch_rows_t(lk_handle *h, int m1, int ngrids, const double2 *grids, int nkeep, double2 *spec, hipStream_t stream) {
    constexpr int A = 1 << LA, Bq = 1 << LB, n = A * Bq, LDT = Bq + 1, FST = A * LDT + 1;
    const int N1 = 1 << m1;
    const int RT = std::max(1, std::min(N1, std::min(4096 / n, 256 / std::max(A, Bq))));
    const int nt = ((RT * std::max(A, Bq) + 63) / 64) * 64;
    (void)want_lds(h, reinterpret_cast<const void *>(fft_rows_reg_kernel<LA, LB>), 100 * 1024);
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
    PeakPart *peaks;  // nullptr: spectra only
};

// rows per workgroup of the fused step 2 for an N2 = 2^(LA+LB)-point row transform
template <int LA, int LB>
static int rows_power_rt(int m1) {
    constexpr int A = 1 << LA, Bq = 1 << LB, n = A * Bq;
    return std::max(1, std::min(1 << m1, std::min(4096 / n, 256 / std::max(A, Bq))));
}

template <int LA, int LB, int KB>
static void launch_rows_power_t(lk_handle *h, int m1, int ntargets, const double2 *grids, const FusedArgs &a, int tw,
                                hipStream_t stream) {
    constexpr int A = 1 << LA, Bq = 1 << LB, LDT = Bq + 1, FST = A * LDT + 1;
    const int N1 = 1 << m1;
    const int RT = rows_power_rt<LA, LB>(m1);
    const int nt = ((RT * std::max(A, Bq) + 63) / 64) * 64;
    (void)want_lds(h, reinterpret_cast<const void *>(fft_rows_power_kernel<LA, LB, KB>), 160 * 1024);
    hipLaunchKernelGGL((fft_rows_power_kernel<LA, LB, KB>), dim3(N1 / RT, ntargets), dim3(nt), (size_t)RT * FST * 16,
                       stream, grids, m1, RT, a.n_off, a.stats, a.b0, a.f0, a.df, a.M, a.fit_mean, a.norm, a.scale,
                       a.power, tw, a.peaks);
}

// returns false if the (m2, outputs-per-thread) combination has no fused instantiation
static bool rows_power_available(int m1, int m2, int64_t M) {
    const int LA = (m2 + 1) / 2, Aa = 1 << LA;
    const long long k2need = (M + ((long long)1 << m1) - 1) >> m1;
    return m2 >= 4 && m2 <= 10 && (k2need + Aa - 1) / Aa <= 8;
}

#define LK_M2_SWITCH(m2, X) \
    switch (m2) {           \
        case 4: X(2, 2)     \
        case 5: X(3, 2)     \
        case 6: X(3, 3)     \
        case 7: X(4, 3)     \
        case 8: X(4, 4)     \
        case 9: X(5, 4)     \
        default: X(5, 5)    \
    }

// per-target peak partials of the fused step 2: its workgroups per target x waves per workgroup
static int rows_power_parts(int m1, int m2) {
#define LK_X(la, lb)                                                                                   \
    {                                                                                                  \
        const int rt = rows_power_rt<la, lb>(m1);                                                      \
        return ((1 << m1) / rt) * ((rt * std::max(1 << (la), 1 << (lb)) + 63) / 64);                   \
    }
    LK_M2_SWITCH(m2, LK_X)
#undef LK_X
}

static bool launch_rows_power(lk_handle *h, int m1, int m2, int ntargets, const double2 *grids, const FusedArgs &a, int tw,
                              hipStream_t stream) {
    const int LA = (m2 + 1) / 2, Aa = 1 << LA;
    const long long k2need = (a.M + ((long long)1 << m1) - 1) >> m1;
    const int kb = (int)((k2need + Aa - 1) / Aa);
    if (kb > 8 || m2 < 4 || m2 > 10) return false;
#define LK_X(la, lb)                                                            \
    {                                                                           \
        if (kb <= 4)                                                            \
            launch_rows_power_t<la, lb, 4>(h, m1, ntargets, grids, a, tw, stream); \
        else                                                                    \
            launch_rows_power_t<la, lb, 8>(h, m1, ntargets, grids, a, tw, stream); \
        return true;                                                            \
    }
    LK_M2_SWITCH(m2, LK_X)
#undef LK_X
}

static void launch_cols_reg(lk_handle *h, int m1, int m2, int ngrids, double2 *grids, const int *rows_used, double2 *gout,
                            int tw, hipStream_t stream) {
#define LK_X(la, lb)                                                                  \
    {                                                                                 \
        launch_cols_t<la, lb>(h, m2, ngrids, grids, rows_used, gout, tw, stream);      \
        break;                                                                        \
    }
    LK_M2_SWITCH(m1, LK_X)
#undef LK_X
}

static void launch_rows_reg(lk_handle *h, int m1, int m2, int ngrids, const double2 *grids, int nkeep, double2 *spec,
                            hipStream_t stream) {
#define LK_X(la, lb)                                                          \
    {                                                                         \
        launch_rows_t<la, lb>(h, m1, ngrids, grids, nkeep, spec, stream);      \
        break;                                                                \
    }
    LK_M2_SWITCH(m2, LK_X)
#undef LK_X
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
static void launch_cols_pruned_t(lk_handle *h, int m1, int m2, int ngrids, const double2 *grids, const int *rows_used,
                                 double2 *gout, hipStream_t stream) {
    constexpr int LA = (LP + 1) / 2, LB = LP / 2, A = 1 << LA, Bq = 1 << LB, LDT = Bq + 1, FST = A * LDT + 1;
    (void)want_lds(h, reinterpret_cast<const void *>(fft_cols_pruned_kernel<LP>), 160 * 1024);
    hipLaunchKernelGGL((fft_cols_pruned_kernel<LP>), dim3((1 << m2) / PRUNED_CT, ngrids), dim3(PRUNED_CT * A),
                       (size_t)PRUNED_CT * FST * 16, stream, grids, m1, m2, rows_used, gout);
}

static bool launch_cols_pruned(lk_handle *h, int lp, int m1, int m2, int ngrids, const double2 *grids, const int *rows_used,
                               double2 *gout, hipStream_t stream) {
    switch (lp) {
        case 5: launch_cols_pruned_t<5>(h, m1, m2, ngrids, grids, rows_used, gout, stream); return true;
        case 6: launch_cols_pruned_t<6>(h, m1, m2, ngrids, grids, rows_used, gout, stream); return true;
        case 7: launch_cols_pruned_t<7>(h, m1, m2, ngrids, grids, rows_used, gout, stream); return true;
        case 8: launch_cols_pruned_t<8>(h, m1, m2, ngrids, grids, rows_used, gout, stream); return true;
        default: return false;
    }
}

static int ilog2_ceil(long long v) {
    int m = 0;
    while (((long long)1 << m) < v) ++m;
    return m;
}

// max_out / arg_out (both or neither): per-target nanmax / nanargmax of the spectra, from the fused kernel's partials where
// that kernel runs, by argmax_launch over `power` otherwise.
int lsfast_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *dy,
                  double f0, double df, int64_t M, int fit_mean, int center_data, int normalization,
                  const double *scale, int oversampling, double *power, hipStream_t stream, double *max_out,
                  int64_t *arg_out) {
    LK_REQUIRE(B >= 0 && n_off_host != nullptr, "bad batch description");
    LK_REQUIRE(M >= 0, "M must be >= 0");
    if (B == 0 || M == 0) return LK_OK;
    LK_REQUIRE(t && y && power, "t, y, power must be non-NULL");
    LK_REQUIRE((max_out == nullptr) == (arg_out == nullptr), "max_power and argmax must both be given (or both NULL)");
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
    // targets per chunk: the grids (3 x 16 B x Nfft per target) stay within 2 GiB (smaller chunks leave too few workgroups
    // per launch, larger ones fall out of the Infinity Cache: measured)
    const size_t chunk_bytes = (size_t)2 << 30;
    const int Bc = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, chunk_bytes / ((size_t)48 * nfft)));
    const bool reg_path = m1 >= 4 && m1 <= 10 && m2 >= 4 && m2 <= 10;
    const bool fused = reg_path && rows_power_available(m1, m2, M);
    const int nparts = fused ? rows_power_parts(m1, m2) : 0;
    const int ntab = (nfft + SPREAD_W - 1) / SPREAD_W + 2;
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 8 + (size_t)B * sizeof(FastStats) + 2 * (ntot * 8 + 256) +
                           (size_t)Bc * 3 * nfft * 16 * 3 + (size_t)Bc * 3 * M * 16 + (size_t)B * 16 +
                           (size_t)B * 6 * ntab * 4 + (size_t)Bc * (nparts + 1) * sizeof(PeakPart) + 16384);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    FastStats *d_stats = (FastStats *)h->ws.alloc((size_t)B * sizeof(FastStats));
    double *d_w = (double *)h->ws.alloc(ntot * 8), *d_wy = (double *)h->ws.alloc(ntot * 8);
    double2 *d_grids = (double2 *)h->ws.alloc((size_t)Bc * 3 * nfft * 16);
    double2 *d_spec = fused ? nullptr : (double2 *)h->ws.alloc((size_t)Bc * 3 * M * 16);
    rc = h->stage.copy(d_off, n_off_host, (size_t)(B + 1) * 8, stream);
    if (rc) return rc;
    const int tw = tile_width(m1, m2);
    double2 *d_grids2 = fused ? (double2 *)h->ws.alloc((size_t)Bc * 3 * nfft * 16) : nullptr;
    LK_REQUIRE(!fused || d_grids2 != nullptr, "workspace exhausted");
    int *d_rows = (int *)h->ws.alloc((size_t)B * 4 * 4);
    int *d_plan = (int *)h->ws.alloc(64);
    int *d_tab = reg_path ? (int *)h->ws.alloc((size_t)B * 6 * ntab * 4) : nullptr;
    PeakPart *d_peaks = (fused && max_out) ? (PeakPart *)h->ws.alloc((size_t)Bc * nparts * sizeof(PeakPart)) : nullptr;
    hipLaunchKernelGGL(lsf_prep_kernel, dim3(B), dim3(PREP_NT), 0, stream, t, y, dy, d_off, (fit_mean || center_data) ? 1 : 0,
                       d_w, d_wy, d_stats, df, nfft, m2, d_rows, d_tab, ntab);
    // ---- plan: the pruned column kernel applies when every grid of every target keeps its samples in the first
    // P <= 256 rows (P < N1) and the row kernel can read 16-column tiles.  The decision needs two device words, so
    // the call synchronises `stream` once here (20-30 us against a >= 1 ms step).
    int lp = 0, n_unordered = B;
    if (fused) {
        if (!h->h_plan) LK_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h->h_plan), 64, hipHostMallocDefault));
        hipLaunchKernelGGL(lsf_plan_kernel, dim3(1), dim3(256), 0, stream, d_rows, B, d_plan);
        LK_HIP_CHECK(hipMemcpyAsync(h->h_plan, d_plan, 8, hipMemcpyDeviceToHost, stream));
        LK_HIP_CHECK(hipStreamSynchronize(stream));
        n_unordered = h->h_plan[1];
        if (m2 >= 8 && m2 <= 10 && N2 >= PRUNED_CT) {
            const int want = std::max(5, ilog2_ceil(std::max(1, h->h_plan[0])));
            if (want <= 8 && want < m1) lp = want;
        }
    }
    // ---- two streams: the spreader of chunk k+1 (LDS atomics, latency bound) runs on h->s_aux under the FFT kernels
    // of chunk k (HBM / VALU bound) on the caller's stream; the spread grids are double buffered, events order the
    // hand-overs.  All s_aux work is consumed through events by `stream`, so the caller still sees one stream.
    const bool two_streams = fused && B > Bc;
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
    hipEvent_t *ev_spread = &h->ev_aux[0], *ev_cols = &h->ev_aux[2];  // [2] each, indexed by the grid buffer
    if (!reg_path) {
        (void)want_lds(h, reinterpret_cast<const void *>(fft_cols_kernel), 100 * 1024);
        (void)want_lds(h, reinterpret_cast<const void *>(fft_rows_kernel), 100 * 1024);
    }
    const size_t ldsA = ((size_t)CT * N1 + N1 / 2 + 1) * 16, ldsB = ((size_t)RT * N2 + N2 / 2 + 1) * 16;
    int chunk = 0;
    for (int b0 = 0; b0 < B; b0 += Bc, ++chunk) {
        const int nb = std::min(Bc, B - b0);
        const int buf = two_streams ? (chunk & 1) : 0;
        double2 *gr = buf ? d_gridsB : d_grids;
        hipStream_t ss = two_streams ? h->s_aux : stream;  // the spreader's stream
        if (two_streams && chunk >= 2) LK_HIP_CHECK(hipStreamWaitEvent(ss, ev_cols[buf], 0));  // chunk-2's step 1 has read gr
        // targets that are not "ordered" (unsorted time, or a 2f grid that wraps): zero their live rows, scatter with global
        // atomics.  Skipped when the plan found none (the usual batch).
        if (!reg_path) {
            LK_HIP_CHECK(hipMemsetAsync(gr, 0, (size_t)nb * 3 * nfft * 16, ss));
        } else if (n_unordered > 0) {
            hipLaunchKernelGGL(lsf_zero_kernel, dim3((N1 + 7) / 8, nb * 3), dim3(256), 0, ss, gr, m1, m2,
                               d_rows + (size_t)b0 * 4);
        }
        if (!reg_path || n_unordered > 0)
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
        if (fused) {
            // step 1 into the second buffer in the tiled layout, step 2 + closed form straight to `power`
            const FusedArgs fa{d_off, d_stats, b0, f0, df, M, fit_mean, normalization, scale, power, d_peaks};
            if (lp) {
                LK_REQUIRE(launch_cols_pruned(h, lp, m1, m2, nb * 3, gr, d_rows + (size_t)b0 * 4, d_grids2, stream),
                           "no pruned column kernel for 2^%d rows", lp);
            } else {
                launch_cols_reg(h, m1, m2, nb * 3, gr, d_rows + (size_t)b0 * 4, d_grids2, tw, stream);
            }
            if (two_streams) LK_HIP_CHECK(hipEventRecord(ev_cols[buf], stream));
            LK_REQUIRE(launch_rows_power(h, m1, m2, nb, d_grids2, fa, lp ? PRUNED_CT : tw, stream),
                       "no step-2 kernel for this layout");
            if (d_peaks)
                hipLaunchKernelGGL(lsf_peaks_kernel, dim3(nb), dim3(64), 0, stream, d_peaks, nparts, b0, max_out, arg_out);
            continue;
        }
        if (reg_path) {
            launch_cols_reg(h, m1, m2, nb * 3, gr, d_rows + (size_t)b0 * 4, nullptr, 1, stream);
            launch_rows_reg(h, m1, m2, nb * 3, gr, (int)M, d_spec, stream);
        } else {
            hipLaunchKernelGGL(fft_cols_kernel, dim3(N2 / CT, nb * 3), dim3(256), ldsA, stream, gr, m1, m2, CT);
            hipLaunchKernelGGL(fft_rows_kernel, dim3(N1 / RT, nb * 3), dim3(256), ldsB, stream, gr, m1, m2, RT,
                               (int)M, d_spec);
        }
        hipLaunchKernelGGL(lsf_power_kernel, dim3((unsigned)((M + 255) / 256), nb), dim3(256), 0, stream, d_spec,
                           d_off, d_stats, b0, f0, df, M, fit_mean, normalization, scale, power);
    }
    LK_HIP_CHECK(hipGetLastError());
    if (max_out && !fused) return argmax_launch(h, B, M, power, max_out, arg_out, stream);
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ fastchi2 launcher
int lsfastchi2_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *dy,
                      double f0, double df, int64_t M, int nterms, int fit_mean, int center_data, int normalization,
                      const double *scale, int oversampling, double *power, hipStream_t stream) {
    LK_REQUIRE(nterms >= 1 && nterms <= LK_MAX_NTERMS, "nterms must be between 1 and %d (got %d)", LK_MAX_NTERMS, nterms);
    if (nterms == 1)
        return lsfast_launch(h, B, n_off_host, t, y, dy, f0, df, M, fit_mean, center_data, normalization, scale,
                             oversampling, power, stream, nullptr, nullptr);
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
    hipLaunchKernelGGL(lsf_prep_kernel, dim3(B), dim3(PREP_NT), 0, stream, t, y, dy, d_off, (fit_mean || center_data) ? 1 : 0,
                       d_w, d_wy, d_stats, df, nfft, m2, (int *)nullptr, (int *)nullptr, 0);
    const bool reg_path = m1 >= 4 && m1 <= 10 && m2 >= 4 && m2 <= 10;
    if (!reg_path) {
        (void)want_lds(h, reinterpret_cast<const void *>(fft_cols_kernel), 100 * 1024);
        (void)want_lds(h, reinterpret_cast<const void *>(fft_rows_kernel), 100 * 1024);
    }
    const int CT = std::max(1, std::min(N2, 4096 / N1)), RT = std::max(1, std::min(N1, 4096 / N2));
    const size_t ldsA = ((size_t)CT * N1 + N1 / 2 + 1) * 16, ldsB = ((size_t)RT * N2 + N2 / 2 + 1) * 16;
    for (int b0 = 0; b0 < B; b0 += Bc) {
        const int nb = std::min(Bc, B - b0);
        LK_HIP_CHECK(hipMemsetAsync(d_grids, 0, (size_t)nb * per_target, stream));
        hipLaunchKernelGGL(lsf_scatter_multi_kernel, dim3((unsigned)((nmax + 255) / 256), nb), dim3(256), 0, stream, t, d_w,
                           d_wy, d_off, d_stats, b0, f0, df, nfft, nterms, d_grids);
        if (reg_path) {
            launch_cols_reg(h, m1, m2, nb * NG, d_grids, nullptr, nullptr, 1, stream);
            launch_rows_reg(h, m1, m2, nb * NG, d_grids, (int)M, d_spec, stream);
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
