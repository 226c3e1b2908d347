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
#include <type_traits>
#include <vector>

#include "lk_common.hpp"
#include "ls_epilogue.hpp"

#ifndef LSF_STENCIL_PROD
#define LSF_STENCIL_PROD 1   // extirpolation weights as products of three distances (no IEEE divisions); 0: the reference's quotient form
#endif

namespace lk {

typedef double lk_d2v __attribute__((ext_vector_type(2)));  // for the non-temporal load / store builtins

constexpr int SPREAD_W = 1024;   // cells owned by one workgroup of lsf_spread_owner_kernel
constexpr int SPREAD_WW = 256;   // ... and by each of its four waves: the granularity of the prep kernel's cadence tables

struct FastStats {
    double wsum, ybar, YY, t0;
    double yws;   // sum w (y - ybar) (the bias entry of X^T y in the multi-term solve)
    double wmax;  // max over the cadences of w and of |w (y - ybar)|: the scales of the scatter kernels' two quanta (the w
    double vmax;  // grids and the w y grids each get their own — |w y| is 1e-4 .. 1e-6 of w for a normalised light curve)
};

// the normalised weight of cadence i and the weighted, centred flux: the ONE place these are formed (prep sums, spreaders)
__device__ __forceinline__ void cadence_weights(const double *__restrict__ y, const double *__restrict__ dy, int64_t i,
                                                double wsum, double ybar, double &w, double &wy) {
    const double d = dy ? dy[i] : 1.0;
    w = (1.0 / (d * d)) / wsum;
    wy = w * (y[i] - ybar);
}

// per target: weights, the mean about y[0], YY, t0 = min t, and the rows of each grid that can hold samples — ONE sweep over
// the cadences (the kernel is a streaming reduction; round 3 swept four times): with u = 1 / dy^2, z = y - y[0],
//     A = sum u,  Bz = sum u z,  Cz = sum u z^2   ->   ybar = y[0] + Bz / A,   YY = (Cz - Bz^2 / A) / A
// (the shifted-data form: y[0] sits within a few sigma of the mean, so the subtraction costs a digit at most, and a constant
// light curve still centres to exactly 0).  Targets the owner-computes spreader cannot take (unsorted, wrapping) and the
// multi-term path need the scales of the scatter kernels' quanta and the bias sum: a second sweep, for those only.
constexpr int PREP_NT = 512;
__global__ __launch_bounds__(PREP_NT) void lsf_prep_kernel(const double *__restrict__ t, const double *__restrict__ y,
                                                        const double *__restrict__ dy,
                                                        const int64_t *__restrict__ n_off, int center,
                                                        FastStats *__restrict__ stats, double df, int nfft, int m2,
                                                        int *__restrict__ rows_used) {
    constexpr int NT = PREP_NT;
    __shared__ double sh[NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = n_off[b], n = n_off[b + 1] - lo;
    // block reductions: a butterfly inside each wave (every lane ends with the same bits), then the wave results in order
    auto breduce = [&](double x, auto op) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x = op(x, __shfl_xor(x, o));
        if ((tid & 63) == 0) sh[tid >> 6] = x;
        __syncthreads();
        double r = sh[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) r = op(r, sh[w]);
        __syncthreads();
        return r;
    };
    auto bsum = [&](double x) { return breduce(x, [](double a, double b2) { return a + b2; }); };
    auto bmax = [&](double x) { return breduce(x, [](double a, double b2) { return fmax(a, b2); }); };
    const double y0 = y[lo];
    double A = 0.0, Bz = 0.0, Cz = 0.0, tmin = INFINITY, tmax = -INFINITY;
    int unsorted = 0;
#pragma unroll 4
    for (int64_t i = tid; i < n; i += NT) {
        const double ti = t[lo + i], z = y[lo + i] - y0;
        double u = 1.0;
        if (dy) {
            const double d = dy[lo + i];
            u = 1.0 / (d * d);
        }
        A += u;
        Bz = fma(u, z, Bz);
        Cz = fma(u * z, z, Cz);
        tmin = fmin(tmin, ti);
        tmax = fmax(tmax, ti);
        if (i + 1 < n) unsorted |= (t[lo + i + 1] < ti) ? 1 : 0;
    }
    const double wsum = dy ? bsum(A) : (double)n;
    Bz = bsum(Bz);
    Cz = bsum(Cz);
    const double t0 = -bmax(-tmin);
    const double t1 = bmax(tmax);
    const int any_unsorted = __syncthreads_or(unsorted);
    const bool nowrap = (t1 - t0) * (double)nfft * df * 2.0 < (double)nfft - 8.0;
    const bool ordered = !any_unsorted && nowrap;
    if (rows_used) {
        if (tid < 3) {
            // grid rows (of N2 cells) that can receive a sample: cells <= tnorm_max + 1; everything wraps if the span
            // reaches Nfft.  Grids 0, 1 use df, grid 2 uses 2 df.
            const double span = (t1 - t0) * (double)nfft * df * (tid == 2 ? 2.0 : 1.0);
            const int nrows = nfft >> m2;
            rows_used[b * 4 + tid] = (span >= (double)nfft - 8.0) ? nrows : min(nrows, (int)((span + 4.0) / (double)(1 << m2)) + 1);
        }
        // "ordered" targets (time sorted, no wrap of the 2 df grid): grid positions are monotone in the cadence
        // index, so the spreading kernel can own cell ranges and use plain stores instead of global atomics
        if (tid == 0) rows_used[b * 4 + 3] = ordered ? 1 : 0;
    }
    const double delta = center ? Bz / wsum : 0.0;
    const double ybar = center ? y0 + delta : 0.0;
    // sum w (y - ybar)^2 with w = u / wsum: about y0 when centred (then ybar - y0 = delta), about 0 otherwise
    double YY;
    if (center)
        YY = fmax(0.0, (Cz - Bz * delta) / wsum);
    else
        YY = (Cz + y0 * (2.0 * Bz + y0 * wsum)) / wsum;  // sum u (z + y0)^2
    double yws = 0.0, wmx = 0.0, vmx = 0.0;
    if (!(rows_used && ordered)) {
        double acc2 = 0.0;
        for (int64_t i = tid; i < n; i += NT) {
            double w, wy;
            cadence_weights(y, dy, lo + i, wsum, ybar, w, wy);
            acc2 += wy;
            wmx = fmax(wmx, w);
            vmx = fmax(vmx, fabs(wy));
        }
        yws = bsum(acc2);
        wmx = bmax(wmx);
        vmx = bmax(vmx);
    }
    if (tid == 0) stats[b] = FastStats{wsum, ybar, YY, t0, yws, wmx, vmx};
}

// Cadence tables of the owner-computes spreaders (ordered targets: grid positions grow with the cadence index), one thread
// per cadence.  Per frequency step (df: the w y and w grids; 2 df: the third grid) and block k of W = 2^logW cells:
//   lo_tab[k] = first cadence with position >= k W - 4,   hi_tab[k] = first cadence with position >= k W + 3,
// so whoever owns cells [k W, (k + 1) W) walks cadences [lo_tab[k], hi_tab[k + 1]) and needs no search.  W = 256: the waves of
// lsf_spread_owner_kernel; W = 16: the tile rows of the column kernel with the extirpolation fused in.  Every cadence fills
// the thresholds that fall between its predecessor's position and its own; the last one also fills those up to the end of
// the rows that can hold samples (no block beyond them is ever looked up).
constexpr int TAB_PER_LANE = 8;  // cadences per lane of lsf_tables_kernel: a wave takes 512 consecutive ones
__global__ __launch_bounds__(256) void lsf_tables_kernel(const double *__restrict__ t, const int64_t *__restrict__ n_off,
                                                          const FastStats *__restrict__ stats,
                                                          const int *__restrict__ rows_used, double df, int nfft, int m2,
                                                          int logW, int *__restrict__ spread_tab, int ntab) {
    const int b = blockIdx.y;
    if (!rows_used[b * 4 + 3]) return;  // not an owner-spreader target
    const int64_t lo = n_off[b], n = n_off[b + 1] - lo;
    const int lane = threadIdx.x & 63;
    const int64_t base = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * TAB_PER_LANE);  // this wave's first cadence
    if (base >= n) return;
    const double t0 = stats[b].t0;
    int *tab = spread_tab + (size_t)b * 4 * ntab;
    const double iW = 1.0 / (double)(1 << logW), dn = (double)nfft;
    auto posn = [&](double tt) {
        const double x = (tt - t0) * dn * df;
        return x < dn ? x : fmod(x, dn);  // the reference's fmod; the identity for every target the spreader takes
    };
    // positions at df of the wave's cadences (short-lived one-cadence waves were bound by their two dependent loads); at
    // 2 df a position is exactly twice as large (a scaling by 2 commutes with every rounding; ordered targets do not wrap)
    double p1[TAB_PER_LANE];
#pragma unroll
    for (int u = 0; u < TAB_PER_LANE; ++u) {
        const int64_t i = base + 64 * u + lane;
        p1[u] = posn(t[lo + (i < n ? i : n - 1)]);
    }
    const int nf0 = min(ntab, (int)((((long long)rows_used[b * 4 + 0] << m2) >> logW) + 3));
    const int nf2 = min(ntab, (int)((((long long)rows_used[b * 4 + 2] << m2) >> logW) + 3));
    double prev_last = base > 0 ? posn(t[lo + base - 1]) : -1e300;  // the position before this wave's first cadence
#pragma unroll
    for (int u = 0; u < TAB_PER_LANE; ++u) {
        const int64_t i = base + 64 * u + lane;
        double pp1 = __shfl_up(p1[u], 1);
        if (lane == 0) pp1 = prev_last;
        prev_last = __shfl(p1[u], 63);
        if (i >= n) continue;
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const double p = gg ? 2.0 * p1[u] : p1[u], pp = i > 0 ? (gg ? 2.0 * pp1 : pp1) : -1e300;
            int *lo_a = tab + (size_t)(2 * gg) * ntab, *hi_a = lo_a + ntab;
            const int nfill = gg ? nf2 : nf0;
            // thresholds x_k = k W - 4 with pp < x_k <= p
            int k0 = i > 0 ? (int)floor((pp + 4.0) * iW) + 1 : 0, k1 = (int)floor((p + 4.0) * iW);
            for (int k = max(k0, 0); k <= min(k1, ntab - 1); ++k) lo_a[k] = (int)i;
            if (i == n - 1)
                for (int k = max(k1 + 1, 0); k < nfill; ++k) lo_a[k] = (int)n;
            // thresholds x_k = k W + 3
            k0 = i > 0 ? (int)floor((pp - 3.0) * iW) + 1 : 0;
            k1 = (int)floor((p - 3.0) * iW);
            for (int k = max(k0, 0); k <= min(k1, ntab - 1); ++k) hi_a[k] = (int)i;
            if (i == n - 1)
                for (int k = max(k1 + 1, 0); k < nfill; ++k) hi_a[k] = (int)n;
        }
    }
}

// e^{2 pi i f t}: the phase in cycles is reduced exactly first (the rounding error of the product comes back through fma),
// then one sincospi on [-1, 1] — cheaper than sincos with its argument reduction, and at least as close to the reference's
// np.exp(2j pi f0 (t - t0)) as that expression is to the true value
__device__ __forceinline__ void phase_factor(double f, double tt, double &c, double &s) {
    const double p = f * tt, e = fma(f, tt, -p);
    const double r = (p - rint(p)) + e;
    sincospi(2.0 * r, &s, &c);
}

// astropy extirpolate (M = 4), utils.py:14-78: a sample at position x goes to cells i0 .. i0 + n - 1 with weights wt[.]
// (n = 1, weight 1 when x is an integer; else the four Lagrange weights prod_k (x - ilo - k) / (denominator_j (x - ind_j)))
struct Stencil4 {
    int i0, n;
    double wt[4];
};
__device__ __forceinline__ Stencil4 stencil4(double x, int nfft) {
    Stencil4 st;
#if LSF_STENCIL_PROD
    if (x == floor(x)) {  // (what fmod(x, 1.0) == 0.0 says, without the call)
#else
    if (fmod(x, 1.0) == 0.0) {
#endif
        st.i0 = (int)x;
        st.n = 1;
        st.wt[0] = 1.0;
        st.wt[1] = st.wt[2] = st.wt[3] = 0.0;
        return st;
    }
    int ilo = (int)(x - 2.0);  // astype(int): truncation toward zero
    ilo = min(max(ilo, 0), nfft - 4);
    // d1..d3 as the reference forms them: x - ilo - k (same value: ilo + k is exact in double)
    const double d0 = x - (double)ilo, d1 = d0 - 1.0, d2 = d0 - 2.0, d3 = d0 - 3.0;
    const double prod = ((d0 * d1) * d2) * d3;
    // j = 0..3: ind = ilo + 3 - j, denominators 6, -2, 2, -6
    st.i0 = ilo;
    st.n = 4;
#if LSF_STENCIL_PROD
    // prod / (c_j d_j) = the product of the OTHER three distances over c_j: the Lagrange weights without the four IEEE divisions
    // (~25 instructions each in the extirpolation phase of every column tile; same value to an ulp or two, better conditioned
    // next to a grid point)
    (void)prod;
    st.wt[3] = ((d0 * d1) * d2) * (1.0 / 6.0);
    st.wt[2] = ((d0 * d1) * d3) * -0.5;
    st.wt[1] = ((d0 * d2) * d3) * 0.5;
    st.wt[0] = ((d1 * d2) * d3) * (-1.0 / 6.0);
#else
    st.wt[3] = prod / (6.0 * d3);
    st.wt[2] = prod / (-2.0 * d2);
    st.wt[1] = prod / (2.0 * d1);
    st.wt[0] = prod / (-6.0 * d0);
#endif
    return st;
}

// The scatter kernels add with GLOBAL atomics, whose order is not fixed.  To be reproducible bit for bit all the same,
// every addend is first rounded to a multiple of a quantum q = 2^e chosen per target AND per kind of grid (2^-50 of the
// largest possible addend, i.e. about the addend's own last bit) and the multiples are accumulated as 64-BIT INTEGERS in the
// grid cells (atomic add on the cell's bits): integer additions are exact and commute whatever piles up in a cell — 2^13 of
// the largest possible addends fit — and lsf_unquantize_kernel turns the sums into doubles (one rounding, of the final sum)
// before the transform reads them.  (Rounds 2-5 added the rounded addends as doubles: exact, hence order-independent, only
// while a cell's sum stayed below 2^53 q = 4 x the largest addend — true on the 5-fold oversampled grids of real
// periodograms, ~0.15 cadences per cell, but a small grid with a few coinciding stencils gave run-to-run differences in the
// last bit: tests/test_batch_api_gpu.py caught one in round 6.)  The w grids and the w (y - ybar) grids have their own quanta (FastStats::wmax, ::vmax): one shared
// quantum, scaled by w, would leave the w y addends of a normalised low-amplitude light curve only 2^-30 of relative
// precision.  (A coarser quantum with more headroom was tried first: 2^-46 cost 1e-9 of the power where the five-point fit
// is ill-conditioned.)
struct Quantum {
    double q, iq;
    // add v to the cell: as an integer multiple of q on the cell's bits, or (no quantum: a zero or non-finite scale) as a double
    __device__ __forceinline__ void add(double *cell, double v) const {
        if (q > 0.0)
            atomicAdd(reinterpret_cast<unsigned long long *>(cell), (unsigned long long)(long long)rint(v * iq));
        else
            unsafeAtomicAdd(cell, v);
    }
};
__device__ __forceinline__ Quantum make_quantum(double vmax) {
    const double q = (vmax > 0.0 && isfinite(vmax)) ? ldexp(1.0, ilogb(1.25 * vmax) - 49) : 0.0;
    return Quantum{q, q > 0.0 ? 1.0 / q : 0.0};
}

__device__ __forceinline__ void extirpolate4(double2 *__restrict__ grid, const Stencil4 &sp, double hr, double hi,
                                             const Quantum &Q) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < sp.n) {
            Q.add(&grid[sp.i0 + k].x, hr * sp.wt[k]);
            Q.add(&grid[sp.i0 + k].y, hi * sp.wt[k]);
        }
}

// spread every cadence of targets [b0, b0 + nb) into its three grids: 0: w*y at f, 1: w at f, 2: w at 2f
__global__ __launch_bounds__(256) void lsf_scatter_kernel(const double *__restrict__ t, const double *__restrict__ y,
                                                           const double *__restrict__ dy,
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
    const FastStats st = stats[b];
    const double tt = t[lo + i] - st.t0;
    double2 *g0 = grids + (size_t)blockIdx.y * 3 * nfft, *g1 = g0 + nfft, *g2 = g1 + nfft;
    double wi, wyi;
    cadence_weights(y, dy, lo + i, st.wsum, st.ybar, wi, wyi);
    const Quantum Qy = make_quantum(st.vmax), Qw = make_quantum(st.wmax);
    for (int fac = 1; fac <= 2; ++fac) {
        const double dff = df * (double)fac, f0f = f0 * (double)fac;
        double c = 1.0, s = 0.0;
        if (f0f > 0.0) phase_factor(f0f, tt, c, s);
        const Stencil4 sp = stencil4(fmod(tt * (double)nfft * dff, (double)nfft), nfft);
        if (fac == 1) {
            extirpolate4(g0, sp, wyi * c, wyi * s, Qy);
            if (fit_mean) extirpolate4(g1, sp, wi * c, wi * s, Qw);
        } else {
            extirpolate4(g2, sp, wi * c, wi * s, Qw);
        }
    }
}

// Owner-computes spreading for ordered targets (sorted time, no wrap): grid positions grow with the cadence index, so
// workgroup (x, target) owns cells [x W, (x + 1) W) and each of its four waves 256 of them; the prep kernel's tables give
// every wave the cadences whose stencils reach its cells — no search.  A lane takes a cadence, forms its weight, phase
// factor and stencil ONCE and adds the stencil points that fall into the wave's cells with LDS atomics.  blockIdx.z = 0
// serves the w y AND the w grid at df (same positions, same stencil, same phase: one pass, two accumulators), z = 1 the w
// grid at 2 df.  Same-address lanes of one ds_add_f64 are applied in lane order and a wave's LDS instructions execute in
// program order (tools/microbench/lds_atomic_order.hip), and no cell is touched by two waves, so the accumulation order is
// fixed: two runs give bit-identical grids.  The cells are written once with plain, coalesced stores — zeros included, so no
// memset and no global atomics.  Weights come straight from y, dy and the prep statistics (nothing per cadence is staged).
__global__ __launch_bounds__(256) void lsf_spread_owner_kernel(const double *__restrict__ t, const double *__restrict__ y,
                                                                const double *__restrict__ dy,
                                                                const int64_t *__restrict__ n_off,
                                                                const FastStats *__restrict__ stats, int b0, double f0,
                                                                double df, int nfft, int m2, int fit_mean,
                                                                double2 *__restrict__ grids,
                                                                const int *__restrict__ rows_used,
                                                                const int *__restrict__ spread_tab, int ntab) {
    __shared__ double2 acc0[SPREAD_W], acc1[SPREAD_W];
    const int lb = blockIdx.y, tid = threadIdx.x;
    const bool pair = blockIdx.z == 0;
    if (!rows_used[lb * 4 + 3]) return;
    const int g = pair ? 0 : 2;
    const int ncell = rows_used[lb * 4 + g] << m2;  // cells the column transform will read (grids 0 and 1: the same count)
    const int c_lo = blockIdx.x * SPREAD_W, c_hi = min(c_lo + SPREAD_W, ncell);
    if (c_lo >= ncell) return;
    double2 *G0 = grids + ((size_t)lb * 3 + g) * (size_t)nfft, *G1 = G0 + nfft;
    const int b = b0 + lb;
    const int64_t lo = n_off[b];
    const FastStats st = stats[b];
    const double fac = pair ? 1.0 : 2.0;
    const double dff = df * fac, f0f = f0 * fac;
    for (int c = tid; c < SPREAD_W; c += 256) {
        acc0[c] = make_double2(0.0, 0.0);
        acc1[c] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    const int w_lo = c_lo + wv * SPREAD_WW, w_hi = min(w_lo + SPREAD_WW, c_hi);
    if (w_lo < w_hi) {
        const int *tab = spread_tab + ((size_t)b * 4 + (pair ? 0 : 2)) * ntab;
        const int kblk = blockIdx.x * (SPREAD_W / SPREAD_WW) + wv;
        const int i_lo = tab[kblk], i_hi = tab[ntab + min(kblk + 1, ntab - 1)];
        for (int i = i_lo + lane; i < i_hi; i += 64) {
            const double tt = t[lo + i] - st.t0;
            double wi, wyi;
            cadence_weights(y, dy, lo + i, st.wsum, st.ybar, wi, wyi);
            double c = 1.0, s = 0.0;
            if (f0f > 0.0) phase_factor(f0f, tt, c, s);
            const Stencil4 sp = stencil4(fmod(tt * (double)nfft * dff, (double)nfft), nfft);
            const double ar = (pair ? wyi : wi) * c, ai = (pair ? wyi : wi) * s, br = wi * c, bi = wi * s;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cell = sp.i0 + k;
                if (k < sp.n && cell >= w_lo && cell < w_hi) {  // this wave's cells only
                    unsafeAtomicAdd(&acc0[cell - c_lo].x, ar * sp.wt[k]);
                    unsafeAtomicAdd(&acc0[cell - c_lo].y, ai * sp.wt[k]);
                    if (pair && fit_mean) {
                        unsafeAtomicAdd(&acc1[cell - c_lo].x, br * sp.wt[k]);
                        unsafeAtomicAdd(&acc1[cell - c_lo].y, bi * sp.wt[k]);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int c = c_lo + tid; c < c_hi; c += 256) {
        G0[c] = acc0[c - c_lo];
        if (pair) G1[c] = acc1[c - c_lo];  // (fit_mean = 0: zeros — the unused grid stays defined)
    }
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

// Scheduling experiments of round 6 (tools/build_variant.sh ... "-DLSF_PRIO=<bits>"; profiles/r06_lsfast_setprio_ab.txt): wave
// priority 3 over bit 0 the store phase of a pass, bit 1 the extirpolation phase of the column kernel, bit 2 the loads of the
// row kernel.  Release builds compile none of them (measured: no gain at two waves per SIMD).
#ifndef LSF_PRIO
#define LSF_PRIO 0
#endif
#ifndef LSF_TWO_STREAMS
#define LSF_TWO_STREAMS 1   // number of EXTRA streams of the chunk loop (0 .. 3); 1: -2 % on the step (profiles/r06_lsfast_two_streams_ab.txt)
#endif
#ifndef LSF_STAGGER
#define LSF_STAGGER 0
#endif
#ifndef LSF_CHUNK_HALF_GB
#define LSF_CHUNK_HALF_GB 3  // bytes of grids per chunk, in units of 2^29 (1.5 GiB = 60 targets at Nfft = 2^19)
#endif
#define LSF_SETPRIO(bit, p)                                         \
    do {                                                            \
        if (LSF_PRIO & (bit)) __builtin_amdgcn_s_setprio(p);        \
    } while (0)

// Extirpolation fused in (ordered targets, `tab16` given): the workgroup's input — the cells of its 16 columns in the rows
// that can hold samples — is not read from a spread grid but built here: the 16-cell-granular tables of lsf_tables_kernel
// name, for every tile row, the cadences whose 4-point stencils reach it; a wave owns a contiguous range of rows, a group of
// S lanes serves one row (lane k takes the row's k-th cadence: weight, phase factor, stencil), and the stencil points that
// fall into the tile are added to an LDS image [row][16] with ds_add_f64 — same-address lanes in lane order, a wave's LDS
// instructions in program order, no row touched by two waves: a fixed order, bitwise reproducible.  The sample-bearing
// rows (3.4 MB per target) are then never written to or read from HBM, and no spreader kernel runs beside the FFT kernels
// (round 3's ran on a second stream and cost the row kernel as much as it took alone).  Other targets (unsorted times,
// wrapping grids) still come through `grids`, filled by lsf_zero_kernel + lsf_scatter_kernel.
template <int LP>
__global__ __launch_bounds__(PRUNED_CT * (1 << ((LP + 1) / 2)), 2) void fft_cols_pruned_kernel(
    const double2 *__restrict__ grids, int m1, int m2, const int *__restrict__ rows_used, double2 *__restrict__ gout,
    const double *__restrict__ t, const double *__restrict__ y, const double *__restrict__ dy,
    const int64_t *__restrict__ n_off, const FastStats *__restrict__ stats, int b0, double f0, double df, int fit_mean,
    const int *__restrict__ tab16, int ntab16) {
    extern __shared__ __attribute__((aligned(16))) double2 lds2[];
    constexpr int LA = (LP + 1) / 2, LB = LP / 2, A = 1 << LA, Bq = 1 << LB, P = 1 << LP, LDT = Bq + 1, FST = A * LDT + 1;
    constexpr int CT = PRUNED_CT, NT = CT * A;
    const int N1 = 1 << m1, N2 = 1 << m2, Q = N1 >> LP;
    const double2 *G = grids + ((size_t)blockIdx.y << (m1 + m2));
    const size_t tstride = (size_t)CT << m1, gstride = (size_t)(N2 / CT) * tstride;
    double2 *O = gout + (size_t)blockIdx.y * gstride + (size_t)blockIdx.x * tstride;  // this column tile
    const int c0 = blockIdx.x * CT;
    const int lbt = blockIdx.y / 3, g = blockIdx.y % 3;
    const int ru = rows_used[lbt * 4 + g];
    const int tid = threadIdx.x;
    const int f = tid % CT, jk = tid / CT;  // column of the tile; j (phase 1) or ka (phase 2)
    const bool p1 = jk < Bq, p2 = jk < A;
    const double invN1 = 1.0 / (double)N1, invN = 1.0 / (double)((size_t)1 << (m1 + m2));
    double2 xin[A];
    if (tab16 != nullptr && rows_used[lbt * 4 + 3]) {
        double2 *acc = lds2;  // [ru][CT]: ru <= P rows of 256 B fit the exchange tile
        for (int e = tid; e < ru * CT; e += NT) acc[e] = make_double2(0.0, 0.0);
        __syncthreads();
        if (!(g == 1 && !fit_mean)) {  // (the unused grid stays zero)
            LSF_SETPRIO(2, 3);
            const int b = b0 + lbt;
            const int64_t lo = n_off[b];
            const FastStats st = stats[b];
            const double fac = g == 2 ? 2.0 : 1.0, dff = df * fac, f0f = f0 * fac, dn = (double)((size_t)1 << (m1 + m2));
            const int nfft = 1 << (m1 + m2);
            const int *tlo = tab16 + ((size_t)b * 4 + (g == 2 ? 2 : 0)) * ntab16, *thi = tlo + ntab16;
            // S lanes per row: 19 cells hold ~4 cadences at the usual 5 cells per cadence of the df grids, ~2 at 2 df
            const int S = g == 2 ? 4 : 8, rpi = 64 / S;
            const int lane = tid & 63, wv = tid >> 6, rr = lane / S, k = lane - rr * S;
            const int rpw = (ru + NT / 64 - 1) / (NT / 64);
            const int r_end = min(ru, (wv + 1) * rpw);
            // one visit: cadence i's stencil points that fall into tile row r
            auto visit = [&](int r, double ti, double yi, double di) {
                const double tt = ti - st.t0;
                const double wi = (1.0 / (di * di)) / st.wsum, wyi = wi * (yi - st.ybar);  // (= cadence_weights)
                double c = 1.0, sn = 0.0;
                if (f0f > 0.0) phase_factor(f0f, tt, c, sn);
                const double xp = tt * dn * dff;
                const Stencil4 sp = stencil4(xp < dn ? xp : fmod(xp, dn), nfft);
                const double amp = g == 0 ? wyi : wi, ar = amp * c, ai = amp * sn;
                const int cbase = (r << m2) + c0;  // first cell of this tile row
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned col = (unsigned)(sp.i0 + q - cbase);
                    if (q < sp.n && col < (unsigned)CT) {
                        unsafeAtomicAdd(&acc[r * CT + col].x, ar * sp.wt[q]);
                        unsafeAtomicAdd(&acc[r * CT + col].y, ai * sp.wt[q]);
                    }
                }
            };
            // four row groups at a time: their table entries, then their cadences, are all requested before the first is used
            constexpr int UN = 4;
            for (int rb = wv * rpw; rb < r_end; rb += UN * rpi) {
                int ii[UN], ih[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    // clamped row, unconditional loads (under `if (r < r_end)` each pair waits for its own round trip)
                    const int r = rb + u * rpi + rr;
                    const int blk = (min(r, r_end - 1) << (m2 - 4)) + blockIdx.x;
                    ii[u] = tlo[blk] + k;
                    ih[u] = thi[blk + 1];
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    asm volatile("" : "+v"(ii[u]), "+v"(ih[u]));
                    if (rb + u * rpi + rr >= r_end) ii[u] = ih[u] = 0;
                }
                double tv[UN], yv[UN], dv[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const bool ok = ii[u] < ih[u];
                    const int64_t ix = lo + (ok ? ii[u] : 0);
                    tv[u] = t[ix];
                    yv[u] = y[ix];
                    dv[u] = dy ? dy[ix] : 1.0;
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int r = rb + u * rpi + rr;
                    if (ii[u] < ih[u]) visit(r, tv[u], yv[u], dv[u]);
                    // rows with more than S cadences (dense sampling): further rounds, still in cadence order
                    int i = ii[u] + S;
                    while (__any(i < ih[u])) {
                        if (i < ih[u]) visit(r, t[lo + i], y[lo + i], dy ? dy[lo + i] : 1.0);
                        i += S;
                    }
                }
            }
            LSF_SETPRIO(2, 0);
        }
        __syncthreads();
        if (p1) {
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const int r = i * Bq + jk;
                xin[i] = r < ru ? acc[r * CT + f] : make_double2(0.0, 0.0);
            }
        }
        __syncthreads();  // the image is read before the first pass overwrites the tile
    } else if (p1) {
#pragma unroll
        for (int i = 0; i < A; ++i) {
            const int r = i * Bq + jk;
            xin[i] = r < ru ? G[(size_t)r * N2 + c0 + f] : make_double2(0.0, 0.0);
        }
    }
    // Every twiddle of the Q passes is a running product of a handful of roots taken once (the passes used to spend five
    // sincospi each):   pre-twiddle W_N1^{(i Bq + j) s} = (W_N1^j)^s (W_N1^{Bq s})^i,   intra-transform W_P^{j ka},
    // inter-step W_N^{c (Q (ka + A kb) + s)} = W_N^{c Q ka} (W_N^c)^s (W_N^{c Q A})^kb
    auto root = [](double x) {
        double sn, cs;
        sincospi(2.0 * x, &sn, &cs);
        return make_double2(cs, sn);
    };
    const double2 wj = root((double)jk * invN1), wq = root((double)Bq * invN1);  // W_N1^j, W_N1^Bq
    const double2 tw_j = root((double)jk / (double)P);
    const double2 wc = root((double)(c0 + f) * invN), stepc = root((double)((long long)(c0 + f) * Q * A) * invN);
    double2 wjs = make_double2(1.0, 0.0), wqs = wjs;                                  // (W_N1^j)^s, (W_N1^Bq)^s
    double2 wout = root((double)((long long)(c0 + f) * Q * jk) * invN);               // W_N^{c (Q ka + s)}
    for (int s = 0; s < Q; ++s) {
        if (p1) {
            double2 v[A];
            if (s == 0) {
#pragma unroll
                for (int i = 0; i < A; ++i) v[i] = xin[i];
            } else {
                double2 w = wjs;
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    v[i] = cmul(xin[i], w);
                    w = cmul(w, wqs);
                }
            }
            reg_fft<LA>(v);
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
            double2 w = wout;
            LSF_SETPRIO(1, 3);
#pragma unroll
            for (int kb = 0; kb < Bq; ++kb) {
                const int q = jk + A * kb;
                O[(size_t)(Q * q + s) * CT + f] = cmul(u[brev_c(kb, LB)], w);
                w = cmul(w, stepc);
            }
            LSF_SETPRIO(1, 0);
        }
        wjs = cmul(wjs, wj);
        wqs = cmul(wqs, wq);
        wout = cmul(wout, wc);
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

// ---- step 2 for 512-point rows (N2 = 2^9, the configs[1] shape), built for bytes in flight rather than for few LDS passes.
// The generic kernel above runs the 512-point transform as 32 x 16 with a 32-point register FFT: 222 VGPRs and a 70-KB
// exchange tile, i.e. two 4-wave workgroups per CU, each with one grid's 64 KB in flight at a time and only half of its
// threads loading — it waits on HBM with ~40-60 KB outstanding per CU (3.6 TB/s in the round-3 trace).  Here the transform
// is 16 x 32:
//   phase 1  thread (row f, j < 32) loads x[32 i + j], i < 16 — EVERY thread loads, 1 KB contiguous per wave instruction
//            from the [c / 16][k1][c % 16] intermediate — runs a 16-point register FFT and multiplies by W_512^{j ka};
//   exchange through LDS as two 8-byte planes (real, then imaginary): 35 KB instead of 70;
//   phase 2  X[ka + 16 kb] = sum_{j < 32} T[ka][j] W_32^{j kb} is needed for kb < 8 only (k2 < 128 covers M <= 128 N1):
//            a PAIR of threads (ka, e) splits the sum by the parity of j — thread e runs two 8-point FFTs over
//            j = j1 + 4 j2, j1 = e, e + 2, combines them with constant 32nd roots, and the two partial sums meet through
//            one DPP row rotation (lane ^ 8); thread e keeps kb = 4 e .. 4 e + 3 of each grid for the closed form.
// No thread ever holds more than 16 points, so the next grid's 64 KB are requested the moment a grid's points sit in LDS,
// under phase 2.  Loads are non-temporal (each byte is read once), targets go last-written first (the tail of what step 1
// has just written is still in the Infinity Cache).  Measured alone on an 85-target chunk: 430 us = 5.1 TB/s against 560-610 us for the generic kernel
// (tools/microbench/lsfast_rows.hip has the variants that were tried: no prefetch at 3 or 4 waves per SIMD — spills —, two
// register sets, 16-row tiles).
__device__ __forceinline__ double dpp_ror8(double x) {  // the value of lane ^ 8 (rotation by 8 inside rows of 16 lanes)
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

struct Rows512 {
    static constexpr int RT = 8;             // rows per tile
    static constexpr int SK = 34;            // doubles per (row, ka) line: 32 + 2 -> pairs of ka tile the banks
    static constexpr int SF = 16 * SK + 4;   // doubles per row: the 32 lanes of a ds_read_b64 group hit 32 bank pairs
    static constexpr int NT = RT * 32;
    static constexpr size_t LDS_BYTES = (size_t)RT * SF * 8;
};

__global__ __launch_bounds__(Rows512::NT, 2) void fft_rows512_power_kernel(
    const double2 *__restrict__ grids, int m1, int ntargets, const int64_t *__restrict__ n_off,
    const FastStats *__restrict__ stats, int b0, double f0, double df, int64_t M, int fit_mean, int norm,
    const double *__restrict__ scale, double *__restrict__ power, PeakPart *__restrict__ peaks) {
    using R = Rows512;
    constexpr int RT = R::RT, SK = R::SK, SF = R::SF;
    extern __shared__ __attribute__((aligned(16))) double tile512[];
    const int tid = threadIdx.x;
    const size_t gstride = (size_t)1 << (m1 + 9);
    const int tiles_per_target = (1 << m1) / RT;
    // phase-1 identity: lanes (j % 16, row) make 1-KB runs, j / 16 selects the neighbouring column tile
    const int jl = tid & 15, f1 = (tid >> 4) & (RT - 1), jh = tid >> 7, j = jl + 16 * jh;
    // phase-2 identity: row fastest (neighbouring lanes write neighbouring frequencies), the pair bit at lane bit 3
    const int f2 = tid & 7, e = (tid >> 3) & 1, ka = tid >> 4;
    // tile T: targets last-written first
    auto tile_base = [&](int T) {
        const int lbT = ntargets - 1 - T / tiles_per_target, r0T = (T % tiles_per_target) * RT;
        return grids + (size_t)lbT * 3 * gstride + ((size_t)jh << (m1 + 4)) + ((size_t)(r0T + f1) << 4) + jl;
    };
    double2 v[16];
    auto load = [&](const double2 *G) {
        LSF_SETPRIO(4, 3);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const lk_d2v q = __builtin_nontemporal_load(reinterpret_cast<const lk_d2v *>(G + ((size_t)i << (m1 + 5))));
            v[i] = make_double2(q.x, q.y);
        }
        LSF_SETPRIO(4, 0);
    };
    double2 step1, step4;  // W_512^j, W_512^{4 j}
    {
        double s, c;
        sincospi((double)j * (1.0 / 256.0), &s, &c);
        step1 = make_double2(c, s);
        sincospi((double)j * (1.0 / 64.0), &s, &c);
        step4 = make_double2(c, s);
    }
    double2 keep[3][4];
    double *row1 = tile512 + f1 * SF + j;
    const double *row2 = tile512 + f2 * SF + ka * SK + e;
    // one grid: the 16 loaded points of v -> keep[G]; `refill` runs once v's last LDS write has been issued (v is dead)
    auto process = [&](auto Gc, auto refill) {
        constexpr int G = decltype(Gc)::value;
        __builtin_amdgcn_sched_barrier(0);
        reg_fft<4>(v);
        __builtin_amdgcn_sched_barrier(0);
        // W_512^{j ka}, ka = 4 a + b, as (W^{4 j})^a (W^j)^b: a few twiddles live instead of a chain of fifteen
        {
            double2 w = step1;
#pragma unroll
            for (int bb = 1; bb < 4; ++bb) {
#pragma unroll
                for (int a = 0; a < 4; ++a) v[brev_c(4 * a + bb, 4)] = cmul(v[brev_c(4 * a + bb, 4)], w);
                if (bb < 3) w = cmul(w, step1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            double2 w = step4;
#pragma unroll
            for (int a = 1; a < 4; ++a) {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) v[brev_c(4 * a + bb, 4)] = cmul(v[brev_c(4 * a + bb, 4)], w);
                if (a < 3) w = cmul(w, step4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        double ure[16], uim[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) row1[k * SK] = v[brev_c(k, 4)].x;
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) ure[jj] = row2[2 * jj];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) row1[k * SK] = v[brev_c(k, 4)].y;
        __builtin_amdgcn_sched_barrier(0);
        refill();
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) uim[jj] = row2[2 * jj];
        __builtin_amdgcn_sched_barrier(0);
        double2 s0[8], s1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            s0[q] = make_double2(ure[2 * q], uim[2 * q]);
            s1[q] = make_double2(ure[2 * q + 1], uim[2 * q + 1]);
        }
        reg_fft<3>(s0);
        __builtin_amdgcn_sched_barrier(0);
        reg_fft<3>(s1);
        __builtin_amdgcn_sched_barrier(0);
        // P[kb] = (e ? W_32^kb : 1) (G0[kb] + W_32^{2 kb} G1[kb]);  X[ka + 16 kb] = P of this thread + P of its partner
        auto pval = [&](int kb) -> double2 {
            const double2 a = s0[brev_c(kb, 3)], bq = s1[brev_c(kb, 3)];
            double2 hsum;
            if (kb == 0)
                hsum = make_double2(a.x + bq.x, a.y + bq.y);
            else if (kb == 4)
                hsum = make_double2(a.x - bq.y, a.y + bq.x);
            else
                hsum = make_double2(a.x + (bq.x * R32C[2 * kb] - bq.y * R32S[2 * kb]),
                                    a.y + (bq.x * R32S[2 * kb] + bq.y * R32C[2 * kb]));
            if (kb == 0) return hsum;
            const double wc = e ? R32C[kb] : 1.0, ws = e ? R32S[kb] : 0.0;
            return make_double2(hsum.x * wc - hsum.y * ws, hsum.x * ws + hsum.y * wc);
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 plo = pval(q), phi = pval(q + 4);
            const double2 mine = e ? phi : plo, send = e ? plo : phi;
            keep[G][q] = make_double2(mine.x + dpp_ror8(send.x), mine.y + dpp_ror8(send.y));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // one tile per workgroup.  (Persistent workgroups that request the next tile's first grid under the closed form were
    // measured: the 64 prefetch registers live across the closed form push the kernel into scratch, 620 against 430 us.)
    const int T = blockIdx.x;
    const double2 *Gt = tile_base(T);
    load(Gt);
    {
        const int lb = ntargets - 1 - T / tiles_per_target, r0 = (T % tiles_per_target) * RT;
        // (fit_mean == 0: grid 1 holds zeros — step 1 transforms the zero rows — and is transformed like the others; its
        // sums come out as exact zeros)
        process(I0{}, [&] { load(Gt + gstride); });
        __syncthreads();
        process(I1{}, [&] { load(Gt + 2 * gstride); });
        __syncthreads();
        process(I2{}, [] {});
        double best_v = 0.0;
        long long best_k = -1;
        {
            const int b = b0 + lb;
            const FastStats st = stats[b];
            const double nn = (double)(n_off[b + 1] - n_off[b]);
            const double sc = scale ? scale[b] : 1.0;
            const int k1 = r0 + f2;
            const double twopi = 6.283185307179586;
            double ph_c = 1.0, ph_s = 0.0, st_c = 1.0, st_s = 0.0;
            if (st.t0 != 0.0) {
                // e^{2 pi i t0 f} for this thread's outputs: one sincos for the first and one for the step between consecutive
                // ones (a rotation by 2 pi t0 df 16 N1), the 2f phase by the double-angle formulas
                const long long kfirst = (long long)k1 + ((long long)(ka + 64 * e) << m1);
                sincos(twopi * st.t0 * (f0 + df * (double)kfirst), &ph_s, &ph_c);
                sincos(twopi * st.t0 * (df * (double)((long long)16 << m1)), &st_s, &st_c);
            }
            // a rolled loop (the closed form is ~350 instructions per output): the kept outputs rotate through slot 0
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                const long long k = (long long)k1 + ((long long)(ka + 16 * (4 * e + q)) << m1);
                if (k < M) {
                    double2 a = keep[0][0], bq = keep[1][0], c2 = keep[2][0];
                    if (st.t0 != 0.0) {
                        const double c = ph_c, s = ph_s;
                        a = make_double2(a.x * c - a.y * s, a.x * s + a.y * c);
                        bq = make_double2(bq.x * c - bq.y * s, bq.x * s + bq.y * c);
                        const double cc = c * c - s * s, ss = 2.0 * s * c;
                        c2 = make_double2(c2.x * cc - c2.y * ss, c2.x * ss + c2.y * cc);
                    }
                    const double pw = gls_power_sums(a.y, a.x, bq.y, bq.x, c2.y, c2.x, fit_mean, norm, st.YY, 0.5 * st.wsum, nn, sc);
                    power[(size_t)b * (size_t)M + k] = pw;
                    if (pw == pw && peak_better(pw, k, best_v, best_k)) {  // ascending k within the thread
                        best_v = pw;
                        best_k = k;
                    }
                }
                const double nc = ph_c * st_c - ph_s * st_s, ns = ph_s * st_c + ph_c * st_s;
                ph_c = nc;
                ph_s = ns;
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int r = 0; r < 3; ++r) keep[g][r] = keep[g][r + 1];
            }
        }
        if (peaks != nullptr) {
            // one partial per wave and tile (shuffles only): lsf_peaks_kernel reduces a target's tiles x waves
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
                peaks[((size_t)lb * tiles_per_target + (T % tiles_per_target)) * (R::NT / 64) + (tid >> 6)] = PeakPart{best_v, best_k};
        }
    }
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
__global__ __launch_bounds__(256) void lsf_scatter_multi_kernel(const double *__restrict__ t, const double *__restrict__ y,
                                                                 const double *__restrict__ dy,
                                                                 const int64_t *__restrict__ n_off,
                                                                 const FastStats *__restrict__ stats, int b0, double f0,
                                                                 double df, int nfft, int nterms,
                                                                 double2 *__restrict__ grids) {
    const int b = b0 + blockIdx.y;
    const int64_t lo = n_off[b];
    const int n = (int)(n_off[b + 1] - lo);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const FastStats st = stats[b];
    const double tt = t[lo + i] - st.t0;
    double2 *g0 = grids + (size_t)blockIdx.y * 3 * nterms * nfft;
    double wi, wyi;
    cadence_weights(y, dy, lo + i, st.wsum, st.ybar, wi, wyi);
    const Quantum Qy = make_quantum(st.vmax), Qw = make_quantum(st.wmax);
    for (int fac = 1; fac <= 2 * nterms; ++fac) {
        const double dff = df * (double)fac, f0f = f0 * (double)fac;
        double c = 1.0, s = 0.0;
        if (f0f > 0.0) phase_factor(f0f, tt, c, s);
        const Stencil4 sp = stencil4(fmod(tt * (double)nfft * dff, (double)nfft), nfft);
        extirpolate4(g0 + (size_t)(nterms + fac - 1) * nfft, sp, wi * c, wi * s, Qw);
        if (fac <= nterms) extirpolate4(g0 + (size_t)(fac - 1) * nfft, sp, wyi * c, wyi * s, Qy);
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

// the scatter kernels' integer sums -> doubles, in place: grid g of a target (ngp grids per target, the first n_wy of them
// w (y - ybar) grids) over the rows that can hold samples (rows_used == nullptr: all N1 rows)
__global__ __launch_bounds__(256) void lsf_unquantize_kernel(double2 *__restrict__ grids, int m1, int m2,
                                                              const int *__restrict__ rows_used,
                                                              const FastStats *__restrict__ stats, int b0, int ngp, int n_wy) {
    const int lbt = blockIdx.y / ngp, g = blockIdx.y % ngp;
    if (rows_used && rows_used[lbt * 4 + 3]) return;  // ordered target: spread with LDS atomics in a fixed order, as doubles
    const int ru = rows_used ? rows_used[lbt * 4 + g] : (1 << m1);
    const FastStats st = stats[b0 + lbt];
    const double q = make_quantum(g < n_wy ? st.vmax : st.wmax).q;
    if (!(q > 0.0)) return;  // (the scatter added plain doubles)
    const int N2 = 1 << m2;
    double2 *G = grids + ((size_t)blockIdx.y << (m1 + m2));
    for (int r = blockIdx.x * 8; r < min(blockIdx.x * 8 + 8, ru); ++r)
        for (int c = threadIdx.x; c < N2; c += 256) {
            const double2 v = G[(size_t)r * N2 + c];
            G[(size_t)r * N2 + c] = make_double2((double)__double_as_longlong(v.x) * q, (double)__double_as_longlong(v.y) * q);
        }
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

// the 16 x 32 kernel applies to 512-point rows of the 16-column tiled intermediate with at most 128 kept outputs per row
static bool rows512_applies(int m1, int m2, int64_t M, int tw) {
    const long long k2need = (M + ((long long)1 << m1) - 1) >> m1;
    return m2 == 9 && tw == PRUNED_CT && m1 >= 3 && k2need <= 128;
}

static void launch_rows512(lk_handle *h, int m1, int ntargets, const double2 *grids, const FusedArgs &a, hipStream_t stream) {
    const int ntiles = ((1 << m1) / Rows512::RT) * ntargets;  // two 4-wave workgroups per CU (256 VGPRs, 35 KB of LDS each)
    hipLaunchKernelGGL(fft_rows512_power_kernel, dim3(ntiles), dim3(Rows512::NT), Rows512::LDS_BYTES, stream, grids, m1, ntargets,
                       a.n_off, a.stats, a.b0, a.f0, a.df, a.M, a.fit_mean, a.norm, a.scale, a.power, a.peaks);
}

static bool launch_rows_power(lk_handle *h, int m1, int m2, int ntargets, const double2 *grids, const FusedArgs &a, int tw,
                              hipStream_t stream) {
    const int LA = (m2 + 1) / 2, Aa = 1 << LA;
    const long long k2need = (a.M + ((long long)1 << m1) - 1) >> m1;
    if (rows512_applies(m1, m2, a.M, tw)) {
        launch_rows512(h, m1, ntargets, grids, a, stream);
        return true;
    }
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

// the cadence side of the fused extirpolation (tab16 == nullptr: every target's rows come from `grids`)
struct SpreadArgs {
    const double *t, *y, *dy;
    const int64_t *n_off;
    const FastStats *stats;
    int b0;
    double f0, df;
    int fit_mean;
    const int *tab16;
    int ntab16;
};

template <int LP>
static void launch_cols_pruned_t(lk_handle *h, int m1, int m2, int ngrids, const double2 *grids, const int *rows_used,
                                 double2 *gout, const SpreadArgs &sa, hipStream_t stream) {
    constexpr int LA = (LP + 1) / 2, LB = LP / 2, A = 1 << LA, Bq = 1 << LB, LDT = Bq + 1, FST = A * LDT + 1;
    (void)want_lds(h, reinterpret_cast<const void *>(fft_cols_pruned_kernel<LP>), 160 * 1024);
    hipLaunchKernelGGL((fft_cols_pruned_kernel<LP>), dim3((1 << m2) / PRUNED_CT, ngrids), dim3(PRUNED_CT * A),
                       (size_t)PRUNED_CT * FST * 16, stream, grids, m1, m2, rows_used, gout, sa.t, sa.y, sa.dy, sa.n_off,
                       sa.stats, sa.b0, sa.f0, sa.df, sa.fit_mean, sa.tab16, sa.ntab16);
}

static bool launch_cols_pruned(lk_handle *h, int lp, int m1, int m2, int ngrids, const double2 *grids, const int *rows_used,
                               double2 *gout, const SpreadArgs &sa, hipStream_t stream) {
    switch (lp) {
        case 5: launch_cols_pruned_t<5>(h, m1, m2, ngrids, grids, rows_used, gout, sa, stream); return true;
        case 6: launch_cols_pruned_t<6>(h, m1, m2, ngrids, grids, rows_used, gout, sa, stream); return true;
        case 7: launch_cols_pruned_t<7>(h, m1, m2, ngrids, grids, rows_used, gout, sa, stream); return true;
        case 8: launch_cols_pruned_t<8>(h, m1, m2, ngrids, grids, rows_used, gout, sa, stream); return true;
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
    // targets per chunk: the grids (3 x 16 B x Nfft per target) stay within 1.5 GiB (LSF_CHUNK_HALF_GB; 1, 1.5 and 2 GiB
    // measure the same within the rep noise once two streams share the chunks, 4 GiB no better)
#ifdef LSF_CHUNK_MB   // (experiments: chunks small enough for the 256-MiB Infinity Cache)
    const size_t chunk_bytes = (size_t)LSF_CHUNK_MB << 20;
#else
    const size_t chunk_bytes = (size_t)LSF_CHUNK_HALF_GB << 29;
#endif
    int Bc = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, chunk_bytes / ((size_t)48 * nfft)));
    if (Bc >= 8) Bc &= ~3;  // N1 / 8 row tiles per target x a multiple of 4 targets: a whole number of rounds of 2 x 256 workgroups
    const bool reg_path = m1 >= 4 && m1 <= 10 && m2 >= 4 && m2 <= 10;
    const bool fused = reg_path && rows_power_available(m1, m2, M);
    // peak partials per target: the 16 x 32 kernel leaves one per wave of N1 / 8 four-wave workgroups (the tile width it needs is
    // only known after the plan, so the larger of the two counts is reserved)
    const int nparts512 = (m2 == 9) ? (N1 / 8) * 4 : 0;
    const int nparts_gen = fused ? rows_power_parts(m1, m2) : 0;
    const int nparts_max = std::max(nparts512, nparts_gen);
    // cadence tables: 256-cell blocks over the whole grid for lsf_spread_owner_kernel, or 16-cell blocks over the (at most
    // 256) sample-bearing rows for the column kernel with the extirpolation fused in — which one is known after the plan
    const int ntab256 = (nfft + SPREAD_WW - 1) / SPREAD_WW + 3;
    const int ntab16 = (m2 >= 8 && m2 <= 10) ? (int)((((size_t)256 << m2) >> 4) + 3) : 0;
    const int ntab_max = std::max(ntab256, ntab16);
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 8 + (size_t)B * sizeof(FastStats) + 512 +
                           (size_t)Bc * 3 * nfft * 16 * (2 + (LSF_TWO_STREAMS > 1 ? LSF_TWO_STREAMS - 1 : 0)) + (size_t)Bc * 3 * M * 16 + (size_t)B * 16 +
                           (size_t)B * 4 * ntab_max * 4 + (size_t)(B + 1) * nparts_max * sizeof(PeakPart) + 16384);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    FastStats *d_stats = (FastStats *)h->ws.alloc((size_t)B * sizeof(FastStats));
    double2 *d_grids = (double2 *)h->ws.alloc((size_t)Bc * 3 * nfft * 16);
    double2 *d_spec = fused ? nullptr : (double2 *)h->ws.alloc((size_t)Bc * 3 * M * 16);
    rc = h->stage.copy(d_off, n_off_host, (size_t)(B + 1) * 8, stream);
    if (rc) return rc;
    const int tw = tile_width(m1, m2);
    double2 *d_grids2 = fused ? (double2 *)h->ws.alloc((size_t)Bc * 3 * nfft * 16) : nullptr;
    LK_REQUIRE(!fused || d_grids2 != nullptr, "workspace exhausted");
    double2 *d_grids3[2] = {nullptr, nullptr};
    for (int a = 0; a < LSF_TWO_STREAMS - 1 && a < 2; ++a) {
        d_grids3[a] = fused ? (double2 *)h->ws.alloc((size_t)Bc * 3 * nfft * 16) : nullptr;
        LK_REQUIRE(!fused || d_grids3[a] != nullptr, "workspace exhausted");
    }
    int *d_rows = (int *)h->ws.alloc((size_t)B * 4 * 4);
    int *d_plan = (int *)h->ws.alloc(64);
    int *d_tab = reg_path ? (int *)h->ws.alloc((size_t)B * 4 * ntab_max * 4) : nullptr;
    PeakPart *d_peaks = (fused && max_out) ? (PeakPart *)h->ws.alloc((size_t)B * nparts_max * sizeof(PeakPart)) : nullptr;
    hipLaunchKernelGGL(lsf_prep_kernel, dim3(B), dim3(PREP_NT), 0, stream, t, y, dy, d_off, (fit_mean || center_data) ? 1 : 0,
                       d_stats, df, nfft, m2, d_rows);
    // ---- plan: the pruned column kernel applies when every grid of every target keeps its samples in the first
    // P <= 256 rows (P < N1) and the row kernel can read 16-column tiles.  The decision needs two device words, so
    // the call synchronises `stream` once here (20-30 us against a >= 1 ms step).
    int lp = 0, n_unordered = B, spread_blocks = (nfft + SPREAD_W - 1) / SPREAD_W;
    if (fused) {
        if (!h->h_plan) LK_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h->h_plan), 64, hipHostMallocDefault));
        hipLaunchKernelGGL(lsf_plan_kernel, dim3(1), dim3(256), 0, stream, d_rows, B, d_plan);
        LK_HIP_CHECK(hipMemcpyAsync(h->h_plan, d_plan, 8, hipMemcpyDeviceToHost, stream));
        LK_HIP_CHECK(hipStreamSynchronize(stream));
        n_unordered = h->h_plan[1];
        // the spreader's workgroups only need to cover the rows that can hold samples
        spread_blocks = std::min(spread_blocks, (int)((((size_t)std::max(1, h->h_plan[0]) << m2) + SPREAD_W - 1) / SPREAD_W));
        if (m2 >= 8 && m2 <= 10 && N2 >= PRUNED_CT) {
            const int want = std::max(5, ilog2_ceil(std::max(1, h->h_plan[0])));
            if (want <= 8 && want < m1) lp = want;
        }
    }
    // cadence tables for the ordered targets (any there are): 16-cell blocks when the pruned column kernel spreads itself
    const int ntab = lp ? ntab16 : ntab256;
    if (reg_path && n_unordered < B)
        hipLaunchKernelGGL(lsf_tables_kernel, dim3((unsigned)((nmax + 256 * TAB_PER_LANE - 1) / (256 * TAB_PER_LANE)), B), dim3(256), 0, stream, t, d_off, d_stats,
                           d_rows, df, nfft, m2, lp ? 4 : 8, d_tab, ntab);
    if (!reg_path) {
        (void)want_lds(h, reinterpret_cast<const void *>(fft_cols_kernel), 100 * 1024);
        (void)want_lds(h, reinterpret_cast<const void *>(fft_rows_kernel), 100 * 1024);
    }
    const size_t ldsA = ((size_t)CT * N1 + N1 / 2 + 1) * 16, ldsB = ((size_t)RT * N2 + N2 / 2 + 1) * 16;
    // Chunks alternate between the caller's stream and a second one (each with its own intermediate: the spread-grid buffer is
    // free when every target is ordered and the pruned column kernel spreads by itself), so that the tail of a chunk's row
    // kernel and the head of the next chunk's column kernel — and, throughout, a store-heavy and a load-heavy kernel — share the
    // GPU.  LSF_TWO_STREAMS (experiment of round 6, profiles/r06_lsfast_two_streams_ab.txt).
    const int nstreams = (LSF_TWO_STREAMS && fused && lp != 0 && n_unordered == 0 && B > Bc) ? std::min(LSF_TWO_STREAMS + 1, 4) : 1;
    double2 *inter_buf[4] = {d_grids2, d_grids, d_grids3[0], d_grids3[1]};
    if (nstreams > 1) {
        if (!h->ev_ls_fork) LK_HIP_CHECK(hipEventCreateWithFlags(&h->ev_ls_fork, hipEventDisableTiming));
        LK_HIP_CHECK(hipEventRecord(h->ev_ls_fork, stream));
        for (int a = 0; a < nstreams - 1; ++a) {
            if (!h->s_ls_aux[a]) LK_HIP_CHECK(hipStreamCreateWithFlags(&h->s_ls_aux[a], hipStreamNonBlocking));
            if (!h->ev_ls_join[a]) LK_HIP_CHECK(hipEventCreateWithFlags(&h->ev_ls_join[a], hipEventDisableTiming));
            if (!LSF_STAGGER) LK_HIP_CHECK(hipStreamWaitEvent(h->s_ls_aux[a], h->ev_ls_fork, 0));
        }
    }
    hipStream_t caller_stream = stream;
    // an error return between the fork and the join must not leave work on the second stream behind (the next call resets the
    // workspace those kernels use): the guard drains it
    struct AuxGuard {
        lk_handle *h;
        int n;
        bool joined;
        ~AuxGuard() {
            if (!joined)
                for (int a = 0; a < n; ++a)
                    if (h->s_ls_aux[a]) (void)hipStreamSynchronize(h->s_ls_aux[a]);
        }
    } aux_guard{h, nstreams - 1, nstreams == 1};
    int chunk_no = 0;
    for (int b0 = 0; b0 < B; b0 += Bc, ++chunk_no) {
        const int nb = std::min(Bc, B - b0);
        double2 *gr = d_grids;
        const int lane_s = chunk_no % nstreams;
        stream = lane_s ? h->s_ls_aux[lane_s - 1] : caller_stream;
        double2 *inter = inter_buf[lane_s];
        // targets that are not "ordered" (unsorted time, or a 2f grid that wraps): zero their live rows, scatter with global
        // atomics.  Skipped when the plan found none (the usual batch).
        if (!reg_path) {
            LK_HIP_CHECK(hipMemsetAsync(gr, 0, (size_t)nb * 3 * nfft * 16, stream));
        } else if (n_unordered > 0) {
            hipLaunchKernelGGL(lsf_zero_kernel, dim3((N1 + 7) / 8, nb * 3), dim3(256), 0, stream, gr, m1, m2,
                               d_rows + (size_t)b0 * 4);
        }
        if (!reg_path || n_unordered > 0) {
            hipLaunchKernelGGL(lsf_scatter_kernel, dim3((unsigned)((nmax + 255) / 256), nb), dim3(256), 0, stream, t, y,
                               dy, d_off, d_stats, b0, f0, df, nfft, fit_mean, gr,
                               reg_path ? d_rows + (size_t)b0 * 4 : (const int *)nullptr);
            hipLaunchKernelGGL(lsf_unquantize_kernel, dim3((N1 + 7) / 8, nb * 3), dim3(256), 0, stream, gr, m1, m2,
                               reg_path ? d_rows + (size_t)b0 * 4 : (const int *)nullptr, d_stats, b0, 3, 1);
        }
        // ordered targets: spread by the pruned column kernel itself, or (other shapes) by the owner-computes spreader
        if (reg_path && !lp && n_unordered < B)
            hipLaunchKernelGGL(lsf_spread_owner_kernel, dim3((unsigned)spread_blocks, nb, 2), dim3(256), 0, stream, t, y, dy,
                               d_off, d_stats, b0, f0, df, nfft, m2, fit_mean, gr, d_rows + (size_t)b0 * 4, d_tab, ntab);
        if (fused) {
            // step 1 into the second buffer in the tiled layout, step 2 + closed form straight to `power`;
            // peak partials of every chunk side by side: one reduction launch after the last chunk
            const int nparts = rows512_applies(m1, m2, M, lp ? PRUNED_CT : tw) ? nparts512 : nparts_gen;
            const FusedArgs fa{d_off, d_stats, b0, f0, df, M, fit_mean, normalization, scale, power,
                               d_peaks ? d_peaks + (size_t)b0 * nparts : nullptr};
            if (lp) {
                const SpreadArgs sa{t, y, dy, d_off, d_stats, b0, f0, df, fit_mean, n_unordered < B ? d_tab : nullptr, ntab};
                LK_REQUIRE(launch_cols_pruned(h, lp, m1, m2, nb * 3, gr, d_rows + (size_t)b0 * 4, inter, sa, stream),
                           "no pruned column kernel for 2^%d rows", lp);
                if (LSF_STAGGER && nstreams > 1 && chunk_no == 0) {  // the other streams start one column kernel late: opposite phases
                    LK_HIP_CHECK(hipEventRecord(h->ev_ls_fork, stream));
                    for (int a = 0; a < nstreams - 1; ++a) LK_HIP_CHECK(hipStreamWaitEvent(h->s_ls_aux[a], h->ev_ls_fork, 0));
                }
            } else {
                launch_cols_reg(h, m1, m2, nb * 3, gr, d_rows + (size_t)b0 * 4, inter, tw, stream);
            }
            LK_REQUIRE(launch_rows_power(h, m1, m2, nb, inter, fa, lp ? PRUNED_CT : tw, stream),
                       "no step-2 kernel for this layout");
            if (b0 + nb == B) {
                for (int a = 0; a < nstreams - 1; ++a) {  // join: the caller's stream continues when all have drained
                    LK_HIP_CHECK(hipEventRecord(h->ev_ls_join[a], h->s_ls_aux[a]));
                    LK_HIP_CHECK(hipStreamWaitEvent(caller_stream, h->ev_ls_join[a], 0));
                }
                aux_guard.joined = true;
                stream = caller_stream;
                if (d_peaks)
                    hipLaunchKernelGGL(lsf_peaks_kernel, dim3(B), dim3(64), 0, stream, d_peaks, nparts, 0, max_out, arg_out);
            }
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
    stream = caller_stream;
    if (const int lrc = take_lds_error(h)) return lrc;  // a launch helper could not raise a kernel's dynamic-LDS limit
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
    // 5 .. LK_MAX_NTERMS terms: served by the EXACT sums (ls_chi2_launch) — the reference's `chi2` of the same nterms to 1e-9,
    // i.e. its `fastchi2` up to that method's own extirpolation error; the FFT power kernel is instantiated for <= 4 terms
    if (nterms > 4)
        return ls_chi2_launch(h, B, n_off_host, t, y, dy, nullptr, f0, df, M, nterms, fit_mean, center_data, normalization, scale,
                              power, stream);
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
    const size_t per_target = (size_t)NG * nfft * 16;
    const int Bc = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, ((size_t)2 << 30) / per_target));
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 8 + (size_t)B * sizeof(FastStats) + 512 +
                           (size_t)Bc * per_target + (size_t)Bc * NG * M * 16 + 16384);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    FastStats *d_stats = (FastStats *)h->ws.alloc((size_t)B * sizeof(FastStats));
    double2 *d_grids = (double2 *)h->ws.alloc((size_t)Bc * per_target);
    double2 *d_spec = (double2 *)h->ws.alloc((size_t)Bc * NG * M * 16);
    rc = h->stage.copy(d_off, n_off_host, (size_t)(B + 1) * 8, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(lsf_prep_kernel, dim3(B), dim3(PREP_NT), 0, stream, t, y, dy, d_off, (fit_mean || center_data) ? 1 : 0,
                       d_stats, df, nfft, m2, (int *)nullptr);
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
        hipLaunchKernelGGL(lsf_scatter_multi_kernel, dim3((unsigned)((nmax + 255) / 256), nb), dim3(256), 0, stream, t, y,
                           dy, d_off, d_stats, b0, f0, df, nfft, nterms, d_grids);
        hipLaunchKernelGGL(lsf_unquantize_kernel, dim3((N1 + 7) / 8, nb * NG), dim3(256), 0, stream, d_grids, m1, m2,
                           (const int *)nullptr, d_stats, b0, NG, nterms);
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
    if (const int lrc = take_lds_error(h)) return lrc;
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
