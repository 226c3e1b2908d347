// ls.hip — exact generalised Lomb-Scargle periodogram (floating mean) by direct trig sums, fp64, gfx950.
//
// Replaces astropy METHODS[method](t, y, dy, frequency, ...) behind LombScarglePeriodogram.from_lightcurve
// (reference: src/lightkurve/periodogram.py:961-964) and fuses lightkurve's normalisation (periodogram.py:969-975).
// Closed form follows astropy lombscargle/implementations/fast_impl.py:74-131 with the six trig sums evaluated
// exactly (utils.py:154-156), i.e. the numerics of 'slow' / 'cython' / 'chi2'.
//
// Design (VALU-fp64 bound; HBM traffic is ~1e-4 of the arithmetic):
//   * one 64-lane wavefront per workgroup owns a tile of 64*F consecutive grid frequencies of ONE target;
//     lane l owns frequencies j0 + l*F .. j0 + l*F + F-1 and keeps their 6*F fp64 sums in registers;
//   * cadences stream through in chunks of 16.  For a chunk, lane (i = l&15, g = l>>4) evaluates ONE exactly
//     range-reduced sincos (the phasor of cadence i at the first frequency of lane 16g) and walks it across 16
//     lanes' start frequencies by complex rotation, writing the seeds to a padded LDS tile [16][65];
//   * the consume loop reads the lane's seed (ds_read_b128, conflict free), fetches the cadence's constants
//     through the scalar cache (wave-uniform address -> s_load), and advances the phasor over its F frequencies
//     with the 3-term recurrence a[k+1] = 2cos(th) a[k] - a[k-1]  (2 FMA) + 6 accumulate FMA = 8 v_fma_f64 per
//     (cadence, frequency) pair = the 16 flop/pair algorithmic figure.  Phasors carry amplitude sqrt(w_i) so the
//     weighted and the uniform-weight (lightkurve default) cases cost the same.
//   * blockIdx -> (target, tile) is XCD-aware: all tiles of a target land on one XCD (block b runs on XCD b%8),
//     so the target's 64 B/cadence record stream is served by that XCD's L2.
#include <cstdlib>

#include "lk_common.hpp"
#include "ls_epilogue.hpp"

namespace lk {

typedef int int8_t_v __attribute__((ext_vector_type(8)));
typedef double double2_v __attribute__((ext_vector_type(2)));

struct CadHot {  // 32 B, read wave-uniformly in the consume loop
    double v;    // sqrt(w_i) * (y_i - ybar)
    double u;    // sqrt(w_i)
    double qc;   // cos(2 pi df t_i)
    double qs;   // sin(2 pi df t_i)
};
struct CadGen {  // 32 B, read per lane in the seed generator
    double t;    // t_i
    double u;    // sqrt(w_i)
    double gc;   // cos(2 pi F df t_i)
    double gs;   // sin(2 pi F df t_i)
};
struct CadAny {  // 24 B, arbitrary-frequency kernel
    double t, v, u;
};
struct TargetStats {
    double wsum;  // sum dy^-2 (or N)
    double ybar;  // weighted mean that was subtracted (0 if no centring)
    double YY;    // sum w (y-ybar)^2
    double yws;   // sum w (y-ybar)
};

// sin/cos of 2*pi*(f*t) with the product reduced exactly: p = f*t rounded, e its exact error (fma),
// r = (p - rint(p)) + e in [-0.5, 0.5]: phase error ~1e-17 cycles even for f*t ~ 1e4.
__device__ __forceinline__ void sincos2pi_prod(double f, double t, double *s, double *c) {
    double p = f * t;
    double e = fma(f, t, -p);
    double r = (p - rint(p)) + e;
    sincospi(2.0 * r, s, c);
}

// ------------------------------------------------------------------------------------------------ prep
// One 256-thread workgroup per target: weights, weighted mean (about y[0] so that a constant curve centres
// to exactly 0 — tests/test_periodogram.py:445-457), YY, and the per-cadence records.
template <int NT>
__device__ __forceinline__ double block_sum(double x, double *sh) {
    const int tid = threadIdx.x;
    sh[tid] = x;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    double r = sh[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void ls_prep_kernel(const double *__restrict__ t, const double *__restrict__ y,
                                                       const double *__restrict__ dy,
                                                       const int64_t *__restrict__ n_off, int center, double df,
                                                       int F, CadHot *__restrict__ hot, CadGen *__restrict__ gen,
                                                       CadAny *__restrict__ any, TargetStats *__restrict__ stats) {
    __shared__ double sh[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = n_off[b], n = n_off[b + 1] - lo;
    if (n <= 0) {
        if (tid == 0) stats[b] = TargetStats{0.0, 0.0, 0.0, 0.0};
        return;
    }
    double acc = 0.0;
    if (dy) {
        for (int64_t i = tid; i < n; i += 256) {
            double d = dy[lo + i];
            acc += 1.0 / (d * d);
        }
    }
    const double wsum = dy ? block_sum<256>(acc, sh) : (double)n;
    const double y0 = y[lo];
    double ybar = 0.0;
    if (center) {
        acc = 0.0;
        for (int64_t i = tid; i < n; i += 256) {
            double d = dy ? dy[lo + i] : 1.0;
            double w = (1.0 / (d * d)) / wsum;
            acc = fma(w, y[lo + i] - y0, acc);
        }
        ybar = block_sum<256>(acc, sh) + y0;
    }
    acc = 0.0;
    for (int64_t i = tid; i < n; i += 256) {
        double d = dy ? dy[lo + i] : 1.0;
        double w = (1.0 / (d * d)) / wsum;
        double yc = y[lo + i] - ybar;
        acc = fma(w * yc, yc, acc);
        double u = sqrt(w), ti = t[lo + i];
        if (any) any[lo + i] = CadAny{ti, u * yc, u};
        if (hot) {
            double s, c;
            sincos2pi_prod(df, ti, &s, &c);
            hot[lo + i] = CadHot{u * yc, u, c, s};
            sincos2pi_prod(df * (double)F, ti, &s, &c);  // F is a small power of two: df*F is exact
            gen[lo + i] = CadGen{ti, u, c, s};
        }
    }
    const double YY = block_sum<256>(acc, sh);
    acc = 0.0;  // sum w (y - ybar): ~1e-17, but it is the bias entry of X^T y in the multi-term solve
    for (int64_t i = tid; i < n; i += 256) {
        double d = dy ? dy[lo + i] : 1.0;
        acc = fma((1.0 / (d * d)) / wsum, y[lo + i] - ybar, acc);
    }
    const double yws = block_sum<256>(acc, sh);
    if (tid == 0) stats[b] = TargetStats{wsum, ybar, YY, yws};
}

// closed-form epilogue: gls_power_sums() in ls_epilogue.hpp.  The kernels below accumulate S2acc = sum w sin cos and
// C2acc = sum w cos^2, so S2 = 2 S2acc and C2 = 2 C2acc - 1 (sum w = 1).
__device__ __forceinline__ double gls_power(double Sh, double Ch, double S, double C, double S2acc, double C2acc,
                                            int fit_mean, int norm, double YY, double psd_factor, double nN,
                                            double scale) {
    return gls_power_sums(Sh, Ch, S, C, 2.0 * S2acc, fma(2.0, C2acc, -1.0), fit_mean, norm, YY, psd_factor, nN, scale);
}

// ------------------------------------------------------------------------------------------------ grid kernel
constexpr int LS_CHUNK = 16;  // cadences per seed tile

// one cadence folded into the 6 x F sums of a lane: k = 0 from the seed, k = 1 by rotation, k >= 2 by the
// three-term recurrence; 8 v_fma_f64 per (cadence, frequency) pair
template <int F>
__device__ __forceinline__ void ls_consume(const int8_t_v &hraw, const double2_v &p, double (&Sh)[F], double (&Ch)[F],
                                           double (&S)[F], double (&C)[F], double (&S2)[F], double (&C2)[F]) {
    CadHot h;
    __builtin_memcpy(&h, &hraw, sizeof(h));
    const double alpha = h.qc + h.qc;
    double am = p.x, bm = p.y;  // k-1 (starts as k = 0)
    Sh[0] = fma(h.v, bm, Sh[0]);
    Ch[0] = fma(h.v, am, Ch[0]);
    S[0] = fma(h.u, bm, S[0]);
    C[0] = fma(h.u, am, C[0]);
    S2[0] = fma(am, bm, S2[0]);
    C2[0] = fma(am, am, C2[0]);
    double ac = fma(am, h.qc, -(bm * h.qs));  // k = 1 by rotation
    double bc = fma(bm, h.qc, am * h.qs);
#pragma unroll
    for (int k = 1; k < F; ++k) {
        Sh[k] = fma(h.v, bc, Sh[k]);
        Ch[k] = fma(h.v, ac, Ch[k]);
        S[k] = fma(h.u, bc, S[k]);
        C[k] = fma(h.u, ac, C[k]);
        S2[k] = fma(ac, bc, S2[k]);
        C2[k] = fma(ac, ac, C2[k]);
        if (k + 1 < F) {
            const double an = fma(alpha, ac, -am);
            const double bn = fma(alpha, bc, -bm);
            am = ac;
            bm = bc;
            ac = an;
            bc = bn;
        }
    }
}

template <int F>
__global__ __launch_bounds__(64, (F <= 10 ? 3 : 2)) void ls_grid_kernel(const CadHot *__restrict__ hot,
                                                      const CadGen *__restrict__ gen,
                                                      const int64_t *__restrict__ n_off,
                                                      const TargetStats *__restrict__ stats, int B, double f0,
                                                      double df, int64_t M, int tiles, int norm, int fit_mean,
                                                      const double *__restrict__ scale,
                                                      double *__restrict__ power) {
    __shared__ double2 seeds[LS_CHUNK][65];  // +1 column: the generator's 8 lanes x 1040 B stride hit 8 bank groups

    // XCD-aware decode: block b -> XCD b%8; targets are dealt round-robin to XCDs, tiles of a target stay together
    const unsigned bid = blockIdx.x;
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const int target = (int)((slot / (unsigned)tiles) * 8u + xcd);
    const int tile = (int)(slot % (unsigned)tiles);
    if (target >= B) return;
    const int64_t lo = n_off[target];
    const int n = (int)(n_off[target + 1] - lo);
    const int lane = threadIdx.x;
    const int64_t j0 = (int64_t)tile * (64 * F);
    hot += lo;
    gen += lo;

    double Sh[F], Ch[F], S[F], C[F], S2[F], C2[F];
#pragma unroll
    for (int k = 0; k < F; ++k) Sh[k] = Ch[k] = S[k] = C[k] = S2[k] = C2[k] = 0.0;

    const int gi = lane & 15, gg = lane >> 4;
    // LDS byte address of seeds[0][lane] for the asm ds_read (address space 3 pointers are 32-bit LDS offsets)
    const unsigned seed_lane_addr =
        (unsigned)(size_t)(__attribute__((address_space(3))) char *)(&seeds[0][lane]);
    const double fstart = fma((double)(j0 + (int64_t)(16 * gg) * F), df, f0);

    for (int i0 = 0; i0 < n; i0 += LS_CHUNK) {
        // ---- seed generation: lane (gi, gg) covers lanes 16gg..16gg+15 of cadence i0+gi
        {
            double a = 0.0, b = 0.0, gc = 1.0, gs = 0.0;
            if (i0 + gi < n) {
                const CadGen g = gen[i0 + gi];
                double sn, cs;
                sincos2pi_prod(fstart, g.t, &sn, &cs);
                a = g.u * cs;
                b = g.u * sn;
                gc = g.gc;
                gs = g.gs;
            }
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                seeds[gi][16 * gg + m] = make_double2(a, b);
                const double an = fma(a, gc, -(b * gs));
                b = fma(b, gc, a * gs);
                a = an;
            }
        }
        __syncthreads();
        // ---- consume
        const int ni = min(LS_CHUNK, n - i0);
        // Software prefetch of the next cadence's record (scalar load) and seed (LDS read), hidden from hipcc's own
        // waitcnt bookkeeping with inline asm (cdna_hip_programming.md §5.7 form (ii): "=s"/"=v" loads, then ONE wait
        // statement naming every destination "+s"/"+v" before the first consumer).  hipcc otherwise sinks both loads
        // to the top of the iteration that uses them and stalls ~200 cycles per cadence on s_waitcnt lgkmcnt(0).
        // Two register sets (A, B) ping-pong over a 2x unrolled loop so that no register copy of an in-flight
        // destination is ever needed (audited in the .s: no v_mov/s_mov of hA/hB/pA/pB between a load and its wait).
        int8_t_v hA, hB;
        double2_v pA, pB;
#define LS_PREFETCH(H, P, II)                                                                           \
    {                                                                                                   \
        const CadHot *hp_ = hot + min(i0 + (II), n - 1); /* clamped: stays inside this target */        \
        const unsigned la_ = seed_lane_addr + (unsigned)(min((II), LS_CHUNK - 1) * (65 * 16));          \
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(H) : "s"(hp_));                                \
        asm volatile("ds_read_b128 %0, %1" : "=v"(P) : "v"(la_));                                       \
        __builtin_amdgcn_sched_barrier(0); /* keep the loads ABOVE the consume body they overlap with */ \
    }
        LS_PREFETCH(hA, pA, 0)
        for (int ii = 0; ii < ni; ii += 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(hA), "+v"(pA));
            LS_PREFETCH(hB, pB, ii + 1)
            ls_consume<F>(hA, pA, Sh, Ch, S, C, S2, C2);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(hB), "+v"(pB));
            LS_PREFETCH(hA, pA, ii + 2)
            if (ii + 1 < ni) ls_consume<F>(hB, pB, Sh, Ch, S, C, S2, C2);
        }
#undef LS_PREFETCH
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(hA), "+v"(pA));  // drain the last (clamped) prefetch
        __syncthreads();
    }

    // ---- epilogue
    const TargetStats st = stats[target];
    const double psd_factor = 0.5 * st.wsum;
    const double sc = scale ? scale[target] : 1.0;
    double *out = power + (size_t)target * (size_t)M;
    const int64_t jl = j0 + (int64_t)lane * F;
#pragma unroll
    for (int k = 0; k < F; ++k) {
        if (jl + k < M)
            out[jl + k] = gls_power(Sh[k], Ch[k], S[k], C[k], S2[k], C2[k], fit_mean, norm, st.YY, psd_factor,
                                    (double)n, sc);
    }
}

// ------------------------------------------------------------------------------------------------ arbitrary frequencies
// One thread per frequency, one exactly-reduced sincos per (cadence, frequency) pair (~7x the grid kernel's cost).
// Used when the requested grid is not regular in frequency (lightkurve then picks 'slow': periodogram.py:933-946).
// Cadence-sliced form (round 6): a seam call at B = 1 with a few thousand irregular frequencies — lightkurve's 'fast' -> 'slow'
// switch for every period= request, periodogram.py:933-946 — put 2000 threads on a 256-CU chip and walked all N cadences in each
// (6.7 ms at N = 20 000).  blockIdx.z = slice of the cadences; with nslice > 1 the six sums of a slice go to
// part[target][slice][sum][j] and ls_any_finish_kernel adds the slices in slice order (a fixed order: bitwise reproducible) and
// forms the power.  nslice depends on (B, M, longest target) only.
__global__ __launch_bounds__(256) void ls_any_kernel(const CadAny *__restrict__ cad,
                                                      const int64_t *__restrict__ n_off,
                                                      const TargetStats *__restrict__ stats,
                                                      const double *__restrict__ freq, int64_t M, int norm,
                                                      int fit_mean, const double *__restrict__ scale,
                                                      double *__restrict__ power, int nslice, double *__restrict__ part) {
    const int target = blockIdx.y, slice = blockIdx.z;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t lo = n_off[target];
    const int n = (int)(n_off[target + 1] - lo);
    cad += lo;
    const double f = (j < M) ? freq[j] : 0.0;
    double Sh = 0, Ch = 0, S = 0, C = 0, S2 = 0, C2 = 0;
    const int i_lo = (int)((int64_t)n * slice / nslice), i_hi = (int)((int64_t)n * (slice + 1) / nslice);
    for (int i = i_lo; i < i_hi; ++i) {
        const CadAny q = cad[i];  // uniform -> scalar load
        double sn, cs;
        sincos2pi_prod(f, q.t, &sn, &cs);
        const double a = q.u * cs, b = q.u * sn;
        Sh = fma(q.v, b, Sh);
        Ch = fma(q.v, a, Ch);
        S = fma(q.u, b, S);
        C = fma(q.u, a, C);
        S2 = fma(a, b, S2);
        C2 = fma(a, a, C2);
    }
    if (j >= M) return;
    if (nslice == 1) {
        const TargetStats st = stats[target];
        power[(size_t)target * (size_t)M + j] =
            gls_power(Sh, Ch, S, C, S2, C2, fit_mean, norm, st.YY, 0.5 * st.wsum, (double)n,
                      scale ? scale[target] : 1.0);
    } else {
        double *p = part + ((size_t)target * nslice + slice) * 6 * (size_t)M + j;
        p[0] = Sh;
        p[(size_t)M] = Ch;
        p[2 * (size_t)M] = S;
        p[3 * (size_t)M] = C;
        p[4 * (size_t)M] = S2;
        p[5 * (size_t)M] = C2;
    }
}

__global__ __launch_bounds__(256) void ls_any_finish_kernel(const double *__restrict__ part, const int64_t *__restrict__ n_off,
                                                             const TargetStats *__restrict__ stats, int64_t M, int nslice,
                                                             int norm, int fit_mean, const double *__restrict__ scale,
                                                             double *__restrict__ power) {
    const int target = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    double sm[6] = {0, 0, 0, 0, 0, 0};
    for (int sl = 0; sl < nslice; ++sl) {
        const double *p = part + ((size_t)target * nslice + sl) * 6 * (size_t)M + j;
#pragma unroll
        for (int q = 0; q < 6; ++q) sm[q] += p[(size_t)q * M];
    }
    const TargetStats st = stats[target];
    const int n = (int)(n_off[target + 1] - n_off[target]);
    power[(size_t)target * (size_t)M + j] = gls_power(sm[0], sm[1], sm[2], sm[3], sm[4], sm[5], fit_mean, norm, st.YY, 0.5 * st.wsum,
                                                      (double)n, scale ? scale[target] : 1.0);
}

// ------------------------------------------------------------------------------------------------ multi-term (chi2)
// astropy lombscargle_chi2 / lombscargle_fastchi2 (chi2_impl.py:5-86, fastchi2_impl.py:60-137): at every frequency
// the weighted least-squares fit of  [1,] sin(m w t), cos(m w t), m = 1..nterms.  X^T X and X^T y are assembled from
// the trig sums  Sw[m], Cw[m] = sum w {sin, cos}(m w t), m <= 2 nterms  and  Syw[m], Cyw[m] = sum w (y - ybar) {...},
// m <= nterms, through the product identities of fastchi2_impl.py:88-93; power = (X^T y)^T (X^T X)^-1 (X^T y).
// The sums here are the EXACT direct sums (what 'chi2' computes); harmonics come from the fundamental by the
// Chebyshev recurrence c_m = 2 c_1 c_{m-1} - c_{m-2} (2 FMA each).  Per (cadence, frequency) pair:
// 4 (phasor) + 4 (2 nterms - 1) (harmonics) + 6 nterms (accumulates) FMAs.
constexpr int LS_CHI2_F = 4;  // frequencies per lane of the multi-term grid kernel (6 nterms sums each)

template <int NT>
__global__ __launch_bounds__(64) void ls_chi2_grid_kernel(const CadHot *__restrict__ hot, const CadGen *__restrict__ gen,
                                                          const int64_t *__restrict__ n_off,
                                                          const TargetStats *__restrict__ stats, int B, double f0,
                                                          double df, int64_t M, int tiles, int norm, int fit_mean,
                                                          const double *__restrict__ scale,
                                                          double *__restrict__ power) {
    constexpr int F = LS_CHI2_F;
    __shared__ double2 seeds[LS_CHUNK][65];
    const unsigned bid = blockIdx.x;
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const int target = (int)((slot / (unsigned)tiles) * 8u + xcd);
    const int tile = (int)(slot % (unsigned)tiles);
    if (target >= B) return;
    const int64_t lo = n_off[target];
    const int n = (int)(n_off[target + 1] - lo);
    const int lane = threadIdx.x;
    const int64_t j0 = (int64_t)tile * (64 * F);
    hot += lo;
    gen += lo;
    Chi2Sums<NT> acc[F];
#pragma unroll
    for (int k = 0; k < F; ++k) acc[k].zero();
    const int gi = lane & 15, gg = lane >> 4;
    const double fstart = fma((double)(j0 + (int64_t)(16 * gg) * F), df, f0);
    for (int i0 = 0; i0 < n; i0 += LS_CHUNK) {
        {  // unit-amplitude seeds: lane (gi, gg) covers lanes 16gg..16gg+15 of cadence i0+gi
            double a = 1.0, b = 0.0, gc = 1.0, gs = 0.0;
            if (i0 + gi < n) {
                const CadGen g = gen[i0 + gi];
                sincos2pi_prod(fstart, g.t, &b, &a);
                gc = g.gc;
                gs = g.gs;
            }
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                seeds[gi][16 * gg + m] = make_double2(a, b);
                const double an = fma(a, gc, -(b * gs));
                b = fma(b, gc, a * gs);
                a = an;
            }
        }
        __syncthreads();
        const int ni = min(LS_CHUNK, n - i0);
        for (int ii = 0; ii < ni; ++ii) {
            const CadHot h = hot[i0 + ii];  // wave-uniform: scalar load
            const double w = h.u * h.u, wy = h.u * h.v;
            const double2 p = seeds[ii][lane];
            const double alpha = h.qc + h.qc;
            double am = p.x, bm = p.y;
            acc[0].add(am, bm, w, wy);
            double ac = fma(am, h.qc, -(bm * h.qs)), bc = fma(bm, h.qc, am * h.qs);
#pragma unroll
            for (int k = 1; k < F; ++k) {
                acc[k].add(ac, bc, w, wy);
                if (k + 1 < F) {
                    const double an = fma(alpha, ac, -am), bn = fma(alpha, bc, -bm);
                    am = ac;
                    bm = bc;
                    ac = an;
                    bc = bn;
                }
            }
        }
        __syncthreads();
    }
    const TargetStats st = stats[target];
    const double sc = scale ? scale[target] : 1.0;
    double *out = power + (size_t)target * (size_t)M;
    const int64_t jl = j0 + (int64_t)lane * F;
#pragma unroll
    for (int k = 0; k < F; ++k)
        if (jl + k < M)
            out[jl + k] = chi2_normalise(acc[k].solve(st.yws, fit_mean), norm, st.YY, 0.5 * st.wsum, (double)n, sc);
}

// arbitrary frequencies: one thread per frequency, one exactly-reduced sincos per (cadence, frequency) pair
template <int NT>
__global__ __launch_bounds__(256) void ls_chi2_any_kernel(const CadAny *__restrict__ cad,
                                                          const int64_t *__restrict__ n_off,
                                                          const TargetStats *__restrict__ stats,
                                                          const double *__restrict__ freq, int64_t M, int norm,
                                                          int fit_mean, const double *__restrict__ scale,
                                                          double *__restrict__ power) {
    const int target = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t lo = n_off[target];
    const int n = (int)(n_off[target + 1] - lo);
    cad += lo;
    const double f = (j < M) ? freq[j] : 0.0;
    Chi2Sums<NT> acc;
    acc.zero();
    for (int i = 0; i < n; ++i) {
        const CadAny q = cad[i];
        double sn, cs;
        sincos2pi_prod(f, q.t, &sn, &cs);
        acc.add(cs, sn, q.u * q.u, q.u * q.v);
    }
    if (j < M) {
        const TargetStats st = stats[target];
        power[(size_t)target * (size_t)M + j] = chi2_normalise(acc.solve(st.yws, fit_mean), norm, st.YY, 0.5 * st.wsum,
                                                               (double)n, scale ? scale[target] : 1.0);
    }
}

// the regular grid f0 + df j as an explicit array (nterms > LK_FAST_NTERMS take the one-thread-per-frequency kernel)
__global__ __launch_bounds__(256) void ls_freq_grid_kernel(double f0, double df, int64_t M, double *__restrict__ freq) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < M) freq[j] = f0 + df * (double)j;
}

// ------------------------------------------------------------------------------------------------ launcher
constexpr int LS_F = 16;
constexpr int LK_FAST_NTERMS = 4;  // nterms with a regular-grid kernel (6 nterms accumulators x LS_CHI2_F frequencies per lane)

int ls_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *dy,
              const double *freq, double f0, double df, int64_t M, int fit_mean, int center_data, int normalization,
              const double *scale, double *power, hipStream_t stream) {
    return ls_chi2_launch(h, B, n_off_host, t, y, dy, freq, f0, df, M, 1, fit_mean, center_data, normalization, scale,
                          power, stream);
}

// nterms = 1: the closed-form kernels above; nterms = 2..LK_MAX_NTERMS: the multi-term least-squares kernels
int ls_chi2_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *dy,
                   const double *freq, double f0, double df, int64_t M, int nterms, int fit_mean, int center_data,
                   int normalization, const double *scale, double *power, hipStream_t stream) {
    LK_REQUIRE(nterms >= 1 && nterms <= LK_MAX_NTERMS, "nterms must be between 1 and %d (got %d)", LK_MAX_NTERMS, nterms);
    LK_REQUIRE(B >= 0 && n_off_host != nullptr, "bad batch description");
    LK_REQUIRE(freq == nullptr || B <= 65535, "at most 65535 targets per call with an explicit frequency array (got %d)", B);
    LK_REQUIRE(M >= 0, "M must be >= 0");
    if (B == 0 || M == 0) return LK_OK;
    LK_REQUIRE(t && y && power, "t, y, power must be non-NULL");
    LK_REQUIRE(normalization >= LK_NORM_STANDARD && normalization <= LK_NORM_LK_PSD, "unknown normalization %d",
               normalization);
    for (int b = 0; b < B; ++b)
        LK_REQUIRE(n_off_host[b + 1] > n_off_host[b], "target %d is empty (every light curve needs >= 1 cadence)", b);
    LK_REQUIRE(n_off_host[0] == 0, "n_off[0] must be 0");
    if (!freq) {
        LK_REQUIRE(f0 >= 0.0, "Frequencies must be positive");
        LK_REQUIRE(df > 0.0, "Frequency steps must be positive");
    }
    const size_t ntot = (size_t)n_off_host[B];
    for (int b = 0; b < B; ++b)
        LK_REQUIRE(n_off_host[b + 1] - n_off_host[b] < (int64_t)1 << 30, "target %d too long", b);

    h->ws.reset();
    // irregular single-term grids with few (target, frequency) pairs: slices of the cadences fill the chip (ls_any_kernel)
    int any_slices = 1;
    if (freq && nterms == 1) {
        int64_t nmax = 1;
        for (int b = 0; b < B; ++b) nmax = std::max<int64_t>(nmax, n_off_host[b + 1] - n_off_host[b]);
        const int64_t threads = (int64_t)B * ((M + 255) / 256) * 256;
        any_slices = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(256, nmax / 64), ((int64_t)1 << 17) / threads));
    }
    const size_t need = (size_t)(B + 1) * 8 + (size_t)B * sizeof(TargetStats) + ntot * (sizeof(CadHot) + sizeof(CadGen)) +
                        (size_t)M * 8 + (any_slices > 1 ? (size_t)B * any_slices * 6 * (size_t)M * 8 + 256 : 0) + 4096;
    int rc = h->ws.reserve(need);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    TargetStats *d_stats = (TargetStats *)h->ws.alloc((size_t)B * sizeof(TargetStats));
    if (!freq && nterms > LK_FAST_NTERMS) {
        // 5 .. LK_MAX_NTERMS terms: 6 nterms sums per frequency no longer fit four frequencies per lane — the grid becomes an
        // explicit array and the one-thread-per-frequency kernel runs (exact sums, ~7 x the cost per pair: rare requests)
        LK_REQUIRE(B <= 65535, "at most 65535 targets per call with nterms > %d (got %d)", LK_FAST_NTERMS, B);
        double *d_freq = (double *)h->ws.alloc((size_t)M * 8);
        LK_REQUIRE(d_freq != nullptr, "workspace exhausted (frequency grid)");
        hipLaunchKernelGGL(ls_freq_grid_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, stream, f0, df, M, d_freq);
        freq = d_freq;
    }
    {
        const int rcs = h->stage.copy(d_off, n_off_host, (size_t)(B + 1) * 8, stream);
        if (rcs) return rcs;
    }
    const int center = (fit_mean || center_data) ? 1 : 0;

    if (freq) {
        CadAny *d_any = (CadAny *)h->ws.alloc(ntot * sizeof(CadAny));
        hipLaunchKernelGGL(ls_prep_kernel, dim3(B), dim3(256), 0, stream, t, y, dy, d_off, center, 0.0, 1,
                           (CadHot *)nullptr, (CadGen *)nullptr, d_any, d_stats);
        dim3 grid((unsigned)((M + 255) / 256), (unsigned)B);
#define LK_CHI2_ANY(NT)                                                                                              \
    hipLaunchKernelGGL(ls_chi2_any_kernel<NT>, grid, dim3(256), 0, stream, d_any, d_off, d_stats, freq, M,           \
                       normalization, fit_mean, scale, power)
        switch (nterms) {
            case 1: {
                double *d_part = any_slices > 1 ? (double *)h->ws.alloc((size_t)B * any_slices * 6 * (size_t)M * 8) : nullptr;
                LK_REQUIRE(any_slices == 1 || d_part != nullptr, "workspace exhausted (partial sums)");
                LK_REQUIRE(B <= 65535, "at most 65535 targets per call on an irregular frequency grid (got %d)", B);
                hipLaunchKernelGGL(ls_any_kernel, dim3(grid.x, grid.y, (unsigned)any_slices), dim3(256), 0, stream, d_any, d_off,
                                   d_stats, freq, M, normalization, fit_mean, scale, power, any_slices, d_part);
                if (any_slices > 1)
                    hipLaunchKernelGGL(ls_any_finish_kernel, grid, dim3(256), 0, stream, d_part, d_off, d_stats, M, any_slices,
                                       normalization, fit_mean, scale, power);
                break;
            }
            case 2: LK_CHI2_ANY(2); break;
            case 3: LK_CHI2_ANY(3); break;
            case 4: LK_CHI2_ANY(4); break;
            case 5: LK_CHI2_ANY(5); break;
            case 6: LK_CHI2_ANY(6); break;
            case 7: LK_CHI2_ANY(7); break;
            default: LK_CHI2_ANY(8); break;
        }
#undef LK_CHI2_ANY
    } else if (nterms > 1) {
        CadHot *d_hot = (CadHot *)h->ws.alloc(ntot * sizeof(CadHot));
        CadGen *d_gen = (CadGen *)h->ws.alloc(ntot * sizeof(CadGen));
        const int F = LS_CHI2_F;
        hipLaunchKernelGGL(ls_prep_kernel, dim3(B), dim3(256), 0, stream, t, y, dy, d_off, center, df, F, d_hot,
                           d_gen, (CadAny *)nullptr, d_stats);
        const int tiles = (int)((M + 64 * F - 1) / (64 * F));
        const size_t nblocks = (size_t)((B + 7) / 8) * 8 * (size_t)tiles;
        LK_REQUIRE(nblocks < ((size_t)1 << 31), "grid too large (B=%d, M=%lld)", B, (long long)M);
#define LK_CHI2_GRID(NT)                                                                                             \
    hipLaunchKernelGGL(ls_chi2_grid_kernel<NT>, dim3((unsigned)nblocks), dim3(64), 0, stream, d_hot, d_gen, d_off,   \
                       d_stats, B, f0, df, M, tiles, normalization, fit_mean, scale, power)
        switch (nterms) {
            case 2: LK_CHI2_GRID(2); break;
            case 3: LK_CHI2_GRID(3); break;
            default: LK_CHI2_GRID(4); break;
        }
#undef LK_CHI2_GRID
    } else {
        CadHot *d_hot = (CadHot *)h->ws.alloc(ntot * sizeof(CadHot));
        CadGen *d_gen = (CadGen *)h->ws.alloc(ntot * sizeof(CadGen));
        const int F = LS_F;  // 16 frequencies per lane (8, 10, 12 measured slower; 32 would need 384 accumulator VGPRs)
        hipLaunchKernelGGL(ls_prep_kernel, dim3(B), dim3(256), 0, stream, t, y, dy, d_off, center, df, F, d_hot,
                           d_gen, (CadAny *)nullptr, d_stats);
        const int tiles = (int)((M + 64 * F - 1) / (64 * F));
        const size_t nblocks = (size_t)((B + 7) / 8) * 8 * (size_t)tiles;
        LK_REQUIRE(nblocks < ((size_t)1 << 31), "grid too large (B=%d, M=%lld)", B, (long long)M);
#define LK_LS_LAUNCH(FF)                                                                                       \
    hipLaunchKernelGGL(ls_grid_kernel<FF>, dim3((unsigned)nblocks), dim3(64), 0, stream, d_hot, d_gen, d_off,  \
                       d_stats, B, f0, df, M, tiles, normalization, fit_mean, scale, power)
        if (F == 8)
            LK_LS_LAUNCH(8);
        else if (F == 10)
            LK_LS_LAUNCH(10);
        else
            LK_LS_LAUNCH(16);
#undef LK_LS_LAUNCH
    }
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ nanargmax
__global__ __launch_bounds__(256) void argmax_kernel(const double *__restrict__ x, int64_t M, double *max_out,
                                                      int64_t *arg_out) {
    __shared__ double sv[256];
    __shared__ int64_t si[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double *row = x + (size_t)b * (size_t)M;
    double best = -INFINITY;
    int64_t bi = -1;
    for (int64_t j = tid; j < M; j += 256) {
        const double v = row[j];
        if (v == v && (bi < 0 || v > best)) {  // NaN skipped; strict > keeps the first maximum
            best = v;
            bi = j;
        }
    }
    sv[tid] = best;
    si[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const double v2 = sv[tid + s];
            const int64_t i2 = si[tid + s];
            if (i2 >= 0 && (si[tid] < 0 || v2 > sv[tid] || (v2 == sv[tid] && i2 < si[tid]))) {
                sv[tid] = v2;
                si[tid] = i2;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        max_out[b] = si[0] >= 0 ? sv[0] : NAN;
        arg_out[b] = si[0];
    }
}

int argmax_launch(lk_handle *h, int B, int64_t M, const double *x, double *max_out, int64_t *argmax_out,
                  hipStream_t stream) {
    (void)h;
    LK_REQUIRE(B >= 0 && M >= 1, "need B >= 0 and M >= 1");
    if (B == 0) return LK_OK;
    LK_REQUIRE(x && max_out && argmax_out, "NULL buffer");
    hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(256), 0, stream, x, M, max_out, argmax_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
