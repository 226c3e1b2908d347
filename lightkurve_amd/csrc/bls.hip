// bls.hip — Box Least Squares periodogram with astropy `run_bls` semantics, fp64, gfx950.
//
// Replaces astropy methods.bls_fast -> C run_bls behind BoxLeastSquaresPeriodogram.from_lightcurve
// (reference: src/lightkurve/periodogram.py:1161-1169; algorithm: SURVEY.md App. B.2).  Compiled with
// -ffp-contract=off: every product, sum and quotient rounds separately, exactly as the reference C does,
// and every per-bin sum is accumulated in cadence order, so all seven outputs are
// BIT-IDENTICAL to the reference (tests/test_bls_gpu.py compares with ==).
//
// One team of waves per (target, period); everything lives in LDS (bls_team_body has the details):
//   histogram  every cadence is added into its phase bin with an LDS floating-point atomic.  Same-address lanes of one
//              ds_add_f64 are applied in lane order and a wave's LDS instructions execute in program order (measured:
//              tools/microbench/lds_atomic_order.hip; re-checked on the device by bls_selftest_kernel), so a bin
//              accumulates in cadence order — the reference's order — with no sorting, rounds or leader election.  Up to
//              four waves share the arithmetic; a ticket in LDS keeps their atomics in cadence-group order.
//   wrap       pad, then sequential inclusive prefix sums: a loads-only serial pass keeps every 32nd running sum, then
//              all threads re-run the 32-bin blocks from those exact carries (same operands, same order: same bits).
//   scan       a lane owns a start bin and walks the durations in ascending length.  Block maxima of
//              Z = S*W - E*Y (Nn(n, dur) = Z[n + dur] - Z[n]) bound whole blocks of 64 / 8 end bins at once, a rigorous
//              growth bound skips further, candidates that are reached pass a division-free conservative filter and only
//              survivors run the reference's exact arithmetic (IEEE divisions), which alone decides the result.  Ties are
//              broken by (caller's duration index, start bin) = the reference's "first wins" order.
// Short periods (two thirds of a grid uniform in frequency) pack up to eight one-wave teams into a workgroup that shares
// the duration tables and the serial prefix pass.  blockIdx -> (target, period) is XCD-aware (all periods of a target on
// one XCD; its t / y*ivar / ivar arrays, 24 B per cadence, stay in that XCD's L2).
#include <cfloat>
#include <cstdlib>
#include <cmath>

#include <algorithm>
#include <numeric>

#include "lk_common.hpp"

namespace lk {

struct BlsStats {
    double min_t, sum_y, sum_ivar, sorted;  // sorted: 1.0 if the target's times never decrease
};

// ------------------------------------------------------------------------------------------------ prep
// per target: min_t (exact), sum_y / sum_ivar accumulated SEQUENTIALLY (reference order), tm = t - min_t,
// yw = (y * ivar, ivar) interleaved (one 16-B load per cadence in the histogram pass).
__global__ __launch_bounds__(256) void bls_prep_kernel(const double *__restrict__ t, const double *__restrict__ y,
                                                        const double *__restrict__ ivar,
                                                        const int64_t *__restrict__ n_off,
                                                        double *__restrict__ tm, double2 *__restrict__ yw,
                                                        BlsStats *__restrict__ stats) {
    __shared__ double sh[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = n_off[b], n = n_off[b + 1] - lo;
    double m = INFINITY;
    for (int64_t i = tid; i < n; i += 256) m = fmin(m, t[lo + i]);
    sh[tid] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = fmin(sh[tid], sh[tid + s]);
        __syncthreads();
    }
    const double min_t = sh[0];
    __syncthreads();
    // tm, yw and the two sums in ONE sweep of 1024-cadence chunks: every thread forms its cadences' y * ivar and ivar, parks them in
    // LDS, and one lane per sum adds the chunk SEQUENTIALLY from there (the reference's order: bit-exactness of the outputs rests on
    // it) — eight LDS reads in flight in front of eight dependent additions.  (Round 6: the sums used to walk global memory one
    // dependent load at a time — 640 us for 20 000 cadences, a fifth of a B = 1 call.)
    __shared__ double sa[1024], sb[1024];
    int unsorted = 0;
    double sum_a = 0.0, sum_b = 0.0;  // (lane 0 of wave 0 / of wave 1)
    for (int64_t c0 = 0; c0 < n; c0 += 1024) {
        const int cn = (int)min((int64_t)1024, n - c0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = tid + 256 * u;
            if (k < cn) {
                const int64_t i = c0 + k;
                const double ti = t[lo + i], yi = y[lo + i], wi = ivar[lo + i];
                tm[lo + i] = ti - min_t;
                const double yw_ = yi * wi;
                yw[lo + i] = make_double2(yw_, wi);
                sa[k] = yw_;
                sb[k] = wi;
                if (i + 1 < n && !(ti <= t[lo + i + 1])) unsorted = 1;
            }
        }
        __syncthreads();
        if (tid == 0 || tid == 64) {
            const double *src = tid == 0 ? sa : sb;
            double s = tid == 0 ? sum_a : sum_b;
            int k = 0;
            for (; k + 8 <= cn; k += 8) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = src[k + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) s += v[q];
            }
            for (; k < cn; ++k) s += src[k];
            if (tid == 0)
                sum_a = s;
            else
                sum_b = s;
        }
        __syncthreads();
    }
    unsorted = __syncthreads_or(unsorted);
    if (tid == 64) sh[1] = sum_b;
    __syncthreads();
    if (tid == 0) stats[b] = BlsStats{min_t, sum_a, sh[1], unsorted ? 0.0 : 1.0};
}

// exact k = trunc(t/P), r = fmod(t, P) for t >= 0, P > 0 (fmod results are always representable, so the
// fused remainder is exact once k is right; the reciprocal estimate is off by at most one).
__device__ __forceinline__ void fold_exact(double t, double P, double invP, double *k_out, double *r_out) {
    double k = floor(t * invP);
    double r = fma(-k, P, t);
    if (r < 0.0) {
        k -= 1.0;
        r = fma(-k, P, t);
    }
    if (r >= P) {
        k += 1.0;
        r = fma(-k, P, t);
    }
    *k_out = k;
    *r_out = r;
}

__device__ __forceinline__ int bin_of(double r, double bin_duration) { return (int)(fabs(r) / bin_duration) + 1; }

// Same value as bin_of without the IEEE division on the common path: q = r * fl(1/bd) is within 4.5e-16*q of the
// correctly rounded quotient, so trunc(q) can only differ when q sits within that distance of an integer; those
// (vanishingly rare) cadences take the exact division.
__device__ __forceinline__ int bin_of_fast(double r, double bin_duration, double inv_bd) {
    const double q = r * inv_bd;
    const double f = q - floor(q);
    const double guard = 1e-12 * (q + 1.0);
    if (f > guard && (1.0 - f) > guard) return (int)q + 1;
    return bin_of(r, bin_duration);
}

struct BlsBest {
    double obj;
    int k, n;
};

// ------------------------------------------------------------------------------------------------ team kernel
// One workgroup ("team", NW = 1 .. 16 waves by LDS footprint) per (target, period); everything lives in LDS.
//
//   histogram  ONE wave walks the target's cadences in order, four 64-cadence chunks at a time, and every lane adds its
//              cadence into the bin with an LDS floating-point atomic (ds_add_f64; y*ivar and ivar in two arrays: the
//              8-byte stride is 1.6 x faster than an interleaved pair).  Same-address lanes of one ds_add_f64 are applied
//              in increasing lane order and a wave's LDS instructions execute in program order (measured:
//              tools/microbench/lds_atomic_order.hip; re-checked per handle by bls_selftest_kernel), so every bin
//              accumulates in cadence order = the reference's order, for sorted and unsorted time alike — no rounds, no
//              segment searches, no leader election.  Time-sorted targets: the cycle number of a cadence is the running
//              k0 or k0 + 1, so the phase is one of two exact fused remainders; a quad that contains a longer gap, or a
//              quotient too close to an integer for the reciprocal multiply, is redone with the general arithmetic.
//              The other waves of the team sleep at the barrier meanwhile (other teams of the CU fill the SIMDs).
//   prefix     sequential inclusive sums (lane 0: y, lane 1: ivar), eight bins per step with the next eight loading
//   scan       all waves; start bins are handed out by an LDS counter (a lane whose walk ends early takes the next start
//              bin); skip-ahead bound, conservative filter and the reference's exact arithmetic as described on top
__device__ __forceinline__ double readlane_f64(double x, int l) {
    const int lo_ = __builtin_amdgcn_readlane(__double2loint(x), l), hi_ = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi_, lo_);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The histogram wave handles U chunks of 64 cadences at a time (their arithmetic interleaves: the wave's dependent-issue
// latency, not its instruction count, is what a lone wave pays) with the global loads of PFG further groups in flight.
// <2, 2>: the 64-VGPR build for the groups that fit many teams per CU; <4, 2>: 128 VGPRs, long periods that own a CU.
template <int U, int PFG>
__device__ __forceinline__ void bls_team_body(
    const double *__restrict__ tm, const double2 *__restrict__ yw, const int64_t *__restrict__ n_off,
    const BlsStats *__restrict__ stats, const double *__restrict__ period, const int *__restrict__ pidx, int np_group,
    int64_t nP, int B, const int *__restrict__ dur_tab, int n_dur, int max_dur, double bin_duration, int oversample,
    int obj_flag, double *__restrict__ out7, int cap, int shape, int ablate_arg, unsigned long long *__restrict__ prof_arg) {
#ifdef LK_BLS_DEBUG
    const int ablate = ablate_arg;
    unsigned long long *const prof = prof_arg;
#else   // release build: the switches are constants and every branch on them folds away
    constexpr int ablate = 0;
    constexpr unsigned long long *prof = nullptr;
    (void)ablate_arg;
    (void)prof_arg;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t_last = prof ? wall_clock64() : 0ull;
#define BLS_LAP(slot_)                                                        \
    do {                                                                      \
        if (prof && threadIdx.x == 0) {                                       \
            const unsigned long long now_ = wall_clock64();                   \
            atomicAdd(&prof[slot_], now_ - t_last);                           \
            t_last = now_;                                                    \
        }                                                                     \
    } while (0)
    // Two shapes.  multi == 0: the workgroup is ONE team of NW waves on one period (long periods).  multi != 0: the
    // workgroup holds G = blockDim / 64 one-wave teams on G consecutive periods of the sorted grid (short periods); they
    // share the duration tables, the barriers, and — the point — the prefix pass: lane 2g + c of wave 0 runs the chain
    // of team g's component c, so the serial chain instructions are spent on 2 G lanes instead of 2.
    // shape: bit 0 multi-period workgroup, bit 1 ordered (atomic-free) histogram, bits 8.. histogram waves
    const int nh_cap = shape >> 8, multi = shape & 1;
    const bool serial_hist = (shape & 2) != 0;
    const int wtid = threadIdx.x, lane = wtid & 63;
    const int wwave = __builtin_amdgcn_readfirstlane(wtid >> 6);
    const int G = multi ? ((int)blockDim.x >> 6) : 1;
    const int NT = multi ? 64 : (int)blockDim.x, NW = NT >> 6;  // threads / waves of this team
    const int tid = multi ? lane : wtid;
    const int wave = multi ? 0 : wwave;                          // wave index inside the team
    const int g = multi ? wwave : 0;                             // team index inside the workgroup
    const unsigned bid = blockIdx.x;
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const unsigned nq = ((unsigned)np_group + (unsigned)G - 1u) / (unsigned)G;
    const int target = (int)((slot / nq) * 8u + xcd);
    if (target >= B) return;
    const int pslot_raw = (int)(slot % nq) * G + g;
    const bool emit = pslot_raw < np_group;            // the tail workgroup of a target: surplus teams redo the last
    const int p = pidx[min(pslot_raw, np_group - 1)];  // period in their own LDS region and write nothing
    const double P = period[p];
    const double invP = 1.0 / P;
    const int n_bins = (int)(ceil(P / bin_duration)) + oversample;
    const int64_t lo = n_off[target];
    const int N = (int)(n_off[target + 1] - lo);
    tm += lo;
    yw += lo;

    // LDS: duration tables | per team: ya[cap] | wa[cap] | s_thr | s_red[3] | s_ctr, s_nb | s_best[NW] | zm8 | zm64
    // tables (sorted ascending by length): dur_bins[k] (k = n_dur: a sentinel no window fits), the caller's index
    // korig[k] (tie-break order) and first_kd[L] = k | dur_bins[k] << 16 for the first k with dur_bins[k] >= L
    int *dur_bins = reinterpret_cast<int *>(smem);
    int *korig = dur_bins + n_dur + 1;
    int *first_kd = korig + n_dur;
    const int tab_ints = 2 * n_dur + 1 + max_dur + 2;
    const size_t region = ((size_t)cap * 16 + 48 + (size_t)NW * 16 + (size_t)((cap >> 3) + (cap >> 6) + 4) * 8 + 15) & ~(size_t)15;
    char *region0 = smem + (((size_t)tab_ints * 4 + 15) & ~(size_t)15);
    double *ya = reinterpret_cast<double *>(region0 + (size_t)g * region);
    double *wa = ya + cap;
    long long *s_thr = reinterpret_cast<long long *>(wa + cap);
    long long *s_red = s_thr + 1;                       // gmax, wmax (bit patterns), yabs (double)
    int *s_ctr = reinterpret_cast<int *>(s_red + 3);    // next start bin of the scan; s_ctr[1] = n_bins (for the chain lanes)
    BlsBest *s_best = reinterpret_cast<BlsBest *>(s_red + 5);
    double *zm8 = reinterpret_cast<double *>(s_best + NW);  // [(n_bins >> 3) + 1] max of Z over aligned blocks of 8 bins
    double *zm64 = zm8 + (cap >> 3) + 1;                    // [(n_bins >> 6) + 1] ... of 64 bins (Z: see the scan)
    for (int i = wtid; i < tab_ints; i += (int)blockDim.x) dur_bins[i] = dur_tab[i];
    for (int i = tid; i <= n_bins; i += NT) {
        ya[i] = 0.0;
        wa[i] = 0.0;
    }
    if (tid == 0) {
        *s_thr = __double_as_longlong(-INFINITY);
        s_red[0] = s_red[1] = s_red[2] = 0;
        s_ctr[0] = 0;
        s_ctr[1] = n_bins;
        s_ctr[2] = 0;  // histogram ticket
    }
    __syncthreads();
    BLS_LAP(0);  // setup

    // ---- histogram (wave 0)
    const BlsStats st = stats[target];
    // NH histogram waves (multi-wave teams): wave h takes the cadence groups g = h, h + NH, ... — the arithmetic of NH groups
    // runs on NH SIMDs at once — and the atomics are issued in group order through a ticket in LDS: wave h waits until the
    // ticket says g, issues its atomics and then writes g + 1.  A wave's LDS instructions execute in program order, so
    // the next wave can only see the new ticket after the atomics before it have been applied: the bins still
    // accumulate in cadence order.
    const int NH = (multi || (ablate & 256) != 0) ? 1 : min(NW, nh_cap);
    volatile int *s_ticket = s_ctr + 2;
    if (wave < NH && !(ablate & 1)) {
        const bool tsorted = st.sorted != 0.0;
        const double inv_bd = 1.0 / bin_duration;
        const double guard = 1e-12 * ((double)n_bins + 2.0);  // >= 1e-12 (q + 1) for every quotient of this period
        const double guard_hi = 1.0 - guard;
        constexpr int GC = U * 64;  // cadences per group of U chunks
        double kd0 = 0.0;  // wave-uniform: a cycle number not above that of any cadence still to come (sorted targets)
        // one group: phases, bins, two atomics per cadence.  `full` (a literal at both call sites): every lane holds a
        // cadence, so the fast path carries no masks at all.
        auto do_group = [&](const double(&tv)[U], const double2(&v)[U], int g, bool full) {
            const int i_base = g * GC;
            bool act[U];
            int ind[U];
#pragma unroll
            for (int u = 0; u < U; ++u) act[u] = full || (i_base + (u << 6) + lane < N);
            if (NH > 1 && tsorted) {  // groups are not consecutive in this wave: the group's first cadence gives the cycle
                double k, r;
                fold_exact(tv[0], P, invP, &k, &r);
                kd0 = readlane_f64(k, 0);
            }
            bool bad = false, nxt_last = false;
            const double kd1 = kd0 + 1.0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double r1 = fma(-kd0, P, tv[u]), r2 = fma(-kd1, P, tv[u]);
                const bool nxt = r1 >= P;
                const double r = nxt ? r2 : r1;
                const double q = r * inv_bd;
                const double f = q - floor(q);
                ind[u] = (int)q + 1;
                bad = bad || (act[u] && !(f > guard && f < guard_hi));
                if (full ? (u == U - 1) : true) bad = bad || (act[u] && nxt && r2 >= P);  // sorted: the last cadence decides
                if (u == U - 1) nxt_last = nxt;
            }
            if (!tsorted || __ballot(bad)) {
                // general arithmetic: exact (k, r) by fused remainder with +-1 correction, exact division wherever the
                // reciprocal multiply is within the guard of an integer
                double k_last = 0.0;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    double k, r;
                    fold_exact(tv[u], P, invP, &k, &r);
                    ind[u] = bin_of_fast(r, bin_duration, inv_bd);
                    if (u == U - 1) k_last = k;
                }
                kd0 = readlane_f64(k_last, 63);  // only used after a full group
            } else if (__ballot(nxt_last) >> 63) {
                kd0 = kd1;
            }
            if (NH > 1) {
                while (*s_ticket != g) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
            }
            if (serial_hist) {
                // The fall-back for a device whose ds_add_f64 does not apply same-address lanes in lane order
                // (bls_selftest_kernel): one lane at a time performs a plain read-add-write of its two bins.  A wave's LDS
                // instructions execute in program order, so lane l + 1 reads what lane l wrote: the bins accumulate in
                // cadence order by construction, with no atomic at all.  64 x the LDS instructions of the atomic form.
#pragma unroll
                for (int u = 0; u < U; ++u)
                    for (int l = 0; l < 64; ++l) {
                        if (lane == l && act[u]) {
                            volatile double *yp = ya + ind[u], *wp = wa + ind[u];
                            *yp = *yp + v[u].x;
                            *wp = *wp + v[u].y;
                        }
                        wave_sync();
                    }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (act[u]) {
                        atomicAdd(&ya[ind[u]], v[u].x);
                        atomicAdd(&wa[ind[u]], v[u].y);
                    }
            }
            if (NH > 1) {
                asm volatile("" ::: "memory");
                // lgkmcnt(0): the wave's LDS atomics above have been executed before the ticket store is issued (LDS
                // instructions of one wave complete in order; the wait makes that independent of the queue's behaviour)
                __builtin_amdgcn_s_waitcnt(0xC07F);
                if (lane == 0) *s_ticket = g + 1;
            }
        };
        const int nfull = N / GC;  // full groups: wave-uniform base pointer + lane offset, unconditional loads (a branch
                                   // around a load would make the compiler wait for every load in flight)
        const int nown = nfull > wave ? (nfull - wave + NH - 1) / NH : 0;  // this wave's full groups: wave + j * NH
        if (nown > 0) {
            double tvb[PFG][U];
            double2 vb[PFG][U];
#pragma unroll
            for (int s = 0; s < PFG; ++s) {
                const int gs = wave + min(s, nown - 1) * NH;
                const double *tp = tm + (size_t)gs * GC;
                const double2 *yp = yw + (size_t)gs * GC;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    tvb[s][u] = tp[(u << 6) + lane];
                    vb[s][u] = yp[(u << 6) + lane];
                }
            }
            for (int j0 = 0; j0 < nown; j0 += PFG) {
#pragma unroll
                for (int s = 0; s < PFG; ++s) {
                    const int jd = j0 + s;
                    if (jd < nown) {  // wave-uniform
                        double tv[U];
                        double2 v[U];
                        const int gn = wave + min(jd + PFG, nown - 1) * NH;  // the last group is re-loaded at the end: harmless
                        const double *tp = tm + (size_t)gn * GC;
                        const double2 *yp = yw + (size_t)gn * GC;
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            tv[u] = tvb[s][u];
                            v[u] = vb[s][u];
                            tvb[s][u] = tp[(u << 6) + lane];
                            vb[s][u] = yp[(u << 6) + lane];
                        }
                        do_group(tv, v, wave + jd * NH, true);
                    }
                }
            }
        }
        if (nfull * GC < N && wave == nfull % NH) {  // ragged tail (group nfull): clamped loads, masked lanes
            double tv[U];
            double2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = min(nfull * GC + (u << 6) + lane, N - 1);
                tv[u] = tm[i];
                v[u] = yw[i];
            }
            do_group(tv, v, nfull, false);
        }
    }
    __syncthreads();
    BLS_LAP(1);  // histogram
    // ---- wrap pad (reference: for n=1..oversample: mean[n_bins-oversample+n-1] = mean[n], in that order)
    if (n_bins - oversample > oversample) {  // source [1, os] and destination [n_bins-os, n_bins-1] are disjoint
        for (int q = 1 + tid; q <= oversample; q += NT) {
            ya[n_bins - oversample + q - 1] = ya[q];
            wa[n_bins - oversample + q - 1] = wa[q];
        }
    } else if (tid == 0) {
        for (int q = 1; q <= oversample; ++q) {
            ya[n_bins - oversample + q - 1] = ya[q];
            wa[n_bins - oversample + q - 1] = wa[q];
        }
    }
    __syncthreads();
    // constants of the scan's skip bound, from the per-bin sums while they are still per-bin
    const double sum_y = st.sum_y, sum_ivar = st.sum_ivar;
    {
        double g = 0.0, wm = 0.0, ysum = 0.0;
        for (int i = 1 + tid; i <= n_bins; i += NT) {
            const double vy = ya[i], vw = wa[i];
            g = fmax(g, fabs(sum_y * vw - sum_ivar * vy));
            wm = fmax(wm, vw);
            ysum += fabs(vy);
        }
        for (int o = 32; o > 0; o >>= 1) {
            g = fmax(g, __shfl_xor(g, o));
            wm = fmax(wm, __shfl_xor(wm, o));
            ysum += __shfl_xor(ysum, o);
        }
        if (lane == 0) {
            atomicMax(&s_red[0], __double_as_longlong(g));
            atomicMax(&s_red[1], __double_as_longlong(wm));
            atomicAdd(reinterpret_cast<double *>(&s_red[2]), ysum * (1.0 + 1e-6));
        }
    }
    __syncthreads();
    BLS_LAP(2);  // wrap pad + gmax / wmax reductions
    // Sequential chain acc = bins[i] + acc in index order (same rounding as the reference loop).  bins[0] is always 0, so
    // starting at i = 0 with acc = 0 is the same chain.  Lane 0: y, lane 1: ivar.
    if (!multi && NW >= 2 && !(ablate & 2)) {
        // One team, two chains.  A lone wave can issue an LDS instruction only every ~20 cycles (measured: a lane that
        // loads, adds and stores bin by bin runs at ~25 cycles per bin however the loop is pipelined; moving the operands
        // in with v_readlane costs the same in VALU issue).  So the serial pass does the minimum:
        // pass 1  lane c of wave 0 (0: y, 1: ivar) only LOADS (two bins per ds_read2_b64) and adds, keeping the running
        //         sum at every 32nd bin (`carries`): half an LDS instruction and one dependent add per bin;
        // pass 2  every thread re-runs one 32-bin block from its exact carry-in, all blocks in parallel, and writes the
        //         prefix sums in place.  Same operands in the same order as the single chain: bit-identical values.
        double *carries = zm8;  // [2][nblk32 + 1], free until the block maxima are formed
        const int total = n_bins + 1, nblk32 = (total + 31) >> 5;
        if (wtid < 2) {
            const double *comp = wtid ? wa : ya;
            double *car = carries + wtid * (nblk32 + 1);
            double acc = 0.0;
            car[0] = 0.0;
            const int nfull = total >> 5;  // full 32-bin blocks
            if (nfull > 0) {
                double x0[8], x1[8], x2[8], x3[8];
                const int lastq = nfull * 4 - 1;  // last 8-bin step inside the full blocks
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
                for (int blk = 0; blk < nfull; ++blk) {  // loads run one block (32 dependent adds) ahead
                    const int q = blk << 2;
                    // sched_barrier: keep each set's reload right behind its adds (left alone, the scheduler sinks all
                    // sixteen loads to the bottom of the loop and the next iteration waits out the full LDS round trip)
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
                    car[blk + 1] = acc;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the ragged last block needs no carry-out
        }
        __syncthreads();
        for (int item = tid; item < 2 * nblk32; item += NT) {
            const int c = item >= nblk32, blk = item - c * nblk32;
            double *comp = c ? wa : ya;
            double acc = carries[c * (nblk32 + 1) + blk];
            const int i0 = blk << 5, i1 = min(i0 + 32, total);
            int i = i0;
            for (; i + 8 <= i1; i += 8) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = comp[i + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc = x[u] + acc;
                    x[u] = acc;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) comp[i + u] = x[u];
            }
            for (; i < i1; ++i) {
                acc = comp[i] + acc;
                comp[i] = acc;
            }
        }
    } else if (wwave == 0 && lane < 2 * G && !(ablate & 2)) {
        // lane 2 g' + c: component c (0: y, 1: ivar) of team g'
        char *rg = region0 + (size_t)(lane >> 1) * region;
        double *comp = reinterpret_cast<double *>(rg) + ((lane & 1) ? cap : 0);
        const int n_bins_c = reinterpret_cast<int *>(rg + (size_t)cap * 16 + 32)[1];
        double acc = 0.0;
        // Eight bins per step on a ring of three register sets, software-pipelined by hand: a set's loads are issued two
        // steps (16 dependent adds, ~150-200 cycles: the LDS round trip seen by a lone wave) before its chain runs, and the
        // sched_group_barrier pattern asks for one LDS instruction after every add so loads, stores and the 8-cycle
        // dependent adds share the issue slots.  All loads are unconditional (the last blocks are re-read, unused): a
        // branch around them would cost the interleave.
        const int total = n_bins_c + 1, nstep = total >> 3;
        if (nstep > 0) {
            double x0[8], x1[8], x2[8];
            const int last = nstep - 1;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0[u] = comp[u];
                x1[u] = comp[(min(1, last) << 3) + u];
                x2[u] = comp[(min(2, last) << 3) + u];
            }
            auto step8 = [&](double(&x)[8], int blk, int blk_next) {
                double *cp = comp + (blk << 3);
                const double *np_ = comp + (min(blk_next, last) << 3);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc = x[u] + acc;
                    x[u] = acc;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) cp[u] = x[u];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = np_[u];
            };
            int it = 0;
            for (; it + 3 <= nstep; it += 3) {
                step8(x0, it, it + 3);
                step8(x1, it + 1, it + 4);
                step8(x2, it + 2, it + 5);
#pragma unroll
                for (int r = 0; r < 24; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  // one VALU (the dependent add)
                    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);  // one LDS instruction
                }
            }
            if (it < nstep) step8(x0, it, it);
            if (it + 1 < nstep) step8(x1, it + 1, it + 1);
        }
        for (int i = nstep << 3; i < total; ++i) {
            acc = comp[i] + acc;
            comp[i] = acc;
        }
    }
    __syncthreads();
    // Z[i] = S * W[i] - E * Y[i] (W, Y the prefix sums just formed): the numerator of both objectives is a plain
    // difference, Nn(n, dur) = Z[n + dur] - Z[n], so the maximum of Z over a block of bins bounds Nn for every window
    // of a start bin that ends in that block.  Block maxima over aligned blocks of 8 and 64 bins (1.1 B per bin).
    {
        const int nb8 = (n_bins >> 3) + 1;
        for (int b8 = tid; b8 < nb8; b8 += NT) {
            double m = -INFINITY;
            const int i0 = b8 << 3, i1 = min(i0 + 7, n_bins);
            for (int i = i0; i <= i1; ++i) m = fmax(m, st.sum_y * wa[i] - st.sum_ivar * ya[i]);
            zm8[b8] = m;
        }
        __syncthreads();
        const int nb64 = (n_bins >> 6) + 1;
        for (int b64 = tid; b64 < nb64; b64 += NT) {
            double m = -INFINITY;
            const int i0 = b64 << 3, i1 = min(i0 + 7, nb8 - 1);
            for (int i = i0; i <= i1; ++i) m = fmax(m, zm8[i]);
            zm64[b64] = m;
        }
    }
    __syncthreads();
    BLS_LAP(3);  // prefix chains + block maxima

    // ---- scan
    // A lane owns a start bin n and walks the durations in ascending length.  With a = y_out sum, b = ivar_in,
    // c = y_in sum, e = ivar_out and Nn = a*b - c*e = S*b - E*c (S, E the totals):
    //     likelihood  0.5*b*(a/e - c/b)^2 = 0.5*Nn^2 / (b*e^2)        snr  (a/e - c/b)/sqrt(1/b+1/e) = Nn / sqrt(b*e*E)
    // (1) skip-ahead: growing the window by one bin changes Nn by at most gmax, only raises b, lowers e by at most wmax,
    //     so from one evaluation the largest m with "every window up to m bins longer is STRICTLY below thr" follows
    //     in closed form (1e-4 safety factors, absolute slack far above the prefix-sum rounding); the walk jumps there.
    // (2) candidates that are reached go through a division-free conservative filter with an absolute slack eN >= every
    //     rounding the exact chain can commit, so a rejected candidate is STRICTLY below the threshold and can never be
    //     the winner (ties always reach the exact path).
    // (3) survivors (a handful per team) run the reference's exact arithmetic, which alone decides the result.
    // thr = the best objective seen so far by anyone in the team (s_thr), warm-started from a coarse lattice.
    double best = -INFINITY;
    int bk = -1, bn = -1;
    if (!(ablate & 4)) {
        const double S = sum_y, E = sum_ivar;
        const double gmax = __longlong_as_double(s_red[0]) * (1.0 + 1e-9);  // max_i |S w_i - E y_i|
        const double wmax = __longlong_as_double(s_red[1]);                 // max_i w_i
        const double yabs = __longlong_as_double(s_red[2]);                 // sum_i |y_i|
        // |Nn computed from the rounded prefix sums - Nn of the real bin sums| is below ~n_bins eps E (|S| + yabs);
        // 1e-9 leaves four orders of magnitude
        const double slack = 1e-9 * E * (fabs(S) + yabs);
        const int dmin = dur_bins[0];
        // conservative filter, then the reference's exact arithmetic; keeps the FIRST best in (caller's duration
        // index, start bin) order
        auto finish = [&](int n, int kc, double y_in, double ivar_in, double ivar_out, double Nn, double eN, double thr) {
            if (Nn + eN < 0.0) return;  // certainly y_out < y_in
            {
                const double m = fabs(Nn) + eN;
                double lhs, rhs;
                if (obj_flag) {
                    lhs = 0.5 * m * m;
                    rhs = thr * ivar_in * ivar_out * ivar_out;
                } else {
                    lhs = m * m;
                    rhs = (thr < 0.0 ? -1.0 : thr * thr) * ivar_in * ivar_out * (ivar_in + ivar_out);
                }
                if (lhs * (1.0 + 1e-12) < rhs) return;  // certainly objective < thr
            }
            double y_out = S - y_in;
            y_in /= ivar_in;
            y_out /= ivar_out;
            double obj;
            if (obj_flag) {
                const double arg = y_out - y_in;
                obj = 0.5 * ivar_in * arg * arg;
            } else {
                const double depth = y_out - y_in;
                const double depth_err = sqrt(1.0 / ivar_in + 1.0 / ivar_out);
                obj = depth / depth_err;
            }
            if (y_out >= y_in &&
                (obj > best || (obj == best && (korig[kc] < korig[bk] || (kc == bk && n < bn))))) {
                best = obj;
                bk = kc;
                bn = n;
                atomicMax(s_thr, __double_as_longlong(obj));  // obj >= 0 here: the bit pattern orders like the value
            }
        };
        // ---- warm start: every 8th start bin x every 8th duration (1.6 % of the candidates) gives the walk below a
        //      threshold close to the final best from its first step
        if (!(ablate & 128)) {
            const int cn = (n_bins - dmin) / 8 + 1, ck = (n_dur + 7) / 8;
            for (int c = tid; c < cn * ck; c += NT) {
                const int kc = (c / cn) * 8, n = (c - (c / cn) * cn) * 8;
                const int dur = dur_bins[kc];
                if (n + dur > n_bins) continue;
                const double y_in = ya[n + dur] - ya[n], ivar_in = wa[n + dur] - wa[n], ivar_out = E - ivar_in;
                if ((ivar_in < DBL_EPSILON) || (ivar_out < DBL_EPSILON)) continue;
                const double ab = S * ivar_in, ce = E * y_in;
                finish(n, kc, y_in, ivar_in, ivar_out, ab - ce, (fabs(ab) + fabs(ce)) * 1e-15,
                       fmax(best, __longlong_as_double(*s_thr)));
            }
            __syncthreads();
        }
        // objective(Nn <= ub, ivar_in >= b, ivar_out >= e) strictly below thr (1e-4 safety factor, as in the skip-ahead)
        auto block_below = [&](double ub, double b, double e, double thr) {
            if (obj_flag) return 0.5 * ub * ub < thr * b * e * e * (1.0 - 1e-4);
            return ub < thr * __builtin_amdgcn_sqrt(b * e * E) * (1.0 - 1e-4);
        };
        int n = atomicAdd(s_ctr, 1);
        int k = 0, dur_k = dmin;  // dur_k == dur_bins[k] (the sentinel once k == n_dur)
        int no64 = 0;
        double lwy = 0.0, lww = 0.0, zn = 0.0, thr_sh = __longlong_as_double(*s_thr);
        if (n + dmin <= n_bins) {
            lwy = ya[n];
            lww = wa[n];
            zn = S * lww - E * lwy;
        }
        while (n + dmin <= n_bins) {  // start bins are handed out in ascending order: the first miss ends the lane
            if (n + dur_k > n_bins) {  // durations ascend: once one overruns, every later one does too
                n = atomicAdd(s_ctr, 1);
                k = 0;
                dur_k = dmin;
                no64 = 0;
                if (n + dmin <= n_bins) {
                    lwy = ya[n];
                    lww = wa[n];
                    zn = S * lww - E * lwy;
                }
                continue;
            }
            const int dur = dur_k, kc = k;
            const int j = n + dur;
            const double hw = wa[j];
            const double thr = fmax(best, thr_sh);  // one iteration stale: still a valid (lower) bound
            thr_sh = __longlong_as_double(*s_thr);
            const double ivar_in = hw - lww;
            const double ivar_out = E - ivar_in;
            // ---- block skip: every window of this start bin that ends inside the aligned block of 64 (then 8) bins around
            //      j has Nn <= max(Z over the block) - Z[n], ivar_in >= this window's (ivar >= 0: windows only grow) and
            //      ivar_out >= this one's minus the block's length x wmax.  If that bound is strictly below thr, none of
            //      them can win: jump to the first duration that ends beyond the block.
            if (thr > 0.0 && ivar_in >= DBL_EPSILON && !(ablate & 32)) {
                int jend = ((j >> 6) + 1) << 6;
                double ub, e_lb;
                bool skip = false;
                if (j >= no64) {  // a 64-block that failed once is not asked again while the walk is still inside it
                    ub = zm64[j >> 6] - zn + slack;
                    e_lb = ivar_out - 64.0 * wmax;
                    skip = ub < 0.0 || (e_lb > 0.0 && block_below(ub, ivar_in, e_lb, thr));
                    if (!skip) no64 = jend;
                }
                if (!skip) {
                    jend = ((j >> 3) + 1) << 3;
                    ub = zm8[j >> 3] - zn + slack;
                    e_lb = ivar_out - 8.0 * wmax;
                    skip = ub < 0.0 || (e_lb > 0.0 && block_below(ub, ivar_in, e_lb, thr));
                }
                if (skip) {
                    const int len = jend - n;  // first duration (in bins) that ends beyond the block
                    if (len > max_dur) {
                        k = n_dur;
                        dur_k = 0xffff;  // the sentinel: this start bin is exhausted
                    } else {
                        const int kd = first_kd[len];
                        k = kd & 0xffff;
                        dur_k = (int)((unsigned)kd >> 16);
                    }
                    continue;
                }
            }
            const double hy = ya[j];
            dur_k = dur_bins[++k];
            const double y_in = hy - lwy;
            if ((ivar_in < DBL_EPSILON) || (ivar_out < DBL_EPSILON)) continue;
            // Nn = y_out ivar_in - y_in ivar_out = S ivar_in - E y_in in real arithmetic
            const double ab = S * ivar_in, ce = E * y_in;
            const double Nn = ab - ce;
            // ---- how many more bins this window may grow before it could reach thr
            if (thr > 0.0 && !(ablate & 64)) {
                double mf;
                if (obj_flag) {
                    const double q = __builtin_amdgcn_sqrt(2.0 * thr * ivar_in) * (1.0 - 1e-4);
                    mf = (q * ivar_out - Nn - slack) * __builtin_amdgcn_rcp(gmax + q * wmax) * (1.0 - 1e-4) - 1.0;
                } else {
                    const double e_lo = ivar_out - 64.0 * wmax;
                    const double q = thr * __builtin_amdgcn_sqrt(ivar_in * e_lo * E) * (1.0 - 1e-4);
                    mf = e_lo > 0.0 ? fmin((q - Nn - slack) * __builtin_amdgcn_rcp(gmax) * (1.0 - 1e-4) - 1.0, 64.0) : 0.0;
                }
                if (mf >= 1.0) {
                    const int m = (int)fmin(mf, 1.0e6);
                    const int kd = first_kd[min(dur + m, max_dur) + 1];  // >= kc + 1 since dur_bins[kc] < dur + m + 1
                    k = kd & 0xffff;
                    dur_k = (int)((unsigned)kd >> 16);
                    continue;  // m >= 1 means this candidate itself is below thr as well
                }
            }
            finish(n, kc, y_in, ivar_in, ivar_out, Nn, (fabs(ab) + fabs(ce)) * 1e-15, thr);
        }
    }
    // ---- winner: (objective desc, caller's duration index asc, start bin asc) is a strict total order on distinct
    //      candidates, so the xor butterfly leaves a wave's winner in every lane; lane 0 of each wave posts it
    for (int o = 32; o > 0; o >>= 1) {
        const double oo = __shfl_xor(best, o);
        const int ok = __shfl_xor(bk, o), on = __shfl_xor(bn, o);
        const bool take = ok >= 0 && (bk < 0 || oo > best ||
                                      (oo == best && (korig[ok] < korig[bk] || (ok == bk && on < bn))));
        if (take) {
            best = oo;
            bk = ok;
            bn = on;
        }
    }
    if (lane == 0) s_best[wave] = BlsBest{best, bk, bn};
    __syncthreads();
    BLS_LAP(4);  // scan
    if (tid == 0 && emit) {
        BlsBest w = s_best[0];
        for (int i = 1; i < NW; ++i) {
            const BlsBest o = s_best[i];
            const bool take = o.k >= 0 && (w.k < 0 || o.obj > w.obj ||
                                            (o.obj == w.obj && (korig[o.k] < korig[w.k] || (o.k == w.k && o.n < w.n))));
            if (take) w = o;
        }
        const size_t stride = (size_t)B * (size_t)nP;
        double *o = out7 + (size_t)target * (size_t)nP + (size_t)p;
        if (w.k < 0) {
            o[0] = -INFINITY;
            for (int f = 1; f < 7; ++f) o[f * stride] = 0.0;
        } else {
            const int dur = dur_bins[w.k], n = w.n;
            double y_in = ya[n + dur] - ya[n];
            const double ivar_in = wa[n + dur] - wa[n];
            double y_out = sum_y - y_in;
            const double ivar_out = sum_ivar - ivar_in;
            y_in /= ivar_in;
            y_out /= ivar_out;
            const double arg = y_out - y_in;
            const double log_like = 0.5 * ivar_in * arg * arg;
            const double depth = y_out - y_in;
            const double depth_err = sqrt(1.0 / ivar_in + 1.0 / ivar_out);
            const double depth_snr = depth / depth_err;
            const double duration = dur * bin_duration;
            const double phase = fmod(n * bin_duration + 0.5 * duration + st.min_t, P);
            o[0] = w.obj;
            o[1 * stride] = depth;
            o[2 * stride] = depth_err;
            o[3 * stride] = duration;
            o[4 * stride] = phase;
            o[5 * stride] = depth_snr;
            o[6 * stride] = log_like;
        }
    }
    BLS_LAP(5);  // final reduction + outputs
    if (prof && threadIdx.x == 0) atomicAdd(&prof[7], 1ull);
#undef BLS_LAP
}

#define BLS_TEAM_ARGS                                                                                                  \
    const double *__restrict__ tm, const double2 *__restrict__ yw, const int64_t *__restrict__ n_off,                  \
        const BlsStats *__restrict__ stats, const double *__restrict__ period, const int *__restrict__ pidx,           \
        int np_group, int64_t nP, int B, const int *__restrict__ dur_tab, int n_dur, int max_dur, double bin_duration, \
        int oversample, int obj_flag, double *__restrict__ out7, int cap, int shape, int ablate, unsigned long long *__restrict__ prof
#define BLS_TEAM_PASS \
    tm, yw, n_off, stats, period, pidx, np_group, nP, B, dur_tab, n_dur, max_dur, bin_duration, oversample, obj_flag, out7, cap, shape, ablate, prof

// 80 VGPRs: 6 waves per SIMD, for the groups whose LDS footprint admits >= 3 teams per CU
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6, 6))) void bls_team_kernel(BLS_TEAM_ARGS) {
    bls_team_body<2, 2>(BLS_TEAM_PASS);
}
// 128 VGPRs: the long periods (one or two teams per CU, <= 16 waves): four chunks at a time, eight more in flight
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void bls_team_deep_kernel(BLS_TEAM_ARGS) {
    bls_team_body<4, 2>(BLS_TEAM_PASS);
}

// lk_create-time check of the hardware property the atomic histogram rests on: one ds_add_f64 applies same-address
// lanes in increasing lane order.  256 bins, 24 chunks of adversarial patterns per workgroup; out[0] counts mismatches
// against the sequential sum done by lane 0.
__global__ __launch_bounds__(64) void bls_selftest_kernel(int *__restrict__ bad) {
    __shared__ double bins[256], ref[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) bins[i] = ref[i] = 0.0;
    wave_sync();
    unsigned long long s = 0x9e3779b97f4a7c15ull * (unsigned long long)(blockIdx.x + 1);
    for (int c = 0; c < 24; ++c) {
        // the same pseudo-random stream in every lane; lane l keeps draw l
        int j = 0;
        double v = 0.0;
        int cur = 0;
        for (int l = 0; l < 64; ++l) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const unsigned r = (unsigned)(s >> 33);
            int jj;
            if ((c & 3) == 0) {
                if (l == 0) cur = r & 255;
                else if ((r & 15) < 9) cur = (cur + 1) & 255;
                jj = cur;
            } else if ((c & 3) == 1) jj = (int)(r % 19u);
            else if ((c & 3) == 2) jj = (int)((r >> 5) & 1u) * 7;
            else jj = (int)(r & 255u);
            const double vv = ldexp(1.0 + (double)(r & 0xfffffu) * 9.5367431640625e-07, (int)((r >> 20) % 20u) - 10) *
                              ((r >> 31) ? -1.0 : 1.0);
            if (l == lane) {
                j = jj;
                v = vv;
            }
            if (lane == 0) ref[jj] += vv;  // sequential, lane order
        }
        atomicAdd(&bins[j], v);
        wave_sync();
    }
    int nbad = 0;
    for (int i = lane; i < 256; i += 64)
        nbad += __double_as_longlong(bins[i]) != __double_as_longlong(ref[i]);
    if (nbad) atomicAdd(bad, nbad);
}

// ------------------------------------------------------------------------------------------------ launcher
// The LDS plan of a team: the packed duration tables + 16 B per phase bin (ya | wa) + the block maxima.  bls_plan_max_bins
// answers "how many phase bins fit" for a set of durations — bls_launch's own limit and lk_bls_max_period's answer (the
// seam into astropy sends longer periods to the original implementation: methods.bls_fast has no such limit).

// ------------------------------------------------------------------------------------------------ wide periods
// Periods whose phase bins do not fit LDS (period / (min duration / oversample) beyond ~9 000 bins: a multi-year baseline
// searched with short durations) — astropy's run_bls has no such limit, so neither may the seam (round 5 merged those rows
// from astropy on the CPU).  One 1024-thread workgroup per (target, period), bins in a global-memory slab of the
// workgroup's own, every step arranged so that the bits are the reference's:
//   histogram  thread j OWNS the bins [j c, (j + 1) c): the cadences pass by in order, 2048 at a time through LDS (bin
//              index, y * ivar, ivar), and every thread adds the ones that fall into its range — each bin accumulates in
//              cadence order with no atomics at all (N compares per thread; the price of a path that runs for a handful
//              of periods per search)
//   prefix     2048-bin tiles through LDS, one lane per array runs the sequential chain with the carry in a register
//   scan       every (duration, start bin) candidate with the reference's exact arithmetic, durations in the CALLER's order,
//              a thread's start bins ascending: "first best wins" per thread, then (objective desc, duration index asc,
//              start bin asc) across threads — the reference's loop order
// ~0.1-1 ms per (target, period): 100 x the LDS kernels' cost per period, for periods they cannot take at all.
constexpr int BLSW_NT = 1024, BLSW_CH = 2048;
__global__ __launch_bounds__(BLSW_NT) void bls_wide_kernel(const double *__restrict__ tm, const double2 *__restrict__ yw,
                                                            const int64_t *__restrict__ n_off,
                                                            const BlsStats *__restrict__ stats,
                                                            const double *__restrict__ period, const int *__restrict__ pidx,
                                                            int n_wide, int64_t nP, int B, const int *__restrict__ dur_caller,
                                                            int n_dur, double bin_duration, int oversample, int obj_flag,
                                                            double *__restrict__ out7, double *__restrict__ slabs,
                                                            size_t cap) {
    __shared__ int s_ind[BLSW_CH];
    __shared__ double s_y[BLSW_CH], s_w[BLSW_CH];
    __shared__ BlsBest s_best[BLSW_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *ya = slabs + (size_t)blockIdx.x * 2 * cap, *wa = ya + cap;
    const long long pairs = (long long)B * n_wide;
    for (long long pair = blockIdx.x; pair < pairs; pair += gridDim.x) {
        const int target = (int)(pair / n_wide), p = pidx[pair % n_wide];
        const double P = period[p], invP = 1.0 / P;
        const int64_t lo = n_off[target];
        const int N = (int)(n_off[target + 1] - lo);
        const BlsStats st = stats[target];
        const double sum_y = st.sum_y, sum_ivar = st.sum_ivar;
        const int n_bins = (int)(ceil(P / bin_duration)) + oversample;
        for (int i = tid; i <= n_bins; i += BLSW_NT) ya[i] = 0.0, wa[i] = 0.0;
        __syncthreads();
        // ---- histogram
        const int c = (n_bins + 1 + BLSW_NT - 1) / BLSW_NT;
        const int b_lo = tid * c, b_hi = min(b_lo + c, n_bins + 1);
        for (int c0 = 0; c0 < N; c0 += BLSW_CH) {
            const int cn = min(BLSW_CH, N - c0);
            for (int i = tid; i < cn; i += BLSW_NT) {
                double k, r;
                fold_exact(tm[lo + c0 + i], P, invP, &k, &r);
                const double2 v = yw[lo + c0 + i];
                s_ind[i] = bin_of(r, bin_duration);
                s_y[i] = v.x;
                s_w[i] = v.y;
            }
            __syncthreads();
            if (b_lo < b_hi)
                for (int i = 0; i < cn; ++i) {
                    const int ind = s_ind[i];
                    if (ind >= b_lo && ind < b_hi) {
                        ya[ind] += s_y[i];
                        wa[ind] += s_w[i];
                    }
                }
            __syncthreads();
        }
        // ---- wrap (sources 1 .. oversample and destinations n_bins - oversample .. n_bins - 1 are disjoint: n_bins is large)
        for (int n = 1 + tid; n <= oversample; n += BLSW_NT) {
            const int ind = n_bins - oversample + n - 1;
            ya[ind] = ya[n];
            wa[ind] = wa[n];
        }
        __syncthreads();
        // ---- inclusive prefix sums, sequential like the reference
        double cy = ya[0], cw = wa[0];   // (used by threads 0 and 64 only)
        for (int t0 = 1; t0 <= n_bins; t0 += BLSW_CH) {
            const int tn = min(BLSW_CH, n_bins + 1 - t0);
            for (int i = tid; i < tn; i += BLSW_NT) s_y[i] = ya[t0 + i], s_w[i] = wa[t0 + i];
            __syncthreads();
            if (tid == 0)
                for (int i = 0; i < tn; ++i) {
                    cy = s_y[i] + cy;
                    s_y[i] = cy;
                }
            if (tid == 64)
                for (int i = 0; i < tn; ++i) {
                    cw = s_w[i] + cw;
                    s_w[i] = cw;
                }
            __syncthreads();
            for (int i = tid; i < tn; i += BLSW_NT) ya[t0 + i] = s_y[i], wa[t0 + i] = s_w[i];
            __syncthreads();
        }
        // ---- scan
        double best = -INFINITY;
        int bk = -1, bn = 0;
        for (int k = 0; k < n_dur; ++k) {
            const int dur = dur_caller[k], n_max = n_bins - dur;
            for (int n = tid; n <= n_max; n += BLSW_NT) {
                double y_in = ya[n + dur] - ya[n];
                const double ivar_in = wa[n + dur] - wa[n];
                double y_out = sum_y - y_in;
                const double ivar_out = sum_ivar - ivar_in;
                if ((ivar_in < DBL_EPSILON) || (ivar_out < DBL_EPSILON)) continue;
                y_in /= ivar_in;
                y_out /= ivar_out;
                double obj;
                if (obj_flag) {
                    const double arg = y_out - y_in;
                    obj = 0.5 * ivar_in * arg * arg;
                } else {
                    const double depth = y_out - y_in;
                    const double depth_err = sqrt(1.0 / ivar_in + 1.0 / ivar_out);
                    obj = depth / depth_err;
                }
                if (y_out >= y_in && obj > best) {
                    best = obj;
                    bk = k;
                    bn = n;
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double oo = __shfl_xor(best, o);
            const int ok = __shfl_xor(bk, o), on = __shfl_xor(bn, o);
            const bool take = ok >= 0 && (bk < 0 || oo > best || (oo == best && (ok < bk || (ok == bk && on < bn))));
            if (take) best = oo, bk = ok, bn = on;
        }
        if (lane == 0) s_best[wave] = BlsBest{best, bk, bn};
        __syncthreads();
        if (tid == 0) {
            BlsBest w = s_best[0];
            for (int i = 1; i < BLSW_NT / 64; ++i) {
                const BlsBest o = s_best[i];
                const bool take = o.k >= 0 && (w.k < 0 || o.obj > w.obj || (o.obj == w.obj && (o.k < w.k || (o.k == w.k && o.n < w.n))));
                if (take) w = o;
            }
            const size_t stride = (size_t)B * (size_t)nP;
            double *o = out7 + (size_t)target * (size_t)nP + (size_t)p;
            if (w.k < 0) {
                o[0] = -INFINITY;
                for (int f = 1; f < 7; ++f) o[f * stride] = 0.0;
            } else {
                const int dur = dur_caller[w.k], n = w.n;
                double y_in = ya[n + dur] - ya[n];
                const double ivar_in = wa[n + dur] - wa[n];
                double y_out = sum_y - y_in;
                const double ivar_out = sum_ivar - ivar_in;
                y_in /= ivar_in;
                y_out /= ivar_out;
                const double arg = y_out - y_in;
                const double log_like = 0.5 * ivar_in * arg * arg;
                const double depth = y_out - y_in;
                const double depth_err = sqrt(1.0 / ivar_in + 1.0 / ivar_out);
                const double depth_snr = depth / depth_err;
                const double duration = dur * bin_duration;
                const double phase = fmod(n * bin_duration + 0.5 * duration + st.min_t, P);
                o[0] = w.obj;
                o[1 * stride] = depth;
                o[2 * stride] = depth_err;
                o[3 * stride] = duration;
                o[4 * stride] = phase;
                o[5 * stride] = depth_snr;
                o[6 * stride] = log_like;
            }
        }
        __syncthreads();   // the slab is reused by this workgroup's next (target, period)
    }
}

static size_t bls_region_of(int cap, int nw_) {
    return ((size_t)cap * 16 + 48 + (size_t)nw_ * 16 + (size_t)((cap >> 3) + (cap >> 6) + 4) * 8 + 15) & ~(size_t)15;
}
static int bls_max_bins_for(size_t tab_bytes) {
    int lo = 0, hi = 16384;  // largest even cap with tab_bytes + region_of(cap, 16) <= 156 KB; bins = cap - 2
    while (lo < hi) {
        const int mid = ((lo + hi + 2) / 2) & ~1;
        if (tab_bytes + bls_region_of(mid, 16) <= 156 * 1024)
            lo = mid;
        else
            hi = mid - 2;
    }
    return lo - 2;
}

int bls_max_period_host(const double *duration_host, int nD, int oversample, double *max_period) {
    LK_REQUIRE(duration_host && max_period && nD >= 1 && oversample >= 1, "bad arguments");
    double min_duration = duration_host[0];
    for (int k = 0; k < nD; ++k) {
        LK_REQUIRE(std::isfinite(duration_host[k]) && duration_host[k] >= DBL_EPSILON, "Invalid inputs for period and/or duration");
        min_duration = std::min(min_duration, duration_host[k]);
    }
    const double bin_duration = min_duration / ((double)oversample);
    std::vector<int> dur_bins;
    for (int k = 0; k < nD; ++k) {
        const int d = (int)(std::round(duration_host[k] / bin_duration));
        if (std::find(dur_bins.begin(), dur_bins.end(), d) == dur_bins.end()) dur_bins.push_back(d);
    }
    const int nd = (int)dur_bins.size(), max_dur = *std::max_element(dur_bins.begin(), dur_bins.end());
    LK_REQUIRE(nd < 65535 && max_dur < 65535, "too many / too long durations for the packed duration table");
    const size_t tab_bytes = (((size_t)(2 * nd + 1 + max_dur + 2) * 4 + 15) / 16) * 16;
    LK_REQUIRE(tab_bytes <= 24 * 1024, "%d durations up to %d bins: the LDS plan holds 24 KB of duration tables", nd, max_dur);
    // n_bins(P) = ceil(P / bin_duration) + oversample <= max_bins
    const int max_bins = bls_max_bins_for(tab_bytes);
    *max_period = (double)(max_bins - oversample) * bin_duration;
    return LK_OK;
}

int bls_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *ivar,
               const double *period_host, const double *period_dev, int64_t nP, const double *duration_host, int nD,
               int oversample, int use_likelihood, double *out7, hipStream_t stream) {
    LK_REQUIRE(B >= 0 && n_off_host != nullptr, "bad batch description");
    LK_REQUIRE(nP >= 0 && nD >= 1, "need nP >= 0 and nD >= 1");
    if (B == 0 || nP == 0) return LK_OK;
    LK_REQUIRE(t && y && ivar && period_host && period_dev && duration_host && out7, "NULL buffer");
    LK_REQUIRE(oversample >= 1, "oversample must be greater than or equal to 1");
    LK_REQUIRE(nP < ((int64_t)1 << 31), "too many periods");
    LK_REQUIRE(n_off_host[0] == 0, "n_off[0] must be 0");
    for (int b = 0; b < B; ++b) {
        const int64_t n = n_off_host[b + 1] - n_off_host[b];
        LK_REQUIRE(n >= 1 && n < ((int64_t)1 << 30), "target %d has %lld cadences", b, (long long)n);
    }
    // the reference's run_bls input checks (nonzero return -> ValueError "Invalid inputs for period and/or duration")
    double min_period = period_host[0], max_period = period_host[0];
    for (int64_t k = 0; k < nP; ++k) {
        LK_REQUIRE(std::isfinite(period_host[k]), "Invalid inputs for period and/or duration (non-finite period)");
        min_period = std::min(min_period, period_host[k]);
        max_period = std::max(max_period, period_host[k]);
    }
    double min_duration = duration_host[0], max_duration = duration_host[0];
    for (int k = 0; k < nD; ++k) {
        LK_REQUIRE(std::isfinite(duration_host[k]), "Invalid inputs for period and/or duration (non-finite duration)");
        min_duration = std::min(min_duration, duration_host[k]);
        max_duration = std::max(max_duration, duration_host[k]);
    }
    LK_REQUIRE(min_period >= DBL_EPSILON, "Invalid inputs for period and/or duration");
    LK_REQUIRE(max_duration <= min_period && min_duration >= DBL_EPSILON, "Invalid inputs for period and/or duration");

    const double bin_duration = min_duration / ((double)oversample);
    // duration in bins; duplicates (same dur -> identical candidates, the first one wins anyway) are dropped
    std::vector<int> dur_bins;
    for (int k = 0; k < nD; ++k) {
        const int d = (int)(std::round(duration_host[k] / bin_duration));
        if (std::find(dur_bins.begin(), dur_bins.end(), d) == dur_bins.end()) dur_bins.push_back(d);
    }
    // scan tables: durations sorted by length (the kernel's skip-ahead walks them in ascending order), the caller's
    // index of each (the reference's tie-break order), and first_k[L] = first sorted k with dur_bins[k] >= L
    const int nd = (int)dur_bins.size();
    std::vector<int> sidx((size_t)nd);
    std::iota(sidx.begin(), sidx.end(), 0);
    std::stable_sort(sidx.begin(), sidx.end(), [&](int a, int b) { return dur_bins[a] < dur_bins[b]; });
    const int max_dur = dur_bins[sidx[nd - 1]];
    LK_REQUIRE(dur_bins[sidx[0]] >= 1, "Invalid inputs for period and/or duration (a duration shorter than half a bin)");
    LK_REQUIRE(nd < 65535 && max_dur < 65535, "too many / too long durations for the packed duration table");
    std::vector<int> dur_tab((size_t)(2 * nd + 1 + max_dur + 2));
    for (int k = 0; k < nd; ++k) {
        dur_tab[k] = dur_bins[sidx[k]];
        dur_tab[nd + 1 + k] = sidx[k];
    }
    dur_tab[nd] = 0xffff;  // sentinel: longer than any bin array the LDS plan admits
    for (int L = 0, k = 0; L <= max_dur + 1; ++L) {
        while (k < nd && dur_tab[k] < L) ++k;
        dur_tab[2 * nd + 1 + L] = (int)((unsigned)k | ((unsigned)dur_tab[k] << 16));
    }
    const size_t tab_bytes = ((dur_tab.size() * 4 + 15) / 16) * 16;
    LK_REQUIRE(tab_bytes <= 24 * 1024, "%d durations up to %d bins: the LDS plan holds 24 KB of duration tables", nd, max_dur);
    // periods grouped by LDS need (n_bins), longest first
    std::vector<int> order((size_t)nP);
    std::iota(order.begin(), order.end(), 0);
    auto nbins_of = [&](int p) { return (int)(std::ceil(period_host[p] / bin_duration)) + oversample; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return period_host[a] > period_host[b]; });
    LK_REQUIRE(max_period / bin_duration < 1.0e9, "max period / (min duration / oversample) = %.3g phase bins: too many",
               max_period / bin_duration);
    const int max_bins = nbins_of(order[0]);
    const size_t tab_bytes16 = tab_bytes;
    // LDS of one team: ya | wa | header | s_best[NW] | block maxima (bls_team_body's carve)
    auto region_of = [&](int cap, int nw_) { return bls_region_of(cap, nw_); };
    // periods whose bins do not fit LDS (the head of `order`) take bls_wide_kernel: bins in a global-memory slab per workgroup
    int n_wide = 0;
    while (n_wide < (int)nP && tab_bytes16 + region_of((nbins_of(order[n_wide]) + 2) & ~1, 16) > 156 * 1024) ++n_wide;
    const size_t wide_cap = n_wide ? (((size_t)max_bins + 2 + 31) & ~(size_t)31) : 0;
    const int wide_slabs = n_wide ? (int)std::min<long long>((long long)B * n_wide, 2 * (long long)h->num_cu) : 0;

    const size_t ntot = (size_t)n_off_host[B];
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 8 + (size_t)B * sizeof(BlsStats) + 3 * (ntot * 8 + 256) +
                           (size_t)nP * 4 + dur_tab.size() * 4 + (size_t)nd * 4 + (size_t)wide_slabs * 2 * wide_cap * 8 + 8192);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    BlsStats *d_stats = (BlsStats *)h->ws.alloc((size_t)B * sizeof(BlsStats));
    double *d_tm = (double *)h->ws.alloc(ntot * 8);
    double2 *d_yw = (double2 *)h->ws.alloc(ntot * 16);
    int *d_pidx = (int *)h->ws.alloc((size_t)nP * 4);
    int *d_dur = (int *)h->ws.alloc(dur_tab.size() * 4);
    int *d_dur_caller = (int *)h->ws.alloc((size_t)nd * 4);
    double *d_slabs = n_wide ? (double *)h->ws.alloc((size_t)wide_slabs * 2 * wide_cap * 8) : nullptr;
    LK_REQUIRE(!n_wide || d_slabs, "workspace exhausted (wide-period slabs)");
    LK_HIP_CHECK(hipMemcpyAsync(d_dur_caller, dur_bins.data(), (size_t)nd * 4, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_off, n_off_host, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_pidx, order.data(), (size_t)nP * 4, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_dur, dur_tab.data(), dur_tab.size() * 4, hipMemcpyHostToDevice, stream));
    // pageable-host async copies are staged before return, but be explicit: the vectors die at scope exit
    LK_HIP_CHECK(hipStreamSynchronize(stream));

    hipLaunchKernelGGL(bls_prep_kernel, dim3(B), dim3(256), 0, stream, t, y, ivar, d_off, d_tm, d_yw, d_stats);

    if (h->bls_attr_set != 1) {
        {
            int rc_ = want_lds(h, reinterpret_cast<const void *>(bls_team_kernel), 156 * 1024);
            if (!rc_) rc_ = want_lds(h, reinterpret_cast<const void *>(bls_team_deep_kernel), 156 * 1024);
            if (rc_) return rc_;
        }
        // the histogram rests on one hardware property (same-address lanes of a ds_add_f64 are applied in lane order):
        // check it on this device once per handle and refuse to run without it
        int *d_bad = (int *)h->ws.alloc(256);
        LK_HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, stream));
        hipLaunchKernelGGL(bls_selftest_kernel, dim3(64), dim3(64), 0, stream, d_bad);
        int bad = -1;
        LK_HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, stream));
        LK_HIP_CHECK(hipStreamSynchronize(stream));
        // not lane-ordered: the histogram falls back to its atomic-free form (one lane at a time, see the kernel) — slower,
        // still bit-exact
        h->bls_serial_hist = bad != 0 ? 1 : 0;
        h->bls_attr_set = 1;
    }
    const int serial_hist = (h->bls_serial_hist || h->bls_force_serial_hist) ? 2 : 0;
    // debug knobs, read once per process: LK_BLS_ABLATE skips phases (results wrong; phase costs by difference),
    // LK_BLS_PROF=1 prints the wall time of every launch, =2 adds per-phase clocks (their atomics distort short teams)
#ifdef LK_BLS_DEBUG   // `make DEBUG=1`: the phase-ablation and profiling switches of the development builds
    static const int ablate = getenv("LK_BLS_ABLATE") ? atoi(getenv("LK_BLS_ABLATE")) : 0;
    static const int prof_level = getenv("LK_BLS_PROF") ? atoi(getenv("LK_BLS_PROF")) : 0;
#else
    constexpr int ablate = 0, prof_level = 0;
#endif
    const bool prof_on = prof_level != 0;
    unsigned long long *d_prof = nullptr;
    if (prof_level >= 2) {
        LK_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d_prof), 64));
        LK_HIP_CHECK(hipMemset(d_prof, 0, 64));
    }
#ifdef LK_BLS_DEBUG   // shape experiments of the development builds
    static const int kMultiMinWaves = getenv("LK_BLS_MULTIMIN") ? atoi(getenv("LK_BLS_MULTIMIN")) : 16;
    static const int nh_of_nw2 = getenv("LK_BLS_NH2") ? atoi(getenv("LK_BLS_NH2")) : 2;
    static const int nh_of_nw4 = getenv("LK_BLS_NH4") ? atoi(getenv("LK_BLS_NH4")) : 4;
#else
    constexpr int kMultiMinWaves = 16;  // multi-period workgroups only where they keep this many waves per CU
    constexpr int nh_of_nw2 = 2, nh_of_nw4 = 4;
#endif
    constexpr int kHistWaves = 4;   // ticket-ordered histogram waves per team (8 and 16 measure the same)
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (prof_on) {
        LK_HIP_CHECK(hipEventCreate(&pe0));
        LK_HIP_CHECK(hipEventCreate(&pe1));
    }
    if (n_wide)
        hipLaunchKernelGGL(bls_wide_kernel, dim3((unsigned)wide_slabs), dim3(BLSW_NT), 0, stream, d_tm, d_yw, d_off, d_stats,
                           period_dev, d_pidx, n_wide, nP, B, d_dur_caller, nd, bin_duration, oversample,
                           use_likelihood ? 1 : 0, out7, d_slabs, wide_cap);
    // Small jobs (a seam call at B = 1: ~20 groups of a few hundred teams, every launch less than one wave of workgroups and
    // ~100 us of one team's latency): the groups are independent — they write different periods' outputs — so they are spread
    // over the handle's streams and run side by side (round 6: 2.4 -> 0.7 ms of team kernels per 5000-period call).
    const int n_spread = (!prof_on && (size_t)B * (size_t)nP <= 65536) ? 4 : 1;
    hipStream_t caller_stream = stream;
    struct SpreadGuard {  // an error return between fork and join drains the side streams
        lk_handle *h;
        int n;
        bool joined;
        ~SpreadGuard() {
            if (!joined)
                for (int a = 0; a < n; ++a)
                    if (h->s_ls_aux[a]) (void)hipStreamSynchronize(h->s_ls_aux[a]);
        }
    } spread_guard{h, n_spread - 1, n_spread == 1};
    if (n_spread > 1) {
        if (!h->ev_ls_fork) LK_HIP_CHECK(hipEventCreateWithFlags(&h->ev_ls_fork, hipEventDisableTiming));
        LK_HIP_CHECK(hipEventRecord(h->ev_ls_fork, caller_stream));
        for (int a = 0; a < n_spread - 1; ++a) {
            if (!h->s_ls_aux[a]) LK_HIP_CHECK(hipStreamCreateWithFlags(&h->s_ls_aux[a], hipStreamNonBlocking));
            if (!h->ev_ls_join[a]) LK_HIP_CHECK(hipEventCreateWithFlags(&h->ev_ls_join[a], hipEventDisableTiming));
            LK_HIP_CHECK(hipStreamWaitEvent(h->s_ls_aux[a], h->ev_ls_fork, 0));
        }
    }
    int group_no = 0;
    size_t g0 = (size_t)n_wide;
    while (g0 < (size_t)nP) {
        const int head_bins = nbins_of(order[g0]);
        stream = (group_no % n_spread) ? h->s_ls_aux[group_no % n_spread - 1] : caller_stream;
        ++group_no;
        if (prof_on) LK_HIP_CHECK(hipEventRecord(pe0, stream));
        // groups: cut whenever the LDS need drops below 7/8 of the group's head (occupancy stays close to the need) — and
        // where one more single-period team would fit a CU (a group that straddles such a step runs all its periods at the
        // head's occupancy: the 4983-bin group ran one team per CU although two thirds of its periods leave room for two)
        // (LDS is handed out in granules: a team that needs 54.4 KB does not fit three times into 160 KB — measured: the
        // group ran at two teams per CU — so the estimate rounds the need up to 2 KB)
        auto teams_of = [&](int nb) {
            return (int)((160 * 1024) / ((tab_bytes16 + region_of((nb + 2) & ~1, 16) + 2047) & ~(size_t)2047));
        };
        const int head_teams = teams_of(head_bins);
        size_t g1 = g0 + 1;
        while (g1 < (size_t)nP && nbins_of(order[g1]) * 8 >= head_bins * 7 &&
               (head_teams >= 4 || teams_of(nbins_of(order[g1])) == head_teams))
            ++g1;
        const int npg = (int)(g1 - g0);
        const int cap = (head_bins + 2) & ~1;  // doubles per component array (even: both arrays 16-B aligned)
        // shape: short periods -> G one-wave teams per workgroup (shared tables and prefix pass); long periods -> one
        // team of NW waves.  Either way aim at the 24 wave slots the 80-VGPR build leaves per CU.
        int multi = 0, nw = 1, gsel = 1, waves_cu = 0;
        for (int gc : {8, 4, 2}) {
            const size_t l = tab_bytes16 + (size_t)gc * region_of(cap, 1);
            if (l > 156 * 1024) continue;
            const int w = std::min(24, (int)((160 * 1024) / (l + 64)) * gc);
            if (w > waves_cu) {
                waves_cu = w;
                gsel = gc;
            }
        }
        int nt;
        size_t lds;
        if (waves_cu >= kMultiMinWaves) {
            multi = 1;
            nt = 64 * gsel;
            lds = tab_bytes16 + (size_t)gsel * region_of(cap, 1);
        } else {
            const int teams = std::max(1, teams_of(head_bins));
            nw = 2;  // the two-pass prefix and the ticket histogram want at least two waves
            while (nw < 16 && teams * nw * 2 <= 24) nw *= 2;
            nt = 64 * nw;
            lds = tab_bytes16 + region_of(cap, nw);
            waves_cu = teams * nw;
        }
        const size_t nwg = (size_t)((B + 7) / 8) * 8 * (((size_t)npg + gsel * multi + (1 - multi) - 1) / (multi ? gsel : 1));
        LK_REQUIRE(nwg < ((size_t)1 << 31), "grid too large");
        const bool deep = !multi && waves_cu <= 16;  // <= 4 waves per SIMD: the 128-VGPR build fits
        const int shape = multi | serial_hist | ((nw <= 2 ? nh_of_nw2 : nw <= 4 ? nh_of_nw4 : kHistWaves) << 8);
        if (deep)
            hipLaunchKernelGGL(bls_team_deep_kernel, dim3((unsigned)nwg), dim3(nt), lds, stream, d_tm, d_yw, d_off, d_stats,
                               period_dev, d_pidx + g0, npg, nP, B, d_dur, nd, max_dur, bin_duration, oversample,
                               use_likelihood ? 1 : 0, out7, cap, shape, ablate, d_prof);
        else
            hipLaunchKernelGGL(bls_team_kernel, dim3((unsigned)nwg), dim3(nt), lds, stream, d_tm, d_yw, d_off, d_stats,
                               period_dev, d_pidx + g0, npg, nP, B, d_dur, nd, max_dur, bin_duration, oversample,
                               use_likelihood ? 1 : 0, out7, cap, shape, ablate, d_prof);
        if (prof_on) {
            unsigned long long hp[8];
            float ms = 0.f;
            LK_HIP_CHECK(hipEventRecord(pe1, stream));
            LK_HIP_CHECK(hipEventSynchronize(pe1));
            LK_HIP_CHECK(hipEventElapsedTime(&ms, pe0, pe1));
            memset(hp, 0, sizeof(hp));
            if (d_prof) {
                LK_HIP_CHECK(hipMemcpy(hp, d_prof, 64, hipMemcpyDeviceToHost));
                LK_HIP_CHECK(hipMemset(d_prof, 0, 64));
            }
            const double nb = (double)std::max<unsigned long long>(hp[7], 1);
            fprintf(stderr,
                    "[bls prof] head_bins %5d periods %6d %s nt %4d lds %6zu | %7.2f ms = %6.2f CU-us per (target, period) | us per "
                    "team: setup %.1f  hist %.1f  pad+red %.1f  prefix %.1f  scan %.1f  out %.1f\n",
                    head_bins, npg, multi ? "multi " : (deep ? "deep  " : "single"), nt, lds, ms, ms * 1e3 * 256.0 / ((double)npg * B),
                    hp[0] / nb * 0.01, hp[1] / nb * 0.01, hp[2] / nb * 0.01, hp[3] / nb * 0.01, hp[4] / nb * 0.01, hp[5] / nb * 0.01);
        }
        g0 = g1;
    }
    stream = caller_stream;
    for (int a = 0; a < n_spread - 1; ++a) {
        LK_HIP_CHECK(hipEventRecord(h->ev_ls_join[a], h->s_ls_aux[a]));
        LK_HIP_CHECK(hipStreamWaitEvent(caller_stream, h->ev_ls_join[a], 0));
    }
    spread_guard.joined = true;
    if (d_prof) LK_HIP_CHECK(hipFree(d_prof));
    if (pe0) {
        (void)hipEventDestroy(pe0);
        (void)hipEventDestroy(pe1);
    }
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
