// flatten.hip — the trend LightCurve.flatten divides by: masked, gap-segmented Savitzky-Golay filter with
// iterative sigma clipping and linear re-interpolation (reference: src/lightkurve/lightcurve.py:996-1063 over
// scipy.signal.savgol_filter, scipy/signal/_savitzky_golay.py:230-357 mode='interp', and
// scipy.interpolate.interp1d(kind='linear', fill_value='extrapolate')).
//
// The curve (N ~ 2e4 doubles) lives in a per-target scratch slab; the loop's phases are separate kernels (see "phase-split
// pipeline" below).  The FIR itself is 2*window flop per cadence per iteration (the one HBM-leaning kernel of the path,
// SURVEY.md §8(d)); for long windows its interior is evaluated in moment form (three prefix sums per tile).  Order
// statistics (nanmedian of flux, of the time steps, of short segments) use the sampled select of block_select.hpp; the
// FIR taps and the edge-fit operators are built once on the host in long double (savgol_design) — the edge polynomial
// refit of scipy's mode='interp' is the linear map y_edge = E x_window with E = V_eval (V^T V)^-1 V^T.
#include <algorithm>
#include <cmath>
#include <vector>

#include "block_select.hpp"
#include "lk_common.hpp"

namespace lk {

// ------------------------------------------------------------------------------------------------ host design
// Solve the small SPD system M a = b (n <= 16) by Gaussian elimination with partial pivoting in long double.
static bool solve_small(std::vector<long double> &M, std::vector<long double> &b, int n) {
    for (int j = 0; j < n; ++j) {
        int p = j;
        for (int i = j + 1; i < n; ++i)
            if (fabsl(M[i * n + j]) > fabsl(M[p * n + j])) p = i;
        if (M[p * n + j] == 0.0L) return false;
        if (p != j) {
            for (int c = 0; c < n; ++c) std::swap(M[j * n + c], M[p * n + c]);
            std::swap(b[j], b[p]);
        }
        for (int i = j + 1; i < n; ++i) {
            const long double f = M[i * n + j] / M[j * n + j];
            for (int c = j; c < n; ++c) M[i * n + c] -= f * M[j * n + c];
            b[i] -= f * b[j];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        long double s = b[i];
        for (int c = i + 1; c < n; ++c) s -= M[i * n + c] * b[c];
        b[i] = s / M[i * n + i];
    }
    return true;
}

// coeffs[w]: FIR taps (scipy savgol_coeffs(w, p): min-norm solution of sum_j c_j x_j^i = delta_i0 with
// x = h..-h; symmetric for deriv=0).  edge[2][half][w]: rows of E for outputs 0..half-1 from the first w samples
// and outputs w-half..w-1 from the last w samples.  Abscissae are scaled to [-1, 1] for conditioning.
static bool savgol_design(int w, int p, std::vector<double> &coeffs, std::vector<double> &edge,
                          long double *tap_poly = nullptr,  // tap_poly[a]: c_j = sum_a tap_poly[a] ((half - j) / half)^a
                          std::vector<double> *edge_minv = nullptr) {  // (p+1)^2: inverse moment matrix of the edge fit
    const int half = w / 2, np1 = p + 1;
    std::vector<long double> z(w);
    const long double hs = half > 0 ? (long double)half : 1.0L;
    for (int j = 0; j < w; ++j) z[j] = (long double)(half - j) / hs;  // x = arange(-half, w-half)[::-1], scaled
    std::vector<long double> M(np1 * np1), rhs(np1, 0.0L);
    for (int a = 0; a < np1; ++a)
        for (int b = 0; b < np1; ++b) {
            long double s = 0.0L;
            for (int j = 0; j < w; ++j) s += powl(z[j], a + b);
            M[a * np1 + b] = s;
        }
    rhs[0] = 1.0L;
    {
        std::vector<long double> Mc = M;
        if (!solve_small(Mc, rhs, np1)) return false;
    }
    if (tap_poly)
        for (int a = 0; a < np1; ++a) tap_poly[a] = rhs[a];
    coeffs.assign(w, 0.0);
    for (int j = 0; j < w; ++j) {
        long double s = 0.0L;
        for (int a = 0; a < np1; ++a) s += rhs[a] * powl(z[j], a);
        coeffs[j] = (double)s;
    }
    // edge operators: polyfit over positions 0..w-1 (scaled u), evaluated at r
    std::vector<long double> u(w);
    const long double c0 = (long double)(w - 1) / 2.0L, sc = c0 > 0 ? c0 : 1.0L;
    for (int j = 0; j < w; ++j) u[j] = ((long double)j - c0) / sc;
    for (int a = 0; a < np1; ++a)
        for (int b = 0; b < np1; ++b) {
            long double s = 0.0L;
            for (int j = 0; j < w; ++j) s += powl(u[j], a + b);
            M[a * np1 + b] = s;
        }
    if (edge_minv) {
        edge_minv->assign((size_t)np1 * np1, 0.0);
        for (int b = 0; b < np1; ++b) {
            std::vector<long double> Mc = M, g(np1, 0.0L);
            g[b] = 1.0L;
            if (!solve_small(Mc, g, np1)) return false;
            for (int a = 0; a < np1; ++a) (*edge_minv)[(size_t)a * np1 + b] = (double)g[a];
        }
    }
    edge.assign((size_t)2 * half * w, 0.0);
    for (int side = 0; side < 2; ++side)
        for (int r = 0; r < half; ++r) {
            const int pos = side == 0 ? r : (w - half + r);
            std::vector<long double> Mc = M, g(np1);
            for (int a = 0; a < np1; ++a) g[a] = powl(u[pos], a);
            if (!solve_small(Mc, g, np1)) return false;  // g = M^-1 v(pos)  (M symmetric)
            for (int j = 0; j < w; ++j) {
                long double s = 0.0L;
                for (int a = 0; a < np1; ++a) s += g[a] * powl(u[j], a);
                edge[((size_t)side * half + r) * w + j] = (double)s;
            }
        }
    return true;
}

int savgol_design_host(int window, int polyorder, double *coeffs, double *edge) {
    LK_REQUIRE(window >= 1 && window % 2 == 1, "window_length must be a positive odd integer");
    LK_REQUIRE(polyorder >= 0 && polyorder < window && polyorder <= 15, "polyorder must be less than window_length");
    LK_REQUIRE(coeffs && edge, "NULL buffer");
    std::vector<double> c, e;
    LK_REQUIRE(savgol_design(window, polyorder, c, e), "singular Savitzky-Golay design");
    std::copy(c.begin(), c.end(), coeffs);
    std::copy(e.begin(), e.end(), edge);
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ device helpers
// Number of k in [k_lo + lane, k_hi) step 64 with pred(k) — eight loads in flight per lane (clamped indices, pinned): one
// load per trip made these counting loops chains of ~40 memory round trips per wave.
template <class Pred>
__device__ __forceinline__ int strided_count64(int k_lo, int k_hi, int lane, Pred pred) {
    int c = 0;
    for (int k0 = k_lo + lane; k0 < k_hi; k0 += 8 * 64) {
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pred(min(k0 + 64 * u, k_hi - 1)) ? 1u : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("" : "+v"(v[u]));
            c += (k0 + 64 * u < k_hi) ? (int)v[u] : 0;
        }
    }
    return c;
}

// Order-preserving compaction of {i < n : pred(i)} into out[] (returns the count, the same in every thread) with coalesced accesses: every wave owns one contiguous strip of the index range and walks it 64
// entries at a time (four groups in flight), positions from ballot prefixes; two barriers.
template <class Pred>
__device__ int strip_compact(int n, Pred pred, int *out, int *sh) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nw = nt >> 6, wv = tid >> 6, lane = tid & 63;
    const int strip = ((n + nw - 1) / nw + 63) & ~63;
    const int k_lo = min(wv * strip, n), k_hi = min(k_lo + strip, n);
    int c = strided_count64(k_lo, k_hi, lane, pred);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    __syncthreads();
    if (lane == 0) sh[wv] = c;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < nw; ++w) {
        if (w < wv) base += sh[w];
        total += sh[w];
    }
    for (int k0 = k_lo; k0 < k_hi; k0 += 256) {
        bool m[4];
        unsigned pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pv[u] = pred(min(k0 + 64 * u + lane, k_hi - 1)) ? 1u : 0u;  // clamped: four loads in flight
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("" : "+v"(pv[u]));
            m[u] = k0 + 64 * u + lane < k_hi && pv[u] != 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long bal = __ballot(m[u]);
            if (m[u]) out[base + __popcll(bal & ((1ull << lane) - 1ull))] = k0 + 64 * u + lane;
            base += __popcll(bal);
        }
    }
    __syncthreads();
    return total;
}

// ================================================================================================ phase-split pipeline
// One light curve = one workgroup PER PHASE KERNEL (the trend phase: T workgroups); state between the kernels lives in the
// per-target slab (compacted times / fluxes, trend, index map, segment starts, masks) plus a 64-byte FlatState.
// LDS plan of the select / trend kernels (dynamic): sh[max(nt, 264)] 64-bit words | fir[FIR_LDS + 2] doubles | shi[nt] ints;
// `fir` holds the candidates of the sampled order statistics (block_select.hpp) or the tile arrays of the trend kernel.
//
// Rounds 1-4 ran all ~75 barrier-separated sweeps of a light curve inside ONE 512-thread workgroup at one register budget
// (128 VGPRs with 79 spilled, two workgroups per CU, an LDS-resident variant for <= 1 900 cadences): 2.39 ms per 1000 x
// 20 000 cadences, bound by memory latency at 16 waves per CU.  The same phases as separate kernels give every phase its own
// register budget and occupancy (the streaming and select phases run 4 workgroups per CU: all 1000 light curves of the
// bench in ONE wave of workgroups), per-phase times straight from a kernel trace, and more than one workgroup per light
// curve where a phase is embarrassingly parallel.  A kernel boundary costs ~2 us; 1 + 4 niters + 1 = 14 of them per call.
// With one wave of workgroups per launch the SLOWEST light curve sets the time of every launch — which is what exposed the
// rare slow routes of the sampled select (a 4096-key sort for ~9 % of the light curves, a bracket miss for ~0.4 %) and led
// to the histogram select and the 4-sigma bracket of block_select.hpp.  Measured (A/B on one box, tools/ab_flatten.sh):
// 2.39 -> 1.77 ms (20 000 cadences, window 401), 0.90 -> 0.59 ms (4500, window 101), 1.05 -> 0.50 ms (3500), and it
// beats the LDS-resident monolith on short light curves too (2000 x 1200: 0.89 -> 0.76 ms), so that kernel is gone.
struct FlatState {
    double dmed_prev, dspacing, rs1, rs2;
    int nm, nm_prev, nseg, removed_any, t_nan, done, want_interp, pad;
};

struct FlatSlab {
    double *tm, *fm, *tr;
    int *idx, *segs;
    uint8_t *mask, *mask1;
};

__device__ __forceinline__ FlatSlab flat_slab(char *scratch, const int64_t *scratch_off, int target, int N) {
    const int Npad = (N + 7) & ~7;
    char *s = scratch + scratch_off[target];
    FlatSlab sl;
    sl.tm = reinterpret_cast<double *>(s);
    sl.fm = sl.tm + Npad;
    sl.tr = sl.fm + Npad;
    sl.idx = reinterpret_cast<int *>(sl.tr + Npad);
    sl.segs = sl.idx + Npad;
    sl.mask = reinterpret_cast<uint8_t *>(sl.segs + Npad + 8);
    sl.mask1 = sl.mask + Npad;
    return sl;
}

constexpr int FLAT_NT = 512;
// waves per SIMD the select / interpolation kernels are compiled for (8 = the 64-VGPR cap that keeps FOUR 512-thread workgroups
// on a CU: 1024 slots, the bench's 1000 light curves in one wave of workgroups)
#ifndef FLAT_INIT_WAVES
#define FLAT_INIT_WAVES 8
#endif
#ifndef FLAT_DTSEG_WAVES
#define FLAT_DTSEG_WAVES 8
#endif
#ifndef FLAT_INTERP_WAVES
#define FLAT_INTERP_WAVES 6
#endif

// ---- phase 0: initial mask (finite & |flux - nanmedian| <= sigma nanstd & ~user_mask), lightcurve.py:1002-1010
__global__ __launch_bounds__(FLAT_NT, FLAT_INIT_WAVES) void flat_init_kernel(const double *__restrict__ flux, const uint8_t *__restrict__ user_mask,
                                                            const int64_t *__restrict__ n_off, double sigma,
                                                            char *__restrict__ scratch, const int64_t *__restrict__ scratch_off,
                                                            FlatState *__restrict__ state, int FIR_LDS, int dbg) {
    // dbg (development builds, LK_FLAT_STOP=100+k): return at stop point k of the sampled select — kernel-time differences
    // between successive stop points are the costs of its phases; -1 = run to the end
    extern __shared__ __attribute__((aligned(16))) unsigned long long dyn_lds[];
    unsigned long long *sh = dyn_lds;
    const int sh_words = max((int)blockDim.x, 264);
    double *fir = reinterpret_cast<double *>(sh + sh_words);
    int *shi = reinterpret_cast<int *>(fir + FIR_LDS + 2);
    double *shd = reinterpret_cast<double *>(sh);
    long long *shl = reinterpret_cast<long long *>(sh);
    const int target = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int64_t lo = n_off[target];
    const int N = (int)(n_off[target + 1] - lo);
    flux += lo;
    if (user_mask) user_mask += lo;
    const FlatSlab sl = flat_slab(scratch, scratch_off, target, N);
    uint8_t *mask = sl.mask;
    if (tid == 0) {
        FlatState st;
        st.dmed_prev = __longlong_as_double(0x7ff8000000000000ll);
        st.dspacing = 0.0;
        st.rs1 = st.rs2 = 0.0;
        st.nm = st.nm_prev = st.nseg = 0;
        st.removed_any = 1;
        st.t_nan = 1;
        st.done = st.want_interp = st.pad = 0;
        state[target] = st;
    }
    auto val = [&](int i) { return flux[i]; };
    auto notnan = [&](int i) { return !isnan(flux[i]); };
    int first = N;
    for (int i = tid; i < N; i += nt)
        if (isfinite(flux[i])) {
            first = i;
            break;
        }
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
    if ((tid & 63) == 0) shi[tid >> 6] = first;
    __syncthreads();
    for (int w = 0; w < (nt >> 6); ++w) first = min(first, shi[w]);
    __syncthreads();
    const double shift = first < N ? flux[first] : 0.0;
    long long c = 0, cinf = 0;
    double part = 0.0, part2 = 0.0;
    strided_pass<8>(N, val, [&](int, double f) {
        if (!isnan(f)) {
            ++c;
            if (isinf(f)) {
                ++cinf;
            } else {
                const double d = f - shift;
                part += d;
                part2 = fma(d, d, part2);
            }
        }
    });
    const long long cnt = block_count_fast(c, shl);
    const long long ninf = block_count_fast(cinf, shl);
    const double s1 = block_sum_fast(part, shd), s2 = block_sum_fast(part2, shd);
    const double sd = ninf > 0 ? __longlong_as_double(0x7ff8000000000000ll)
                               : sqrt(fmax(0.0, (s2 - s1 * s1 / (double)cnt) / (double)cnt));
    __syncthreads();
    if (dbg == 99) return;  // (stop point: statistics done, select not started)
    const double med = block_median_sampled(N, cnt, val, notnan, sh, fir, FIR_LDS, dbg);
    if (dbg >= 0) return;
    strided_pass<8>(N, val, [&](int i, double f) {
        bool m = isfinite(f) && (fabs(f - med) <= sd * sigma);
        if (user_mask && user_mask[i]) m = false;
        mask[i] = m ? 1 : 0;
    });
}

// ---- phase 1 of an iteration: order-preserving compaction of the kept cadences fused with the gather of their times / fluxes
__global__ __launch_bounds__(FLAT_NT) void flat_compact_kernel(const double *__restrict__ t, const double *__restrict__ flux,
                                                               const int64_t *__restrict__ n_off, char *__restrict__ scratch,
                                                               const int64_t *__restrict__ scratch_off,
                                                               FlatState *__restrict__ state, double *__restrict__ trend, int it) {
    __shared__ int shi[FLAT_NT / 64];
    const int target = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    if (state[target].done) return;
    const int64_t lo = n_off[target];
    const int N = (int)(n_off[target + 1] - lo);
    t += lo;
    flux += lo;
    trend += lo;
    const FlatSlab sl = flat_slab(scratch, scratch_off, target, N);
    const uint8_t *mask = sl.mask;
    int *idx = sl.idx;
    double *tm = sl.tm, *fm = sl.fm;
    const int nw = nt >> 6, wv = tid >> 6, lane = tid & 63;
    const int strip = ((N + nw - 1) / nw + 63) & ~63;
    const int k_lo = min(wv * strip, N), k_hi = min(k_lo + strip, N);
    int c = strided_count64(k_lo, k_hi, lane, [&](int k) { return mask[k] != 0; });
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) shi[wv] = c;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < nw; ++w) {
        if (w < wv) base += shi[w];
        total += shi[w];
    }
    const int nm = total;
    bool saw_nan = false;
    for (int k0 = k_lo; k0 < k_hi; k0 += 256) {
        bool m[4];
        unsigned mk[4];
        double tv[4], fv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 64 * u + lane, kc = min(k, k_hi - 1);
            mk[u] = mask[kc];
            tv[u] = t[kc];
            fv[u] = flux[kc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("" : "+v"(mk[u]), "+v"(tv[u]), "+v"(fv[u]));
            m[u] = k0 + 64 * u + lane < k_hi && mk[u] != 0;
            saw_nan |= m[u] && isnan(tv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long bal = __ballot(m[u]);
            if (m[u]) {
                const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                idx[pos] = k0 + 64 * u + lane;
                tm[pos] = tv[u];
                fm[pos] = fv[u];
            }
            base += __popcll(bal);
        }
    }
    const int any_nan = __syncthreads_or(saw_nan ? 1 : 0);
    if (nm == 0) {
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        for (int i = tid; i < N; i += nt) trend[i] = qnan;
    }
    if (tid == 0) {
        state[target].nm = nm;
        if (it == 0) state[target].t_nan = any_nan;
        if (nm == 0) state[target].done = 1;  // (want_interp stays 0: the trend is all NaN)
    }
}

// ---- phase 2: gap segmentation — cut where dt > break_tol * nanmedian(dt), lightcurve.py:1022-1027.
// The cuts ride on the median's own pass over the time steps: every step larger than break_tol x (the bracket's lower pivot, a
// lower bound of the median) is noted as a CANDIDATE cut in a small LDS list; once the median is known the few candidates
// are re-tested against the exact threshold and ranked by position.  Two more sweeps over the times (count + compaction of
// the cut predicate) only when that list overflows or no bound was available.
constexpr int FLAT_CUT_CAP = 384;

__global__ __launch_bounds__(FLAT_NT, FLAT_DTSEG_WAVES) void flat_dtseg_kernel(const int64_t *__restrict__ n_off, double break_tol,
                                                             char *__restrict__ scratch, const int64_t *__restrict__ scratch_off,
                                                             FlatState *__restrict__ state, int FIR_LDS, int it, int near_on,
                                                             int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long dyn_lds[];
    __shared__ int cut_n;
    __shared__ int cut_i[FLAT_CUT_CAP];
    unsigned long long *sh = dyn_lds;
    const int sh_words = max((int)blockDim.x, 264);
    double *fir = reinterpret_cast<double *>(sh + sh_words);
    int *shi = reinterpret_cast<int *>(fir + FIR_LDS + 2);
    long long *shl = reinterpret_cast<long long *>(sh);
    const int target = blockIdx.x, tid = threadIdx.x;
    const FlatState st = state[target];
    if (st.done) return;
    const int N = (int)(n_off[target + 1] - n_off[target]);
    const FlatSlab sl = flat_slab(scratch, scratch_off, target, N);
    const double *tm = sl.tm;
    int *segs = sl.segs;
    const int nm = st.nm;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    double dmed = qnan, dspacing = st.dspacing;
    bool have_cuts = false;  // cut_i[0..cut_n) holds every step that can exceed the threshold (workgroup-uniform)
    if (nm >= 2) {
        const bool t_nan = st.t_nan != 0;
        auto dval = [&](int i) { return tm[i + 1] - tm[i]; };
        auto dkeep = [&](int i) { return !t_nan || !isnan(tm[i + 1] - tm[i]); };  // (no NaN time: nothing to re-read)
        // step i (between kept cadences i and i + 1) cuts in front of cadence i + 1 if it exceeds break_tol * median; with
        // break_tol >= 0 and a bound lo <= median, break_tol * lo <= break_tol * median (rounding is monotone)
        const bool bound_ok = break_tol >= 0.0;
        auto note = [&](int i, double d, double lo) {
            if (d > break_tol * lo || !(lo >= 0.0)) {
                const int slot = atomicAdd(&cut_n, 1);
                if (slot < FLAT_CUT_CAP) cut_i[slot] = i + 1;
            }
        };
        long long cnt = (long long)(nm - 1);
        if (st.t_nan) {
            long long c = 0;
            strided_pass<8>(nm - 1, dval, [&](int, double d) { c += isnan(d) ? 0 : 1; });
            cnt = block_count_fast(c, shl);
            __syncthreads();
        }
        if (tid == 0) cut_n = 0;
        __syncthreads();
        bool near_ok = false;
        if (it > 0 && near_on && st.dspacing > 0.0 && st.nm_prev >= nm) {
            const double width = st.dspacing * (3.0 * (double)(st.nm_prev - nm) + 96.0);
            dmed = block_median_near(nm - 1, cnt, dval, dkeep, st.dmed_prev, width, sh, fir, FIR_LDS, &near_ok, note);
            have_cuts = near_ok && bound_ok;
        }
        if (!near_ok) {
            __syncthreads();
            if (tid == 0) cut_n = 0;  // (candidates noted against the failed guess are void)
            __syncthreads();
            bool ran = false;
            dmed = block_median_sampled(nm - 1, cnt, dval, dkeep, sh, fir, FIR_LDS, dbg < 8 ? dbg : -1, &dspacing, note, &ran);
            if (dbg >= 0 && dbg <= 8) return;  // (8: the whole select, fall-back routes included)
            have_cuts = ran && bound_ok;
        }
        __syncthreads();
        if (have_cuts && cut_n > FLAT_CUT_CAP) have_cuts = false;
    }
    const double thr = break_tol * dmed;  // NaN break_tol => every comparison false => no cuts
    int nseg;
    if (have_cuts) {
        // (a NaN step is never a cut: dkeep excluded it from the pass, and `NaN > thr` is false in the reference too)
        const int nc = cut_n;
        int mine = -1;
        if (tid < nc) {
            const int i = cut_i[tid];
            if ((tm[i] - tm[i - 1]) > thr) mine = i;
        }
        __syncthreads();
        if (tid < nc) cut_i[tid] = mine;
        __syncthreads();
        int rank = 0, total = 0;
        for (int j = 0; j < nc; ++j) {
            const int v = cut_i[j];
            if (v >= 0) {
                ++total;
                if (v < mine) ++rank;
            }
        }
        if (tid == 0) segs[0] = 0;
        if (mine >= 0) segs[1 + rank] = mine;
        nseg = 1 + total;
    } else {
        nseg = strip_compact(
            nm, [&](int i) { return i == 0 || (tm[i] - tm[i - 1]) > thr; }, segs, shi);
    }
    if (tid == 0) {
        state[target].nseg = nseg;
        if (nm >= 2) {
            state[target].dmed_prev = dmed;
            state[target].dspacing = dspacing;
            state[target].nm_prev = nm;
        }
    }
}

// Inclusive scan of a double over the 64 lanes on the DPP network (no LDS crossbar): Hillis-Steele inside each row of 16 lanes
// (row_shr 1, 2, 4, 8; a lane without a source adds 0), then lane 15 of a row broadcast to the next row (rows 1 and 3) and
// lane 31 to rows 2 and 3 — the GFX9 scan idiom.  Three of these per tile replaced 6 x 3 ds_bpermute round trips.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double flat_dpp0(double x) {   // the DPP-selected lane's x, 0.0 where there is none / row masked
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double flat_wave_scan(double x) {
    x += flat_dpp0<0x111, 0xF>(x);   // row_shr:1
    x += flat_dpp0<0x112, 0xF>(x);   // row_shr:2
    x += flat_dpp0<0x114, 0xF>(x);   // row_shr:4
    x += flat_dpp0<0x118, 0xF>(x);   // row_shr:8
    x += flat_dpp0<0x142, 0xA>(x);   // row_bcast:15 into rows 1 and 3
    x += flat_dpp0<0x143, 0xC>(x);   // row_bcast:31 into rows 2 and 3
    return x;
}

// median of a short segment (fewer cadences than the window): the rare path of the trend kernel, kept out of line so that its
// select machinery does not set the register budget of the streaming paths
__device__ __noinline__ double flat_segment_median(const double *x, int len, unsigned long long *sh, double *cand, int cap) {
    auto val = [&](int i) { return x[i]; };
    auto keep = [&](int) { return true; };  // masked flux is finite
    return block_median_sampled(len, (long long)len, val, keep, sh, cand, cap);
}

// ---- phase 3: per segment the median (short ones) or the Savitzky-Golay trend, lightcurve.py:1030-1046.  Every kept cadence
// gets its trend exactly once; its residual goes into the two running sums the clip needs.  T = gridDim.y workgroups share a
// light curve: the work items — one per short segment (its median), one per tile of a long segment's interior, one per long
// segment's pair of edges — are dealt round-robin; a workgroup's residual sums go to rs_part[target * T + y] and the clip adds
// the T partials in order (deterministic).  QUAD: the moment form of the interior (taps quadratic in the offset) instead of
// the tap-by-tap FIR — two kernels so that each carries only its own registers.
template <bool QUAD>
__global__ __launch_bounds__(FLAT_NT, QUAD ? 6 : 4) void flat_trend_kernel(const int64_t *__restrict__ n_off, int window, int polyorder,
                                                             double break_tol, const double *__restrict__ coeffs,
                                                             const double *__restrict__ edge, char *__restrict__ scratch,
                                                             const int64_t *__restrict__ scratch_off,
                                                             const FlatState *__restrict__ state, int FIR_LDS, double quad_a,
                                                             double quad_b, const double *__restrict__ edge_minv,
                                                             double2 *__restrict__ rs_part, int dbg) {
    // dbg (development builds, LK_FLAT_STOP=300+k; results wrong, launch time = the cost of what is left — read it off the
    // FIRST iteration's launch in a --timeline, the later ones run on the garbage this leaves): 0 return after the state /
    // segment reads, 1 interior tiles only, 2 edges and short segments only, 3 tiles without their output phase, 4 no trend
    // stores, 6 / 7 / 8 tiles up to the local moments / the wave scans / the barriers around the wave totals
    extern __shared__ __attribute__((aligned(16))) unsigned long long dyn_lds[];
    unsigned long long *sh = dyn_lds;
    const int sh_words = max((int)blockDim.x, 264);
    double *fir = reinterpret_cast<double *>(sh + sh_words);
    double *shd = reinterpret_cast<double *>(sh);
    const int target = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const FlatState st = state[target];
    if (st.done) return;
    const int N = (int)(n_off[target + 1] - n_off[target]);
    const FlatSlab sl = flat_slab(scratch, scratch_off, target, N);
    const double *fm = sl.fm;
    double *tr = sl.tr;
    const int *segs = sl.segs;
    const int nm = st.nm, nseg = st.nseg;
    const int half = window / 2;
    const int T = gridDim.y, y = blockIdx.y;
    if (dbg == 0) {
        if (nseg > 0 && segs[nseg - 1] < 0) tr[0] = 0.0;   // (keeps the reads alive)
        return;
    }
    int item = 0;  // running work-item number (workgroup-uniform): this workgroup takes those with item % T == y
    double rs1 = 0.0, rs2 = 0.0;
    auto put = [&](int i, double v) {
        tr[i] = v;
        const double r = fm[i] - v;
        rs1 += r;
        rs2 = fma(r, r, rs2);
    };
    for (int sg = 0; sg < nseg; ++sg) {
        const int l = segs[sg], h = (sg + 1 < nseg) ? segs[sg + 1] : nm;
        const int len = h - l;
        if (window > len || (double)len < break_tol) {
            if ((item++ % T) != y || dbg == 1 || dbg == 3) continue;
            __syncthreads();
            const double med = flat_segment_median(fm + l, len, sh, fir, FIR_LDS);
            for (int i = l + tid; i < h; i += nt) put(i, med);
        } else {
            // interior: correlate with the taps (window fully inside the segment)
            const int o_lo = l + half, o_hi = h - half;  // outputs [o_lo, o_hi)
            // Tiles of TO outputs; the tile's TO + window - 1 inputs are staged in LDS once (coalesced) as FP = 8
            // interleaved sub-arrays (element e at [e % 8][e / 8], sub-array stride S = 4 mod 32: the staging stores
            // and the reads below are both bank-conflict free).  A thread produces 8 NEIGHBOURING outputs from a
            // sliding 8-value register window: per tap one 8-B LDS read (the value entering the window — lanes read
            // consecutive addresses of one sub-array) and 8 FMAs, so the fp64 pipe is the limit, not the LDS (the
            // 2-outputs-per-16-B-read version this replaces kept the LDS port 100 % busy at 21 % of the FMA peak).
            // Every output accumulates its taps in the order of scipy's correlate1d, as before.
            constexpr int FP = 8;
            const int S = (((FIR_LDS / FP) - 4) / 32) * 32 + 4;
            const int TO = S >= 36 ? ((FP * S - FP - (window - 1)) / FP) * FP : 0;
            // polyorder <= 3: the taps are a quadratic in the offset, c_k = a + b k^2, so an output is
            // a S0 + b S2 with the window moments S0 = sum y, S2 = sum k^2 y — O(1) per output from three prefix sums
            // (y, u y, u^2 y; u = position relative to the tile centre) instead of `window` FMAs.  A tile spans at
            // most 4 windows, which keeps the cancellation in S2 = W2 - 2 v W1 + v^2 W0 to a few bits: the result is
            // within ~1e-14 of the tap-by-tap sum (tests state 1e-10).  Used for long windows only (quad_b != 0).
            const int QCAP = QUAD ? ((FIR_LDS / 4 - 1) * 32) / 33 : FIR_LDS / 3;  // (QUAD: 4 arrays of QCAP + QCAP / 32 + 1 doubles)
            const int QNI = min(QCAP, 4 * window), QTO = QNI - (window - 1);
            if (QUAD && quad_b != 0.0 && QTO >= 64) {
                // p0..p2: the tile's three prefix sums; xs: its inputs (the residual of an output needs its own flux: from LDS,
                // not a second global round trip).  The inputs of this workgroup's NEXT tile are fetched (clamped, unconditional)
                // before the scans of the current one: a tile costs one memory latency, hidden behind the previous tile.
                // A thread writes the prefixes of 4 consecutive inputs: unpadded, lanes l and l + 8 would hit the same LDS banks
                // (8-way conflicts on 16 stores per tile); one pad double per 32 entries (PX) spreads them over all banks.
                auto PX = [](int e) { return e + (e >> 5); };
                const int QS = QCAP + (QCAP >> 5) + 1;
                double *p0 = fir, *p1 = fir + QS, *p2 = fir + 2 * QS, *xs = fir + 3 * QS;
                const int lane = tid & 63, wv = tid >> 6, nwv = nt >> 6;
                const int ntile = (o_hi - o_lo + QTO - 1) / QTO;
                int j = ((y - item) % T + T) % T;  // the first tile of this segment that is this workgroup's
                item += ntile;
                const int CHF = (QNI + nt - 1) / nt;  // inputs per thread: every tile uses chunks of CHF (<= 4 at 512 threads)
                double xn[4] = {0.0, 0.0, 0.0, 0.0};
                auto fetch = [&](int jj) {
                    const int o0n = o_lo + jj * QTO, nin = min(QTO, o_hi - o0n) + window - 1;
                    const double *xq = fm + (o0n - half);
                    const int e0n = min(nin, tid * CHF);
#pragma unroll
                    for (int q = 0; q < 4; ++q) xn[q] = xq[min(e0n + q, nin - 1)];
                };
                if (CHF <= 4 && j < ntile) fetch(j);
                if (dbg == 2) j = ntile;
                for (; j < ntile; j += T) {
                    const int o0 = o_lo + j * QTO;
                    const int no = min(QTO, o_hi - o0), ni = no + window - 1;
                    const double uc = 0.5 * (double)(ni - 1);
                    const double *x = fm + (o0 - half);
                    const int CH = CHF;
                    const int e0 = min(ni, tid * CH), e1 = min(ni, e0 + CH);
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
                    double xr[4] = {0.0, 0.0, 0.0, 0.0};  // CH <= 4: the thread's inputs stay in registers for the second sweep
                    if (CH <= 4) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            asm volatile("" : "+v"(xn[q]));
                            xr[q] = xn[q];
                        }
                        if (j + T < ntile) fetch(j + T);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (q < CH && e0 + q < e1) {
                                const double yv = xr[q], u = (double)(e0 + q) - uc;
                                s0 += yv;
                                s1 = fma(u, yv, s1);
                                s2 = fma(u * u, yv, s2);
                            }
                    } else {
                        for (int e = e0; e < e1; ++e) {
                            const double yv = x[e], u = (double)e - uc;
                            s0 += yv;
                            s1 = fma(u, yv, s1);
                            s2 = fma(u * u, yv, s2);
                        }
                    }
                    if (dbg == 6) {   // (ablation: loads + local moments only)
                        rs1 += s0 + s1 + s2;
                        continue;
                    }
                    // inclusive scan over the wave, then over the waves
                    const double i0 = flat_wave_scan(s0), i1 = flat_wave_scan(s1), i2 = flat_wave_scan(s2);
                    if (dbg == 7) {   // (ablation: + the wave scans, no barrier)
                        rs1 += i0 + i1 + i2;
                        continue;
                    }
                    __syncthreads();  // shd and the prefix arrays of the previous tile are free
                    if (lane == 63) {
                        shd[wv * 3 + 0] = i0;
                        shd[wv * 3 + 1] = i1;
                        shd[wv * 3 + 2] = i2;
                    }
                    __syncthreads();
                    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
                    if (dbg == 8) {   // (ablation: + the two barriers around the wave totals, no carries / LDS stores)
                        rs1 += i0;
                        continue;
                    }
                    for (int w2 = 0; w2 < wv && w2 < nwv; ++w2) {
                        r0 += shd[w2 * 3 + 0];
                        r1 += shd[w2 * 3 + 1];
                        r2 += shd[w2 * 3 + 2];
                    }
                    r0 += flat_dpp0<0x138, 0xF>(i0);   // wave_shr:1 — the previous lane's inclusive sum, 0 in lane 0
                    r1 += flat_dpp0<0x138, 0xF>(i1);
                    r2 += flat_dpp0<0x138, 0xF>(i2);
                    if (CH <= 4) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (q < CH && e0 + q < e1) {
                                const int e = e0 + q;
                                const double yv = xr[q], u = (double)e - uc;
                                r0 += yv;
                                r1 = fma(u, yv, r1);
                                r2 = fma(u * u, yv, r2);
                                p0[PX(e)] = r0;
                                p1[PX(e)] = r1;
                                p2[PX(e)] = r2;
                                xs[PX(e)] = yv;
                            }
                    } else {
                        for (int e = e0; e < e1; ++e) {
                            const double yv = x[e], u = (double)e - uc;
                            r0 += yv;
                            r1 = fma(u, yv, r1);
                            r2 = fma(u * u, yv, r2);
                            p0[PX(e)] = r0;
                            p1[PX(e)] = r1;
                            p2[PX(e)] = r2;
                            xs[PX(e)] = yv;
                        }
                    }
                    __syncthreads();
                    if (dbg == 3) continue;
                    for (int q = tid; q < no; q += nt) {
                        const int hi = q + window - 1;
                        double w0 = p0[PX(hi)], w1 = p1[PX(hi)], w2 = p2[PX(hi)];
                        if (q > 0) {
                            w0 -= p0[PX(q - 1)];
                            w1 -= p1[PX(q - 1)];
                            w2 -= p2[PX(q - 1)];
                        }
                        const double v = (double)(q + half) - uc;
                        const double m2 = fma(v, fma(v, w0, -2.0 * w1), w2);  // sum (u - v)^2 y
                        const double tv = fma(quad_b, m2, quad_a * w0);
                        if (dbg != 4) tr[o0 + q] = tv;
                        const double r = xs[PX(q + half)] - tv;
                        rs1 += r;
                        rs2 = fma(r, r, rs2);
                    }
                }
            } else if (!QUAD && TO >= FP) {
                for (int o0 = o_lo; o0 < o_hi; o0 += TO) {
                    if ((item++ % T) != y) continue;
                    const int no = min(TO, o_hi - o0), ni = no + window - 1;
                    __syncthreads();
                    for (int e = tid; e < ni + FP; e += nt)   // FP look-ahead slots past the end, zero filled
                        fir[(e % FP) * S + e / FP] = e < ni ? fm[o0 - half + e] : 0.0;
                    __syncthreads();
                    for (int q = tid; FP * q < no; q += nt) {
                        double win[FP], acc[FP];
#pragma unroll
                        for (int r = 0; r < FP; ++r) {
                            win[r] = fir[r * S + q];  // x[8 q + r]
                            acc[r] = 0.0;
                        }
                        int jb = 0;
                        for (; jb + FP <= window; jb += FP) {
                            const int nxt = q + 1 + jb / FP;
#pragma unroll
                            for (int jj = 0; jj < FP; ++jj) {
                                const double cj = coeffs[jb + jj];
#pragma unroll
                                for (int r = 0; r < FP; ++r) acc[r] = fma(cj, win[(jj + r) & (FP - 1)], acc[r]);
                                win[jj] = fir[jj * S + nxt];  // x[8 q + jb + jj + 8] replaces x[8 q + jb + jj]
                            }
                        }
                        const int nxt = q + 1 + jb / FP;
#pragma unroll
                        for (int jj = 0; jj < FP - 1; ++jj) {  // the window % 8 last taps
                            if (jb + jj < window) {
                                const double cj = coeffs[jb + jj];
#pragma unroll
                                for (int r = 0; r < FP; ++r) acc[r] = fma(cj, win[(jj + r) & (FP - 1)], acc[r]);
                                win[jj] = fir[jj * S + nxt];
                            }
                        }
#pragma unroll
                        for (int r = 0; r < FP; ++r)
                            if (FP * q + r < no) put(o0 + FP * q + r, acc[r]);
                    }
                }
            } else if ((item++ % T) == y) {
                for (int i = o_lo + tid; i < o_hi; i += nt) {
                    const double *x = fm + (i - half);
                    double acc = 0.0;
                    for (int j = 0; j < window; ++j) acc = fma(coeffs[j], x[j], acc);
                    put(i, acc);
                }
            }
            // edges: polynomial refit of the first / last `window` samples (mode='interp').  The two windows are
            // staged in LDS; thread (side, r) streams row r of the operator (stored transposed, [side][tap][row]:
            // lanes read neighbouring rows) with 8 loads in flight — the plain tap loop was a chain of ~400
            // dependent L2 round trips and had become the longest part of the segment.
            if ((item++ % T) != y || dbg == 1 || dbg == 3) continue;  // the pair of edges of this segment: one work item
            __syncthreads();
            for (int e = tid; e < 2 * window; e += nt)
                fir[e] = e < window ? fm[l + e] : fm[h - window + (e - window)];
            __syncthreads();
            const int np1 = polyorder + 1;
            if (edge_minv && 2 * window + 2 * np1 <= FIR_LDS) {
                // The edge outputs are the least-squares polynomial of the side's `window` samples evaluated at the
                // output's position: p + 1 moments sum_j u_j^b x_j per side (one wave per moment), beta = M^-1 m,
                // then a Horner evaluation per output — O(window p) per segment instead of the half x window
                // operator rows (401 dependent FMAs per output on L2-resident rows: after the moment-form interior
                // this was most of the segment).  Abscissae scaled to [-1, 1] as on the host; polyorder <= 5.
                const double c0 = 0.5 * (double)(window - 1), sc = c0 > 0.0 ? c0 : 1.0;
                double *mom = fir + 2 * window;
                const int lane = tid & 63, wv = tid >> 6, nwv = nt >> 6;
                for (int pair = wv; pair < 2 * np1; pair += nwv) {
                    const int side = pair / np1, b = pair - side * np1;
                    const double *x = fir + side * window;
                    double sm = 0.0;
                    for (int j = lane; j < window; j += 64) {
                        const double u = ((double)j - c0) / sc;
                        double ub = 1.0;
                        for (int q = 0; q < b; ++q) ub *= u;
                        sm = fma(ub, x[j], sm);
                    }
                    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
                    if (lane == 0) mom[pair] = sm;
                }
                __syncthreads();
                for (int e = tid; e < 2 * half; e += nt) {
                    const int side = e >= half, r = e - side * half;
                    const int pos = side ? (window - half + r) : r;
                    const double u = ((double)pos - c0) / sc;
                    double acc = 0.0;
                    for (int a = np1 - 1; a >= 0; --a) {
                        double beta = 0.0;
                        for (int b = 0; b < np1; ++b) beta = fma(edge_minv[a * np1 + b], mom[side * np1 + b], beta);
                        acc = fma(acc, u, beta);
                    }
                    put(side ? (h - half + r) : (l + r), acc);
                }
            } else {
            for (int e = tid; e < 2 * half; e += nt) {
                const int side = e >= half, r = e - side * half;
                const double *x = fir + side * window;
                const double *E = edge + (size_t)side * half * window + r;
                double acc = 0.0;
                int j = 0;
                for (; j + 8 <= window; j += 8) {
                    double ev[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) ev[u] = E[(size_t)(j + u) * half];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = fma(ev[u], x[j + u], acc);
                }
                for (; j < window; ++j) acc = fma(E[(size_t)j * half], x[j], acc);
                put(side ? (h - half + r) : (l + r), acc);
            }
            }
        }
        __syncthreads();
    }
    const double s1 = block_sum_fast(rs1, shd), s2 = block_sum_fast(rs2, shd);
    if (tid == 0) rs_part[(size_t)target * T + y] = make_double2(s1, s2);
}


// ---- phase 4: clip |flux - trend| < sigma nanstd(flux - trend) + 1e-14 and mask[mask] &= mask1, lightcurve.py:1049-1063.
// A light curve whose clip removed nothing (or whose last iteration this is) is frozen: done = 1, want_interp = 1.
__global__ __launch_bounds__(FLAT_NT) void flat_clip_kernel(const int64_t *__restrict__ n_off, double sigma,
                                                            char *__restrict__ scratch, const int64_t *__restrict__ scratch_off,
                                                            FlatState *__restrict__ state, int last,
                                                            const double2 *__restrict__ rs_part, int T) {
    const int target = blockIdx.x, tid = threadIdx.x;
    const FlatState st = state[target];
    if (st.done) return;
    const int N = (int)(n_off[target + 1] - n_off[target]);
    const FlatSlab sl = flat_slab(scratch, scratch_off, target, N);
    const double *fm = sl.fm, *tr = sl.tr;
    const int *idx = sl.idx;
    uint8_t *mask = sl.mask, *mask1 = sl.mask1;
    const int nm = st.nm;
    auto resid = [&](int i) { return fm[i] - tr[i]; };
    double s1 = 0.0, s2 = 0.0;  // the trend workgroups' residual sums, in workgroup order
    for (int y = 0; y < T; ++y) {
        const double2 pr = rs_part[(size_t)target * T + y];
        s1 += pr.x;
        s2 += pr.y;
    }
    const double sd = sqrt(fmax(0.0, (s2 - s1 * s1 / (double)nm) / (double)nm));
    const double lim = sd * sigma + 1e-14;
    int removed = 0;
    strided_pass<8>(nm, resid, [&](int i, double r) {
        const bool keepit = fabs(r) < lim;
        mask1[i] = keepit ? 1 : 0;
        if (!keepit) {
            if (!last) mask[idx[i]] = 0;  // (the last iteration's survivors stay in mask1: flat_interp_kernel reads both)
            removed = 1;
        }
    });
    const int removed_any = __syncthreads_or(removed);
    if (tid == 0) {
        state[target].removed_any = removed_any;
        if (last || !removed_any) {
            state[target].done = 1;
            state[target].want_interp = 1;
        }
    }
}

// ---- phase 5 (once, after the loop): linear interpolation / extrapolation of the kept trend onto every cadence,
// lightcurve.py:1053-1058, from the frozen state of the light curve's last iteration; and the final mask.
// The knots — the kept cadences that survived the last clip — are NOT compacted into arrays of their own: `mask` still marks
// the kept cadences of the frozen iteration (the last clip leaves it alone), i.e. the compacted arrays tm / tr, and `mask1`
// marks the survivors among those.  For cadence c with x = t[c], p = the first compacted entry with time >= x (the number of
// kept cadences before c, less any that share c's time); np.searchsorted(knots, x, 'left') names the first KNOT at or after p
// as the upper end of the interpolation interval and the knot before it as the lower end — p itself and p - 1 unless an entry
// was clipped there, so the six values an output needs (two flags, two times, two trends) are requested in ONE round, from
// arrays neighbouring lanes share.  Clipped entries at p or p - 1, equal times and the two ends (interp1d's clamp of the
// interval to the first / last pair of knots = extrapolation) take a scalar search.  This dropped the knot gather of rounds
// 1-5a (17 B read + 16 B written per kept cadence, and two dependent rounds of gathers from it): 277 -> ~150 us per launch.
__global__ __launch_bounds__(FLAT_NT, FLAT_INTERP_WAVES) void flat_interp_kernel(const double *__restrict__ t, const int64_t *__restrict__ n_off,
                                                              char *__restrict__ scratch, const int64_t *__restrict__ scratch_off,
                                                              const FlatState *__restrict__ state, double *__restrict__ trend,
                                                              uint8_t *__restrict__ final_mask) {
    __shared__ int shi[FLAT_NT / 64], shk[FLAT_NT / 64];
    __shared__ int ends[4];  // first, second, second-last, last knot (indices into the compacted arrays)
    const int target = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const FlatState st = state[target];
    const int64_t lo = n_off[target];
    const int N = (int)(n_off[target + 1] - lo);
    t += lo;
    trend += lo;
    const FlatSlab sl = flat_slab(scratch, scratch_off, target, N);
    const uint8_t *mask = sl.mask, *mask1 = sl.mask1;
    if (final_mask) final_mask += lo;
    if (!st.want_interp) {  // (no kept cadence: the trend is all NaN already; nothing survived)
        if (final_mask)
            for (int i = tid; i < N; i += nt) final_mask[i] = 0;
        return;
    }
    const double *tm = sl.tm, *tr = sl.tr;
    const int nm = st.nm;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const int nw = nt >> 6, wv = tid >> 6, lane = tid & 63;
    // every wave owns a contiguous strip of cadences: kept cadences per strip (-> the compacted index a strip starts at), and
    // the number of knots (strips of the compacted arrays)
    const int strip = ((N + nw - 1) / nw + 63) & ~63;
    const int k_lo = min(wv * strip, N), k_hi = min(k_lo + strip, N);
    int c = strided_count64(k_lo, k_hi, lane, [&](int k) { return mask[k] != 0; });
    const int mstrip = ((nm + nw - 1) / nw + 63) & ~63;
    const int m_lo = min(wv * mstrip, nm), m_hi = min(m_lo + mstrip, nm);
    int ck = strided_count64(m_lo, m_hi, lane, [&](int k) { return mask1[k] != 0; });
    for (int o = 32; o > 0; o >>= 1) {
        c += __shfl_xor(c, o);
        ck += __shfl_xor(ck, o);
    }
    if (lane == 0) {
        shi[wv] = c;
        shk[wv] = ck;
    }
    if (tid == 0) {
        int f1 = 0;
        while (f1 < nm && !mask1[f1]) ++f1;
        int f2 = f1 + 1;
        while (f2 < nm && !mask1[f2]) ++f2;
        int l1 = nm - 1;
        while (l1 >= 0 && !mask1[l1]) --l1;
        int l2 = l1 - 1;
        while (l2 >= 0 && !mask1[l2]) --l2;
        ends[0] = f1;
        ends[1] = f2;
        ends[2] = l2;
        ends[3] = l1;
    }
    __syncthreads();
    int base = 0, n2 = 0;
    for (int w = 0; w < nw; ++w) {
        if (w < wv) base += shi[w];
        n2 += shk[w];
    }
    if (n2 < 2) {  // (interp1d needs two knots: the trend is all NaN; the survivors are still reported)
        for (int i = tid; i < N; i += nt) trend[i] = qnan;
        if (final_mask && tid == 0) {
            int kept = 0;
            for (int i = 0; i < N; ++i) final_mask[i] = (mask[i] && mask1[kept++]) ? 1 : 0;
        }
        return;
    }
    const int F1 = ends[0], F2 = ends[1], L2 = ends[2], L1 = ends[3];
    for (int k0 = k_lo; k0 < k_hi; k0 += 256) {
        // four 64-cadence groups in flight; per group: mask / time -> compacted index p -> one round of neighbour loads
        bool in[4], kf[4];
        double xn[4];
        unsigned mk[4];
        int p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 64 * u + lane, kc = min(k, k_hi - 1);  // clamped, unconditional, pinned loads
            in[u] = k < k_hi;
            mk[u] = mask[kc];
            xn[u] = t[kc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("" : "+v"(mk[u]), "+v"(xn[u]));
            kf[u] = in[u] && mk[u] != 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long bal = __ballot(kf[u]);
            p[u] = base + __popcll(bal & ((1ull << lane) - 1ull));
            base += __popcll(bal);
        }
        double xa[4], xb[4], ya[4], yb[4];
        unsigned fa[4], fb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pa = min(max(p[u] - 1, 0), nm - 1), pb = min(p[u], nm - 1);
            xa[u] = tm[pa];
            xb[u] = tm[pb];
            ya[u] = tr[pa];
            yb[u] = tr[pb];
            fa[u] = mask1[pa];
            fb[u] = mask1[pb];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("" : "+v"(xa[u]), "+v"(xb[u]), "+v"(ya[u]), "+v"(yb[u]), "+v"(fa[u]), "+v"(fb[u]));
            if (!in[u]) continue;
            const int k = k0 + 64 * u + lane;
            if (final_mask) final_mask[k] = (kf[u] && fb[u] != 0) ? 1 : 0;  // (a kept cadence IS entry p: fb is its own flag)
            const double x = xn[u];
            double x0 = xa[u], x1 = xb[u], y0 = ya[u], y1 = yb[u];
            // the common case: entries p - 1 and p exist, both are knots, p is neither the first knot nor past the last, and
            // the entry before p is strictly earlier (no equal times)
            const bool plain = p[u] >= 1 && p[u] < nm && fa[u] != 0 && fb[u] != 0 && xa[u] < x;
            if (!plain) {
                int q = min(p[u], nm);
                while (q > 0 && tm[q - 1] >= x) --q;  // equal times: those entries are not "< x"
                int hi_i = q;
                while (hi_i < nm && !mask1[hi_i]) ++hi_i;  // the first knot at or after q
                int lo_i;
                if (hi_i > L1) {          // no knot at or after q: the last pair (extrapolation)
                    hi_i = L1;
                    lo_i = L2;
                } else if (hi_i == F1) {  // no knot before q: the first pair
                    hi_i = F2;
                    lo_i = F1;
                } else {
                    lo_i = hi_i - 1;
                    while (!mask1[lo_i]) --lo_i;  // (a knot exists below: hi_i > F1)
                }
                x0 = tm[lo_i];
                x1 = tm[hi_i];
                y0 = tr[lo_i];
                y1 = tr[hi_i];
            }
            const double slope = (y1 - y0) / (x1 - x0);
            trend[k] = isnan(x) ? qnan : slope * (x - x0) + y0;
        }
    }
}

int flatten_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux,
                   const uint8_t *user_mask, int window, int polyorder, double break_tol, int niters, double sigma,
                   double *trend, uint8_t *final_mask, hipStream_t stream) {
    LK_REQUIRE(B >= 0 && n_off_host != nullptr, "bad batch description");
    if (B == 0) return LK_OK;
    LK_REQUIRE(t && flux && trend, "NULL buffer");
    LK_REQUIRE(window >= 1, "window_length must be a positive integer");
    LK_REQUIRE(window % 2 == 1, "window_length must be odd");
    LK_REQUIRE(niters >= 1, "niters must be >= 1");
    if (polyorder >= window) polyorder = window - 1;  // the reference clamps with a warning (lightcurve.py:1015-1020)
    LK_REQUIRE(polyorder >= 0 && polyorder <= 15, "polyorder %d outside 0..15", polyorder);
    LK_REQUIRE(n_off_host[0] == 0, "n_off[0] must be 0");
    // the design (taps + edge operators, ~1 MB at window 401) is built once per (device, window, polyorder) in long
    // double on the host and kept resident; it is a few ms of host work that would otherwise dominate small batches
    struct Design {
        int device, window, polyorder;
        double *d_c, *d_e, *d_minv;  // d_minv: inverse moment matrix of the edge fit (polyorder <= 5), else nullptr
        double quad_a, quad_b;  // taps = quad_a + quad_b k^2 (polyorder <= 3), else 0, 0
    };
    static std::vector<Design> cache;
    const Design *des = nullptr;
    for (const Design &d : cache)
        if (d.device == h->device && d.window == window && d.polyorder == polyorder) des = &d;
    if (!des) {
        std::vector<double> coeffs, edge;
        long double tap_poly[16] = {0};
        std::vector<double> minv;
        LK_REQUIRE(savgol_design(window, polyorder, coeffs, edge, tap_poly, &minv),
                   "singular Savitzky-Golay design (window %d, order %d)", window, polyorder);
        Design d{h->device, window, polyorder, nullptr, nullptr, nullptr, 0.0, 0.0};
        if (polyorder <= 5 && window >= 3) {
            LK_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d.d_minv), minv.size() * 8));
            LK_HIP_CHECK(hipMemcpy(d.d_minv, minv.data(), minv.size() * 8, hipMemcpyHostToDevice));
        }
        if (polyorder >= 2 && polyorder <= 3 && window >= 3) {
            const int half = window / 2;
            const long double a = tap_poly[0], b = tap_poly[2] / ((long double)half * (long double)half);
            long double worst = 0.0L;  // the odd terms vanish by symmetry; check the quadratic against the taps anyway
            for (int j = 0; j < window; ++j) {
                const long double k = (long double)(half - j);
                worst = std::max(worst, fabsl(a + b * k * k - (long double)coeffs[j]));
            }
            if (worst <= 1e-15L * fabsl(a)) {
                d.quad_a = (double)a;
                d.quad_b = (double)b;
            }
        }
        LK_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d.d_c), coeffs.size() * 8));
        LK_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&d.d_e), edge.size() * 8 + 8));
        LK_HIP_CHECK(hipMemcpy(d.d_c, coeffs.data(), coeffs.size() * 8, hipMemcpyHostToDevice));
        if (!edge.empty()) {
            const int half = window / 2;
            std::vector<double> et(edge.size());  // [side][row][tap] -> [side][tap][row]
            for (int sd = 0; sd < 2; ++sd)
                for (int r = 0; r < half; ++r)
                    for (int j = 0; j < window; ++j)
                        et[((size_t)sd * window + j) * half + r] = edge[((size_t)sd * half + r) * window + j];
            LK_HIP_CHECK(hipMemcpy(d.d_e, et.data(), et.size() * 8, hipMemcpyHostToDevice));
        }
        cache.push_back(d);
        des = &cache.back();
    }
    double *d_c = des->d_c, *d_e = des->d_e;
    // moment form of the Savitzky-Golay interior for long windows (see the kernel)
    constexpr int quad_min = 201;
    const bool use_quad = quad_min > 0 && window >= quad_min && des->quad_b != 0.0;
    const double quad_a = use_quad ? des->quad_a : 0.0, quad_b = use_quad ? des->quad_b : 0.0;
    constexpr bool edge_moments = true;
    const double *d_minv = edge_moments ? des->d_minv : nullptr;
    constexpr int near_on = 1;  // guided dt median in iterations >= 1
    std::vector<int64_t> soff((size_t)B + 1, 0);
    for (int b = 0; b < B; ++b) {
        const int64_t n = n_off_host[b + 1] - n_off_host[b];
        LK_REQUIRE(n >= 1 && n < ((int64_t)1 << 30), "target %d has %lld cadences", b, (long long)n);
        const int64_t np = (n + 7) & ~(int64_t)7;
        // slab: tm, fm, tr (8 B each) | idx, segs (4 B each, segs + 8 entries) | mask, mask1 (1 B each)
        soff[b + 1] = soff[b] + ((3 * np * 8 + 2 * np * 4 + 32 + 2 * np + 255) & ~(int64_t)255);
    }
    h->ws.reset();
    int rc = h->ws.reserve((size_t)soff[B] + (size_t)B * (64 + 16 * 16) + 8192);
    if (rc) return rc;
    // the two offset tables travel as ONE staged copy through the handle's pinned ring (captured before this function
    // returns, ordered on the caller's stream): no host synchronisation in the launcher (it used to idle the GPU ~50 us per call)
    // They live in a buffer of the handle's own and are re-sent only when they differ from the previous call's (batches of one
    // shape follow each other in a pipeline): the copy and the queue hand-over around it cost ~15 us of idle GPU per call.
    char *d_s = (char *)h->ws.alloc((size_t)soff[B]);
    LK_REQUIRE(d_s, "workspace exhausted");
    std::vector<int64_t> both((size_t)2 * (B + 1));
    std::copy(n_off_host, n_off_host + B + 1, both.begin());
    std::copy(soff.begin(), soff.end(), both.begin() + B + 1);
    if (h->flat_tab_cap < both.size()) {
        h->flat_tab_host.clear();
        if (h->flat_tab_dev) LK_HIP_CHECK(hipFree(h->flat_tab_dev));   // (synchronises: no kernel reads it any more)
        h->flat_tab_dev = nullptr;
        h->flat_tab_cap = 0;
        LK_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&h->flat_tab_dev), (both.size() + 1024) * 8));
        h->flat_tab_cap = both.size() + 1024;
    }
    // (a call on ANOTHER stream re-sends them even when they are equal: the cached copy is ordered on the stream it was
    // queued on, not on this one — ADVICE r5; calls on different streams must still be separated by a synchronisation,
    // include/lkhip.h "Conventions")
    if (h->flat_tab_host != both || h->flat_tab_stream != stream) {
        h->flat_tab_host.clear();
        const int rcs = h->stage.copy(h->flat_tab_dev, both.data(), both.size() * 8, stream);
        if (rcs) return rcs;
        h->flat_tab_host = both;
        h->flat_tab_stream = stream;
    }
    const int64_t *d_off = h->flat_tab_dev, *d_soff = d_off + (B + 1);
    int64_t nmax = 0;
    for (int b = 0; b < B; ++b) nmax = std::max(nmax, n_off_host[b + 1] - n_off_host[b]);
    // FIR / candidate area of the tap-by-tap trend kernel (short windows): 8 x 612 doubles = 4096-output tiles; doubled until
    // the window fits a tile
    int fir_lds = 4896;
    while (fir_lds < 16384 && window > fir_lds / 2 + 1) fir_lds *= 2;
    const size_t lds_sel = (size_t)std::max(FLAT_NT, 264) * 8 + (size_t)(fir_lds + 2) * 8 + (size_t)FLAT_NT * 4;
#ifdef LK_FLAT_PROFILE
    const int stop_at = getenv("LK_FLAT_STOP") ? atoi(getenv("LK_FLAT_STOP")) : -1;  // profiling aid, see flat_init_kernel
#else
    constexpr int stop_at = -1;
#endif
    FlatState *d_state = (FlatState *)h->ws.alloc((size_t)B * sizeof(FlatState));
    // trend workgroups per light curve: enough tiles for each (a 20 000-cadence light curve has ~19 tiles of ~1070 outputs at
    // window 401; a 4500-cadence one 4), and B x T >= ~4 workgroups per CU
    int trend_T = (int)std::max<int64_t>(1, std::min<int64_t>(8, nmax / 4096));
#ifdef LK_FLAT_PROFILE
    if (getenv("LK_FLAT_T")) trend_T = std::max(1, std::min(16, atoi(getenv("LK_FLAT_T"))));
#endif
    double2 *d_rs = (double2 *)h->ws.alloc((size_t)B * trend_T * sizeof(double2));
    LK_REQUIRE(d_state != nullptr && d_rs != nullptr, "workspace exhausted (flatten state)");
    // the moment-form trend kernel keeps four tile arrays (three prefix sums + the inputs) of 1470 doubles: three 512-thread
    // workgroups of it fit a CU's 160 KB of LDS (longer windows: sized from the tap kernel's area)
    const int fir_trend = std::max(4 * 1470, ((window > 1470 / 2 ? fir_lds : 0) / 3) * 4);
    const int quad_cap = ((fir_trend / 4 - 1) * 32) / 33;  // the kernel's QCAP (arrays padded by one double per 32)
    const bool quad_kernel = quad_b != 0.0 && std::min(quad_cap, 4 * window) - (window - 1) >= 64;
    const size_t lds_trend = quad_kernel ? (size_t)std::max(FLAT_NT, 264) * 8 + (size_t)(fir_trend + 2) * 8 + (size_t)FLAT_NT * 4 : lds_sel;
    // (__syncthreads_or keeps a few bytes of static LDS: the dynamic part may not claim all 160 KB)
    int rc_ = want_lds(h, reinterpret_cast<const void *>(flat_init_kernel), 152 * 1024);
    if (!rc_) rc_ = want_lds(h, reinterpret_cast<const void *>(flat_dtseg_kernel), 152 * 1024);
    if (!rc_) rc_ = want_lds(h, reinterpret_cast<const void *>(flat_trend_kernel<true>), 152 * 1024);
    if (!rc_) rc_ = want_lds(h, reinterpret_cast<const void *>(flat_trend_kernel<false>), 152 * 1024);
    if (rc_) return rc_;
    // the select phases: a 4096-double candidate area (the sampled selects collect ~2600 candidates) keeps FOUR of their
    // workgroups on a CU — 1024 slots: the 1000 light curves of the bench shape run as one wave of workgroups
    constexpr int fir_pick = 4096;
    const size_t lds_pick = (size_t)std::max(FLAT_NT, 264) * 8 + (size_t)(fir_pick + 2) * 8 + (size_t)FLAT_NT * 4;
    hipLaunchKernelGGL(flat_init_kernel, dim3(B), dim3(FLAT_NT), lds_pick, stream, flux, user_mask, d_off, sigma, d_s, d_soff,
                       d_state, fir_pick, (stop_at >= 99 && stop_at < 200) ? stop_at - 100 : -1);
    for (int it = 0; it < niters; ++it) {
        hipLaunchKernelGGL(flat_compact_kernel, dim3(B), dim3(FLAT_NT), 0, stream, t, flux, d_off, d_s, d_soff, d_state, trend, it);
        hipLaunchKernelGGL(flat_dtseg_kernel, dim3(B), dim3(FLAT_NT), lds_pick, stream, d_off, break_tol, d_s, d_soff, d_state,
                           fir_pick, it, near_on, (stop_at >= 200 && stop_at < 300) ? stop_at - 200 : -1);
        if (quad_kernel)
            hipLaunchKernelGGL(flat_trend_kernel<true>, dim3(B, trend_T), dim3(FLAT_NT), lds_trend, stream, d_off, window, polyorder,
                               break_tol, d_c, d_e, d_s, d_soff, d_state, fir_trend, quad_a, quad_b, d_minv, d_rs,
                               stop_at >= 300 ? stop_at - 300 : -1);
        else
            hipLaunchKernelGGL(flat_trend_kernel<false>, dim3(B, trend_T), dim3(FLAT_NT), lds_sel, stream, d_off, window, polyorder,
                               break_tol, d_c, d_e, d_s, d_soff, d_state, fir_lds, quad_a, quad_b, d_minv, d_rs,
                               stop_at >= 300 ? stop_at - 300 : -1);
        hipLaunchKernelGGL(flat_clip_kernel, dim3(B), dim3(FLAT_NT), 0, stream, d_off, sigma, d_s, d_soff, d_state,
                           it == niters - 1 ? 1 : 0, d_rs, trend_T);
    }
    hipLaunchKernelGGL(flat_interp_kernel, dim3(B), dim3(FLAT_NT), 0, stream, t, d_off, d_s, d_soff, d_state, trend, final_mask);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
