// regress.hip — RegressionCorrector numerics on gfx950: weighted normal equations with Gaussian priors,
// iterated with sigma clipping (reference: src/lightkurve/correctors/regressioncorrector.py:127-189 and
// 243-279; astropy.stats.sigma_clip defaults maxiters=5, median/std).
//
// Per iteration, for a batch of targets that share K (columns) but have ragged N (cadences):
//   gram_mfma_kernel   G = [X | y]^T diag(m / err^2) [X | y]   (upper 64x64 blocks) on the fp64 matrix cores
//                      (v_mfma_f64_16x16x4_f64); X tiles are staged through LDS with the weight folded into the
//                      left operand.  This is the only MFMA-shaped contraction on the path (2 N K^2 flop).
//   solve_kernel       A = G[:K,:K] + diag(1/sigma_p^2), b = G[:K,K] + mu_p/sigma_p^2, LU with partial pivoting
//                      (the reference calls LAPACK gesv), one workgroup per target, in global scratch.
//   resid_clip_kernel  r = y - X w over ALL cadences, then astropy's sigma_clip on r (median by radix select,
//                      std, <=5 passes) and outlier |= clipped.   (reference quirk, SURVEY App. B.5: the clip
//                      statistics include cadences outside cadence_mask and earlier outliers.)
// Final: model = X w - median(X w).
#include <cstdlib>

#include "block_select.hpp"
#include "lk_common.hpp"

namespace lk {

// Largest number of regressors: the global-memory LU (solve_kernel / invert_kernel) keeps one column of multipliers in LDS.
// 1023 until round 4; wider matrices (a spline with a knot per 5 cadences of a 20 000-cadence light curve, split design
// matrices) are rare and slow — O(K^3) in one workgroup — but they stay on the device.
constexpr int REG_KMAX = 4095;


typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int GR_BLK = 64;    // output block edge
constexpr int GR_RC = 32;     // cadences per LDS stage
constexpr int GR_LD = 80;     // LDS row stride in doubles (== 32 dwords mod 64: conflict-free ds_read_b64 fragments)

// element (n, c) of the augmented matrix [X | y], zero beyond column K
__device__ __forceinline__ double aug(const double *__restrict__ X, const double *__restrict__ y, int K, int64_t n,
                                      int c) {
    if (c < K) return X[n * K + c];
    return (c == K && y) ? y[n] : 0.0;
}

// DELTA (passes 2.. of the clip loop): the fit mask only ever loses cadences (outl |= clipped), so the normal matrix of
// the next pass is the previous one minus the contributions of the cadences the last clip ADDED to the outlier set:
// clip_kernel leaves them as an ascending list (new_idx, new_cnt) and this kernel runs its stages over the list instead of
// over all N cadences and subtracts the result from G in place — tens of rows instead of thousands.  A target whose list
// overflowed (new_cnt > new_cap) is recomputed in full.  The list order and the stage order are fixed: same bits every run.
template <bool DELTA>
__global__ __launch_bounds__(256) void gram_mfma_kernel(const double *__restrict__ X, const double *__restrict__ y,
                                                         const double *__restrict__ err,
                                                         const uint8_t *__restrict__ cmask,
                                                         const uint8_t *__restrict__ outl,
                                                         const int64_t *__restrict__ n_off, int K, int KB,
                                                         double *__restrict__ G, const int *__restrict__ done = nullptr,
                                                         const int *__restrict__ new_cnt = nullptr,
                                                         const int *__restrict__ new_idx = nullptr, int new_cap = 0) {
    if (done && done[blockIdx.y]) return;  // this target's clip loop has converged (see regress_launch): G is final
    __shared__ double sa[GR_RC][GR_LD];  // weighted left tile  (rows: cadence, cols: 64 output rows)
    __shared__ double sb[GR_RC][GR_LD];  // right tile          (cols: 64 output cols)
    // upper-triangular block pair from the linear block index
    int bi = 0, bj = blockIdx.x;
    while (bj >= KB - bi) {
        bj -= KB - bi;
        ++bi;
    }
    bj += bi;
    const int target = blockIdx.y;
    const int64_t lo = n_off[target];
    int n = (int)(n_off[target + 1] - lo);
    const int *rows = nullptr;  // DELTA: the cadences to take away (ascending); nullptr = all cadences
    if (DELTA) {
        const int cnt = new_cnt[target];
        if (cnt <= new_cap) {
            rows = new_idx + (size_t)target * new_cap;
            n = cnt;
        }
    }
    if (n <= 0) return;  // (DELTA: nothing to take away; the clamped loads below need one valid row)
    X += lo * K;
    if (y) y += lo;
    const int Kp = KB * GR_BLK, Ka = K + (y ? 1 : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;  // wave owns the 32x32 sub-block (wi, wj): 2x2 MFMA tiles
    const int i0 = bi * GR_BLK, j0 = bj * GR_BLK;
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const bool live_i0 = i0 + wi * 32 < Ka, live_i1 = i0 + wi * 32 + 16 < Ka;
    const bool live_j0 = j0 + wj * 32 < Ka, live_j1 = j0 + wj * 32 + 16 < Ka;

    __shared__ double sw[GR_RC];  // the stage's 32 cadence weights (0 = masked), computed once per cadence
    // Software pipeline: the global loads of stage s + 1 (8 + 8 values per thread, and the stage's weights on the first
    // 32 threads) are issued before the MFMAs of stage s and land in registers while the matrix cores work.  The loads
    // are unconditional on clamped addresses (a thread's two columns are fixed: base pointer + row stride each; a
    // column beyond [X | y] reads a valid address with stride 0) and the zeros are selected on the way into LDS:
    // `in ? load : 0` ends up as a branch around the load with a full wait behind it, eight round trips per stage.
    const int ccol = tid & 63, crow = tid >> 6;
    const double *pa, *pb;
    int sa_str, sb_str;
    bool la, lb;
    {
        auto col = [&](int c, const double *&base, int &stride, bool &live) {
            const bool isx = c < K, isy = c == K && y != nullptr;
            live = isx || isy;
            base = isx ? X + c : (isy ? y : X);
            stride = isx ? K : (isy ? 1 : 0);
        };
        col(i0 + ccol, pa, sa_str, la);
        col(j0 + ccol, pb, sb_str, lb);
    }
    double ra[8], rb[8], wv = 1.0;
    uint8_t wc = 1, wo = 0;
    auto fetch = [&](int n0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int nn = min(n0 + crow + 4 * q, n - 1);
            if (DELTA && rows) nn = rows[nn];
            ra[q] = pa[(size_t)nn * sa_str];
            rb[q] = pb[(size_t)nn * sb_str];
        }
        if (tid < GR_RC) {
            int nn = min(n0 + tid, n - 1);
            if (DELTA && rows) nn = rows[nn];
            const int64_t g = lo + nn;
            wv = err ? err[g] : 1.0;
            wc = cmask ? cmask[g] : (uint8_t)1;
            // a listed cadence was in the previous fit unless cadence_mask excludes it (the clip runs over all cadences)
            wo = (outl && !(DELTA && rows)) ? outl[g] : (uint8_t)0;
        }
    };
    fetch(0);
    for (int n0 = 0; n0 < n; n0 += GR_RC) {
        if (tid < GR_RC) sw[tid] = (n0 + tid < n && wc != 0 && wo == 0) ? 1.0 / (wv * wv) : 0.0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = crow + 4 * q;
            const bool in = n0 + r < n;
            sa[r][ccol] = ((in && la) ? ra[q] : 0.0) * sw[r];   // X / err^2 exactly as the reference forms it (:166)
            sb[r][ccol] = (in && lb) ? rb[q] : 0.0;
        }
        __syncthreads();
        if (n0 + GR_RC < n) fetch(n0 + GR_RC);
#pragma unroll
        for (int kk = 0; kk < GR_RC; kk += 4) {
            const int kr = kk + (lane >> 4), cc = lane & 15;
            const double a0 = sa[kr][wi * 32 + cc], a1 = sa[kr][wi * 32 + 16 + cc];
            const double b0 = sb[kr][wj * 32 + cc], b1 = sb[kr][wj * 32 + 16 + cc];
            if (live_i0 && live_j0) acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            if (live_i0 && live_j1) acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            if (live_i1 && live_j0) acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            if (live_i1 && live_j1) acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    double *Gt = G + (size_t)target * Kp * Kp;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wi * 32 + a * 16 + (lane >> 4) + 4 * r;
                const int col = j0 + wj * 32 + b * 16 + (lane & 15);
                if (DELTA && rows)
                    Gt[(size_t)row * Kp + col] -= acc[a][b][r];
                else
                    Gt[(size_t)row * Kp + col] = acc[a][b][r];
            }
}

// ------------------------------------------------------------------------------------------------ narrow Grams (<= 144 columns)
// gram_mfma_kernel spends 48 (121 columns) or 57 (136 + 1) MFMA tiles per cadence step on a matrix whose upper triangle has
// 36 or 45, stages every 64-column strip through LDS once per block pair and meets two barriers per 32 cadences with
// four waves: 40 % of the fp64 MFMA rate.  Here ONE 8-wave workgroup owns a matrix: the upper-triangular 16 x 16 tiles
// (row-major) are dealt to the waves in contiguous runs of <= GT_NS; a stage of 32 cadences x all columns sits in LDS
// column-major ([column][cadence], leading dimension 40: a lane's 16-byte read is conflict-free across the wave), so one
// ds_read_b128 feeds TWO MFMA steps — the summation index of v_mfma_f64_16x16x4_f64 is permuted (step 2m + h takes
// cadences 8m + 2q + h from lane group q) — double-buffered with one barrier per stage.  The left operand of a tile row is
// the same LDS value as the right operand of that tile column, times the cadence weight (X / err^2 as the reference forms
// it): no second tile.  Off-diagonal tiles are written to both triangles.
constexpr int GT_RC = 32, GT_LDC = 34, GT_NT = 512, GT_MAXT = 9;
// optional float32 source of gram_tri_kernel: element (row, col) = (double)(mode == 0 ? pix : pix / div[row]) - mean[col]
// (div: per-cadence float32 divisors of the batch, mean: per-target column means, either may be null = no division / no centring)
struct GramF32Source {
    const float *pix, *div;
    const double *mean;
    int mode;
};

// NS = tiles per wave = ceil(T (T + 1) / 2 / 8).  The hot loop is branch-free: a wave with fewer tiles than NS repeats its
// last one (the copy is not written), every slot reads both its operands, every global load is unconditional (clamped
// address, value selected afterwards) — a branch around a load or an MFMA makes the compiler wait for everything in flight.
template <bool WEIGHTED, int NS, bool F32SRC = false>  // F32SRC: the matrix comes from float32 pixels (GramF32Source); a template so the float64 instances keep their registers
__global__ __launch_bounds__(GT_NT) void gram_tri_kernel(const double *__restrict__ X, const double *__restrict__ y,
                                                         const double *__restrict__ err,
                                                         const uint8_t *__restrict__ cmask,
                                                         const uint8_t *__restrict__ outl,
                                                         const int64_t *__restrict__ n_off, int K, int ldg,
                                                         double *__restrict__ G, const int *__restrict__ done = nullptr,
                                                         GramF32Source fs = GramF32Source{nullptr, nullptr, nullptr, 0}) {
    if (done && done[blockIdx.x]) return;
    extern __shared__ __attribute__((aligned(16))) double gt_sm[];
    const int target = blockIdx.x;
    const int64_t lo = n_off[target];
    const int n = (int)(n_off[target + 1] - lo);
    // fs.pix: the matrix is not X but (double)(pix / div[row]) - mean[col] formed on the way into LDS from float32 pixels (the PLD
    // pixel blocks: what pld_ratio_kernel used to write out as float64, 1.7 GB per block, for this kernel and the projection to read back)
    constexpr bool f32src = F32SRC;
    const float *pixb = f32src ? fs.pix + lo * K : nullptr;
    const float *divb = (f32src && fs.div) ? fs.div + lo : nullptr;
    const double *meanb = (f32src && fs.mean) ? fs.mean + (size_t)target * K : nullptr;
    if (!f32src) X += lo * K;
    if (y) y += lo;
    const int Ka = K + (y ? 1 : 0), T = (Ka + 15) >> 4, Tc = T << 4;
    double *tile = gt_sm;                               // [2][Tc][GT_LDC]
    double *sw = gt_sm + 2 * (size_t)Tc * GT_LDC;       // [2][GT_RC] cadence weights of the stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, cc = lane & 15;
    // this wave's run of upper-triangular tiles (row-major): waves 0 .. rem-1 take per + 1, the others per
    const int nti = T * (T + 1) / 2, per = nti >> 3, rem = nti & 7;
    const int first = wave * per + min(wave, rem), cnt = per + (wave < rem ? 1 : 0);
    int aoff[NS], boff[NS], ti[NS], tj[NS];  // LDS offsets (doubles) of a slot's operands for lane group 0, cadence 0
    {
        int i = 0, j = first;
        while (i < T - 1 && j >= T - i) {
            j -= T - i;
            ++i;
        }
        j = min(j + i, T - 1);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            ti[s] = i;
            tj[s] = j;
            aoff[s] = ((i << 4) + cc) * GT_LDC + 2 * q;
            boff[s] = ((j << 4) + cc) * GT_LDC + 2 * q;
            if (s + 1 < cnt) {  // slots beyond the run repeat its last tile
                if (++j == T) {
                    ++i;
                    j = i;
                }
            }
        }
    }
    double4_t acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // global -> registers -> LDS: thread (c16, row pair rp, column-tile parity) takes cadences 2 rp, 2 rp + 1 of column
    // 16 tc + c16 for tc = parity, parity + 2, ...: one 16-byte LDS write each.  Per column: base pointer and row stride
    // (X: K, y: 1, nothing: a valid address with stride 0 whose value is dropped).
    const int c16 = tid & 15, rp = (tid >> 4) & 15, tcs = tid >> 8;
    constexpr int NP = (GT_MAXT + 1) / 2;
    const double *cbase[NP];
    const float *pbase[NP];
    int cstride[NP];
    bool clive[NP];
    double cmean[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int col = ((tcs + 2 * u) << 4) + c16;
        const bool isx = col < K, isy = (col == K) && y != nullptr;
        clive[u] = isx || isy;
        cbase[u] = f32src ? y : (isx ? X + col : (isy ? y : X));           // (float32 source: only the y column is float64)
        pbase[u] = f32src ? (isx ? pixb + col : pixb) : nullptr;
        cstride[u] = isx ? K : (isy ? 1 : 0);
        cmean[u] = (meanb && isx) ? meanb[col] : 0.0;
    }
    double f0[NP], f1[NP], wv = 1.0;
    float g0[NP], g1[NP], da = 1.0f, db = 1.0f;  // float32 source: raw pixels and the two rows' divisors (nothing is consumed here)
    uint8_t wc = 1, wo = 0;
    auto fetch = [&](int n0) {
        const int r0 = n0 + 2 * rp, ra = min(r0, n - 1), rb = min(r0 + 1, n - 1);
        if (f32src) {
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                g0[u] = pbase[u][(size_t)ra * cstride[u]];
                g1[u] = pbase[u][(size_t)rb * cstride[u]];
            }
            if (divb) {
                da = divb[ra];
                db = divb[rb];
            }
        } else {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            f0[u] = cbase[u][(size_t)ra * cstride[u]];
            f1[u] = cbase[u][(size_t)rb * cstride[u]];
        }
        }
        if (WEIGHTED && tid < GT_RC) {  // raw values only: nothing here may wait for a load (the MFMAs of the stage follow)
            const int64_t g = lo + min(n0 + tid, n - 1);
            wv = err ? err[g] : 1.0;
            wc = cmask ? cmask[g] : (uint8_t)1;
            wo = outl ? outl[g] : (uint8_t)0;
        }
    };
    auto stash = [&](int buf, int n0) {  // values beyond the matrix or the batch become zeros on their way into LDS
        double *tb = tile + (size_t)buf * Tc * GT_LDC;
        const int r0 = n0 + 2 * rp;
        if (f32src) {  // the ratio in float32 as the reference forms it, then float64, then the column mean (pld_ratio_kernel's arithmetic)
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                f0[u] = (double)(fs.mode == 0 ? g0[u] : g0[u] / da) - cmean[u];
                f1[u] = (double)(fs.mode == 0 ? g1[u] : g1[u] / db) - cmean[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NP; ++u)
            if (tcs + 2 * u < T) {
                const int col = ((tcs + 2 * u) << 4) + c16;
                *reinterpret_cast<double2 *>(tb + (size_t)col * GT_LDC + 2 * rp) =
                    make_double2((clive[u] && r0 < n) ? f0[u] : 0.0, (clive[u] && r0 + 1 < n) ? f1[u] : 0.0);
            }
        if (WEIGHTED && tid < GT_RC) sw[buf * GT_RC + tid] = (wc != 0 && wo == 0 && n0 + tid < n) ? 1.0 / (wv * wv) : 0.0;
    };
    fetch(0);
    stash(0, 0);
    __syncthreads();
    int buf = 0;
    for (int n0 = 0; n0 < n; n0 += GT_RC) {
        if (n0 + GT_RC < n) fetch(n0 + GT_RC);  // wave-uniform; the last stage stashes stale registers into the idle buffer
        const double *tb = tile + (size_t)buf * Tc * GT_LDC;
#pragma unroll
        for (int m = 0; m < GT_RC / 8; ++m) {
            // lane group q: cadences 8m + 2q, 8m + 2q + 1 of the step pair (2m, 2m + 1)
            double2 w2 = make_double2(1.0, 1.0);
            if (WEIGHTED) w2 = *reinterpret_cast<const double2 *>(sw + buf * GT_RC + 8 * m + 2 * q);
            double2 a2[NS], b2[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                a2[s] = *reinterpret_cast<const double2 *>(tb + aoff[s] + 8 * m);
                b2[s] = *reinterpret_cast<const double2 *>(tb + boff[s] + 8 * m);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (WEIGHTED) {
                    a2[s].x *= w2.x;
                    a2[s].y *= w2.y;
                }
                acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[s].x, b2[s].x, acc[s], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[s].y, b2[s].y, acc[s], 0, 0, 0);
        }
        stash(buf ^ 1, n0 + GT_RC);
        __syncthreads();
        buf ^= 1;
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    double *Gt = G + (size_t)target * ldg * ldg;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < cnt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (ti[s] << 4) + q + 4 * r, col = (tj[s] << 4) + cc;
                Gt[(size_t)row * ldg + col] = acc[s][r];
                if (ti[s] != tj[s]) Gt[(size_t)col * ldg + row] = acc[s][r];
            }
        }
}

template <bool WEIGHTED, int NS>
static void gram_tri_go(lk_handle *h, int *rc, size_t lds, const double *X, const double *y, const double *err,
                        const uint8_t *cmask, const uint8_t *outl, const int64_t *d_off, int B, int K, int ldg, double *G,
                        const int *done, hipStream_t stream, GramF32Source fs) {
    if (!WEIGHTED && fs.pix != nullptr) {  // (the float32 source exists for plain Grams only)
        *rc = want_lds(h, reinterpret_cast<const void *>(gram_tri_kernel<false, NS, true>), 160 * 1024);
        if (*rc) return;
        hipLaunchKernelGGL((gram_tri_kernel<false, NS, true>), dim3(B), dim3(GT_NT), lds, stream, X, y, err, cmask, outl, d_off, K,
                           ldg, G, done, fs);
        return;
    }
    *rc = want_lds(h, reinterpret_cast<const void *>(gram_tri_kernel<WEIGHTED, NS>), 160 * 1024);
    if (*rc) return;
    hipLaunchKernelGGL((gram_tri_kernel<WEIGHTED, NS>), dim3(B), dim3(GT_NT), lds, stream, X, y, err, cmask, outl, d_off, K, ldg,
                       G, done, fs);
}

// launch helper: true if the narrow kernel took the job
static bool gram_tri_try(lk_handle *h, const double *X, const double *y, const double *err, const uint8_t *cmask,
                         const uint8_t *outl, const int64_t *d_off, int B, int K, int ldg, double *G, const int *done,
                         hipStream_t stream, GramF32Source fs = GramF32Source{nullptr, nullptr, nullptr, 0}) {
    const int Ka = K + (y ? 1 : 0), T = (Ka + 15) / 16;
    if (!h || T > GT_MAXT || ldg < 16 * T) return false;
    const size_t lds = ((size_t)2 * 16 * T * GT_LDC + 2 * GT_RC) * 8;
    const bool weighted = err || cmask || outl;
    const int ns = (T * (T + 1) / 2 + 7) / 8;
    int rc = 0;
#define GT_CASE(NS_)                                                                                              \
    case NS_:                                                                                                     \
        if (weighted)                                                                                             \
            gram_tri_go<true, NS_>(h, &rc, lds, X, y, err, cmask, outl, d_off, B, K, ldg, G, done, stream, fs);       \
        else                                                                                                      \
            gram_tri_go<false, NS_>(h, &rc, lds, X, y, err, cmask, outl, d_off, B, K, ldg, G, done, stream, fs);      \
        break;
    switch (ns) {
        GT_CASE(1)
        GT_CASE(2)
        GT_CASE(3)
        GT_CASE(4)
        GT_CASE(5)
        GT_CASE(6)
        default:
            return false;
    }
#undef GT_CASE
    return rc == 0;
}

// A w = b by LU with partial pivoting, one 256-thread workgroup per target.  A lives in global scratch
// (K x (K+1) augmented, row-major); trailing updates are spread over the workgroup.
__global__ __launch_bounds__(256) void solve_kernel(const double *__restrict__ G, int K, int Kp,
                                                     const double *__restrict__ prior_mu,
                                                     const double *__restrict__ prior_sigma,
                                                     double *__restrict__ Awork, double *__restrict__ w,
                                                     const int *__restrict__ done = nullptr) {
    if (done && done[blockIdx.x]) return;
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    __shared__ double s_col[REG_KMAX + 1];  // pivot-column multipliers
    const int target = blockIdx.x, tid = threadIdx.x;
    const double *Gt = G + (size_t)target * Kp * Kp;
    const int Ka = K + 1;
    double *A = Awork + (size_t)target * K * Ka;
    // build the augmented system from the upper-triangular Gram blocks
    for (int e = tid; e < K * Ka; e += 256) {
        const int i = e / Ka, j = e - i * Ka;
        double v;
        if (j < K) {
            v = (j >= i || (j / GR_BLK) == (i / GR_BLK)) ? Gt[(size_t)i * Kp + j] : Gt[(size_t)j * Kp + i];
            // blocks on the diagonal were computed in full; off-diagonal lower blocks come from the transpose
            if (i == j && prior_sigma) {
                const double s = prior_sigma[(size_t)target * K + i];
                v += 1.0 / (s * s);
            }
        } else {
            v = Gt[(size_t)i * Kp + K];
            if (prior_sigma) {
                const double s = prior_sigma[(size_t)target * K + i];
                v += prior_mu[(size_t)target * K + i] / (s * s);
            }
        }
        A[e] = v;
    }
    __syncthreads();
    for (int j = 0; j < K; ++j) {
        // pivot search in column j, rows j..K-1 (first maximum wins, like LAPACK idamax)
        double best = -1.0;
        int bi = j;
        for (int i = j + tid; i < K; i += 256) {
            const double v = fabs(A[(size_t)i * Ka + j]);
            if (v > best) {
                best = v;
                bi = i;
            }
        }
        s_val[tid] = best;
        s_idx[tid] = bi;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                const double v2 = s_val[tid + s];
                const int i2 = s_idx[tid + s];
                if (v2 > s_val[tid] || (v2 == s_val[tid] && i2 < s_idx[tid])) {
                    s_val[tid] = v2;
                    s_idx[tid] = i2;
                }
            }
            __syncthreads();
        }
        const int p = s_idx[0];
        __syncthreads();
        if (p != j) {
            for (int c = tid; c < Ka; c += 256) {
                const double t0 = A[(size_t)j * Ka + c];
                A[(size_t)j * Ka + c] = A[(size_t)p * Ka + c];
                A[(size_t)p * Ka + c] = t0;
            }
        }
        __syncthreads();
        const double piv = A[(size_t)j * Ka + j];
        for (int i = j + 1 + tid; i < K; i += 256) s_col[i] = A[(size_t)i * Ka + j] / piv;
        __syncthreads();
        // trailing update rows j+1..K-1, cols j+1..K (including the right-hand side)
        const int nr = K - j - 1, nc = Ka - j - 1;
        for (int e = tid; e < nr * nc; e += 256) {
            const int i = j + 1 + e / nc, c = j + 1 + e % nc;
            A[(size_t)i * Ka + c] = fma(-s_col[i], A[(size_t)j * Ka + c], A[(size_t)i * Ka + c]);
        }
        __syncthreads();
    }
    // back substitution (serial in i, parallel dot products)
    for (int i = K - 1; i >= 0; --i) {
        double part = 0.0;
        for (int c = i + 1 + tid; c < K; c += 256) part = fma(A[(size_t)i * Ka + c], s_col[c], part);
        s_val[tid] = part;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) s_val[tid] += s_val[tid + s];
            __syncthreads();
        }
        if (tid == 0) s_col[i] = (A[(size_t)i * Ka + K] - s_val[0]) / A[(size_t)i * Ka + i];
        __syncthreads();
    }
    for (int i = tid; i < K; i += 256) w[(size_t)target * K + i] = s_col[i];
}

// The same solve for K <= ~138 with the whole augmented system in LDS: no global round trips between the phases of a
// column, the pivot found by a wave reduction + one 4-entry LDS exchange (2 barriers instead of 9), and a column-oriented
// back substitution (1 barrier per unknown instead of 9).  Same pivot rule; the update order per element is unchanged.
constexpr int SOLVE_NT = 1024, SOLVE_NW = SOLVE_NT / 64;  // sixteen waves: the trailing update of a column is rows / waves deep
__global__ __launch_bounds__(SOLVE_NT) void solve_lds_kernel(const double *__restrict__ G, int K, int Kp,
                                                         const double *__restrict__ prior_mu,
                                                         const double *__restrict__ prior_sigma,
                                                         double *__restrict__ w, const int *__restrict__ done = nullptr) {
    if (done && done[blockIdx.x]) return;
    extern __shared__ __attribute__((aligned(16))) double s_A[];  // K x (K + 1) augmented system | K multipliers | pivots
    const int target = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *Gt = G + (size_t)target * Kp * Kp;
    const int Ka = K + 1;
    double *A = s_A, *x = s_A + (size_t)K * Ka;
    double *s_pv = x + K;
    int *s_pi = reinterpret_cast<int *>(s_pv + SOLVE_NW);
    // prior terms once per row (x, s_pv.. are free until the elimination starts): 1 / sigma^2 and mu / sigma^2
    double *p_d = x, *p_b = s_A + (size_t)K * Ka + K + 2 * SOLVE_NW;  // p_b: behind x and the pivot exchange words
    for (int i = tid; i < K; i += SOLVE_NT) {
        double d = 0.0, b = 0.0;
        if (prior_sigma) {
            const double sg = prior_sigma[(size_t)target * K + i];
            d = 1.0 / (sg * sg);
            b = prior_mu[(size_t)target * K + i] / (sg * sg);
        }
        p_d[i] = d;
        p_b[i] = b;
    }
    __syncthreads();
    // the augmented system: element e = (i, j), four loads in flight per thread, the source ADDRESS selected (upper 64 x 64
    // blocks hold the matrix; the mirrored part is read transposed), never the loaded value — a select between two loads
    // becomes two guarded loads with a wait each
    const int tot = K * Ka;
    for (int e0 = tid; e0 < tot; e0 += 4 * SOLVE_NT) {
        double v[4];
        int ii[4], jj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + SOLVE_NT * u, tot - 1);
            const int i = e / Ka, j = e - i * Ka;
            ii[u] = i;
            jj[u] = j;
            const bool direct = j >= i || (j / GR_BLK) == (i / GR_BLK);  // (j == K: the right-hand side column, direct)
            const double *src = direct ? Gt + (size_t)i * Kp + j : Gt + (size_t)j * Kp + i;
            v[u] = *src;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + SOLVE_NT * u < tot) {
                double a = v[u];
                if (jj[u] == ii[u]) a += p_d[ii[u]];
                if (jj[u] == K) a += p_b[ii[u]];
                A[ii[u] * Ka + jj[u]] = a;
            }
    }
    __syncthreads();
    // The pivot of column j + 1 is found INSIDE the trailing update of column j: lane 0 of the wave that updates row i holds
    // the new A[i][j + 1], so every wave leaves the best of its rows (first one wins: ascending rows, strict >) and the
    // search costs no extra pass, no shuffles and no barrier of its own.  Same values compared, same winner as the
    // separate search (LAPACK idamax: largest magnitude, smallest index).  Column 0 is searched the plain way.
    {
        double best = -1.0;
        int bi = 0;
        for (int i = tid; i < K; i += SOLVE_NT) {
            const double v = fabs(A[i * Ka]);
            if (v > best) {
                best = v;
                bi = i;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double v2 = __shfl_xor(best, o);
            const int i2 = __shfl_xor(bi, o);
            if (v2 > best || (v2 == best && i2 < bi)) {
                best = v2;
                bi = i2;
            }
        }
        if (lane == 0) {
            s_pv[wave] = best;
            s_pi[wave] = bi;
        }
        __syncthreads();
    }
    for (int j = 0; j < K; ++j) {
        int p = s_pi[0];
        double pb = s_pv[0];
#pragma unroll
        for (int q = 1; q < SOLVE_NW; ++q)
            if (s_pv[q] > pb || (s_pv[q] == pb && s_pi[q] < p)) {
                pb = s_pv[q];
                p = s_pi[q];
            }
        // an all-NaN column leaves no candidate (every comparison with NaN is false, p = INT_MAX): keep row j — the result is
        // NaN either way (numpy.linalg.solve gives NaN coefficients there too), but every LDS access stays in range
        if (p < j || p >= K) p = j;
        // multipliers straight from the un-swapped rows (row p plays the role of row j and vice versa), then the swap
        const double piv = A[p * Ka + j];
        for (int i = j + 1 + tid; i < K; i += SOLVE_NT) x[i] = A[(i == p ? j : i) * Ka + j] / piv;
        __syncthreads();
        if (p != j) {
            for (int c = tid; c < Ka; c += SOLVE_NT) {
                const double t0 = A[j * Ka + c];
                A[j * Ka + c] = A[p * Ka + c];
                A[p * Ka + c] = t0;
            }
            __syncthreads();
        }
        // trailing update: wave w takes rows j+1+w, j+5+w, ...; lanes over the columns
        double pr[3];  // Ka - j - 1 <= 143 < 192 columns for every K the LDS plan admits
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = j + 1 + lane + 64 * u;
            pr[u] = c < Ka ? A[j * Ka + c] : 0.0;
        }
        // (the barrier behind the multipliers separates this column's reads of the candidate words from the next column's writes)
        double nbest = -1.0;
        int nbi = 0x7fffffff;
        for (int i = j + 1 + wave; i < K; i += SOLVE_NW) {
            const double m = x[i];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int c = j + 1 + lane + 64 * u;
                if (c < Ka) {
                    const double v = fma(-m, pr[u], A[i * Ka + c]);
                    A[i * Ka + c] = v;
                    if (u == 0 && lane == 0 && fabs(v) > nbest) {  // (lane 0, u = 0: column j + 1)
                        nbest = fabs(v);
                        nbi = i;
                    }
                }
            }
        }
        if (lane == 0) {
            s_pv[wave] = nbest;  // -1 from a wave without rows: never the winner while a row is left
            s_pi[wave] = nbi;
        }
        __syncthreads();
    }
    // back substitution, column oriented: x_i = b_i / U_ii, then b_r -= U_ri x_i for r < i (the right-hand side lives in
    // column K); one barrier per unknown.  (Sum order differs from the dot-product form of the global kernel: results
    // agree to rounding.)
    for (int i = K - 1; i >= 0; --i) {
        const double xi = A[i * Ka + K] / A[i * Ka + i];  // (row i is not written below: one barrier per unknown is enough)
        if (tid == 0) x[i] = xi;
        for (int r = tid; r < i; r += SOLVE_NT) A[r * Ka + K] = fma(-A[r * Ka + i], xi, A[r * Ka + K]);
        __syncthreads();
    }
    for (int i = tid; i < K; i += SOLVE_NT) w[(size_t)target * K + i] = x[i];
}

// model[n] = sum_k X[n][k] w[k]; one wavefront per cadence row, lanes over k (coalesced), wave reduction.
__global__ __launch_bounds__(256) void model_kernel(const double *__restrict__ X, const double *__restrict__ w,
                                                     const int64_t *__restrict__ n_off, int K,
                                                     double *__restrict__ model, const int *__restrict__ done = nullptr) {
    if (done && done[blockIdx.y]) return;
    extern __shared__ __attribute__((aligned(16))) double s_w[];
    const int target = blockIdx.y;
    const int64_t lo = n_off[target];
    const int n = (int)(n_off[target + 1] - lo);
    for (int k = threadIdx.x; k < K; k += 256) s_w[k] = w[(size_t)target * K + k];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // two rows per trip, four columns per lane and row requested together (round 6: the plain loop waited for each of a row's
    // ceil(K / 64) loads in turn — 3.7 TB/s at K = 135); the multiply-adds and the shuffle tree keep their order: same bits
    const int stride = gridDim.x * 4;
    for (int r = blockIdx.x * 4 + wave; r < n; r += 2 * stride) {
        const int r2 = r + stride;
        const bool two = r2 < n;  // (wave-uniform)
        const double *rowa = X + (lo + r) * (int64_t)K, *rowb = X + (lo + (two ? r2 : r)) * (int64_t)K;
        double acca = 0.0, accb = 0.0;
        for (int k0 = lane; k0 < K; k0 += 256) {
            double xa[4], xb[4], wk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = min(k0 + 64 * u, K - 1);
                xa[u] = rowa[k];
                xb[u] = rowb[k];
                wk[u] = s_w[k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + 64 * u < K) {
                    acca = fma(xa[u], wk[u], acca);
                    accb = fma(xb[u], wk[u], accb);
                }
        }
        for (int off = 32; off > 0; off >>= 1) {
            acca += __shfl_down(acca, off);
            accb += __shfl_down(accb, off);
        }
        if (lane == 0) {
            model[lo + r] = acca;
            if (two) model[lo + r2] = accb;
        }
    }
}

// residuals over all cadences + astropy sigma_clip; outl |= clipped.  One 1024-thread workgroup per target.
__global__ __launch_bounds__(1024) void clip_kernel(const double *__restrict__ y, const double *__restrict__ model,
                                                     const int64_t *__restrict__ n_off, double sigma, int maxiters,
                                                     uint8_t *__restrict__ flag, uint8_t *__restrict__ outl,
                                                     int *__restrict__ done = nullptr, int *__restrict__ new_cnt = nullptr,
                                                     int *__restrict__ new_idx = nullptr, int new_cap = 0) {
    // done[target] != 0: an earlier pass of the clip loop added no outlier, so the fit mask, the fit, the residuals and
    // this clip would all repeat exactly — the remaining passes of the reference's `for count in range(niters)` are
    // no-ops for this target and are skipped (regressioncorrector.py:245-272 has no early exit; the fixed point is exact)
    if (done && done[blockIdx.x]) return;
    __shared__ unsigned long long sh[1024];
    const int target = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = n_off[target];
    const int n = (int)(n_off[target + 1] - lo);
    y += lo;
    model += lo;
    flag += lo;
    outl += lo;
    auto val = [&](int i) { return y[i] - model[i]; };
    auto keep = [&](int i) { return flag[i] != 0; };
    long long cnt = 0;
    for (int i = tid; i < n; i += 1024) {
        const double r = val(i);
        const bool fin = isfinite(r);
        flag[i] = fin ? 1 : 0;
        cnt += fin;
    }
    __syncthreads();
    long long count = block_count_dyn(cnt, reinterpret_cast<long long *>(sh));
    double lo_b = -INFINITY, hi_b = INFINITY;
    for (int it = 0; it < maxiters && count > 0; ++it) {
        const double cen = block_median(n, count, val, keep, sh);
        double part = 0.0;
        for (int i = tid; i < n; i += 1024)
            if (flag[i]) part += val(i);
        const double mean = block_sum_dyn(part, reinterpret_cast<double *>(sh)) / (double)count;
        part = 0.0;
        for (int i = tid; i < n; i += 1024)
            if (flag[i]) {
                const double d = val(i) - mean;
                part = fma(d, d, part);
            }
        const double sd = sqrt(block_sum_dyn(part, reinterpret_cast<double *>(sh)) / (double)count);
        lo_b = cen - sd * sigma;
        hi_b = cen + sd * sigma;
        cnt = 0;
        for (int i = tid; i < n; i += 1024) {
            if (flag[i]) {
                const double r = val(i);
                const bool in = (r >= lo_b) && (r <= hi_b);
                flag[i] = in ? 1 : 0;
                cnt += in;
            }
        }
        __syncthreads();
        const long long newcount = block_count_dyn(cnt, reinterpret_cast<long long *>(sh));
        const bool changed = newcount != count;
        count = newcount;
        if (!changed) break;
    }
    // outl |= clipped; the cadences that are NEW in the outlier set go to new_idx in ascending order (block-wide
    // compaction per sweep of 1024 cadences: the list order must not depend on scheduling, the DELTA Gram sums over it)
    int added = 0;
    int *wtot = reinterpret_cast<int *>(sh);  // [16] wave totals of the sweep
    if (new_idx) new_idx += (size_t)target * new_cap;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        bool isnew = false;
        if (i < n) {
            const double r = val(i);
            const bool clipped = !isfinite(r) || r < lo_b || r > hi_b;
            if (clipped) {
                isnew = outl[i] == 0;
                outl[i] = 1;
            }
        }
        if (new_idx) {
            const unsigned long long bal = __ballot(isnew);
            const int lane = tid & 63, wv = tid >> 6;
            __syncthreads();
            if (lane == 0) wtot[wv] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < 16; ++w) {
                const int c = wtot[w];
                before += w < wv ? c : 0;
                total += c;
            }
            const int pos = added + before + __popcll(bal & ((1ull << lane) - 1ull));
            if (isnew && pos < new_cap) new_idx[pos] = i;
            added += total;
        } else {
            added += isnew ? 1 : 0;
        }
    }
    if (new_idx) {
        if (tid == 0) new_cnt[target] = added;  // > new_cap: the list overflowed, the next Gram is computed in full
    } else if (done) {
        added = __syncthreads_or(added);
    }
    if (done && tid == 0 && !added) done[target] = 1;
}

// model -= median(model)
__global__ __launch_bounds__(1024) void demedian_kernel(const int64_t *__restrict__ n_off,
                                                         double *__restrict__ model) {
    __shared__ unsigned long long sh[1024];
    const int target = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = n_off[target];
    const int n = (int)(n_off[target + 1] - lo);
    model += lo;
    auto val = [&](int i) { return model[i]; };
    auto keep = [&](int) { return true; };
    const double med = block_median(n, (long long)n, val, keep, sh);
    for (int i = tid; i < n; i += 1024) model[i] -= med;
}

// (X^T S^-1 X + diag(1 / prior_sigma^2))^-1, the coefficient covariance RegressionCorrector keeps when
// propagate_errors=True (regressioncorrector.py:183-185, np.linalg.inv): Gauss-Jordan with partial pivoting on [A | I]
// in global scratch (K x 2K doubles per target), one 256-thread workgroup per target.
__global__ __launch_bounds__(256) void invert_kernel(const double *__restrict__ G, int K, int Kp,
                                                      const double *__restrict__ prior_sigma,
                                                      double *__restrict__ Awork, double *__restrict__ inv) {
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    __shared__ double s_col[REG_KMAX + 1];  // column j of every row
    const int target = blockIdx.x, tid = threadIdx.x;
    const double *Gt = G + (size_t)target * Kp * Kp;
    const int W = 2 * K;
    double *A = Awork + (size_t)target * K * W;
    for (int e = tid; e < K * W; e += 256) {
        const int i = e / W, j = e - i * W;
        double v;
        if (j < K) {
            v = (j >= i || (j / GR_BLK) == (i / GR_BLK)) ? Gt[(size_t)i * Kp + j] : Gt[(size_t)j * Kp + i];
            if (i == j && prior_sigma) {
                const double sg = prior_sigma[(size_t)target * K + i];
                v += 1.0 / (sg * sg);
            }
        } else {
            v = (j - K == i) ? 1.0 : 0.0;
        }
        A[e] = v;
    }
    __syncthreads();
    for (int j = 0; j < K; ++j) {
        double best = -1.0;
        int bi = j;
        for (int i = j + tid; i < K; i += 256) {
            const double v = fabs(A[(size_t)i * W + j]);
            if (v > best) {
                best = v;
                bi = i;
            }
        }
        s_val[tid] = best;
        s_idx[tid] = bi;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                const double v2 = s_val[tid + s];
                const int i2 = s_idx[tid + s];
                if (v2 > s_val[tid] || (v2 == s_val[tid] && i2 < s_idx[tid])) {
                    s_val[tid] = v2;
                    s_idx[tid] = i2;
                }
            }
            __syncthreads();
        }
        const int p = s_idx[0];
        __syncthreads();
        if (p != j) {
            for (int c = tid; c < W; c += 256) {
                const double t0 = A[(size_t)j * W + c];
                A[(size_t)j * W + c] = A[(size_t)p * W + c];
                A[(size_t)p * W + c] = t0;
            }
        }
        __syncthreads();
        const double piv = A[(size_t)j * W + j];
        for (int i = tid; i < K; i += 256) s_col[i] = A[(size_t)i * W + j];
        __syncthreads();
        // scale the pivot row (columns right of j; column j itself is never read again)
        for (int c = j + 1 + tid; c < W; c += 256) A[(size_t)j * W + c] /= piv;
        __syncthreads();
        // eliminate column j from every other row
        const int nc = W - j - 1;
        for (int e = tid; e < K * nc; e += 256) {
            const int i = e / nc, c = j + 1 + e % nc;
            if (i != j) A[(size_t)i * W + c] = fma(-s_col[i], A[(size_t)j * W + c], A[(size_t)i * W + c]);
        }
        __syncthreads();
    }
    for (int e = tid; e < K * K; e += 256) {
        const int i = e / K, c = e - i * K;
        inv[(size_t)target * K * K + e] = A[(size_t)i * W + K + c];
    }
}

// Plain Gram matrix G = A^T A of a wide block (PCA of the PLD product blocks, K in the hundreds): 128 x 128 output blocks,
// each wave a 64 x 64 sub-block = 4 x 4 MFMA tiles, 16 cadences per LDS stage.  Against the 64 x 64 kernel above (2 x 2
// tiles per wave, three barriers around 32 MFMAs) a stage here is two barriers around 64 MFMAs per wave with 8 LDS
// fragment reads per 16 MFMAs, which is what the matrix cores need to stay fed.  Only the blocks on or above the
// diagonal are computed; on a diagonal block the strictly-lower wave idles and both operands come from one tile.
// Output layout as gram_mfma_kernel's: G[row * ldg + col] for every 64 x 64 block on or above the diagonal.
constexpr int G2_BLK = 128, G2_RC = 16, G2_LD = 144;  // LD == 16 mod 32 doubles: conflict-free ds_read_b64 fragments
typedef double double2_t __attribute__((ext_vector_type(2)));
template <bool VEC2>  // VEC2: K even, so every row of A is 16-byte aligned and a thread moves pairs of columns
__global__ __launch_bounds__(256) void gram128_kernel(const double *__restrict__ A, const int64_t *__restrict__ n_off, int K,
                                                       int KB2, int ldg, double *__restrict__ G) {
    __shared__ __attribute__((aligned(16))) double sa[G2_RC][G2_LD];
    __shared__ __attribute__((aligned(16))) double sb[G2_RC][G2_LD];
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (8 private L2s) by their linear index, so the
    // index is re-read such that consecutive VIRTUAL indices — the blocks of one matrix, which walk the same cadences of
    // the same column panels at the same time — land on one XCD and share its L2.
    const int nb = gridDim.x, total = nb * gridDim.y, lin = blockIdx.x + nb * blockIdx.y, chunk = total >> 3;
    const int virt = lin < chunk * 8 ? (lin & 7) * chunk + (lin >> 3) : lin;
    int bi = 0, bj = virt % nb;
    while (bj >= KB2 - bi) {
        bj -= KB2 - bi;
        ++bi;
    }
    bj += bi;
    const int target = virt / nb;
    const int64_t lo = n_off[target];
    const int n = (int)(n_off[target + 1] - lo);
    A += lo * K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int i0 = bi * G2_BLK, j0 = bj * G2_BLK;
    const bool diag = bi == bj;
    const bool idle = diag && wi > wj;  // strictly-lower 64 x 64 sub-block of a diagonal block
    // 16-wide tiles of this wave that hold real columns (the last 128-block of K = 816 has 48 of 128): the others are
    // skipped — wave-uniform, so these are scalar branches around whole MFMAs
    const int na_live = min(4, max(0, (K - (i0 + wi * 64) + 15) >> 4)), nb_live = min(4, max(0, (K - (j0 + wj * 64) + 15) >> 4));
    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // stage = 16 cadences x 128 columns per operand = 4 pairs (VEC2) or 8 single values per thread and operand.  Loads
    // are unconditional on clamped addresses and masked afterwards (guarded loads compile to one branch per load).
    double2_t ra[4], rb[4];
    auto fetch = [&](int n0) {
        if (VEC2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = tid + 256 * q, r = e >> 6, c = (e & 63) * 2;
                const int nn = n0 + r;
                const size_t rowo = (size_t)max(0, min(nn, n - 1)) * K;  // n == 0: stay inside A (values are masked)
                const double2_t va = *reinterpret_cast<const double2_t *>(A + rowo + min(i0 + c, K - 2));
                const double2_t vb = *reinterpret_cast<const double2_t *>(A + rowo + min(j0 + c, K - 2));
                ra[q] = (nn < n && i0 + c < K) ? va : (double2_t){0.0, 0.0};
                rb[q] = (nn < n && j0 + c < K) ? vb : (double2_t){0.0, 0.0};
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = tid + 256 * (2 * q + h), r = e >> 7, c = e & 127;
                    const int nn = n0 + r;
                    const size_t rowo = (size_t)max(0, min(nn, n - 1)) * K;
                    const double va = A[rowo + min(i0 + c, K - 1)], vb = A[rowo + min(j0 + c, K - 1)];
                    ra[q][h] = (nn < n && i0 + c < K) ? va : 0.0;
                    rb[q][h] = (nn < n && j0 + c < K) ? vb : 0.0;
                }
        }
    };
    fetch(0);
    for (int n0 = 0; n0 < n; n0 += G2_RC) {
        __syncthreads();  // the previous stage's fragments have been read
        if (VEC2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = tid + 256 * q, r = e >> 6, c = (e & 63) * 2;
                *reinterpret_cast<double2_t *>(&sa[r][c]) = ra[q];
                *reinterpret_cast<double2_t *>(&sb[r][c]) = rb[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = tid + 256 * (2 * q + h), r = e >> 7, c = e & 127;
                    sa[r][c] = ra[q][h];
                    sb[r][c] = rb[q][h];
                }
        }
        __syncthreads();
        fetch(n0 + G2_RC);  // in flight while the matrix cores work (past the end: clamped, masked, unused)
        if (!idle && na_live > 0 && nb_live > 0) {
#pragma unroll
            for (int kk = 0; kk < G2_RC; kk += 4) {
                const int kr = kk + (lane >> 4), cc = lane & 15;
                double av[4], bv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    av[t] = sa[kr][wi * 64 + t * 16 + cc];
                    bv[t] = sb[kr][wj * 64 + t * 16 + cc];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (a < na_live && b < nb_live)
                            acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
        }
    }
    if (idle) return;
    double *Gt = G + (size_t)target * ldg * ldg;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wi * 64 + a * 16 + (lane >> 4) + 4 * r;
                const int col = j0 + wj * 64 + b * 16 + (lane & 15);
                if (row < ldg && col < ldg) Gt[(size_t)row * ldg + col] = acc[a][b][r];
            }
}

// plain Gram matrices G_b = A_b^T A_b of B row-major (N_b x K) blocks (no weights, no masks), for PCA (pld.hip)
int gram_plain_launch(const double *A, const int64_t *d_off, int B, int K, double *G, hipStream_t stream, lk_handle *h) {
    const int KB = (K + GR_BLK - 1) / GR_BLK;
    if (gram_tri_try(h, A, nullptr, nullptr, nullptr, nullptr, d_off, B, K, KB * GR_BLK, G, nullptr, stream))
        return KB * GR_BLK;
    constexpr int wide_min = 192;  // from this width the 128 x 128-tile kernel wins
    if (K >= wide_min) {
        const int KB2 = (K + G2_BLK - 1) / G2_BLK;
        if (K % 2 == 0)
            hipLaunchKernelGGL(gram128_kernel<true>, dim3(KB2 * (KB2 + 1) / 2, B), dim3(256), 0, stream, A, d_off, K, KB2,
                               KB * GR_BLK, G);
        else
            hipLaunchKernelGGL(gram128_kernel<false>, dim3(KB2 * (KB2 + 1) / 2, B), dim3(256), 0, stream, A, d_off, K, KB2,
                               KB * GR_BLK, G);
        return KB * GR_BLK;
    }
    hipLaunchKernelGGL(gram_mfma_kernel<false>, dim3(KB * (KB + 1) / 2, B), dim3(256), 0, stream, A, (const double *)nullptr,
                       (const double *)nullptr, (const uint8_t *)nullptr, (const uint8_t *)nullptr, d_off, K, KB, G);
    return KB * GR_BLK;  // leading dimension of each G_b
}

// Plain Gram of the matrix (double)(mode == 0 ? pix : pix / div[row]) - mean[col] formed from float32 pixels on the fly (narrow
// blocks only: returns 0 if the narrow kernel cannot take the shape, and the caller materialises the matrix instead).
int gram_plain_f32_launch(const float *pix, const float *div, const double *mean, int mode, const int64_t *d_off, int B, int K,
                          double *G, hipStream_t stream, lk_handle *h) {
    const int KB = (K + GR_BLK - 1) / GR_BLK;
    if (gram_tri_try(h, nullptr, nullptr, nullptr, nullptr, nullptr, d_off, B, K, KB * GR_BLK, G, nullptr, stream,
                     GramF32Source{pix, div, mean, mode}))
        return KB * GR_BLK;
    return 0;
}

// astropy.stats.sigma_clip(y, sigma, maxiters, cenfunc=median, stdfunc=std).mask for B ragged arrays — what
// LightCurve.remove_outliers (src/lightkurve/lightcurve.py:1430-1556) keeps its cadences by: 1 = clipped or non-finite.
int sigma_clip_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *y, double sigma, int maxiters,
                      uint8_t *outlier, hipStream_t stream) {
    LK_REQUIRE(B >= 0 && n_off_host != nullptr, "bad batch description");
    if (B == 0 || n_off_host[B] == 0) return LK_OK;
    LK_REQUIRE(y && outlier, "NULL buffer");
    LK_REQUIRE(maxiters >= 0, "maxiters must be >= 0");
    const size_t ntot = (size_t)n_off_host[B];
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 8 + ntot * 9 + 3 * 256 + 4096);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    double *d_zero = (double *)h->ws.alloc(ntot * 8);
    uint8_t *d_flag = (uint8_t *)h->ws.alloc(ntot);
    {
        const int rcs = h->stage.copy(d_off, n_off_host, (size_t)(B + 1) * 8, stream);
        if (rcs) return rcs;
    }
    LK_HIP_CHECK(hipMemsetAsync(d_zero, 0, ntot * 8, stream));
    LK_HIP_CHECK(hipMemsetAsync(outlier, 0, ntot, stream));
    hipLaunchKernelGGL(clip_kernel, dim3(B), dim3(1024), 0, stream, y, d_zero, d_off, sigma, maxiters, d_flag, outlier);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

int regress_launch(lk_handle *h, int B, const int64_t *n_off_host, int K, const double *X, const double *y,
                   const double *err, const uint8_t *cmask, const double *prior_mu, const double *prior_sigma,
                   double clip_sigma, int niters, double *w, double *model, uint8_t *outl, hipStream_t stream,
                   double *w_cov) {
    LK_REQUIRE(B >= 0 && n_off_host != nullptr, "bad batch description");
    if (B == 0) return LK_OK;
    LK_REQUIRE(K >= 1 && K <= REG_KMAX, "K=%d outside the supported range 1..%d", K, REG_KMAX);
    LK_REQUIRE(B <= 65535, "at most 65535 targets per call on this path (got %d): split the batch", B);
    LK_REQUIRE(X && y && w && model && outl, "NULL buffer");
    LK_REQUIRE((prior_mu == nullptr) == (prior_sigma == nullptr), "Please specify both `prior_mu` and `prior_sigma`");
    LK_REQUIRE(niters >= 1, "niters must be >= 1");
    LK_REQUIRE(n_off_host[0] == 0, "n_off[0] must be 0");
    for (int b = 0; b < B; ++b) {
        const int64_t n = n_off_host[b + 1] - n_off_host[b];
        LK_REQUIRE(n >= 1 && n < ((int64_t)1 << 30), "target %d has %lld cadences", b, (long long)n);
    }
    const size_t ntot = (size_t)n_off_host[B];
    const int KB = (K + 1 + GR_BLK - 1) / GR_BLK, Kp = KB * GR_BLK;
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 8 + (size_t)B * Kp * Kp * 8 + (size_t)B * K * (K + 1) * 8 * (w_cov ? 2 : 1) +
                           ntot + (size_t)B * 4 + (size_t)B * 4 * (256 + 1) + 4096);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    double *d_G = (double *)h->ws.alloc((size_t)B * Kp * Kp * 8);
    double *d_A = (double *)h->ws.alloc((size_t)B * K * (K + 1) * 8 * (w_cov ? 2 : 1));  // K x 2K per target for the inverse
    uint8_t *d_flag = (uint8_t *)h->ws.alloc(ntot);
    {
        const int rcs = h->stage.copy(d_off, n_off_host, (size_t)(B + 1) * 8, stream);
        if (rcs) return rcs;
    }
    LK_HIP_CHECK(hipMemsetAsync(outl, 0, ntot, stream));
    // per-target convergence flags of the clip loop: a pass that adds no outlier ends the target's loop
    constexpr bool early = true;
    int *d_done = early ? (int *)h->ws.alloc((size_t)B * 4) : nullptr;
    if (d_done) LK_HIP_CHECK(hipMemsetAsync(d_done, 0, (size_t)B * 4, stream));
    // the cadences each clip adds to the outlier set, for the DELTA Gram of the next pass
    constexpr int kNewCap = 256;
    int *d_newcnt = (int *)h->ws.alloc((size_t)B * 4);
    int *d_newidx = (int *)h->ws.alloc((size_t)B * kNewCap * 4);
    LK_REQUIRE(d_newcnt && d_newidx, "workspace exhausted");
    const int nblk = KB * (KB + 1) / 2;
    for (int it = 0; it < niters; ++it) {
        if (it == 0) {
            if (!gram_tri_try(h, X, y, err, cmask, outl, d_off, B, K, Kp, d_G, (const int *)d_done, stream))
                hipLaunchKernelGGL(gram_mfma_kernel<false>, dim3(nblk, B), dim3(256), 0, stream, X, y, err, cmask, outl, d_off,
                                   K, KB, d_G, (const int *)d_done);
        }
        else
            hipLaunchKernelGGL(gram_mfma_kernel<true>, dim3(nblk, B), dim3(256), 0, stream, X, y, err, cmask, outl, d_off, K,
                               KB, d_G, (const int *)d_done, (const int *)d_newcnt, (const int *)d_newidx, kNewCap);
        const size_t solve_lds = ((size_t)K * (K + 1) + 2 * K + 2 * SOLVE_NW + 2) * 8;  // system | x | pivot exchange | prior terms
        if (solve_lds <= 160 * 1024) {
            {
                const int rc_ = want_lds(h, reinterpret_cast<const void *>(solve_lds_kernel), 160 * 1024);
                if (rc_) return rc_;
            }
            hipLaunchKernelGGL(solve_lds_kernel, dim3(B), dim3(SOLVE_NT), solve_lds, stream, d_G, K, Kp, prior_mu, prior_sigma, w,
                               (const int *)d_done);
        } else {
            hipLaunchKernelGGL(solve_kernel, dim3(B), dim3(256), 0, stream, d_G, K, Kp, prior_mu, prior_sigma, d_A, w,
                               (const int *)d_done);
        }
        hipLaunchKernelGGL(model_kernel, dim3(64, B), dim3(256), (size_t)K * 8, stream, X, w, d_off, K, model,
                           (const int *)d_done);
        hipLaunchKernelGGL(clip_kernel, dim3(B), dim3(1024), 0, stream, y, model, d_off, clip_sigma, 5, d_flag, outl, d_done,
                           d_newcnt, d_newidx, kNewCap);
    }
    hipLaunchKernelGGL(demedian_kernel, dim3(B), dim3(1024), 0, stream, d_off, model);
    // d_G still holds the normal matrix of the LAST iteration's fit: its inverse is the coefficient covariance
    if (w_cov) hipLaunchKernelGGL(invert_kernel, dim3(B), dim3(256), 0, stream, d_G, K, Kp, prior_sigma, d_A, w_cov);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ a sub-matrix's share of the model
// out[b][n] = sum_{c0 <= k < c1} X[b][n][k] w[b][k] for B same-shaped matrices: the `diagnostic_lightcurves[name]` of one
// sub-matrix of the collection (reference regressioncorrector.py:281-307, before its median is taken off) — PLDCorrector's
// restore_trend needs the spline block's (pldcorrector.py:418-420).  One wavefront per cadence row, lanes over the columns.
__global__ __launch_bounds__(256) void model_part_kernel(const double *__restrict__ X, const double *__restrict__ w, int N, int K,
                                                          int c0, int c1, double *__restrict__ out) {
    const int target = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *wt = w + (size_t)target * K;
    for (int r = blockIdx.x * 4 + wave; r < N; r += gridDim.x * 4) {
        const double *row = X + ((size_t)target * N + r) * K;
        double acc = 0.0;
        for (int k = c0 + lane; k < c1; k += 64) acc = fma(row[k], wt[k], acc);
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if (lane == 0) out[(size_t)target * N + r] = acc;
    }
}

int model_part_launch(lk_handle *h, int B, int N, int K, int c0, int c1, const double *X, const double *w, double *out,
                      hipStream_t stream) {
    (void)h;
    LK_REQUIRE(B >= 1 && B <= 65535 && N >= 1 && K >= 1, "bad shapes");
    LK_REQUIRE(0 <= c0 && c0 <= c1 && c1 <= K, "column range [%d, %d) outside 0..%d", c0, c1, K);
    LK_REQUIRE(X && w && out, "NULL buffer");
    hipLaunchKernelGGL(model_part_kernel, dim3(64, B), dim3(256), 0, stream, X, w, N, K, c0, c1, out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ DesignMatrix.standardize
// correctors/designmatrix.py:215-250: per column, zeros count as missing; subtract the nanmedian and divide by the nanstd of
// the remaining values, missing values come back as 0; a column whose values are all equal (nanstd == 0) is left unchanged.
// One workgroup per (column, matrix); the column is read through a strided view (no staging copy).
__global__ __launch_bounds__(256) void dm_standardize_kernel(const double *__restrict__ A, int N, int P, double *__restrict__ out) {
    __shared__ unsigned long long sh[264];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const double *col = A + (size_t)b * N * P + c;
    double *oc = out + (size_t)b * N * P + c;
    auto val = [&](int i) { return col[(size_t)i * P]; };
    auto keep = [&](int i) {
        const double v = col[(size_t)i * P];
        return v == v && v != 0.0;
    };
    long long cnt = 0;
    double sum = 0.0;
    for (int i = tid; i < N; i += 256)
        if (keep(i)) {
            ++cnt;
            sum += val(i);
        }
    const long long count = block_count_dyn(cnt, reinterpret_cast<long long *>(sh));
    if (count == 0) {  // nothing but zeros / NaN: every entry is "missing" -> 0
        for (int i = tid; i < N; i += 256) oc[(size_t)i * P] = 0.0;
        return;
    }
    const double mean = block_sum_dyn(sum, reinterpret_cast<double *>(sh)) / (double)count;
    double ss = 0.0;
    for (int i = tid; i < N; i += 256)
        if (keep(i)) {
            const double d = val(i) - mean;
            ss = fma(d, d, ss);
        }
    const double sd = sqrt(block_sum_dyn(ss, reinterpret_cast<double *>(sh)) / (double)count);  // numpy nanstd (ddof = 0)
    if (sd == 0.0) {  // constant column: unchanged (its zeros were "missing" and come back as zeros)
        for (int i = tid; i < N; i += 256) {
            const double v = val(i);
            oc[(size_t)i * P] = v == v ? v : 0.0;
        }
        return;
    }
    const double med = block_median(N, count, val, keep, sh);
    for (int i = tid; i < N; i += 256) oc[(size_t)i * P] = keep(i) ? (val(i) - med) / sd : 0.0;
}

int dm_standardize_launch(lk_handle *h, int B, int N, int P, const double *A, double *out, hipStream_t stream) {
    LK_REQUIRE(B >= 1 && N >= 1 && P >= 1, "bad shapes");
    LK_REQUIRE(B <= 65535, "at most 65535 matrices per call (grid.y; got %d): split the batch", B);
    LK_REQUIRE(A && out, "NULL buffer");
    (void)h;
    hipLaunchKernelGGL(dm_standardize_kernel, dim3(P, B), dim3(256), 0, stream, A, N, P, out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
