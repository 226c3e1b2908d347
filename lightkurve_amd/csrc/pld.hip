// pld.hip — PLDCorrector.create_design_matrix on gfx950 (reference: src/lightkurve/correctors/pldcorrector.py:186-287,
// DesignMatrix.pca correctors/designmatrix.py:252-282, create_spline_matrix :952-997).
//
// For a batch of B same-shaped cutouts (N cadences, P PLD pixels, Pb background pixels) build
//     X = [ pld_order_1 | ... | pld_order_o | background | spline | 1 ]      (N x K, float64)
// where every PLD/background block is the PCA (top-k left singular vectors of the column-centred matrix) of
//   order 1: pixel flux / SAP flux;  order n: all n-fold products of the order-1 components;  background pixels.
//
// PCA = Gram + eigen.  Pixel and background blocks: C = A^T A on the fp64 matrix cores (gram_mfma_kernel; gram128_kernel
// — 128 x 128 output blocks, 4 x 4 MFMA tiles per wave — from 192 columns), A written already centred
// (pld_colmean_kernel + pld_ratio_kernel).  Product blocks (order >= 2, at least 100 columns): the products are never
// materialised; their Gram matrix is expanded from the canonical staircase of the 2o-th moments of the first-order
// components (pld_moment_gram_kernel + pld_moment_expand_kernel, see there: 6.4 x fewer MFMAs at k = 16, o = 3) and the
// projection generates them on the fly (pld_project_products_kernel).  Top-k eigenpairs of C by blocked subspace
// iteration with Rayleigh-Ritz.  Every dense step of the
// iteration is MFMA work in one workgroup per matrix: C Q (32-byte row loads two steps ahead of the matrix cores), the
// skinny products Q^T Z and X M, Ritz vectors + residual in one pass; three products with C per Ritz step (eight for the
// mid-size product blocks, whose flat spectrum tolerates it), column-scaled Cholesky-QR between steps and for the random
// start (SVQB as the fallback; small l x l eigenproblems by parallel cyclic Jacobi in LDS).  Pixel blocks of at
// most 138 columns whose iteration does not converge in eight steps fall through to a direct Jacobi on C held in LDS;
// for P <= 64 the Jacobi always runs on C itself.
// Then U = A V diag(lambda)^-1/2 (pld_project_kernel, MFMA) written straight into X.  The reference uses fbpca
// (randomised range finder, 10 power iterations, 2 oversampling columns, unseeded RNG => not reproducible); the oracle
// uses an exact SVD; the iteration here converges the residual ||C r - theta r|| to 1e-10 theta_max, i.e. to ten digits
// wherever the spectrum has a gap.  PCA bases are only defined up to rotation inside a block and the
// regression is invariant to it, so parity is stated on the corrected flux.
// The order-1 block skips the reference's redundant re-PCA of an already orthonormal basis (same subspace).
#include <algorithm>
#include <type_traits>
#include <mutex>
#include <map>
#include <string>
#include <vector>

#include "block_select.hpp"
#include "lk_common.hpp"

namespace lk {

constexpr int PLD_LMAX = 64;  // largest small eigenproblem kept in LDS
constexpr int PLD_DIRECT_MAX = 138;  // largest P whose Gram matrix fits LDS for the direct Jacobi (512 threads)
// Stop of the subspace iteration: ||C r - theta r|| <= tol * theta_max * sqrt(k).  PLD design matrices: 1e-7 — the residual falls
// 4.8 -> 1.5e-2 -> 1.5e-5 -> 8.5e-9 -> 3.2e-12 per Rayleigh-Ritz step on the 816-column blocks, and the corrected flux / the outlier
// masks of every reference golden are UNCHANGED for any threshold down to 1e-6 (profiles/r05_pld_tol_sweep.txt: 1.5e-8 / 4.7e-8 /
// 1.4e-8 at 1e-10 ... 1e-6, first movement — 2.2e-7 on pld_k2sin_order3 — at 1e-5; masks identical to 1e-3): the fifth step
// bought digits nothing downstream sees.  Stated parity 1e-6 on the corrected flux.  The standalone PCA (lk_pca_batch /
// DesignMatrix.pca), whose OUTPUT is the basis, keeps 1e-10.
#define PLD_EIG_TOL 1e-7
#define PCA_EIG_TOL 1e-10
#ifndef PLD_F32_RES
#define PLD_F32_RES 1e-5  // relative residual above which the Chebyshev filter reads the float32 copy of C (1e-3 until round 5: with the
                          // stop at 1e-7 the last filter step may read it too — same steps, same corrected flux, -0.4 ms per 500 cutouts)
#endif
constexpr int PLD_KC = 64;    // rows of the basis staged in LDS per step of the MFMA product C Q (64 x 66 doubles also hold
                              // eig_xty's 16 partial tiles; 128 rows left no room for a second workgroup on the CU)
constexpr int PLD_QS = PLD_LMAX + 2;  // LDS row stride of that stage (doubles)
// row stride (floats) of the float32 stage of eig_cq32 for a basis of NA 16-column tiles: 16 NA + 16, so the four k-rows of an
// MFMA operand read start 16 banks apart and the 64 lanes hit 64 different banks
__host__ __device__ constexpr int pld_qs32(int na) { return 16 * na + 16; }
// Relative residual above which the Rayleigh-Ritz product itself reads the float32 copy of C on the float32 matrix cores (round 6):
// a step that starts three decades above the float32 noise floor cannot pass the 1e-7 stop, so its Ritz pairs only have to
// steer the filter; convergence is only ever declared from a float64 product.
#ifndef PLD_RR32_RES
#define PLD_RR32_RES 1e-3
#endif
typedef double pld_d4 __attribute__((ext_vector_type(4)));

static int ncombos(int k, int order) {  // C(k + order - 1, order)
    long long r = 1;
    for (int i = 1; i <= order; ++i) r = r * (k + i - 1) / i;
    return (int)r;
}

int pld_design_width(int P, int Pb, int pld_order, int pca_components, int n_knots) {
    int K = 0;
    if (P > 0 && pld_order > 0) {
        const int k1 = std::min(pca_components, P);
        for (int o = 1; o <= pld_order; ++o) K += std::min(pca_components, ncombos(k1, o));
    }
    K += std::min(pca_components, Pb);
    return K + n_knots + 1;
}

// ------------------------------------------------------------------------------------------------ elementwise
// out[b][n][p] = (double)( pix[b][n][p] / div[b][n] ) with the division in float32 like numpy's float32 / float32;
// div = SAP flux (PLD pixels) or the float32 row sum of the background pixels (normalize) or 1.
__global__ __launch_bounds__(256) void pld_ratio_kernel(const float *__restrict__ pix, const float *__restrict__ lc,
                                                         int mode, int N, int P, double *__restrict__ out,
                                                         const double *__restrict__ colmean,
                                                         const float *__restrict__ rowdiv) {
    constexpr int ROWS = 32;  // cadences per workgroup: 32 x P contiguous floats in, 32 x P contiguous doubles out
    __shared__ float div[ROWS];
    const int b = blockIdx.y, n0 = blockIdx.x * ROWS, tid = threadIdx.x;
    const int nr = min(ROWS, N - n0);
    if (mode == 2 && rowdiv) {
        if (tid < nr) div[tid] = rowdiv[(size_t)b * N + n0 + tid];  // pld_rowdiv_kernel's sums
    } else if (mode == 2) {
        // np.nansum over the float32 pixels of a cadence (accumulated in double, rounded once: <= 1 ulp(f32) from numpy's
        // pairwise float32 sum).  A wave per cadence, lanes over pixels; the per-lane partials are added in lane order.
        const int wave = tid >> 6, lane = tid & 63;
        for (int r = wave; r < nr; r += 4) {
            const float *row = pix + ((size_t)b * N + n0 + r) * P;
            double sm = 0.0;
            for (int p = lane; p < P; p += 64) {
                const float v = row[p];
                if (v == v) sm += (double)v;
            }
            for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
            if (lane == 0) div[r] = (float)sm;
        }
    } else if (tid < nr) {
        div[tid] = mode == 1 ? lc[(size_t)b * N + n0 + tid] : 1.0f;
    }
    __syncthreads();
    const size_t base = ((size_t)b * N + n0) * P;
    const double *cm = colmean ? colmean + (size_t)b * P : nullptr;  // column means (pld_colmean_kernel): write A centred
    // two row phases x 128 columns: no integer division per element, a thread's column mean loaded once per column chunk
    const int cx = tid & 127, ry = tid >> 7;
    for (int c0 = 0; c0 < P; c0 += 128) {
        const int c = c0 + cx;
        if (c >= P) continue;
        const double m = cm ? cm[c] : 0.0;
        for (int r = ry; r < nr; r += 2) {
            const float v = pix[base + (size_t)r * P + c];
            const double q = (double)(mode == 0 ? v : v / div[r]);
            out[base + (size_t)r * P + c] = cm ? q - m : q;
        }
    }
}

// float32 row sums of the background pixels (the divisor of mode 2 above, same arithmetic), one wave per cadence
__global__ __launch_bounds__(256) void pld_rowdiv_kernel(const float *__restrict__ pix, int N, int P, float *__restrict__ div) {
    // SIXTEEN lanes per cadence (round 6; a wave per cadence spent six 64-wide shuffle steps on 121 pixels — 2 loads per lane — and the
    // kernel ran at 2.5 TB/s): a lane's pixels l, l + 16, ... are requested together, summed in double in that order, then four
    // xor steps inside the group (the partial sums meet in a fixed order: reproducible; <= 1 ulp(f32) from numpy's pairwise sum).
    const int b = blockIdx.y, grp = threadIdx.x >> 4, l = threadIdx.x & 15;
#pragma unroll 1
    for (int q = 0; q < 2; ++q) {  // 32 cadences per workgroup, 16 at a time
        const int n = blockIdx.x * 32 + q * 16 + grp;
        if (n >= N) return;  // (whole 16-lane groups leave together; shuffles below stay inside a group)
        const float *row = pix + ((size_t)b * N + n) * P;
        double sm = 0.0;
        for (int p0 = l; p0 < P; p0 += 128) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[min(p0 + 16 * u, P - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + 16 * u < P && v[u] == v[u]) sm += (double)v[u];
        }
        for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
        if (l == 0) div[(size_t)b * N + n] = (float)sm;
    }
}

// Column means of the ratio matrix straight from the float32 pixels (one workgroup per cutout), so that pld_ratio_kernel
// can write A already centred: the centring pass over the float64 matrix (read, read, write: 1.2 ms per block at
// configs[4]) becomes one read of the float32 input.  Eight row phases per column summed in the order pld_center_kernel
// uses, the same float32 division: the means, hence A, are bit-identical to the two-pass form.
__global__ __launch_bounds__(1024) void pld_colmean_kernel(const float *__restrict__ pix, const float *__restrict__ divv,
                                                            int mode, int N, int P, double *__restrict__ mean) {
    __shared__ double sh[8][129];
    const int b = blockIdx.x, cx = threadIdx.x & 127, ry = threadIdx.x >> 7;
    const float *pb = pix + (size_t)b * N * P;
    const float *db = divv ? divv + (size_t)b * N : nullptr;
    for (int c0 = 0; c0 < P; c0 += 128) {
        const int c = c0 + cx;
        double s = 0.0;
        if (c < P) {
            int n = ry;
            for (; n + 24 < N; n += 32) {  // four loads in flight
                const float v0 = pb[(size_t)n * P + c], v1 = pb[(size_t)(n + 8) * P + c], v2 = pb[(size_t)(n + 16) * P + c],
                            v3 = pb[(size_t)(n + 24) * P + c];
                if (mode == 0) {
                    s += (double)v0;
                    s += (double)v1;
                    s += (double)v2;
                    s += (double)v3;
                } else {
                    const float d0 = db[n], d1 = db[n + 8], d2 = db[n + 16], d3 = db[n + 24];
                    s += (double)(v0 / d0);
                    s += (double)(v1 / d1);
                    s += (double)(v2 / d2);
                    s += (double)(v3 / d3);
                }
            }
            for (; n < N; n += 8) {
                const float v = pb[(size_t)n * P + c];
                s += (double)(mode == 0 ? v : v / db[n]);
            }
        }
        __syncthreads();
        sh[ry][cx] = s;
        __syncthreads();
        if (ry == 0 && c < P) {
            double t = 0.0;
            for (int r = 0; r < 8; ++r) t += sh[r][cx];
            mean[(size_t)b * P + c] = t / (double)N;
        }
    }
}

// column means of A_b (N x P) subtracted in place; one workgroup per (column tile, b)
__global__ __launch_bounds__(256) void pld_center_kernel(double *__restrict__ A, int N, int P) {
    __shared__ double sh[8][33];
    const int b = blockIdx.y, c0 = blockIdx.x * 32;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    double *Ab = A + (size_t)b * N * P;
    const int c = c0 + cx;
    double s = 0.0;
    if (c < P)
        for (int n = ry; n < N; n += 8) s += Ab[(size_t)n * P + c];
    sh[ry][cx] = s;
    __syncthreads();
    if (ry == 0) {
        double t = 0.0;
        for (int r = 0; r < 8; ++r) t += sh[r][cx];
        sh[0][cx] = t / (double)N;
    }
    __syncthreads();
    const double mean = sh[0][cx];
    if (c < P)
        for (int n = ry; n < N; n += 8) Ab[(size_t)n * P + c] -= mean;
}

// all `order`-fold products (combinations with replacement, lexicographic: itertools order, as the reference's
// multichoose loop) of the k columns of U = X[:, col0 : col0 + k].  comb[idx * order + pos] = column of factor pos of
// product idx (built once per (k, order) on the host).  Two kernels: the column means of the products (nothing is
// written), then the CENTRED products — the PCA input — in one pass over A.
constexpr int PM_ROWS = 128;  // rows of U staged per step of the products-mean kernel
__global__ __launch_bounds__(1024) void pld_products_mean_kernel(const double *__restrict__ X, int ldx, int col0, int k,
                                                                  int order, int N, int Pc,
                                                                  const uint8_t *__restrict__ comb,
                                                                  double *__restrict__ mean) {
    // workgroup = 256 products x 4 row phases; the rows of U go through LDS in chunks, the four phase sums are added in
    // a fixed order (the one-thread-per-product version walked all N rows serially with strided global loads)
    __shared__ double us[PM_ROWS * 49];  // k <= 48 components, row stride k | 1
    __shared__ double red[4][256];
    const int b = blockIdx.y, t = threadIdx.x & 255, g = threadIdx.x >> 8, idx = blockIdx.x * 256 + t;
    const int ks = k | 1;  // odd row stride
    int a[4] = {0, 0, 0, 0};
    if (idx < Pc)
        for (int pos = 0; pos < order; ++pos) a[pos] = comb[idx * order + pos];
    const double *u = X + (size_t)b * N * ldx + col0;
    double s = 0.0;
    for (int r0 = 0; r0 < N; r0 += PM_ROWS) {
        __syncthreads();
        for (int e = threadIdx.x; e < PM_ROWS * k; e += 1024) {
            const int r = e / k, c = e - r * k;
            us[r * ks + c] = r0 + r < N ? u[(size_t)(r0 + r) * ldx + c] : 0.0;
        }
        __syncthreads();
        const int rend = min(PM_ROWS, N - r0);
        for (int r = g; r < rend; r += 4) {
            const double *rr = us + r * ks;
            double prod = rr[a[0]];
            for (int pos = 1; pos < order; ++pos) prod *= rr[a[pos]];
            s += prod;
        }
    }
    red[g][t] = s;
    __syncthreads();
    if (g == 0 && idx < Pc) mean[(size_t)b * Pc + idx] = (((red[0][t] + red[1][t]) + red[2][t]) + red[3][t]) / (double)N;
}

constexpr int PP_ROWS = 16;  // cadences per workgroup of the products kernel
__global__ __launch_bounds__(256) void pld_products_kernel(const double *__restrict__ X, int ldx, int col0, int k, int order,
                                                            int N, int Pc, const uint8_t *__restrict__ comb,
                                                            const double *__restrict__ mean, double *__restrict__ out) {
    // 16 cadences per workgroup: a product's factor indices and mean are fetched once per 16 outputs (the 4-row version
    // spent more on those fetches than on the 3.6 GB it writes: 3.2 ms against 0.6 ms of HBM time at P = 816)
    __shared__ double us[PP_ROWS][65];
    const int b = blockIdx.y, n0 = blockIdx.x * PP_ROWS;
    for (int e = threadIdx.x; e < PP_ROWS * k; e += 256) {
        const int r = e / k, c = e - r * k;
        us[r][c] = n0 + r < N ? X[((size_t)b * N + n0 + r) * ldx + col0 + c] : 0.0;
    }
    __syncthreads();
    const int nr = min(PP_ROWS, N - n0);
    for (int idx = threadIdx.x; idx < Pc; idx += 256) {
        int a[4] = {0, 0, 0, 0};
        for (int pos = 0; pos < order; ++pos) a[pos] = comb[idx * order + pos];
        const double m = mean[(size_t)b * Pc + idx];
        double *o = out + ((size_t)b * N + n0) * Pc + idx;
#pragma unroll 4
        for (int r = 0; r < nr; ++r) {
            double prod = us[r][a[0]];
            for (int pos = 1; pos < order; ++pos) prod *= us[r][a[pos]];
            o[(size_t)r * Pc] = prod - m;
        }
    }
}

// ------------------------------------------------------------------------------------------------ moment-form Gram
// The Gram matrix of an order-o product block never needs the N x Pc matrix of products: its entry for the columns
// (a1..ao), (b1..bo) is the 2o-th moment sum_n u_a1 .. u_ao u_b1 .. u_bo of the k first-order components, which depends
// only on the MULTISET of the 2o factors.  Every multiset has one canonical split — its o smallest factors | its o
// largest — so it suffices to compute the pairs (row tuple x, column tuple y) with max(x) <= min(y): with the rows
// ordered by their largest factor and the columns in natural (lexicographic = smallest-factor-first) order, that is a
// staircase of 54 264 entries for k = 16, o = 3 (71 168 in whole 16 x 16 MFMA tiles) against the 458 752 of the upper
// 128 x 128 blocks of the 816 x 816 matrix — 6.4 x fewer matrix-core instructions, and the 22.8 MB per cutout of
// materialised products are neither written nor read.  Operands are generated on the fly from a 64-cadence stage of U in
// LDS (o reads and o - 1 multiplications per element).  A wave owns up to 4 x 4 tiles (a "wave tile" of the host-built
// list: first row in row order, first column, 16-bit mask of the tiles that hold canonical pairs).
constexpr int MG_CH = 64;  // cadences per LDS stage
// Round 6, the masked launch (profiles/r06_pld_moment_gram.txt: 39 % of its wave cycles parked, 46 exec-mask branches per step):
//   MG_UNIFORM   the wave-tile descriptor is read through readfirstlane — the compiler could not see through `threadIdx.x >> 6` that
//                a wave's 64 lanes share it, and guarded every MFMA and every operand row with s_and_saveexec + s_cbranch_execz;
//   MG_ALL_ROWS  all 4 + 4 operand rows of a wave tile are generated, needed or not: the reads of a row no tile uses sat behind a
//                branch of their own, each with its wait — 24 unconditional ds_read_b64 per step cost less than the serialised few.
// A/B on one box, PLD step of 500 cutouts, 3 interleaved reps: neither 29.49 ms | ALL_ROWS 28.67 | UNIFORM 29.33 | both 28.59.
#ifndef MG_UNIFORM
#define MG_UNIFORM 1
#endif
#ifndef MG_ALL_ROWS
#define MG_ALL_ROWS 1
#endif
constexpr int kMomentMinCols = 100;  // product blocks at least this wide take the moment form
template <int O, bool FULL, int PRE, int KS>  // KS: LDS row stride (k | 1) when known at compile time, else 0; PRE: MG_CH * k / 256 elements of the next stage wait in registers (4: k <= 16, 12: k <= 48)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pld_moment_gram_kernel(
    const double *__restrict__ X, int ldx, int col0, int k, int N, int Pc, int ldm, const uint8_t *__restrict__ rcomb,
    const uint8_t *__restrict__ comb, const int4 *__restrict__ wt, int nwt, double *__restrict__ Mcan,
    const int *__restrict__ rperm, double *__restrict__ mean) {
    extern __shared__ __attribute__((aligned(16))) double mg_us[];  // 2 x MG_CH x ks
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, lq = lane >> 4, lr = lane & 15;
#if MG_UNIFORM
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int wave = tid >> 6;
#endif
    const int ks = KS > 0 ? KS : (k | 1);
    const int w = blockIdx.x * 4 + wave;
    const int4 t = wt[min(w, nwt - 1)];
#if MG_UNIFORM
    // the wave tile is the same for the 64 lanes of a wave: say so (the compiler cannot see it through `threadIdx.x >> 6` and
    // guarded each MFMA and each operand row with an exec-mask branch — s_and_saveexec + s_cbranch_execz, 46 per step)
    const int r0 = __builtin_amdgcn_readfirstlane(t.x), c0 = __builtin_amdgcn_readfirstlane(t.y);
    const unsigned mask = w < nwt ? (unsigned)__builtin_amdgcn_readfirstlane(t.z) : 0u;
#else
    const int r0 = t.x, c0 = t.y;
    const unsigned mask = w < nwt ? (unsigned)t.z : 0u;
#endif
    // bit i: this wave tile is the one that also sums the products of row tile i over the cadences (their column means:
    // every (row, cadence) pair passes through exactly one lane of the A operand)
#if MG_UNIFORM
    const unsigned mflag = (!FULL && w < nwt) ? (unsigned)__builtin_amdgcn_readfirstlane(t.w) : 0u;  // (the host keeps such wave tiles out of the FULL launch)
#else
    const unsigned mflag = (!FULL && w < nwt) ? (unsigned)t.w : 0u;  // (the host keeps such wave tiles out of the FULL launch)
#endif
    double msum[4] = {0.0, 0.0, 0.0, 0.0};
    // rows / columns past the end are clamped, not zeroed: their accumulator entries are simply never stored
    int ia[4][O], ib[4][O];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 16 * i + lr, c = c0 + 16 * i + lr;
#pragma unroll
        for (int pos = 0; pos < O; ++pos) {
            ia[i][pos] = rcomb[(size_t)min(r, Pc - 1) * O + pos];
            ib[i][pos] = comb[(size_t)min(c, Pc - 1) * O + pos];
        }
    }
    pld_d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = pld_d4{0.0, 0.0, 0.0, 0.0};
    const double *u = X + (size_t)b * N * ldx + col0;
    double pre[PRE];
    auto fetch = [&](int n0) {
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
            const int e = tid + 256 * q;
            const int r = e / k, c = e - r * k;
            pre[q] = (e < MG_CH * k && n0 + r < N) ? u[(size_t)(n0 + r) * ldx + c] : 0.0;
        }
    };
    auto park = [&](double *buf) {
#pragma unroll
        for (int q = 0; q < PRE; ++q) {
            const int e = tid + 256 * q;
            const int r = e / k, c = e - r * k;
            if (e < MG_CH * k) buf[r * ks + c] = pre[q];
        }
    };
    fetch(0);
    park(mg_us);
    __syncthreads();
    // byte offsets of the factors inside a stage row; rows / columns that no needed tile touches are never generated
    int oa[4][O], ob[4][O];
    bool ua[4], ub[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ua[i] = FULL || MG_ALL_ROWS || (mask & (0xfu << (4 * i))) != 0;
        ub[i] = FULL || MG_ALL_ROWS || (mask & (0x1111u << i)) != 0;
#pragma unroll
        for (int pos = 0; pos < O; ++pos) {
            oa[i][pos] = ia[i][pos] * 8;
            ob[i][pos] = ib[i][pos] * 8;
        }
    }
    // Software pipeline: the 8 x O LDS reads of step s + 1 are issued before the 16 MFMAs of step s and multiplied out
    // after them (left to itself the compiler chains read -> wait -> multiply 17 times per step: 1700 cycles of exposed
    // LDS latency next to 1024 cycles of matrix-core work).
    double ra[4][O], rb[4][O], av[4], bv[4];
    auto issue = [&](const char *rowp) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (ua[i]) {
#pragma unroll
                for (int pos = 0; pos < O; ++pos) ra[i][pos] = *reinterpret_cast<const double *>(rowp + oa[i][pos]);
            }
            if (ub[i]) {
#pragma unroll
                for (int pos = 0; pos < O; ++pos) rb[i][pos] = *reinterpret_cast<const double *>(rowp + ob[i][pos]);
            }
        }
    };
    auto multiply = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double pa = ra[i][0], pb = rb[i][0];
#pragma unroll
            for (int pos = 1; pos < O; ++pos) {
                pa *= ra[i][pos];
                pb *= rb[i][pos];
            }
            av[i] = pa;
            bv[i] = pb;
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int pos = 0; pos < O; ++pos) ra[i][pos] = rb[i][pos] = 0.0;
    int cur = 0;
    for (int n0 = 0; n0 < N; n0 += MG_CH) {
        const bool more = n0 + MG_CH < N;
        if (more) fetch(n0 + MG_CH);
        const char *stage = reinterpret_cast<const char *>(mg_us + cur * MG_CH * ks) + lq * ks * 8;
        issue(stage);
        multiply();
#pragma unroll 1
        for (int sst = 0; sst < MG_CH / 4; ++sst) {
            __builtin_amdgcn_sched_barrier(0);
            if (sst + 1 < MG_CH / 4) issue(stage + (sst + 1) * 4 * ks * 8);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (FULL || (mask & (1u << (4 * i + j))))
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
            if (!FULL && mflag) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (mflag & (1u << i)) msum[i] += av[i];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (sst + 1 < MG_CH / 4) multiply();
        }
        if (more) park(mg_us + (cur ^ 1) * MG_CH * ks);
        __syncthreads();
        cur ^= 1;
    }
    if (!FULL && mflag) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (mflag & (1u << i)) {
                double m = msum[i];
                m += __shfl_xor(m, 16);
                m += __shfl_xor(m, 32);
                const int r = r0 + 16 * i + lr;
                if (lq == 0 && r < Pc) mean[(size_t)b * Pc + rperm[r]] = m / (double)N;
            }
    }
    double *Mb = Mcan + (size_t)b * ldm * ldm;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (mask & (1u << (4 * i + j))) {
                const int col = c0 + 16 * j + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rowi = r0 + 16 * i + lq + 4 * r;
                    if (rowi < Pc && col < Pc) Mb[(size_t)rowi * ldm + col] = acc[i][j][r];
                }
            }
}

// G[i][j] = moment of the merged multiset of columns i and j (looked up through the host-built index table) minus
// N mean_i mean_j: the Gram matrix of the CENTRED products, written in full (both triangles).
#ifndef MEXP_U
#define MEXP_U 8
#endif
__global__ __launch_bounds__(256) void pld_moment_expand_kernel(const double *__restrict__ Mcan, size_t mstride,
                                                                 const uint32_t *__restrict__ src,
                                                                 const double *__restrict__ mean, int Pc, int ldg, double Nd,
                                                                 double *__restrict__ G, float *__restrict__ G32 = nullptr) {
    // a thread writes MEXP_U groups of four neighbouring columns (16-byte reads of the index table, 32-byte stores); all
    // groups' index reads are issued before the gathers of any (the kernel is a chain index -> gather -> store: one
    // group per thread left it at 2 TB/s of stores, two at 3)
    const int b = blockIdx.y;
    const int q4 = (Pc + 3) >> 2, tot = Pc * q4;
    constexpr int U = MEXP_U;
    const int e0 = blockIdx.x * (256 * U) + threadIdx.x;  // the thread's groups: e0, e0 + 256, ... (full-density wave stores)
    if (e0 >= tot) return;
    const double *mb = mean + (size_t)b * Pc, *Mb = Mcan + (size_t)b * mstride;
    double *Gb = G + (size_t)b * ldg * ldg;
    if ((Pc & 3) == 0) {
        int ii[U], jj[U];
        uint4 sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + 256 * u, tot - 1);
            ii[u] = e / q4;
            jj[u] = (e - ii[u] * q4) * 4;
            sv[u] = *reinterpret_cast<const uint4 *>(src + (size_t)ii[u] * Pc + jj[u]);
        }
        double v[U][4], mj[U][4], mi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u][0] = Mb[sv[u].x];
            v[u][1] = Mb[sv[u].y];
            v[u][2] = Mb[sv[u].z];
            v[u][3] = Mb[sv[u].w];
            mi[u] = Nd * mb[ii[u]];
#pragma unroll
            for (int t = 0; t < 4; ++t) mj[u][t] = mb[jj[u] + t];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e0 + 256 * u < tot) {
                pld_d4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = v[u][t] - mi[u] * mj[u][t];
                *reinterpret_cast<pld_d4 *>(Gb + (size_t)ii[u] * ldg + jj[u]) = o;
                if (G32)  // the float32 copy the early filter products of the eigen-solver stream instead
                    *reinterpret_cast<float4 *>(G32 + (size_t)b * ldg * ldg + (size_t)ii[u] * ldg + jj[u]) =
                        make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
            }
    } else {
        for (int u = 0; u < U && e0 + 256 * u < tot; ++u) {
            const int e = e0 + 256 * u, i = e / q4, j0 = (e - i * q4) * 4;
            const double mi = Nd * mb[i];
            const uint32_t *sp = src + (size_t)i * Pc + j0;
            double *g = Gb + (size_t)i * ldg + j0;
            for (int t = 0; t < 4 && j0 + t < Pc; ++t) g[t] = Mb[sp[t]] - mi * mb[j0 + t];
        }
    }
}

// The same expansion by SYMMETRIC 64 x 64 tiles (Pc a multiple of 4): the kernel above is bound by its gathers — 64 divergent 8-byte
// reads per instruction keep the texture addresser busy a cycle per lane (contiguous reads in their place: 680 -> 381 us per launch) —
// and C is symmetric.  A workgroup takes a tile pair (ti <= tj): gathers the block (rows of ti, columns of tj) once, writes it, and
// writes its transpose to (tj, ti) through LDS (rows of 65 doubles): half the gathers and half the index reads, every store still a
// 32-byte (float64) / 16-byte (float32) run.  Same values, same bits.
#ifndef MEXP_SYM
#define MEXP_SYM 1
#endif
__global__ __launch_bounds__(256) void pld_moment_expand_sym_kernel(const double *__restrict__ Mcan, size_t mstride,
                                                                     const uint32_t *__restrict__ src,
                                                                     const double *__restrict__ mean, int Pc, int ldg, double Nd,
                                                                     double *__restrict__ G, float *__restrict__ G32, int T) {
    __shared__ double tile[64][65];
    const int b = blockIdx.y, tid = threadIdx.x;
    // tile pair p -> (ti <= tj), row-major over the upper triangle of T x T tiles
    int ti = 0, rem = blockIdx.x;
    while (rem >= T - ti) {
        rem -= T - ti;
        ++ti;
    }
    const int tj = ti + rem;
    const double *mb = mean + (size_t)b * Pc, *Mb = Mcan + (size_t)b * mstride;
    double *Gb = G + (size_t)b * ldg * ldg;
    float *G32b = G32 ? G32 + (size_t)b * ldg * ldg : nullptr;
    const int c4 = tid & 15, r0 = tid >> 4;  // a thread: columns 4 c4 .. 4 c4 + 3 of rows r0, r0 + 16, r0 + 32, r0 + 48
    const int jg = tj * 64 + 4 * c4;
    const bool jok = jg < Pc;  // (Pc % 4 == 0: a group is inside or outside as a whole)
    int ii[4];
    uint4 sv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        ii[u] = ti * 64 + r0 + 16 * u;
        sv[u] = *reinterpret_cast<const uint4 *>(src + (size_t)min(ii[u], Pc - 1) * Pc + min(jg, Pc - 4));
    }
    double v[4][4], mj[4], mi[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) mj[t] = mb[min(jg, Pc - 4) + t];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        v[u][0] = Mb[sv[u].x];
        v[u][1] = Mb[sv[u].y];
        v[u][2] = Mb[sv[u].z];
        v[u][3] = Mb[sv[u].w];
        mi[u] = Nd * mb[min(ii[u], Pc - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        pld_d4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            o[t] = v[u][t] - mi[u] * mj[t];
            tile[r0 + 16 * u][4 * c4 + t] = o[t];
        }
        if (ii[u] < Pc && jok) {
            *reinterpret_cast<pld_d4 *>(Gb + (size_t)ii[u] * ldg + jg) = o;
            if (G32b)
                *reinterpret_cast<float4 *>(G32b + (size_t)ii[u] * ldg + jg) =
                    make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        }
    }
    if (ti == tj) return;  // (workgroup-uniform; the diagonal tile is its own transpose, gathered in full)
    __syncthreads();
    // the transpose: element (row j of tile tj, column i of tile ti) = tile[i][j]
    const int ig = ti * 64 + 4 * c4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int jl = r0 + 16 * u, jr = tj * 64 + jl;
        if (jr < Pc && ig < Pc) {
            pld_d4 o;
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = tile[4 * c4 + t][jl];
            *reinterpret_cast<pld_d4 *>(Gb + (size_t)jr * ldg + ig) = o;
            if (G32b)
                *reinterpret_cast<float4 *>(G32b + (size_t)jr * ldg + ig) =
                    make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        }
    }
}

// mv[b][a] = sum_p mean[b][p] V[b][p][a]: the projection of the column means, subtracted from every cadence by the
// projection kernel below (one workgroup per cutout; inside the projection kernel this was a 204-step dependent chain
// of global loads in front of every workgroup's MFMA loop)
__global__ __launch_bounds__(256) void pld_mean_proj_kernel(const double *__restrict__ mean, const double *__restrict__ V,
                                                             int Pc, int kk, double *__restrict__ mv) {
    __shared__ double red[4][64];
    const int b = blockIdx.x, a = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const double *Vb = V + (size_t)b * Pc * kk, *mb = mean + (size_t)b * Pc;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four independent chains: the loads of 16 steps are in flight together
    if (a < kk) {
        int p = sl;
        for (; p + 12 < Pc; p += 16) {
            s0 = fma(mb[p], Vb[(size_t)p * kk + a], s0);
            s1 = fma(mb[p + 4], Vb[(size_t)(p + 4) * kk + a], s1);
            s2 = fma(mb[p + 8], Vb[(size_t)(p + 8) * kk + a], s2);
            s3 = fma(mb[p + 12], Vb[(size_t)(p + 12) * kk + a], s3);
        }
        for (; p < Pc; p += 4) s0 = fma(mb[p], Vb[(size_t)p * kk + a], s0);
    }
    red[sl][a] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < 64 && a < kk) mv[(size_t)b * kk + a] = ((red[0][a] + red[1][a]) + red[2][a]) + red[3][a];
}

// U = (products - mean) V diag(lam)^-1/2 into X[:, col0 : col0 + kk] with the products generated on the fly from the
// first-order components (X[:, col1 : col1 + k1]).  One wave = 16 cadences; the summation index of an MFMA step is the
// product column p = p0 + (lane >> 4), its factor tuple one packed dword of LDS.
// WR row tiles of 16 cadences per wave share every operand of V and every factor tuple (round 6: with one row tile a step was one
// global load, four LDS reads and two multiplications per MFMA and the kernel ran at 0.41 of the fp64 MFMA rate; four row tiles for
// the 816-column block, two for the 136-column one: 1.47 -> 1.17 ms and 0.30 -> 0.28 ms per 500 cutouts.  Requesting the next step's
// operands of V ahead of a step's MFMAs — two register sets, a scheduling fence — measured nothing: four to five waves per SIMD
// already cover that round trip.  What is left is the operand generation: ~64 fp64 multiplications and ~130 integer
// instructions per 16 MFMAs on the same SIMD.)
template <int O, int KT, int WR>
__global__ __launch_bounds__(256) void pld_project_products_kernel(const double *__restrict__ Xin, int ldx, int col1, int k1,
                                                                    int N, int Pc, const uint32_t *__restrict__ packed,
                                                                    const double *__restrict__ mvg,
                                                                    const double *__restrict__ V,
                                                                    const double *__restrict__ lam, int kk, int col0,
                                                                    double *__restrict__ Xout) {
    extern __shared__ __attribute__((aligned(16))) double pp_lds[];  // us[64 WR][ks] | tup[Pc] (dwords)
    const int b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane >> 4, lr = lane & 15;
    const int ks = k1 | 1;
    constexpr int ROWS = 64 * WR;
    double *us = pp_lds;
    uint32_t *tup = reinterpret_cast<uint32_t *>(us + ROWS * ks);
    const int nb = blockIdx.x * ROWS;
    for (int e = tid; e < ROWS * k1; e += 256) {
        const int r = e / k1, c = e - r * k1;
        us[r * ks + c] = nb + r < N ? Xin[((size_t)b * N + nb + r) * ldx + col1 + c] : 0.0;
    }
    for (int e = tid; e < Pc; e += 256) tup[e] = packed[e];
    const double *Vb = V + (size_t)b * Pc * kk, *mv = mvg + (size_t)b * kk;
    __syncthreads();
    const int n0 = nb + wave * 16 * WR;
    if (n0 >= N) return;
    const double *row[WR];
#pragma unroll
    for (int w = 0; w < WR; ++w) row[w] = us + (wave * 16 * WR + 16 * w + lr) * ks;
    pld_d4 acc[WR][KT];
#pragma unroll
    for (int w = 0; w < WR; ++w)
#pragma unroll
        for (int c = 0; c < KT; ++c) acc[w][c] = pld_d4{0.0, 0.0, 0.0, 0.0};
    // Out-of-range columns are handled by CLAMPED unconditional loads times a 0 / 1 factor: a guarded load ends up behind a
    // branch with a full s_waitcnt right after it, one L2 round trip per MFMA (all operands are finite).
    double cflag[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) cflag[c] = c * 16 + lr < kk ? 1.0 : 0.0;
    for (int p0 = 0; p0 < Pc; p0 += 16) {
        double prod[WR][4], bv[4][KT];
        uint32_t tp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) tp[q] = tup[min(p0 + 4 * q + lq, Pc - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p0 + 4 * q + lq, pc = min(p, Pc - 1);
            const double live = p < Pc ? 1.0 : 0.0;
#pragma unroll
            for (int w = 0; w < WR; ++w) {
                double pr = row[w][tp[q] & 255u];
#pragma unroll
                for (int pos = 1; pos < O; ++pos) pr *= row[w][(tp[q] >> (8 * pos)) & 255u];
                prod[w][q] = pr * live;
            }
#pragma unroll
            for (int c = 0; c < KT; ++c) bv[q][c] = Vb[(size_t)pc * kk + min(c * 16 + lr, kk - 1)] * cflag[c];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int w = 0; w < WR; ++w)
#pragma unroll
                for (int c = 0; c < KT; ++c) acc[w][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(prod[w][q], bv[q][c], acc[w][c], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < KT; ++c) {
        const int a = c * 16 + lr;
        if (a < kk) {
            const double sc = sqrt(fmax(lam[(size_t)b * kk + a], 1e-300));
#pragma unroll
            for (int w = 0; w < WR; ++w)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + 16 * w + lq + 4 * r;
                    if (n < N) Xout[((size_t)b * N + n) * ldx + col0 + a] = (acc[w][c][r] - mv[a]) / sc;
                }
        }
    }
}

// prior_sigma blocks: 10 * nanstd(lc float32) (np.nanstd of a float32 array -> float32), / pca for the PLD blocks
__global__ __launch_bounds__(256) void pld_prior_kernel(const float *__restrict__ lc, int N, int K, int n_pld_cols,
                                                         int pca, double *__restrict__ prior_sigma) {
    __shared__ double sh[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *f = lc + (size_t)b * N;
    double s = 0.0;
    long long c = 0;
    for (int i = tid; i < N; i += 256)
        if (f[i] == f[i]) {
            s += (double)f[i];
            ++c;
        }
    const long long cnt = block_count_dyn(c, reinterpret_cast<long long *>(sh));
    const double mean = block_sum_dyn(s, sh) / (double)cnt;
    s = 0.0;
    for (int i = tid; i < N; i += 256)
        if (f[i] == f[i]) {
            const double d = (double)f[i] - mean;
            s = fma(d, d, s);
        }
    const double sd = (double)(float)sqrt(block_sum_dyn(s, sh) / (double)cnt) * 10.0;
    for (int j = tid; j < K; j += 256) prior_sigma[(size_t)b * K + j] = j < n_pld_cols ? sd / (double)pca : sd;
}

// clamped B-spline basis (patsy bs(x, df, degree, include_intercept=True)) (+ constant column).  A thread evaluates ONE row
// (de Boor: the degree + 1 non-zero values and the span they start at) and parks it in LDS; then the workgroup writes the
// rows out with the lanes across the COLUMNS, so a row's nb (+ 1) doubles go out as one contiguous run instead of one
// 8-byte store per lane at a 1-KB stride (the block is mostly zeros: 0.5 ms -> the write bandwidth of its 1 GB).
__global__ __launch_bounds__(256) void pld_spline_kernel(const double *__restrict__ time, const double *__restrict__ knots,
                                                          int n_inner, int degree, int N, int ldx, int col0,
                                                          double *__restrict__ X, int with_const) {
    __shared__ double s_nv[256][9];
    __shared__ int s_mu[256];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int n = blockIdx.x * 256 + tid;
    const double *t = time + (size_t)b * N;
    const double *kn = knots + (size_t)b * (n_inner + 2);  // [lo, inner..., hi]
    const int order = degree + 1, nb = n_inner + order;
    if (n < N) {
        const double x = t[n], lo = kn[0], hi = kn[n_inner + 1];
        auto T = [&](int i) -> double {  // full knot vector: order copies of lo, inner, order copies of hi
            if (i < order) return lo;
            if (i >= order + n_inner) return hi;
            return kn[1 + i - order];
        };
        // knot span: largest mu with T(mu) <= x < T(mu+1), the right end belongs to the last non-empty span
        int mu = order - 1;
        const int last = order + n_inner - 1;
        while (mu < last && !(x < T(mu + 1))) ++mu;
        while (mu > order - 1 && T(mu) == T(mu + 1)) --mu;  // skip empty spans at repeated interior knots
        double Nv[8] = {1.0, 0, 0, 0, 0, 0, 0, 0}, left[8], right[8];
        for (int j = 1; j <= degree; ++j) {
            left[j] = x - T(mu + 1 - j);
            right[j] = T(mu + j) - x;
            double saved = 0.0;
            for (int r = 0; r < j; ++r) {
                const double den = right[r + 1] + left[j - r];
                const double tmp = den != 0.0 ? Nv[r] / den : 0.0;
                Nv[r] = saved + right[r + 1] * tmp;
                saved = left[j - r] * tmp;
            }
            Nv[j] = saved;
        }
        s_mu[tid] = mu - degree;  // column of Nv[0]
#pragma unroll
        for (int j = 0; j < 8; ++j) s_nv[tid][j] = Nv[j];
    }
    __syncthreads();
    const int ncol = nb + with_const, nrow = min(256, N - blockIdx.x * 256);
    double *base = X + ((size_t)b * N + (size_t)blockIdx.x * 256) * ldx + col0;
    for (int e = tid; e < nrow * ncol; e += 256) {
        const int r = e / ncol, c = e - r * ncol;
        const int j = c - s_mu[r];
        double v = 0.0;
        if (c == nb)
            v = 1.0;
        else if (j >= 0 && j <= degree)
            v = s_nv[r][j];
        base[(size_t)r * ldx + c] = v;
    }
}

// ------------------------------------------------------------------------------------------------ small eigenproblems
// Parallel cyclic Jacobi on a symmetric n x n matrix M (LDS, leading dim ld, n even); W <- eigenvectors (columns),
// diag(M) <- eigenvalues.  All threads of the workgroup participate.  rot: n/2 x 4 doubles of LDS scratch.
// Wt != nullptr: the eigenvectors are kept TRANSPOSED in global memory (Wt[p * n + i] = W[i][p], n x n) instead of in W
// (LDS) — rows p and q of Wt are what a rotation touches, so the accesses stay coalesced.
// M, W, rot, shred are OFFSETS (in doubles) into the workgroup's dynamic LDS: a non-inlined function that took them as
// plain pointers would address LDS through flat loads and stores.
__device__ __forceinline__ double jac_rsq(double x) {  // 1 / sqrt(x), x > 0 and normal
    double y = __builtin_amdgcn_rsq(x);
    y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
    return fma(0.5 * y, fma(-x * y, y, 1.0), y);
}
__device__ __forceinline__ double jac_rcp(double x) {  // 1 / x, x > 0 and normal
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    return fma(y, fma(-x, y, 1.0), y);
}

template <bool WT>
static __device__ __noinline__ void jacobi_eig_lds(int oM, int oW, int n, int ld, int orot, int oshred, double *Wt = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    double *M = lds_dyn + oM, *W = lds_dyn + oW, *rot = lds_dyn + orot, *shred = lds_dyn + oshred;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (WT)
        for (int e = tid; e < n * n; e += nt) Wt[e] = (e / n == e % n) ? 1.0 : 0.0;
    else
        for (int e = tid; e < n * n; e += nt) W[(e / n) * ld + (e % n)] = (e / n == e % n) ? 1.0 : 0.0;
    __syncthreads();
    const int half = n >> 1;
    // e / n by one multiplication (the loops below split ~1000 indices per round; the compiler's
    // 32-bit division by a run-time n is ~40 instructions, and this function is nothing but short, latency-bound phases)
    const unsigned magic = (unsigned)((((unsigned long long)1 << 32) + (unsigned)n - 1u) / (unsigned)n);  // exact while e (n - 1) < 2^32
    auto divn = [&](int e) { return (int)__umulhi((unsigned)e, magic); };
    int *roti = reinterpret_cast<int *>(rot);  // (p, q) of the round's pairs as ints, (c, s) as doubles behind them
    double *rotcs = rot + half;
    for (int sweep = 0; sweep < 40; ++sweep) {
        // convergence: off-diagonal mass vs diagonal mass
        double off = 0.0, dia = 0.0;
        for (int e = tid; e < n * n; e += nt) {
            const int i = divn(e), j = e - i * n;
            const double v = M[i * ld + j];
            if (i == j)
                dia = fma(v, v, dia);
            else
                off = fma(v, v, off);
        }
        const double offs = block_sum_dyn(off, shred), dias = block_sum_dyn(dia, shred);
        // converged: off-diagonal norm below 1e-13 of the diagonal's (1e-15 until round 3: one more sweep for nothing the
        // caller's 1e-10 residual test can see).  The cost of a sweep is the n - 1 dependent rotation computations (one
        // fp64 division and two square roots each, ~1.5 us per round with its three barriers), not the barriers: a
        // one-barrier form with 2 x 2 blocks in registers and per-wave rotation tables measured 25 % SLOWER.
        if (offs <= 1e-26 * dias || offs == 0.0) break;
        for (int r = 0; r < n - 1; ++r) {
            if (tid < half) {
                int p, q;
                if (tid == 0) {
                    p = n - 1;
                    q = r;
                } else {  // r, tid < n - 1: one conditional subtraction each instead of two divisions
                    p = r + tid;
                    if (p >= n - 1) p -= n - 1;
                    q = r - tid + n - 1;
                    if (q >= n - 1) q -= n - 1;
                }
                if (p > q) {
                    const int t0 = p;
                    p = q;
                    q = t0;
                }
                const double app = M[p * ld + p], aqq = M[q * ld + q], apq = M[p * ld + q];
                double c = 1.0, s = 0.0;
                if (fabs(apq) > 1e-140) {  // (its square below must not underflow)
                    // t = sgn(tau) / (|tau| + sqrt(1 + tau^2)), tau = d / b, written as sgn(d) b / (|d| + hypot(d, b)): one
                    // square root, one division and one reciprocal square root on the round's critical path instead of
                    // three divisions and two square roots (the same rotation up to rounding)
                    // (round 6: the square root, the division and the reciprocal square root are the hardware estimates + two Newton
                    // steps each instead of the IEEE expansions with their scaling and fix-up code — this chain is the critical
                    // path of every round; c^2 + s^2 = c^2 (1 + t^2) stays 1 to an ulp or two, and a rotation angle that is off in
                    // its last bits still annihilates a_pq to ~1e-16 of its size)
                    const double d = aqq - app, b2 = 2.0 * apq;
                    const double h2 = fma(d, d, b2 * b2);
                    const double hh = h2 * jac_rsq(h2);
                    const double t = (d >= 0.0 ? b2 : -b2) * jac_rcp(fabs(d) + hh);
                    c = jac_rsq(fma(t, t, 1.0));
                    s = t * c;
                }
                roti[2 * tid] = p;
                roti[2 * tid + 1] = q;
                rotcs[2 * tid] = c;
                rotcs[2 * tid + 1] = s;
            }
            __syncthreads();
            // M <- J^T M J in ONE phase: a thread owns the 2 x 2 block (pair a, pair b), applies the row rotation of pair
            // a and then the column rotation of pair b to it in registers (the arithmetic of the separate row and column
            // sweeps this replaces) and writes its own four entries back — no other thread touches them, so no barrier
            // between "rows" and "columns".  W <- W J by (row, pair) threads in the same phase.
            const unsigned magic_h = (unsigned)((((unsigned long long)1 << 32) + (unsigned)half - 1u) / (unsigned)half);
            for (int blk = tid; blk < half * half; blk += nt) {
                const int a = (int)__umulhi((unsigned)blk, magic_h), b2 = blk - a * half;
                const int p = roti[2 * a], q = roti[2 * a + 1], p2 = roti[2 * b2], q2 = roti[2 * b2 + 1];
                const double c = rotcs[2 * a], s = rotcs[2 * a + 1], c2 = rotcs[2 * b2], s2 = rotcs[2 * b2 + 1];
                const double mpp = M[p * ld + p2], mpq = M[p * ld + q2], mqp = M[q * ld + p2], mqq = M[q * ld + q2];
                const double rpp = c * mpp - s * mqp, rpq = c * mpq - s * mqq;  // rows
                const double rqp = s * mpp + c * mqp, rqq = s * mpq + c * mqq;
                M[p * ld + p2] = c2 * rpp - s2 * rpq;                            // columns
                M[p * ld + q2] = s2 * rpp + c2 * rpq;
                M[q * ld + p2] = c2 * rqp - s2 * rqq;
                M[q * ld + q2] = s2 * rqp + c2 * rqq;
            }
            for (int e = tid; e < half * n; e += nt) {
                const int pr = divn(e), i = e - pr * n;
                const int p = roti[2 * pr], q = roti[2 * pr + 1];
                const double c = rotcs[2 * pr], s = rotcs[2 * pr + 1];
                if (WT) {
                    const double wp = Wt[(size_t)p * n + i], wq = Wt[(size_t)q * n + i];
                    Wt[(size_t)p * n + i] = c * wp - s * wq;
                    Wt[(size_t)q * n + i] = s * wp + c * wq;
                } else {
                    const double wp = W[i * ld + p], wq = W[i * ld + q];
                    W[i * ld + p] = c * wp - s * wq;
                    W[i * ld + q] = s * wp + c * wq;
                }
            }
            if (WT) __threadfence_block();
            __syncthreads();
        }
    }
}

// order[j] = index of the j-th largest diagonal entry of M (n <= 64); ties by index
__device__ void sort_desc_lds(const double *M, int n, int ld, int *order) {
    const int tid = threadIdx.x;
    if (tid < n) {
        const double v = M[tid * ld + tid];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double u = M[j * ld + j];
            rank += (u > v || (u == v && j < tid)) ? 1 : 0;
        }
        order[rank] = tid;
    }
    __syncthreads();
}

// element (i, j) of the symmetric Gram matrix stored as upper 64x64 blocks with leading dimension ldg
__device__ __forceinline__ double gsym(const double *__restrict__ G, int ldg, int i, int j) {
    return (j >= i || (j >> 6) == (i >> 6)) ? G[(size_t)i * ldg + j] : G[(size_t)j * ldg + i];
}

// ------------------------------------------------------------------------------------------------ top-k eigenpairs
// The pieces of the subspace iteration are separate NON-inlined device functions: inlined into one kernel body (the
// first version) the compiler carried ~200 spilled registers through the hot loops.  EigCtx travels by value.
// The LDS regions are OFFSETS (in doubles) into the dynamic LDS block, turned back into pointers inside each function:
// a pointer argument would lose its address space and every LDS access would become a flat load/store.
struct EigCtx {
    int T, W, rot, shred, vec, qstage;  // LDS: two l x ld matrices, rotation table, reduction scratch, 2 l scalars, stage
    int order;                          // LDS: l ints (offset in doubles)
    double *Gb;                         // this matrix (global, leading dimension ldg)
    const float *G32b;                  // its float32 copy for the filter products of the early steps (or nullptr)
    int ldg, P, k, l, ld;
    int kc;                             // rows of the basis per LDS stage of eig_cq (a multiple of 8)
    int kc32;                           // rows per stage of eig_cq32 (float32 stage in the same LDS region; a multiple of 24 = 8 PF)
};
#define LK_EIG_LDS extern __shared__ __attribute__((aligned(16))) double lds_dyn[]
// likewise the global operands: as plain pointer arguments they would be read with flat loads
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) const double cgdouble;
typedef __attribute__((address_space(1))) const pld_d4 cgd4;

// Every dense step runs on the fp64 matrix cores (v_mfma_f64_16x16x4: A operand [row = lane & 15][k = lane >> 4],
// B operand [k = lane >> 4][col = lane & 15], D [row = (lane >> 4) + 4 r][col = lane & 15]).  The first version used one
// thread per output entry with a P-long serial fma chain for the skinny products (X^T Y, X M): at 1024 threads and 4
// waves per SIMD those loops were latency-bound and cost as much as the products with C.

// out (LDS, l x ld) = X^T Yv for two P x l row-major matrices.  Wave = (16 x 16 output tile, slice of the P rows); the
// slices are summed through LDS (qstage doubles as the buffer of partial tiles).
static __device__ __noinline__ void eig_xty(EigCtx c, const double *X_, const double *Yv_, int oout) {
    LK_EIG_LDS;
    cgdouble *X = (cgdouble *)X_, *Yv = (cgdouble *)Yv_;
    double *out = lds_dyn + oout;
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63, nwv = nt >> 6;
    const int lq = lane >> 4, lr = lane & 15, l = c.l, P = c.P, na = (l + 15) >> 4;
    const int tiles = na * na, parts = max(1, nwv / tiles);
    double *pbuf = lds_dyn + c.qstage;
    for (int unit = wave; unit < tiles * parts; unit += nwv) {  // (tile, row slice) units; more tiles than waves: several each
        const int tile = unit % tiles, part = unit / tiles;
        const int acol = (tile / na) * 16 + lr, ccol = (tile % na) * 16 + lr;
        const int steps = (P + 3) >> 2, per = (steps + parts - 1) / parts;
        const int s0 = part * per, s1 = min(steps, s0 + per);
        pld_d4 acc = pld_d4{0.0, 0.0, 0.0, 0.0};
        const bool aok = acol < l, cok = ccol < l;
        const int acl = min(acol, l - 1), ccl = min(ccol, l - 1);
        int st = s0;
        for (; st + 4 <= s1; st += 4) {  // four steps of loads in flight
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // clamped addresses, 0/1 factors (a select would put each load back under a branch)
                const int i = (st + u) * 4 + lq, ic = min(i, P - 1);
                av[u] = X[(size_t)ic * l + acl] * ((i < P && aok) ? 1.0 : 0.0);
                bv[u] = Yv[(size_t)ic * l + ccl] * ((i < P && cok) ? 1.0 : 0.0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
        for (; st < s1; ++st) {
            const int i = st * 4 + lq, ic = min(i, P - 1);
            const double av = X[(size_t)ic * l + acl] * ((i < P && aok) ? 1.0 : 0.0);
            const double bv = Yv[(size_t)ic * l + ccl] * ((i < P && cok) ? 1.0 : 0.0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pbuf[(part * tiles + tile) * 256 + (lq + 4 * r) * 16 + lr] = acc[r];
    }
    __syncthreads();
    for (int e = tid; e < l * l; e += nt) {
        const int a = e / l, cc = e % l;
        const int idx = ((a >> 4) * na + (cc >> 4)) * 256 + (a & 15) * 16 + (cc & 15);
        double sm = 0.0;
        for (int pt = 0; pt < parts; ++pt) sm += pbuf[pt * tiles * 256 + idx];
        out[a * c.ld + cc] = sm;
    }
    __syncthreads();
}

// out = X Mm (P x l times the l x l matrix Mm in LDS, leading dimension ld), optionally out2 = X2 Mm in the same pass.
// A wave owns 16-row strips; the single-matrix form loads a whole strip of X before it stores, so out may alias X.  With
// theta != nullptr the return value is this thread's share of sum_{c < k} || out2[:, c] - theta[c] out[:, c] ||^2.
// sc_ie != 0 (TWO only): out2 is stored as (X2 Mm - sc_c X Mm) sc_ie — the first term of the Chebyshev recurrence, which used to be
// a sweep of its own over the two outputs (the residual is still taken from the unscaled products).
template <int NA, bool TWO>
static __device__ __noinline__ double eig_xm(EigCtx c, const double *X_, double *out_, const double *X2_, double *out2_,
                                             int oMm, int otheta, double sc_c = 0.0, double sc_ie = 0.0) {
    LK_EIG_LDS;
    cgdouble *X = (cgdouble *)X_, *X2 = (cgdouble *)X2_;
    gdouble *out = (gdouble *)out_, *out2 = (gdouble *)out2_;
    const double *Mm = lds_dyn + oMm, *theta = otheta >= 0 ? lds_dyn + otheta : nullptr;
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63, nwv = nt >> 6;
    const int lq = lane >> 4, lr = lane & 15, l = c.l, P = c.P, ld = c.ld;
    double part = 0.0;
    const int nstrip = (P + 15) >> 4;
    for (int strip = wave; strip < nstrip; strip += nwv) {
        const int irow = strip * 16 + lr;
        pld_d4 acc[NA], acc2[TWO ? NA : 1];
#pragma unroll
        for (int t = 0; t < NA; ++t) acc[t] = pld_d4{0.0, 0.0, 0.0, 0.0};
        if (TWO) {
#pragma unroll
            for (int t = 0; t < NA; ++t) acc2[t] = pld_d4{0.0, 0.0, 0.0, 0.0};
        }
        double av[4 * NA], av2s[TWO ? 4 * NA : 1];
        const int irc = min(irow, P - 1);
#pragma unroll
        for (int ks = 0; ks < 4 * NA; ++ks) {  // clamped addresses, 0/1 factors: all loads of the strip in flight at once
            const int a = ks * 4 + lq, ac = min(a, l - 1);
            const double f = (irow < P && a < l) ? 1.0 : 0.0;
            av[ks] = X[(size_t)irc * l + ac] * f;
            if (TWO) av2s[ks] = X2[(size_t)irc * l + ac] * f;
        }
#pragma unroll
        for (int ks = 0; ks < 4 * NA; ++ks) {
            const int a = ks * 4 + lq;
            double av2 = 0.0;
            if (TWO) av2 = av2s[ks];
#pragma unroll
            for (int t = 0; t < NA; ++t) {
                const int col = t * 16 + lr;
                const double bv = (a < l && col < l) ? Mm[a * ld + col] : 0.0;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv, acc[t], 0, 0, 0);
                if (TWO) acc2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av2, bv, acc2[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int col = t * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = strip * 16 + lq + 4 * r;
                if (i < P && col < l) {
                    out[(size_t)i * l + col] = acc[t][r];
                    if (TWO) {
                        out2[(size_t)i * l + col] = sc_ie != 0.0 ? (acc2[t][r] - sc_c * acc[t][r]) * sc_ie : acc2[t][r];
                        if (theta && col < c.k) {
                            const double d = acc2[t][r] - theta[col] * acc[t][r];
                            part = fma(d, d, part);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    return part;
}

// Optional epilogue of a product (the phase-split iteration's filter steps): dst = alpha C src - beta src - gamma prev, i.e. one
// step of the Chebyshev recurrence written by the product's own stores instead of a separate sweep over three P x l arrays.
struct EigEpi {
    double alpha, beta, gamma;
    const double *prev;
};

// dst = C src (P x l, row-major): dst^T (l x P) = src^T (l x P) C (P x P).  The A operand (src^T) comes from an LDS stage
// of PLD_KC rows of src; the B operand straight from global: a wave owns 64 consecutive columns of C and a lane loads
// FOUR of them per row (32 B, so one load instruction covers 4 rows x 512 contiguous bytes) — component j of the load
// feeds MFMA tile j, whose column (lane & 15) is therefore column 64 w + 4 (lane & 15) + j of C (two and 32 w + 2 (lane &
// 15) + j for the wide basis).  Loads run two steps
// ahead of the matrix cores.  Every element of C is streamed exactly once per product.
// One pass of dst = C src over the column tiles [tile0, tile0 + nwv * NT): wave w owns NT consecutive tiles (16 NT columns),
// a lane NT neighbouring columns of each row.  All waves of the workgroup call it together (it stages src through LDS).
template <int NA, int NT, bool C32, bool EPI = false>
static __device__ __noinline__ void eig_cq_pass(EigCtx c, const double *src_, double *dst_, int tile0, EigEpi ep = EigEpi{1.0, 0.0, 0.0, nullptr}) {
    cgdouble *src = (cgdouble *)src_;
    gdouble *dst = (gdouble *)dst_;
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63;
    const int lq = lane >> 4, lr = lane & 15, l = c.l, P = c.P, ldg = c.ldg, na = (l + 15) >> 4;
    LK_EIG_LDS;
    double *qstage = lds_dyn + c.qstage;
    typedef double bvec __attribute__((ext_vector_type(4)));  // (NT of its components are used)
    typedef __attribute__((address_space(1))) const float cgfloat;
    const int nsteps = (P + 3) >> 2, KC = c.kc;
    const int col_w = (tile0 + wave * NT) << 4;  // first column of this wave
    const bool active = col_w < P;
    const int n0 = col_w + NT * lr;
    // The loads of C are UNCONDITIONAL: rows clamped to P - 1 (the matching rows of the staged basis are zeros), an idle
    // wave of the last pass re-reads row 0 of the first columns (L2).  Written as `(active && krow < P) ? load : 0` the
    // loads went under a branch and the compiler issued them AFTER the step's MFMAs with a full wait at the top of the
    // next trip — the "two steps ahead" of the source was no prefetch at all (round 4, from the ISA).
    cgdouble *cp = (cgdouble *)c.Gb + (active ? n0 : NT * lr);
    cgfloat *cp32 = (cgfloat *)c.G32b + (active ? n0 : NT * lr);  // C32: half the bytes of the stream
    const size_t rstride = active ? (size_t)ldg : 0;
    bool colok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) colok[j] = active && n0 + j < P;
    auto load_b = [&](int st) -> bvec {
        const int krow = min(st * 4 + lq, P - 1);
        bvec r = bvec(0.0);
        if (C32) {
            cgfloat *q = cp32 + (size_t)krow * rstride;
            if (NT == 4) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 f = *(__attribute__((address_space(1))) const f4 *)q;
                r = bvec{(double)f[0], (double)f[1], (double)f[2], (double)f[3]};
            } else if (NT >= 2) {  // (dword-aligned multi-dword loads are fine on global memory)
                typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
                const f2 f = *(__attribute__((address_space(1))) const f2 *)q;
                r[0] = (double)f[0];
                r[1] = (double)f[1];
                if (NT == 3) r[2] = (double)q[2];
            } else {
                r[0] = (double)q[0];
            }
        } else {
            cgdouble *q = cp + (size_t)krow * rstride;
            if (NT == 4) {
                r = *(cgd4 *)q;
            } else if (NT >= 2) {
                typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
                const d2 f = *(__attribute__((address_space(1))) const d2 *)q;
                r[0] = f[0];
                r[1] = f[1];
                if (NT == 3) r[2] = q[2];
            } else {
                r[0] = q[0];
            }
        }
        return r;
    };
    pld_d4 acc[NT][NA];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ai = 0; ai < NA; ++ai) acc[t][ai] = pld_d4{0.0, 0.0, 0.0, 0.0};
    bvec b0 = load_b(0), b1 = load_b(1);
    for (int k0 = 0; k0 < P; k0 += KC) {
        __syncthreads();
        for (int e = tid; e < KC * 16 * na; e += nt) {
            const int kk = e / (16 * na), a = e - kk * (16 * na);
            qstage[kk * PLD_QS + a] = (k0 + kk < P && a < l) ? src[(size_t)(k0 + kk) * l + a] : 0.0;
        }
        __syncthreads();
        const int s_lo = k0 >> 2, s_hi = min(nsteps, (k0 + KC) >> 2);
        for (int st = s_lo; st < s_hi; st += 2) {  // PLD_KC is a multiple of 8: step st + 1 stays inside the stage
            const bvec c0 = b0, c1 = b1;
            b0 = load_b(st + 2);
            b1 = load_b(st + 3);
            const int kk = st * 4 - k0;
            double av[NA];
#pragma unroll
            for (int ai = 0; ai < NA; ++ai) av[ai] = qstage[(kk + lq) * PLD_QS + ai * 16 + lr];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double bv = colok[t] ? c0[t] : 0.0;
#pragma unroll
                for (int ai = 0; ai < NA; ++ai)
                    acc[t][ai] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ai], bv, acc[t][ai], 0, 0, 0);
            }
#pragma unroll
            for (int ai = 0; ai < NA; ++ai) av[ai] = qstage[(kk + 4 + lq) * PLD_QS + ai * 16 + lr];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double bv = colok[t] ? c1[t] : 0.0;
#pragma unroll
                for (int ai = 0; ai < NA; ++ai)
                    acc[t][ai] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ai], bv, acc[t][ai], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t;
#pragma unroll
        for (int ai = 0; ai < NA; ++ai)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = ai * 16 + lq + 4 * r;
                if (colok[t] && a < l) {
                    if (EPI)
                        dst[(size_t)n * l + a] = ep.alpha * acc[t][ai][r] - ep.beta * src[(size_t)n * l + a] -
                                                 ep.gamma * ((cgdouble *)ep.prev)[(size_t)n * l + a];
                    else
                        dst[(size_t)n * l + a] = acc[t][ai][r];
                }
            }
    }
}

// dst = C src: full passes of NTMAX tiles per wave, then ONE last pass with as few tiles per wave as cover the rest —
// 816 columns = 51 tiles on 8 waves ran as 4 + 4 tiles per wave (the second pass with 5 of 8 waves busy); 4 + 3 is an eighth
// less (a pass costs what its busiest wave costs).
template <int NA, bool C32 = false, bool EPI = false>
static __device__ __forceinline__ void eig_cq(EigCtx c, const double *src, double *dst, EigEpi ep = EigEpi{1.0, 0.0, 0.0, nullptr}) {
    const int nwv = (int)blockDim.x >> 6, tiles = (c.P + 15) >> 4;
    // NTMAX tiles (= consecutive columns per lane) per wave: 4 while the 4 x NA accumulator tiles fit the register budget
    // of a 1024-thread workgroup, 2 for the wide basis (NA = 4: 4 x 4 tiles would be all 128 VGPRs)
    constexpr int NTMAX = NA <= 2 ? 4 : 2;
    int t0 = 0;
    for (; tiles - t0 > nwv * (NTMAX - 1); t0 += nwv * NTMAX) eig_cq_pass<NA, NTMAX, C32, EPI>(c, src, dst, t0, ep);
    const int rest = tiles - t0;
    if (rest > 0) {
        const int ntl = (rest + nwv - 1) / nwv;  // 1 .. NTMAX - 1
        if (ntl == 1)
            eig_cq_pass<NA, 1, C32, EPI>(c, src, dst, t0, ep);
        else if (NTMAX > 2 && ntl == 2)
            eig_cq_pass<NA, 2, C32, EPI>(c, src, dst, t0, ep);
        else if (NTMAX > 3)
            eig_cq_pass<NA, 3, C32, EPI>(c, src, dst, t0, ep);
    }
    __syncthreads();
}

// dst = C32 src on the FLOAT32 matrix cores (v_mfma_f32_16x16x4_f32: twice the fp64 MFMA rate on gfx950, half the bytes of C,
// half the accumulator registers): the products of the Chebyshev filter and of the Rayleigh-Ritz steps that are still far
// from the stop.  Same dataflow as eig_cq_pass — dst^T (l x P) = src^T C, the A operand (src^T, rounded to float32 while it is
// staged) from LDS, the B operand straight from the float32 copy of C, NT neighbouring columns per lane — but with up to 8
// column tiles per wave, so the 816-column blocks (51 tiles on 8 waves) take ONE pass over the staged basis instead of two.
// Operand layout of the instruction: A [row = lane & 15][k = lane >> 4], B [k = lane >> 4][col = lane & 15],
// D [row = 4 (lane >> 4) + r][col = lane & 15] — a lane ends with FOUR CONSECUTIVE basis columns of one column of C, i.e. one
// 32-byte store into the row-major dst.  Products accumulate in float32: a relative 1e-6 on quantities that only steer the
// iteration (the float64 Rayleigh-Ritz step before the stop restores every digit).
typedef float pld_f4 __attribute__((ext_vector_type(4)));
template <int NA, int NT, bool EPI = false>
static __device__ __noinline__ void eig_cq32_pass(EigCtx c, const double *src_, double *dst_, int tile0, EigEpi ep = EigEpi{1.0, 0.0, 0.0, nullptr}) {
    constexpr int QS32 = pld_qs32(NA);
    cgdouble *src = (cgdouble *)src_;
    gdouble *dst = (gdouble *)dst_;
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63;
    const int lq = lane >> 4, lr = lane & 15, l = c.l, P = c.P, ldg = c.ldg, na = (l + 15) >> 4;
    LK_EIG_LDS;
    float *qs = reinterpret_cast<float *>(lds_dyn + c.qstage);
    typedef __attribute__((address_space(1))) const float cgfloat;
    const int nsteps = (P + 3) >> 2, KC = c.kc32;
    const int col_w = (tile0 + wave * NT) << 4;  // first column of this wave
    const bool active = col_w < P;
    const int n0 = col_w + NT * lr;
    // unconditional loads (see eig_cq_pass): rows clamped to P - 1, idle waves re-read the first columns of row 0
    cgfloat *cp = (cgfloat *)c.G32b + (active ? n0 : NT * lr);
    const size_t rstride = active ? (size_t)ldg : 0;
    bool colok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) colok[j] = active && n0 + j < P;
    struct BRow { float v[NT]; };
    auto load_b = [&](int st) -> BRow {
        const int krow = min(st * 4 + lq, P - 1);
        cgfloat *q = cp + (size_t)krow * rstride;
        BRow r;
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // (dword-aligned multi-dword global loads)
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
        int j = 0;
#pragma unroll
        for (; j + 4 <= NT; j += 4) {
            const f4u f = *(__attribute__((address_space(1))) const f4u *)(q + j);
            r.v[j] = f[0];
            r.v[j + 1] = f[1];
            r.v[j + 2] = f[2];
            r.v[j + 3] = f[3];
        }
        if (NT - j >= 2) {
            const f2u f = *(__attribute__((address_space(1))) const f2u *)(q + j);
            r.v[j] = f[0];
            r.v[j + 1] = f[1];
            j += 2;
        }
        if (NT - j >= 1) r.v[j] = q[j];
        return r;
    };
    pld_f4 acc[NT][NA];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ai = 0; ai < NA; ++ai) acc[t][ai] = pld_f4{0.f, 0.f, 0.f, 0.f};
    // The loads of C run PF k-steps (4 PF rows x 64 NT bytes per wave) ahead of the matrix cores, in TWO register sets used in
    // turn: a step consumes one set while the load of the step PF ahead lands in the other.  (With one set the loop-carried
    // registers had to be copied before their reload, the compiler put all copies — and so the waits for ALL outstanding loads —
    // at the top of the trip, and sank the loads below the MFMAs: no load ever overlapped an MFMA, a product of 500 matrices
    // took 370-446 us against the 250 us of its bytes whether 250 or 500 workgroups ran.)  The scheduling fences keep each
    // load in front of its step's MFMAs.
    constexpr int PF = 3;
    BRow qa[PF], qb[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) qa[u] = load_b(u);
    auto mstep = [&](const BRow &cb, int kk) {
        float av[NA];
#pragma unroll
        for (int ai = 0; ai < NA; ++ai) av[ai] = qs[(kk + lq) * QS32 + ai * 16 + lr];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float bv = colok[t] ? cb.v[t] : 0.f;
#pragma unroll
            for (int ai = 0; ai < NA; ++ai)
                acc[t][ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ai], bv, acc[t][ai], 0, 0, 0);
        }
    };
    for (int k0 = 0; k0 < P; k0 += KC) {
        __syncthreads();
        for (int e = tid; e < KC * 16 * na; e += nt) {
            const int kk = e / (16 * na), a = e - kk * (16 * na);
            qs[kk * QS32 + a] = (k0 + kk < P && a < l) ? (float)src[(size_t)(k0 + kk) * l + a] : 0.f;
        }
        __syncthreads();
        const int s_lo = k0 >> 2, s_hi = min(nsteps, (k0 + KC) >> 2);
        for (int st = s_lo; st < s_hi; st += 2 * PF) {  // kc32 is a multiple of 8 PF: the 2 PF steps of a trip stay inside the stage
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                qb[u] = load_b(st + PF + u);
                __builtin_amdgcn_sched_barrier(0);
                mstep(qa[u], (st + u) * 4 - k0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                qa[u] = load_b(st + 2 * PF + u);
                __builtin_amdgcn_sched_barrier(0);
                mstep(qb[u], (st + PF + u) * 4 - k0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t;
#pragma unroll
        for (int ai = 0; ai < NA; ++ai) {
            const int a0 = ai * 16 + 4 * lq;
            if (colok[t]) {
                if (a0 + 3 < l && (l & 3) == 0) {  // rows of dst are 32-byte aligned: one store
                    pld_d4 o = pld_d4{(double)acc[t][ai][0], (double)acc[t][ai][1], (double)acc[t][ai][2], (double)acc[t][ai][3]};
                    if (EPI) {
                        const pld_d4 cu = *(cgd4 *)(src + (size_t)n * l + a0), pv = *(cgd4 *)((cgdouble *)ep.prev + (size_t)n * l + a0);
                        o = ep.alpha * o - ep.beta * cu - ep.gamma * pv;
                    }
                    *(__attribute__((address_space(1))) pld_d4 *)(dst + (size_t)n * l + a0) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (a0 + r < l) {
                            double o = (double)acc[t][ai][r];
                            if (EPI)
                                o = ep.alpha * o - ep.beta * src[(size_t)n * l + a0 + r] -
                                    ep.gamma * ((cgdouble *)ep.prev)[(size_t)n * l + a0 + r];
                            dst[(size_t)n * l + a0 + r] = o;
                        }
                }
            }
        }
    }
}

template <int NA, bool EPI = false>
static __device__ __forceinline__ void eig_cq32(EigCtx c, const double *src, double *dst, EigEpi ep = EigEpi{1.0, 0.0, 0.0, nullptr}) {
    const int nwv = (int)blockDim.x >> 6, tiles = (c.P + 15) >> 4;
    constexpr int NTMAX = NA <= 2 ? 8 : 4;  // accumulators: NT x NA x 4 VGPRs
    for (int t0 = 0; t0 < tiles;) {
        const int ntl = min(NTMAX, (tiles - t0 + nwv - 1) / nwv);
        switch (ntl) {
            case 1: eig_cq32_pass<NA, 1, EPI>(c, src, dst, t0, ep); break;
            case 2: eig_cq32_pass<NA, 2, EPI>(c, src, dst, t0, ep); break;
            case 3: eig_cq32_pass<NA, 3, EPI>(c, src, dst, t0, ep); break;
            case 4: eig_cq32_pass<NA, 4, EPI>(c, src, dst, t0, ep); break;
            case 5: if (NTMAX >= 5) eig_cq32_pass<NA, NTMAX >= 5 ? 5 : 1, EPI>(c, src, dst, t0, ep); break;
            case 6: if (NTMAX >= 6) eig_cq32_pass<NA, NTMAX >= 6 ? 6 : 1, EPI>(c, src, dst, t0, ep); break;
            case 7: if (NTMAX >= 7) eig_cq32_pass<NA, NTMAX >= 7 ? 7 : 1, EPI>(c, src, dst, t0, ep); break;
            default: if (NTMAX >= 8) eig_cq32_pass<NA, NTMAX >= 8 ? 8 : 1, EPI>(c, src, dst, t0, ep); break;
        }
        t0 += nwv * ntl;
    }
    __syncthreads();
}

// SVQB orthonormalisation of the P x l matrix Yv (in place): unit-scale columns, eig of the l x l Gram, Yv S Phi^-1/2
template <int NA>
static __device__ __noinline__ void eig_svqb(EigCtx c, double *Yv) {
    LK_EIG_LDS;
    const int tid = threadIdx.x, nt = blockDim.x, l = c.l, ld = c.ld;
    double *T = lds_dyn + c.T, *W = lds_dyn + c.W, *vec = lds_dyn + c.vec;
    for (int pass = 0; pass < 2; ++pass) {
        eig_xty(c, Yv, Yv, c.T);
        if (tid < l) vec[tid] = T[tid * ld + tid] > 0.0 ? 1.0 / sqrt(T[tid * ld + tid]) : 0.0;
        __syncthreads();
        for (int e = tid; e < l * l; e += nt) {
            const int a = e / l, cc = e % l;
            if (cc >= a) {
                const double v = 0.5 * (T[a * ld + cc] + T[cc * ld + a]) * vec[a] * vec[cc];
                W[a * ld + cc] = v;
                W[cc * ld + a] = v;
            }
        }
        __syncthreads();
        for (int e = tid; e < l * l; e += nt) T[(e / l) * ld + (e % l)] = W[(e / l) * ld + (e % l)];
        __syncthreads();
        jacobi_eig_lds<false>(c.T, c.W, l, ld, c.rot, c.shred);
        double mx = 0.0;
        for (int a = 0; a < l; ++a) mx = fmax(mx, T[a * ld + a]);
        if (tid < l) {
            const double ph = fmax(T[tid * ld + tid], 1e-15 * mx);
            vec[l + tid] = 1.0 / sqrt(ph);
        }
        __syncthreads();
        // Mm = diag(vec) W diag(vec2) into T, then Yv <- Yv Mm in place
        for (int e = tid; e < l * l; e += nt) {
            const int a = e / l, cc = e % l;
            T[a * ld + cc] = vec[a] * W[a * ld + cc] * vec[l + cc];
        }
        __syncthreads();
        eig_xm<NA, false>(c, Yv, Yv, nullptr, nullptr, c.T, -1);
    }
}

// Cholesky-QR of the P x l matrix Yin -> Qout (two passes; Qout may be Yin).  The columns handed in are images C^q r of
// Ritz vectors, i.e. nearly orthogonal with wildly different norms: after scaling them to unit length the Gram matrix is
// close to the identity and its Cholesky factor is benign.  Returns false (in every thread) on a breakdown (pivot <=
// 1e-12), in which case the caller falls back to SVQB.
template <int NA>
static __device__ __noinline__ bool eig_cholqr(EigCtx c, const double *Yin, double *Qout) {
    LK_EIG_LDS;
    const int tid = threadIdx.x, nt = blockDim.x, l = c.l, ld = c.ld;
    double *T = lds_dyn + c.T, *W = lds_dyn + c.W, *vec = lds_dyn + c.vec;
    int *order = reinterpret_cast<int *>(lds_dyn + c.order);
    const double *cur = Yin;
    bool one_pass = false;
    for (int pass = 0; pass < 2; ++pass) {
        eig_xty(c, cur, cur, c.W);
        if (tid < l) vec[tid] = W[tid * ld + tid] > 0.0 ? 1.0 / sqrt(W[tid * ld + tid]) : 0.0;
        __syncthreads();
        for (int e = tid; e < l * l; e += nt) {
            const int a = e / l, cc = e % l;
            if (cc >= a) {
                const double v = 0.5 * (W[a * ld + cc] + W[cc * ld + a]) * vec[a] * vec[cc];
                T[a * ld + cc] = v;
                T[cc * ld + a] = v;
            }
        }
        __syncthreads();
        // (The smallest pivot of the unit-diagonal Gram matrix decides below whether a second pass is needed.)
        // T = L L^T in place (lower), then W = L^-1, both on one wave (l <= 64: lane = row)
        if (tid < 64) {
            int ok = 1;
            double dmin = 1.0;
            for (int j = 0; j < l; ++j) {
                const double d = T[j * ld + j];
                dmin = fmin(dmin, d);
                if (!(d > 1e-12)) {
                    ok = 0;
                    break;
                }
                const double rs = 1.0 / sqrt(d);
                __builtin_amdgcn_wave_barrier();
                if (tid >= j && tid < l) T[tid * ld + j] *= rs;  // column j of L (diagonal included: sqrt(d))
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (tid > j && tid < l) {
                    const double lij = T[tid * ld + j];
                    for (int cc = j + 1; cc <= tid; ++cc) T[tid * ld + cc] -= lij * T[cc * ld + j];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (tid == 0) {
                order[0] = ok;
                order[1] = (pass == 0 && dmin >= 0.1) ? 1 : 0;
            }
            if (ok && tid < l) {
                // COLUMN cc of L^-1 solves L x = e_cc: lane = column, forward substitution down the rows
                const int cc = tid;
                for (int i = 0; i < l; ++i) {
                    double x = (i == cc) ? 1.0 : 0.0;
                    for (int m = cc; m < i; ++m) x -= T[i * ld + m] * W[m * ld + cc];
                    W[i * ld + cc] = i < cc ? 0.0 : x / T[i * ld + i];
                }
            }
        }
        __syncthreads();
        const bool ok = order[0] != 0;
        // Well conditioned already?  With unit diagonal and every pivot >= 0.1 the Gram matrix's condition number is a few
        // tens at most: one pass leaves ||Q^T Q - I|| ~ 1e-14, five decades below anything the iteration resolves, and the second
        // pass (a Gram sweep, a factorisation and a product: half of this function) is skipped.  True for the pseudo-random
        // start and for filtered Ritz vectors.
        if (pass == 0) one_pass = order[1] != 0;
        __syncthreads();
        if (!ok) return false;
        // Mm[a][cc] = vec[a] Linv[cc][a] (a <= cc): out = cur diag(vec) L^-T
        for (int e = tid; e < l * l; e += nt) {
            const int a = e / l, cc = e % l;
            T[a * ld + cc] = a <= cc ? vec[a] * W[cc * ld + a] : 0.0;
        }
        __syncthreads();
        eig_xm<NA, false>(c, cur, Qout, nullptr, nullptr, c.T, -1);
        cur = Qout;
        if (one_pass) break;
    }
    return true;
}

// Per-matrix state of the phase-split iteration (pld_eigs_* kernels below), 64 bytes in global memory.
struct EigsState {
    int done;       // nothing left to do here: converged, or handed to the one-kernel iteration (converged == 0)
    int converged;
    int it;         // Rayleigh-Ritz steps taken
    int rr_var;     // product variant of the NEXT Rayleigh-Ritz step: 0 = float32 C on the float32 matrix cores, 2 = float64
    int f_var;      // variant of this step's two filter products: 0, 1 = float32 C with float64 products, 2
    int cheb;       // Chebyshev filter (else plain powers of C)
    int pad[2];
    double cc, ie;  // filter interval [0, theta_cut]: centre = half-width = cc, ie = 1 / cc
    double res, th0;
};
static_assert(sizeof(EigsState) == 64, "EigsState is indexed with a 64-byte stride");

// Top-k eigenpairs of the P x P Gram matrix of matrix b -> V (P x k, row-major), lam (k).  One workgroup per matrix.
// scratch per matrix: 4 * P * l doubles (Q, Z, R, Y).  NA = number of 16-column tiles of the basis (l <= 16 NA): a
// compile-time constant, because with a run-time bound the compiler keeps all 4 x 4 accumulator tiles of the product
// with C (128 VGPRs = the whole budget of a 1024-thread workgroup) and spills around every MFMA.
template <int NA>
__global__ __launch_bounds__(1024) void pld_topk_eig_kernel(double *__restrict__ G, int ldg, int P, int k, int l, int npow,
                                                             double *__restrict__ scratch, double *__restrict__ V,
                                                             double *__restrict__ lam, long long *__restrict__ iters_out,
                                                             int max_it, int *__restrict__ status, int cheb_on, int kc,
                                                             int mirror, double tol, const float *__restrict__ G32 = nullptr,
                                                             const EigsState *__restrict__ skip = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
    if (skip && skip[b].converged) return;  // the phase-split iteration already delivered this matrix
    double *Gb = G + (size_t)b * ldg * ldg;
    double *Vb = V + (size_t)b * P * k, *lamb = lam + (size_t)b * k;
    const int ld = l + 1;                      // odd leading dimension: conflict-free column walks
    const int oT = 0, oW = l * ld, orot = 2 * l * ld, oshred = orot + 2 * l, ovec = oshred + nt, oorder = ovec + 2 * l,
              oqstage = oorder + (l + 1) / 2 + 1;
    double *T = lds + oT;                      // l x ld
    double *W = lds + oW;                      // l x ld
    double *shred = lds + oshred;              // nt doubles (after the (l/2) x 4 rotation table)
    double *vec = lds + ovec;                  // 2 * l  (norms / theta)
    int *order = reinterpret_cast<int *>(lds + oorder);  // l ints; then qstage: PLD_KC x PLD_QS (subspace path only)

    if (P <= l && l > PLD_LMAX) {
        if (status && status[b]) return;  // the short subspace pass already converged this matrix
        // ---- direct, mid-size (PLD_LMAX < P <= PLD_DIRECT_MAX): Jacobi on C itself with C in LDS (l x (l + 1) doubles
        // fill it) and the eigenvectors transposed in global scratch.  A few ms per matrix, against tens of Rayleigh-
        // Ritz steps of the subspace iteration when the spectrum of a 2nd-order product block decays slowly.
        double *Wt = scratch + (size_t)b * l * l;
        const int orot2 = l * ld, oshred2 = orot2 + 2 * l;
        double *vec2 = lds + oshred2 + nt;
        int *order2 = reinterpret_cast<int *>(vec2 + 2 * l);
        for (int e = tid; e < l * l; e += nt) {
            const int i = e / l, j = e % l;
            T[i * ld + j] = (i < P && j < P) ? gsym(Gb, ldg, i, j) : 0.0;
        }
        __syncthreads();
        jacobi_eig_lds<true>(oT, oT, l, ld, orot2, oshred2, Wt);
        if (tid < l && tid >= P) T[tid * ld + tid] = -1.0;  // pad eigenvalue sorts last
        __syncthreads();
        // order by eigenvalue, l may exceed 64: rank by counting, one thread per entry
        for (int a = tid; a < l; a += nt) {
            const double v = T[a * ld + a];
            int rank = 0;
            for (int j = 0; j < l; ++j) {
                const double u = T[j * ld + j];
                rank += (u > v || (u == v && j < a)) ? 1 : 0;
            }
            order2[rank] = a;
        }
        __syncthreads();
        for (int e = tid; e < P * k; e += nt) {
            const int a = e / P, i = e - a * P;  // i fastest: rows of Wt are read contiguously
            Vb[(size_t)i * k + a] = Wt[(size_t)order2[a] * l + i];
        }
        if (tid < k) lamb[tid] = T[order2[tid] * ld + order2[tid]];
        if (tid == 0 && iters_out) iters_out[(size_t)b * 8] = 0;
        return;
    }
    if (P <= l) {
        // ---- direct: Jacobi on C itself (l = P rounded up to even; the pad row/col is zero)
        for (int e = tid; e < l * l; e += nt) {
            const int i = e / l, j = e % l;
            T[i * ld + j] = (i < P && j < P) ? gsym(Gb, ldg, i, j) : 0.0;
        }
        __syncthreads();
        jacobi_eig_lds<false>(oT, oW, l, ld, orot, oshred);
        if (tid < l && tid >= P) T[tid * ld + tid] = -1.0;  // pad eigenvalue sorts last
        __syncthreads();
        sort_desc_lds(T, l, ld, order);
        for (int e = tid; e < P * k; e += nt) {
            const int i = e / k, a = e % k;
            Vb[e] = W[i * ld + order[a]];
        }
        if (tid < k) lamb[tid] = T[order[tid] * ld + order[tid]];
        if (tid == 0 && iters_out) iters_out[(size_t)b * 8] = 0;
        return;
    }

    // ---- subspace iteration with Rayleigh-Ritz steps
    double *Q = scratch + (size_t)b * 4 * P * l, *Z = Q + (size_t)P * l, *R = Z + (size_t)P * l, *Y = R + (size_t)P * l;
    const int kc32 = (int)(((size_t)kc * PLD_QS * 8) / ((size_t)pld_qs32(NA) * 4)) / 24 * 24;  // the float32 stage fills the same LDS region
    const EigCtx ctx{oT, oW, orot, oshred, ovec, oqstage, oorder, Gb, G32 ? G32 + (size_t)b * ldg * ldg : nullptr, ldg, P, k, l, ld, kc, kc32};
    long long tprof[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = iters_out ? (long long)wall_clock64() : 0;
    auto lap = [&](int slot) {  // debug (LK_PLD_ITERS=1): per-phase 100 MHz ticks
        if (iters_out) {
            const long long now = (long long)wall_clock64();
            tprof[slot] += now - tlast;
            tlast = now;
        }
    };
    // the Gram kernel wrote the upper 64x64 blocks only: mirror them so the products below read plain rows
    if (mirror)
        for (int e = tid; e < P * P; e += nt) {
            const int i = e / P, j = e - i * P;
            if ((j >> 6) < (i >> 6)) Gb[(size_t)i * ldg + j] = Gb[(size_t)j * ldg + i];
        }
    // deterministic pseudo-random start
    for (int e = tid; e < P * l; e += nt) {
        unsigned int x = (unsigned int)(e + 1) * 2654435761u;
        x ^= x >> 15;
        x *= 2246822519u;
        x ^= x >> 13;
        Q[e] = (double)(x & 0xffffffu) / 8388608.0 - 1.0;
    }
    __syncthreads();
    // the pseudo-random columns are close to orthogonal already: Cholesky-QR (no l x l eigenproblem) orthonormalises them;
    // SVQB only if the Cholesky factor breaks down
    if (!eig_cholqr<NA>(ctx, Q, Q)) eig_svqb<NA>(ctx, Q);
    lap(0);
    int it = 0;
    bool converged = false;
    double *theta = vec + l;  // sorted Ritz values of the current step
    bool rr32 = ctx.G32b != nullptr;  // this step's Rayleigh-Ritz product on the float32 matrix cores (PLD_RR32_RES)
    for (; it < max_it; ++it) {
        if (rr32)
            eig_cq32<NA>(ctx, Q, Z);  // Z = C32 Q
        else
            eig_cq<NA>(ctx, Q, Z);  // Z = C Q
        lap(1);
        eig_xty(ctx, Q, Z, oW);  // T = Q^T Z (symmetrised)
        for (int e = tid; e < l * l; e += nt) {
            const int a = e / l, c = e % l;
            T[a * ld + c] = 0.5 * (W[a * ld + c] + W[c * ld + a]);
        }
        __syncthreads();
        lap(2);
        jacobi_eig_lds<false>(oT, oW, l, ld, orot, oshred);
        sort_desc_lds(T, l, ld, order);
        lap(3);
        // R = Q W (Ritz vectors, sorted by Ritz value), Y = Z W = C R, and the residual || C r - theta r || of the k
        // wanted pairs, in one pass over Q and Z
        if (tid < l) theta[tid] = T[order[tid] * ld + order[tid]];
        __syncthreads();
        for (int e = tid; e < l * l; e += nt) {
            const int a = e / l, c = e % l;
            T[a * ld + c] = W[a * ld + order[c]];
        }
        __syncthreads();
        const double th0 = fabs(theta[0]), th_k = theta[k - 1], th_cut = theta[l - 1];
        // (the filter decision needs the Ritz values only, so it is taken BEFORE the pass that writes R and Y: with the Chebyshev
        // filter that pass stores Y as the recurrence's first term X1 = (C R - c R) / e straight away — one sweep over two P x l
        // arrays less per step; unused if this step converges)
        const bool cheb = npow >= 2 && th_cut > 0.0 &&
                          (((cheb_on & 1) && th_k < 1.5 * th_cut && th0 < 30.0 * th_cut) || (cheb_on & 2));  // bit 1 (LK_PLD_CHEB=2/3): always
        const double cc = 0.5 * th_cut, ie = cheb ? 1.0 / cc : 0.0;
        const double part = eig_xm<NA, true>(ctx, Q, R, Z, Y, oT, ovec + l, cc, ie);
        const double res = sqrt(block_sum_dyn(part, shred));
        if (tid < k) lamb[tid] = theta[tid];
        __syncthreads();
        lap(4);
        if (!rr32 && res <= tol * th0 * sqrt((double)k)) {  // (only a float64 product can declare convergence)
            converged = true;
            break;
        }
        rr32 = ctx.G32b != nullptr && res > PLD_RR32_RES * th0;
        // next basis: orthonormalised p(C) R with npow - 1 more products between two Rayleigh-Ritz steps.
        //  * flat spectrum (the wanted Ritz values sit close to the first unwanted one: product blocks): p = the Chebyshev
        //    polynomial of degree npow that is bounded by 1 on the unwanted interval [0, theta_cut] and grows fastest
        //    outside it — for theta_k / theta_cut = 1.2 a degree-3 step damps the unwanted part 6.8x instead of the
        //    1.7x of C^3, so the slowly converging blocks need a fraction of the Rayleigh-Ritz steps;
        //  * steep spectrum: the same filter (option bit 1, default) — the factors between the columns differ by many
        //    orders of magnitude, but the Cholesky-QR scales the columns to unit length first, and measured it needs 5.0
        //    steps where p = C^npow (option bit 1 off) needs 7.2.
        double *src = Y, *dst = Z;
        if (cheb) {
            // x = (C - c) / e with c = e = theta_cut / 2:  X0 = R, X1 = (C R - c R) / e (stored by eig_xm above),
            // X_{j+1} = (2/e)(C X_j - c X_j) - X_{j-1} (written by the product's own stores: EigEpi)
            double *prev = R, *cur = Y, *free1 = Z, *free2 = Q;
            // While the residual is far from the tolerance the filter's products read the FLOAT32 copy of C (half the bytes
            // of a stream that is all of the product's time): the filter only has to amplify the wanted directions — the
            // Ritz values, the residual and the convergence test come from the float64 product of the Rayleigh-Ritz step,
            // and a float32 C limits the reachable residual to ~1e-7 theta_max, so the last steps filter in float64.
            const bool f32 = ctx.G32b != nullptr && res > PLD_F32_RES * th0;
            for (int pw = 1; pw < npow; ++pw) {
                double *nxt = free1;
                const EigEpi ep{2.0 * ie, 2.0 * ie * cc, 1.0, prev};
                if (f32 && res > PLD_RR32_RES * th0)  // far from the stop: float32 matrix cores (noise floor ~2e-6 of the subspace)
                    eig_cq32<NA, true>(ctx, cur, nxt, ep);
                else if (f32)                         // float32 C, float64 products: a FIXED perturbation of C, floor ~1e-8
                    eig_cq<NA, true, true>(ctx, cur, nxt, ep);
                else
                    eig_cq<NA, false, true>(ctx, cur, nxt, ep);
                free1 = free2;
                free2 = prev;
                prev = cur;
                cur = nxt;
            }
            src = cur;
        } else {
            for (int pw = 1; pw < npow; ++pw) {
                eig_cq<NA>(ctx, src, dst);
                double *t2 = src;
                src = dst;
                dst = t2;
            }
        }
        lap(5);
        if (!eig_cholqr<NA>(ctx, src, Q)) {  // breakdown: one plain step from the (orthonormal) Ritz vectors, SVQB
            eig_cq<NA>(ctx, R, Q);
            eig_svqb<NA>(ctx, Q);
        }
        lap(6);
    }
    for (int e = tid; e < P * k; e += nt) {
        const int i = e / k, a = e % k;
        Vb[e] = R[(size_t)i * l + a];
    }
    if (tid == 0 && iters_out) {  // debug (LK_PLD_ITERS=1): step count and per-phase 100 MHz ticks of this workgroup
        iters_out[(size_t)b * 8] = it;
        for (int s2 = 0; s2 < 7; ++s2) iters_out[(size_t)b * 8 + 1 + s2] = tprof[s2];
    }
    if (tid == 0 && status) status[b] = converged ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ phase-split iteration
// Round 6.  The one-kernel iteration above keeps a matrix on one workgroup from the random start to the last Ritz vector: its
// products with C (HBM streams) and its l x l eigenproblems / Cholesky factorisations (latency chains on one wave) share one
// register budget (128 VGPRs, ~60 spilled around the calls), and since the 500 workgroups of a batch start together and do
// identical work, the two workgroups of a CU sit in the SAME phase at the same time — the serial stretches were never hidden
// behind the other workgroup's stream (per matrix 2.5 ms of serial phases + 4.95 ms of products = the kernel's 7.45 ms).
// Here every phase is its own launch over all matrices, state in global memory:
//   pld_eigs_init_kernel   mirror the Gram blocks, pseudo-random start, Cholesky-QR            (once)
//   pld_eigs_prod_kernel   dst = C src with the filter recurrence fused into its stores        (3 per step: Rayleigh-Ritz, filter 1, 2)
//   pld_eigs_rr_kernel     Q^T Z, l x l Jacobi, Ritz vectors + residual, stop test, filter plan
//   pld_eigs_orth_kernel   Cholesky-QR of the filtered block
// A product launch is nothing but streams (deep register prefetch, no serial phase inside); the small phases run two workgroups
// per CU with nobody streaming beside them.  Per-matrix control flow lives in EigsState: converged matrices drop out of every
// later launch by their `done` flag, precision follows the residual (float32 matrix cores -> float32 C with float64 products ->
// float64), and whatever has not converged after the fixed number of steps the launcher queues — or hits a Cholesky breakdown —
// is handed to the one-kernel iteration (which restarts it; `skip` = the converged flags).  No host synchronisation.
struct EigsLds {
    int T, W, rot, shred, vec, order, qstage;
};
__device__ __forceinline__ EigsLds eigs_lds(int l, int nt) {
    const int ld = l + 1;
    EigsLds o;
    o.T = 0;
    o.W = l * ld;
    o.rot = 2 * l * ld;
    o.shred = o.rot + 2 * l;
    o.vec = o.shred + nt;
    o.order = o.vec + 2 * l;
    o.qstage = o.order + (l + 1) / 2 + 1;
    return o;
}
static size_t eigs_small_lds_bytes(int l, int nt, int kc) {
    const int ld = l + 1;
    return ((size_t)2 * l * ld + 2 * l + nt + 2 * l + (l + 1) / 2 + 1 + (size_t)kc * PLD_QS) * 8 + 64;
}

template <int NA>
__global__ __launch_bounds__(512) void pld_eigs_init_kernel(double *__restrict__ G, int ldg, int P, int k, int l,
                                                            double *__restrict__ scratch, EigsState *__restrict__ state, int kc,
                                                            int mirror, int have32) {
    const int tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
    double *Gb = G + (size_t)b * ldg * ldg;
    const EigsLds o = eigs_lds(l, nt);
    const EigCtx ctx{o.T, o.W, o.rot, o.shred, o.vec, o.qstage, o.order, Gb, nullptr, ldg, P, k, l, l + 1, kc, 0};
    double *Q = scratch + (size_t)b * 4 * P * l;
    if (mirror)
        for (int e = tid; e < P * P; e += nt) {
            const int i = e / P, j = e - i * P;
            if ((j >> 6) < (i >> 6)) Gb[(size_t)i * ldg + j] = Gb[(size_t)j * ldg + i];
        }
    for (int e = tid; e < P * l; e += nt) {  // the deterministic pseudo-random start of the one-kernel iteration
        unsigned int x = (unsigned int)(e + 1) * 2654435761u;
        x ^= x >> 15;
        x *= 2246822519u;
        x ^= x >> 13;
        Q[e] = (double)(x & 0xffffffu) / 8388608.0 - 1.0;
    }
    __syncthreads();
    if (!eig_cholqr<NA>(ctx, Q, Q)) eig_svqb<NA>(ctx, Q);
    if (tid == 0) {
        EigsState st;
        st.done = 0;
        st.converged = 0;
        st.it = 0;
        st.rr_var = have32 ? 0 : 2;
        st.f_var = 2;
        st.cheb = 0;
        st.pad[0] = st.pad[1] = 0;
        st.cc = st.ie = 0.0;
        st.res = 1e300;
        st.th0 = 0.0;
        state[b] = st;
    }
}

// which = 0: Z = C Q (Rayleigh-Ritz product, variant rr_var); 1: Z = a C Y - b Y - g R; 2: Q = a C Z - b Z - g Y (filter, f_var)
// with (a, b, g) = (2 / e, 2 c / e, 1) for the Chebyshev recurrence and (1, 0, 0) for plain powers.
template <int NA>
__global__ __launch_bounds__(512, 4) void pld_eigs_prod_kernel(const double *__restrict__ G, const float *__restrict__ G32, int ldg,
                                                               int P, int l, double *__restrict__ scratch,
                                                               const EigsState *__restrict__ state, int which, int kc, int kc32) {
    const int b = blockIdx.x;
    const EigsState st = state[b];
    if (st.done) return;
    double *Q = scratch + (size_t)b * 4 * P * l, *Z = Q + (size_t)P * l, *R = Z + (size_t)P * l, *Y = R + (size_t)P * l;
    const EigCtx ctx{0, 0, 0, 0, 0, 0, 0, const_cast<double *>(G) + (size_t)b * ldg * ldg, G32 ? G32 + (size_t)b * ldg * ldg : nullptr,
                     ldg, P, 0, l, l + 1, kc, kc32};
    const double *src = which == 0 ? Q : which == 1 ? Y : Z, *prev = which == 1 ? R : Y;
    double *dst = which == 2 ? Q : Z;
    const int var = which == 0 ? st.rr_var : st.f_var;
    if (which == 0) {
        if (var == 0)
            eig_cq32<NA>(ctx, src, dst);
        else
            eig_cq<NA>(ctx, src, dst);
        return;
    }
    const EigEpi ep = st.cheb ? EigEpi{2.0 * st.ie, 2.0 * st.ie * st.cc, 1.0, prev} : EigEpi{1.0, 0.0, 0.0, prev};
    if (var == 0)
        eig_cq32<NA, true>(ctx, src, dst, ep);
    else if (var == 1)
        eig_cq<NA, true, true>(ctx, src, dst, ep);
    else
        eig_cq<NA, false, true>(ctx, src, dst, ep);
}

template <int NA>
__global__ __launch_bounds__(512) void pld_eigs_rr_kernel(int P, int k, int l, double *__restrict__ scratch,
                                                          EigsState *__restrict__ state, double *__restrict__ V,
                                                          double *__restrict__ lam, int kc, double tol, int cheb_on, int last,
                                                          int have32) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
    const EigsState st = state[b];
    if (st.done) return;
    const EigsLds o = eigs_lds(l, nt);
    const int ld = l + 1;
    const EigCtx ctx{o.T, o.W, o.rot, o.shred, o.vec, o.qstage, o.order, nullptr, nullptr, 0, P, k, l, ld, kc, 0};
    double *T = lds + o.T, *W = lds + o.W, *shred = lds + o.shred, *vec = lds + o.vec, *theta = vec + l;
    int *order = reinterpret_cast<int *>(lds + o.order);
    double *Q = scratch + (size_t)b * 4 * P * l, *Z = Q + (size_t)P * l, *R = Z + (size_t)P * l, *Y = R + (size_t)P * l;
    eig_xty(ctx, Q, Z, o.W);  // T = Q^T Z (symmetrised)
    for (int e = tid; e < l * l; e += nt) {
        const int a = e / l, c = e % l;
        T[a * ld + c] = 0.5 * (W[a * ld + c] + W[c * ld + a]);
    }
    __syncthreads();
    jacobi_eig_lds<false>(o.T, o.W, l, ld, o.rot, o.shred);
    sort_desc_lds(T, l, ld, order);
    if (tid < l) theta[tid] = T[order[tid] * ld + order[tid]];
    __syncthreads();
    for (int e = tid; e < l * l; e += nt) {
        const int a = e / l, c = e % l;
        T[a * ld + c] = W[a * ld + order[c]];
    }
    __syncthreads();
    const double th0 = fabs(theta[0]), th_k = theta[k - 1], th_cut = theta[l - 1];
    const bool cheb = th_cut > 0.0 && (((cheb_on & 1) && th_k < 1.5 * th_cut && th0 < 30.0 * th_cut) || (cheb_on & 2));
    const double cc = 0.5 * th_cut, ie = cheb ? 1.0 / cc : 0.0;
    // R = Q W (Ritz vectors by Ritz value), Y = Z W = C R and the residual of the k wanted pairs in one pass; with the Chebyshev
    // filter Y is stored as the recurrence's first term X1 = (C R - c R) / e straight away (unused if this step converges)
    const double part = eig_xm<NA, true>(ctx, Q, R, Z, Y, o.T, o.vec + l, cc, ie);
    const double res = sqrt(block_sum_dyn(part, shred));
    // only a float64 product can declare convergence
    const bool conv = st.rr_var == 2 && res <= tol * th0 * sqrt((double)k);
    if (conv) {
        double *Vb = V + (size_t)b * P * k;
        for (int e = tid; e < P * k; e += nt) Vb[e] = R[(size_t)(e / k) * l + (e % k)];
        if (tid < k) lam[(size_t)b * k + tid] = theta[tid];
    }
    if (tid == 0) {
        EigsState nx = st;
        nx.it = st.it + 1;
        nx.res = res;
        nx.th0 = th0;
        nx.converged = conv ? 1 : 0;
        nx.done = (conv || last) ? 1 : 0;
        nx.cheb = cheb ? 1 : 0;
        nx.cc = cc;
        nx.ie = ie;
        const bool far = have32 && res > PLD_RR32_RES * th0;
        nx.rr_var = far ? 0 : 2;
        nx.f_var = far ? 0 : (have32 && res > PLD_F32_RES * th0) ? 1 : 2;
        state[b] = nx;
    }
}

template <int NA>
__global__ __launch_bounds__(512) void pld_eigs_orth_kernel(int P, int k, int l, double *__restrict__ scratch,
                                                            EigsState *__restrict__ state, int kc) {
    const int tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
    if (state[b].done) return;
    const EigsLds o = eigs_lds(l, nt);
    const EigCtx ctx{o.T, o.W, o.rot, o.shred, o.vec, o.qstage, o.order, nullptr, nullptr, 0, P, k, l, l + 1, kc, 0};
    double *Q = scratch + (size_t)b * 4 * P * l;
    if (!eig_cholqr<NA>(ctx, Q, Q)) {  // breakdown (never seen on PLD blocks): the one-kernel iteration restarts this matrix
        if (tid == 0) state[b].done = 1;
    }
}

// U = A V diag(lam)^-1/2 into X[:, col0 : col0 + k] on the fp64 matrix cores.  One WAVE = 16 rows of A against all of V
// (k <= 64 components = 4 column tiles), no LDS and no barriers: the first version staged 64 x 64 tiles of A and V through
// LDS with two workgroup barriers around 16 MFMAs per wave and ran 15x off the HBM time of A.  v_mfma_f64_16x16x4:
// A operand [row = lane & 15][kk = lane >> 4], B operand [kk = lane >> 4][col = lane & 15], D [row = (lane >> 4) + 4 r]
// [col = lane & 15].  The summation index of an MFMA step is free to permute: lane group q = lane >> 4 owns columns
// p0 + 4 q .. + 3 of A (one 32-byte load when P % 4 == 0 — the four groups cover a 128-byte line of each row) and feeds
// component j at step j, with the matching row p0 + 4 q + j of V as the B operand (4 x 128-byte rows, from L1/L2).
#ifndef PROJ_U4
#define PROJ_U4 1
#endif
typedef double pld_d4u __attribute__((ext_vector_type(4), aligned(8)));
template <int KT, bool VEC4>  // KT = 16-column tiles of V (k <= 16 KT); VEC4: P % 4 == 0, rows of A are 32-byte aligned
__global__ __launch_bounds__(256) void pld_project_kernel(const double *__restrict__ A, const double *__restrict__ V,
                                                           const double *__restrict__ lam, int N, int P, int k, int ldx,
                                                           int col0, double *__restrict__ X) {
    const int b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = (blockIdx.x * 4 + wave) * 16;
    if (n0 >= N) return;
    const int lq = lane >> 4, lr = lane & 15, row = n0 + lr;
    const double *Ar = A + ((size_t)b * N + min(row, N - 1)) * P;
    const double *Vb = V + (size_t)b * P * k;
    // loads are UNCONDITIONAL on clamped addresses and masked afterwards BY A 0/1 FACTOR: guarded loads end up behind
    // branches and are no longer issued back to back — and so does `cond ? loaded : 0.0`, which the compiler turns back
    // into a branch around the load with a full wait behind it (round 4: every load of A and V of this kernel was
    // serialised that way, 32 round trips per 64 columns).  A product cannot be if-converted.  Clamped addresses hold
    // finite values of the same matrix, so 0 x value is 0.
    auto load_a = [&](int p0) -> pld_d4 {
        const int p = p0 + 4 * lq;
        pld_d4 v;
        if (VEC4) {
            v = *reinterpret_cast<const pld_d4 *>(Ar + min(p, P - 4));
            const double f = (row < N && p < P) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= f;
        } else {
#if PROJ_U4
            // P not a multiple of four (121: rows neither 16- nor 32-byte aligned): still ONE 32-byte request per lane — four
            // 8-byte loads per lane put ~24 cache lines under every instruction and fetched each line four times through an L1
            // that 32 waves' rows do not fit.  The last group starts at P - 4 and its elements are shifted into place.
            const int st = min(p, P - 4), sh = p - st;  // sh = 0 except in the row's last group
            const pld_d4u raw = *reinterpret_cast<const pld_d4u *>(Ar + st);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double x = sh == 0 ? raw[j] : (sh == 1 ? raw[min(j + 1, 3)] : (sh == 2 ? raw[min(j + 2, 3)] : raw[3]));
                v[j] = x * ((row < N && p + j < P) ? 1.0 : 0.0);
            }
#else
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = Ar[min(p + j, P - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= (row < N && p + j < P) ? 1.0 : 0.0;
#endif
        }
        return v;
    };
    pld_d4 acc[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) acc[c] = pld_d4{0.0, 0.0, 0.0, 0.0};
    for (int p0 = 0; p0 < P; p0 += 64) {  // 64 columns per trip: four 32-byte loads of A and 16 kt loads of V in flight
        pld_d4 a4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a4[u] = load_a(p0 + 16 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = p0 + 16 * u + 4 * lq + j;
#pragma unroll
                for (int c = 0; c < KT; ++c) {
                    const int col = c * 16 + lr;
                    const double vraw = Vb[(size_t)min(p, P - 1) * k + min(col, k - 1)];
                    const double bv = vraw * ((p < P && col < k) ? 1.0 : 0.0);
                    acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[u][j], bv, acc[c], 0, 0, 0);
                }
            }
    }
#pragma unroll
    for (int c = 0; c < KT; ++c) {
        const int a = c * 16 + lr;
        if (a < k) {
            const double sc = sqrt(fmax(lam[(size_t)b * k + a], 1e-300));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + lq + 4 * r;
                if (n < N) X[((size_t)b * N + n) * ldx + col0 + a] = acc[c][r] / sc;
            }
        }
    }
}

// The same projection straight from the float32 pixels (round 6): A = (double)(pix / div[row]) - mean[col] is formed in the lane
// that feeds it to the matrix cores — pld_ratio_kernel used to write A out as float64 (1.7 GB per 121-column block of 500 cutouts)
// for the Gram kernel and this one to read back.  Same arithmetic, same summation order: the same bits.  Needs P >= 4.
typedef float pld_f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int KT>
__global__ __launch_bounds__(256) void pld_project_f32_kernel(const float *__restrict__ pix, const float *__restrict__ divv,
                                                               const double *__restrict__ mean, int mode,
                                                               const double *__restrict__ V, const double *__restrict__ lam,
                                                               int N, int P, int k, int ldx, int col0, double *__restrict__ X) {
    const int b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = (blockIdx.x * 4 + wave) * 16;
    if (n0 >= N) return;
    const int lq = lane >> 4, lr = lane & 15, row = n0 + lr, rowc = min(row, N - 1);
    const float *Pr = pix + ((size_t)b * N + rowc) * P;
    const double *mb = mean + (size_t)b * P, *Vb = V + (size_t)b * P * k;
    const float dv = (mode != 0 && divv) ? divv[(size_t)b * N + rowc] : 1.0f;
    auto load_a = [&](int p0) -> pld_d4 {
        const int p = p0 + 4 * lq;
        const int st = min(p, P - 4), sh = p - st;  // (sh = 0 except in the row's last group)
        const pld_f4u raw = *reinterpret_cast<const pld_f4u *>(Pr + st);
        const pld_d4u mr = *reinterpret_cast<const pld_d4u *>(mb + st);
        pld_d4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int js = min(j + sh, 3);
            const float x = js == 0 ? raw[0] : (js == 1 ? raw[1] : (js == 2 ? raw[2] : raw[3]));
            const double m = js == 0 ? mr[0] : (js == 1 ? mr[1] : (js == 2 ? mr[2] : mr[3]));
            const double q = (double)(mode == 0 ? x : x / dv) - m;
            v[j] = (row < N && p + j < P) ? q : 0.0;  // (a select, not a product: q may be Inf / NaN where the reference's is, but only inside the matrix)
        }
        return v;
    };
    pld_d4 acc[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) acc[c] = pld_d4{0.0, 0.0, 0.0, 0.0};
    for (int p0 = 0; p0 < P; p0 += 64) {
        pld_d4 a4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a4[u] = load_a(p0 + 16 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = p0 + 16 * u + 4 * lq + j;
#pragma unroll
                for (int c = 0; c < KT; ++c) {
                    const int col = c * 16 + lr;
                    const double vraw = Vb[(size_t)min(p, P - 1) * k + min(col, k - 1)];
                    const double bv = vraw * ((p < P && col < k) ? 1.0 : 0.0);
                    acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[u][j], bv, acc[c], 0, 0, 0);
                }
            }
    }
#pragma unroll
    for (int c = 0; c < KT; ++c) {
        const int a = c * 16 + lr;
        if (a < k) {
            const double sc = sqrt(fmax(lam[(size_t)b * k + a], 1e-300));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + lq + 4 * r;
                if (n < N) X[((size_t)b * N + n) * ldx + col0 + a] = acc[c][r] / sc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ direct solver, P <= 138
// Top-k eigenpairs of a Gram matrix that fits LDS (the 121-column pixel and background blocks, the 136-column 2nd-order
// product block), computed DIRECTLY instead of by subspace iteration (whose Rayleigh-Ritz steps — a 32 x 32 Jacobi and a
// Cholesky-QR each — cost 2-3 ms per launch in barrier-separated latency-bound phases): the LAPACK route for a few
// eigenpairs of a dense symmetric matrix,
//   1. Householder tridiagonalisation of the packed lower triangle in LDS (two barriers per column: the symmetric
//      matrix-vector product and the rank-2 update; norms and inner products are recomputed by every wave — cheaper than a
//      barrier); the reflectors stay in the eliminated columns (unscaled, with their scale factors beside them);
//   2. the k largest eigenvalues of the tridiagonal matrix by multisection on Sturm counts, one wave per eigenvalue
//      (64 trial points per round: 10 rounds reach the last bit);
//   3. their eigenvectors from the twisted factorisation of T - lambda I (one lane per vector, one pass), modified
//      Gram-Schmidt inside clusters of close eigenvalues;
//   4. back-transformation by the reflectors, one wave per vector (the vector in registers), no barrier.
// Residuals ||C v - lambda v|| <= 1e-15 lambda_max in the numpy restatement this was written from (steep pixel-block
// spectra, flat product-block spectra, repeated eigenvalues).  One 1024-thread workgroup per matrix (LDS: 76 KB triangle +
// 18 KB vectors + 36 KB scratch of the twisted factorisations).
constexpr int TD_NT = 1024, TD_LANES = 16;  // eigenvectors factorised at a time (LDS scratch 2 P doubles each); 8 where that lets two
                                            // workgroups share a CU (td_plan)

// Wave-wide sum in every lane without the LDS crossbar: four DPP levels inside each row of 16 lanes (quad swaps, half-row and
// row mirrors), then the four row sums through v_readlane.  ~10x shorter than six dependent ds_bpermute round trips — the
// direct solver's phases are chains of such reductions.
template <int CTRL>
__device__ __forceinline__ double td_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double td_readlane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ double td_quad_sum(double x) {  // sum over each aligned group of 4 lanes, in all 4
    x += td_dpp<0xB1>(x);   // quad_perm [1, 0, 3, 2]
    x += td_dpp<0x4E>(x);   // quad_perm [2, 3, 0, 1]
    return x;
}
__device__ __forceinline__ double td_wave_sum(double x) {
    x = td_quad_sum(x);
    x += td_dpp<0x141>(x);  // row_half_mirror: the other quad of each group of 8
    x += td_dpp<0x140>(x);  // row_mirror: the other half of the row of 16
    return (td_readlane(x, 0) + td_readlane(x, 16)) + (td_readlane(x, 32) + td_readlane(x, 48));
}

__device__ __forceinline__ int td_tri(int i, int j) { return ((i * (i + 1)) >> 1) + j; }  // j <= i

// Round 6: the eigenvectors live in their output array V (global, L2-resident) instead of LDS and the twisted factorisations run
// `tdl` (8 or 16) at a time, so that a 121-column matrix needs 79 KB of LDS instead of 111 — with the kernel at 64 VGPRs TWO
// 1024-thread workgroups share a CU and a batch of 500 runs as one wave of workgroups whose barrier-separated Householder steps
// interleave, instead of two rounds of one latency-bound workgroup per CU.
__global__ __launch_bounds__(TD_NT, 8) void pld_tridiag_eig_kernel(const double *__restrict__ G, int ldg, int P, int k,
                                                                    double *__restrict__ V, double *__restrict__ lam,
                                                                    unsigned long long *__restrict__ clk, int tdl) {
#ifdef LK_PLD_DEBUG   // per-phase clocks of matrix 0 (100 MHz wall clock), `make DEBUG=1`
#define TD_CLK(slot)                                                                 \
    do {                                                                             \
        if (clk && threadIdx.x == 0) {                                               \
            const unsigned long long c_ = wall_clock64();                            \
            if (blockIdx.x == 0) clk[slot] = c_;                                     \
            clk[8 + 4 * (size_t)gridDim.x + 8 * (size_t)blockIdx.x + slot] = c_;    \
        }                                                                            \
    } while (0)
#else
#define TD_CLK(slot) do { } while (0)
    (void)clk;
#endif
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    constexpr int NW = TD_NT / 64;
#ifdef LK_PLD_DEBUG   // every workgroup's start, end and place (XCC, HW_ID): clk[8 + 4 b ..]
    if (clk && tid == 0) {
        clk[8 + 4 * (size_t)b] = wall_clock64();
        clk[8 + 4 * (size_t)b + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
    }
#endif
    const double *Gb = G + (size_t)b * ldg * ldg;
    const int ntri = (P * (P + 1)) >> 1;
    double *At = lds;                       // packed lower triangle
    double *dd = At + ((ntri + 1) & ~1);    // diagonal of T
    double *ee = dd + P;                    // ee[i] = T[i + 1][i]
    double *tau = ee + P;                   // reflector factors
    double *scl = tau + P;                  // v = [1, x[1:] * scl]
    double *pb = scl + P;                   // p = tau S v
    double *vb = pb + P;                    // the current reflector
    double *e2 = pb;                        // squared off-diagonal of T (phase 2 on)
    double *lamv = vb + P;                  // k eigenvalues (descending)
    double *zinv = lamv + ((k + 1) & ~1);   // 1 / norm of the k twisted-factorisation vectors
    double *ws = zinv + ((k + 1) & ~1);     // tdl x 2 x P scratch of the twisted factorisations, [which][i][lane]
    double *Z = V + (size_t)b * P * k;      // P x k eigenvectors (row-major): the output array itself
    // ---- load the lower triangle (G's upper triangle is always valid: element (i, j <= i) = G[j][i])
    for (int j = wave; j < P; j += NW)  // row j of G from its diagonal on: coalesced
        for (int i = j + lane; i < P; i += 64) At[td_tri(i, j)] = Gb[(size_t)j * ldg + i];
    __syncthreads();
    TD_CLK(0);
    // ---- 1. tridiagonalisation
    // What bounds it (round 6, the workgroup timeline of a 500-matrix launch in the development build: every workgroup's start,
    // end and phase clocks): INSTRUCTION ISSUE.  Alone on a CU a workgroup takes ~370 us here; of two sharing a CU the first
    // takes ~400 and the second ~800 — the older waves go first and leave it nothing.  So instructions are what to save, in
    // every wave: the reflector (norm, square root, two divisions) is formed by wave 0 alone — the others pick tau up behind the
    // barrier they wait at anyway — and waves whose rows lie beyond the trailing block skip the product.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int kk = 0; kk < P - 2; ++kk) {
        const int m = P - kk - 1, r0 = kk + 1;  // trailing block: rows / columns r0 .. P - 1; x_i = A[r0 + i][kk]
        if (wave_u == 0) {
            // alpha, sigma = sum_{i >= 1} x_i^2, the column in registers (m <= 137: three entries per lane)
            double x0 = At[td_tri(r0 + min(lane, m - 1), kk)];
            double x1 = At[td_tri(r0 + min(lane + 64, m - 1), kk)];
            double x2 = At[td_tri(r0 + min(lane + 128, m - 1), kk)];
            if (lane >= m) x0 = 0.0;
            if (lane + 64 >= m) x1 = 0.0;
            if (lane + 128 >= m) x2 = 0.0;
            const double alpha = td_readlane(x0, 0);
            const double sigma = td_wave_sum(fma(lane == 0 ? 0.0 : x0, x0, fma(x1, x1, x2 * x2)));
            double tk0 = 0.0, sc = 0.0, beta = alpha;
            if (sigma != 0.0) {
                const double nrm = sqrt(fma(alpha, alpha, sigma));
                beta = alpha >= 0.0 ? -nrm : nrm;
                tk0 = (beta - alpha) / beta;
                sc = 1.0 / (alpha - beta);
            }
            if (lane == 0) {
                dd[kk] = At[td_tri(kk, kk)];
                ee[kk] = beta;
                tau[kk] = tk0;
                scl[kk] = sc;
            }
            // v = [1, x[1:] * scale] into LDS (vb)
            if (lane < m) vb[lane] = lane == 0 ? 1.0 : x0 * sc;
            if (lane + 64 < m) vb[lane + 64] = x1 * sc;
            if (lane + 128 < m) vb[lane + 128] = x2 * sc;
        }
        __syncthreads();
        const double tk = tau[kk];  // (broadcast read: workgroup-uniform)
        if (tk != 0.0) {
            // p = tau S v with EIGHT threads per row (an eighth of the columns each, three DPP steps).  The row's chunk is
            // walked four columns at a time with all eight LDS reads issued before the first multiply-add (clamped addresses,
            // masked operands): the plain loop compiled to one LDS round trip per column — ds_read x 2, s_waitcnt lgkmcnt(0),
            // v_fmac — twice over (row part, column part, each to the wave's longest trip).
            {
                const int q = tid & 7;
                const int jq = (m + 7) >> 3, j_lo = q * jq, j_hi = min(m, j_lo + jq);
                for (int ib = wave_u * 8; ib < m; ib += 128) {  // (wave-uniform trips: all lanes reach the DPP sums)
                    const int i = ib + (lane >> 3);
                    double acc = 0.0;
                    if (i < m) {
                        const int rowbase = td_tri(r0 + i, r0), colbase = r0 + i;
                        for (int j0 = j_lo; j0 < j_hi; j0 += 4) {
                            double a[4], v[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int jc = min(j0 + u, j_hi - 1), rj = r0 + jc;
                                // S_ij: row i for j <= i, column i (row j) for j > i
                                const int addr = jc <= i ? rowbase + jc : (int)(__umul24(rj, rj + 1) >> 1) + colbase;
                                a[u] = At[addr];
                                v[u] = vb[jc];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) acc = fma(a[u], j0 + u < j_hi ? v[u] : 0.0, acc);
                        }
                    }
                    acc = td_quad_sum(acc);
                    acc += td_dpp<0x141>(acc);  // row_half_mirror: the other quad of the row's eight threads
                    if (i < m && q == 0) pb[i] = tk * acc;
                }
            }
            __syncthreads();
            // every wave with rows to update: K = tau / 2 p^T v;  w = p - K v
            if (wave_u * 4 < ((m + 1) >> 1)) {
            double dot = 0.0;
            {
                double pp[3], vv[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int i = lane + 64 * u, ic = min(i, m - 1);
                    pp[u] = i < m ? pb[ic] : 0.0;
                    vv[u] = vb[ic];
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) dot = fma(pp[u], vv[u], dot);
            }
            const double Kc = 0.5 * tk * td_wave_sum(dot);
            // rank-2 update of the lower triangle, S_ij -= v_i w_j + w_i v_j.  Row i holds i + 1 entries: rows i and m - 1 - i are
            // walked as ONE strip of m + 1 entries by sixteen threads, four entries each per batch — every thread has the same
            // two batches at m = 120 where a thread-group per row left the last rows five (the phase ends with its slowest wave).
            // The operands of a batch are all loaded before its read-modify-writes (left to the compiler, every store to the
            // triangle fences the loads behind it: they may alias).  The arithmetic per entry is unchanged.
            {
                const int q = tid & 15, npair = (m + 1) >> 1;
                for (int pi = tid >> 4; pi < npair; pi += TD_NT / 16) {
                    const int ia = pi, ib = m - 1 - pi;
                    const double via = vb[ia], vib = vb[ib];
                    const double wia = pb[ia] - Kc * via, wib = pb[ib] - Kc * vib;
                    const int base_a = td_tri(r0 + ia, r0), base_b = td_tri(r0 + ib, r0) - (ia + 1);
                    const int len = ia == ib ? ia + 1 : m + 1;
                    for (int p0 = q; p0 < len; p0 += 64) {
                        double vj[4], pj[4], av[4];
                        int ad[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int pos = min(p0 + 16 * u, len - 1);
                            const bool in_a = pos <= ia;
                            const int c = in_a ? pos : pos - ia - 1;
                            ad[u] = (in_a ? base_a : base_b) + pos;
                            vj[u] = vb[c];
                            pj[u] = pb[c];
                            av[u] = At[ad[u]];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int pos = p0 + 16 * u;
                            if (pos < len) {
                                const bool in_a = pos <= ia;
                                const double vr = in_a ? via : vib, wr = in_a ? wia : wib;
                                At[ad[u]] = av[u] - fma(vr, pj[u] - Kc * vj[u], wr * vj[u]);
                            }
                        }
                    }
                }
            }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (P >= 2) {
            dd[P - 2] = At[td_tri(P - 2, P - 2)];
            ee[P - 2] = At[td_tri(P - 1, P - 2)];
        }
        dd[P - 1] = At[td_tri(P - 1, P - 1)];
        ee[P - 1] = 0.0;
    }
    __syncthreads();
    TD_CLK(1);
    // ---- 2. eigenvalues: Gershgorin bounds, then multisection on Sturm counts (a wave per eigenvalue)
    double gl = INFINITY, gu = -INFINITY, e2max = 0.0;
    for (int i = 0; i < P; ++i) {  // (every thread: P <= 138 LDS broadcasts)
        const double r = (i > 0 ? fabs(ee[i - 1]) : 0.0) + (i < P - 1 ? fabs(ee[i]) : 0.0);
        gl = fmin(gl, dd[i] - r);
        gu = fmax(gu, dd[i] + r);
        if (i < P - 1) e2max = fmax(e2max, ee[i] * ee[i]);
    }
    // Sturm counts in the PRODUCT form on the matrix scaled to norm ~1 (ddn = d / |T|, e2 = (e / |T|)^2; pb and vb are free
    // after the tridiagonalisation): p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2}, eigenvalues below x = sign changes of p.
    // Three fp64 operations per step where the quotient form (q_i = d_i - x - e^2 / q_{i-1}, a v_rcp_f64 + Newton step)
    // needs seven: with 1024 trial points on the CU this phase is bound by fp64 issue, not by the chain's latency
    // (measured: four steps of operands in flight changed nothing).  p is rescaled every 8 steps (it grows by at most ~3
    // per step on the scaled matrix); a zero takes the sign opposite to its predecessor's, which is what the quotient
    // form's q = -pivmin meant.
    const double tnorm = fmax(fabs(gl), fabs(gu)), eps = 2.220446049250313e-16;
    const double tinv = tnorm > 0.0 ? 1.0 / tnorm : 1.0;
    double *ddn = vb;
    // (round 6) e2 is kept >= 1e-140: after an exactly vanishing p_i the next term -e_i^2 p_{i-1} is then non-zero, so the loop
    // below needs no test for zeros at all (a zero may carry either sign: the sign changes over it and its successor add up to
    // one, the Sturm convention) — and a block that decouples exactly (a constant pixel: e = 0) cannot stall the recurrence.  The
    // shift of the eigenvalues is < 1e-70 |T|.
    for (int i = tid; i < P; i += TD_NT) {
        ddn[i] = dd[i] * tinv;
        if (i < P - 1) e2[i] = fmax((ee[i] * tinv) * (ee[i] * tinv), 1e-140);
    }
    __syncthreads();
    const double pivmin = 2.2250738585072014e-308 * fmax(1.0, e2max);
    // Sign changes are counted on the sign BITS (xor of the high words, shift, add: three 32-bit operations) and p is rescaled once per
    // 8 steps outside the unrolled body: 6 VALU instructions per step.  Round 5's form (a zero test and the rescale's two compares,
    // two products and eight selects evaluated EVERY step: 45 instructions, profiles/r06_pld_tridiag.txt) made this phase 300-380 us
    // of the kernel's ~950 per matrix, bound by instruction issue with one wave per eigenvalue.
    auto count_below = [&](double x) {  // eigenvalues of T below x
        const double xs = x * tinv;
        double p0 = 1.0, p1 = ddn[0] - xs;
        int cnt = (int)((unsigned)__double2hiint(p1) >> 31);
        auto step = [&](double di, double ei) {
            const double pn = fma(di - xs, p1, -(ei * p0));
            cnt += (int)((unsigned)(__double2hiint(pn) ^ __double2hiint(p1)) >> 31);
            p0 = p1;
            p1 = pn;
        };
        int i = 1;
        for (; i + 8 <= P; i += 8) {
            double d8[8], e8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {  // (LDS broadcasts, all requested before the chain starts)
                d8[u] = ddn[i + u];
                e8[u] = e2[i + u - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) step(d8[u], e8[u]);
            // |p| grows or shrinks by at most ~3 per step on the scaled matrix (8 steps: < 1e4): one test per block keeps it in range
            const double ap = fmax(fabs(p1), fabs(p0));
            const double sc = ap > 1e150 ? 1e-150 : (ap < 1e-150 ? 1e150 : 1.0);
            p1 *= sc;
            p0 *= sc;
        }
        for (; i < P; ++i) step(ddn[i], e2[i - 1]);
        return cnt;
    };
    for (int j = wave; j < k; j += NW) {
        const int want = P - 1 - j;  // ascending index of the j-th largest
        const double pad = 2.0 * tnorm * eps * P + 2.0 * pivmin;
        double lo = gl - pad, hi = gu + pad;  // invariant: count_below(lo) <= want < count_below(hi)
        for (int round = 0; round < 14; ++round) {
            const double step = (hi - lo) / 65.0;
            const double x = lo + step * (double)(lane + 1);
            const bool below = !(x < hi) ? false : (count_below(x) <= want);  // trial points that are still "lo" candidates
            const unsigned long long bal = __ballot(below);
            const int nlo = __popcll(bal);  // monotone in x: the first nlo points are <= the eigenvalue's side
            const double nlo_x = lo + step * (double)nlo, nhi_x = nlo < 64 ? lo + step * (double)(nlo + 1) : hi;
            const double new_lo = nlo > 0 ? nlo_x : lo;
            if (!(new_lo > lo) && !(nhi_x < hi)) break;
            lo = new_lo;
            hi = fmin(hi, nhi_x);
            if (hi - lo <= 2.0 * eps * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin) break;
        }
        if (lane == 0) lamv[j] = 0.5 * (lo + hi);
    }
    __syncthreads();
    TD_CLK(2);
    // ---- 3. eigenvectors of T from the twisted factorisation of T - lambda I (Parlett & Dhillon), one lane per vector:
    // D+ / L from the top, D- / U from the bottom, the twist index r where |gamma_i| = |D+_i + D-_i - (d_i - lambda)| is smallest,
    // z_r = 1, z_{i+1} = -U_i z_i below it and z_{i-1} = -L_{i-1} z_i above it.  One pass, 2 P doubles of scratch per vector
    // (the LU-based inverse iteration this replaced kept 4 P and swept the matrix four times).
    const double cl_tol = 1e-3 * tnorm;  // "close" eigenvalues: orthogonalised against each other afterwards
    for (int j0 = 0; j0 < k; j0 += tdl) {
        if (wave == 0 && lane < tdl && j0 + lane < k) {
            const int j = j0 + lane;
            double x0 = lamv[j];
            // equal eigenvalues are perturbed apart so that their factorisations (hence their vectors) differ
            int rank_in_cluster = 0;
            for (int jj = 0; jj < j; ++jj) rank_in_cluster += fabs(lamv[jj] - lamv[j]) < 10.0 * eps * tnorm ? 1 : 0;
            x0 -= 10.0 * eps * tnorm * (double)rank_in_cluster;
            double *Ls = ws + lane, *Ds = Ls + (size_t)P * tdl;  // [i][lane]
            const double floor_ = eps * eps * tnorm;                    // breakdown guard for a vanishing pivot
            double dp = dd[0] - x0;
            for (int i = 0; i < P - 1; ++i) {
                if (fabs(dp) < floor_) dp = dp < 0.0 ? -floor_ : floor_;
                const double ei = ee[i], li = ei / dp;
                Ds[(size_t)i * tdl] = dp;
                Ls[(size_t)i * tdl] = li;
                dp = (dd[i + 1] - x0) - li * ei;
            }
            Ds[(size_t)(P - 1) * tdl] = dp;
            // from the bottom: D-_i, gamma_i, the twist; U_i overwrites D+_{i+1} (no longer needed once gamma_{i+1} is known)
            double dm = dd[P - 1] - x0;
            double gbest = fabs(dp + dm - (dd[P - 1] - x0));
            int r = P - 1;
            for (int i = P - 2; i >= 0; --i) {
                if (fabs(dm) < floor_) dm = dm < 0.0 ? -floor_ : floor_;
                const double ei = ee[i], ui = ei / dm;
                Ds[(size_t)(i + 1) * tdl] = ui;  // U_i lives at slot i + 1
                dm = (dd[i] - x0) - ui * ei;
                const double g = fabs(Ds[(size_t)i * tdl] + dm - (dd[i] - x0));
                if (g < gbest) {
                    gbest = g;
                    r = i;
                }
            }
            double nrm2 = 1.0, zi = 1.0;
            Z[(size_t)r * k + j] = 1.0;
            for (int i = r; i < P - 1; ++i) {
                zi = -Ds[(size_t)(i + 1) * tdl] * zi;
                Z[(size_t)(i + 1) * k + j] = zi;
                nrm2 = fma(zi, zi, nrm2);
            }
            zi = 1.0;
            for (int i = r; i > 0; --i) {
                zi = -Ls[(size_t)(i - 1) * tdl] * zi;
                Z[(size_t)(i - 1) * k + j] = zi;
                nrm2 = fma(zi, zi, nrm2);
            }
            zinv[j] = 1.0 / sqrt(nrm2);
        }
        __syncthreads();
        // normalise the batch's vectors (every thread; the vectors sit in global memory)
        const int nb = min(tdl, k - j0);
        for (int e = tid; e < P * nb; e += TD_NT) {
            const int i = e / nb, jj = j0 + (e - i * nb);
            Z[(size_t)i * k + jj] *= zinv[jj];
        }
    }
    __syncthreads();
    TD_CLK(3);
    // Modified Gram-Schmidt inside clusters of close eigenvalues, in eigenvalue order, and 4. the back-transformation
    // V = H_0 H_1 ... H_{P-3} Z — a wave per vector with the vector in REGISTERS (P <= 192: three entries per lane), no LDS
    // traffic but the reflector itself.
    // k <= 16 (round 6): wave j owns vector j through BOTH phases.  Round jj of the Gram-Schmidt: wave jj (by then orthogonal to
    // its earlier neighbours; normalised now if it changed) publishes its vector in LDS, the waves of the later vectors within
    // cl_tol of it subtract their component along it — the arithmetic of the sequential loop it replaces (vector j against
    // jj = 0 .. j - 1 in order), but the steep spectra of pixel blocks put 8-10 of the 16 wanted eigenvalues within 1e-3 |T|
    // of each other, and one wave walking ~30 (j, jj) pairs through L2 took 82-96 us of the ~1000 per matrix; rounds without a
    // later neighbour are skipped (the masks are wave-uniform: ballots over the eigenvalues held one per lane).
    if (k <= NW) {
        const int j = wave;
        const bool own = j < k;
        double z0 = own && lane < P ? Z[(size_t)lane * k + j] : 0.0;
        double z1 = own && lane + 64 < P ? Z[(size_t)(lane + 64) * k + j] : 0.0;
        double z2 = own && lane + 128 < P ? Z[(size_t)(lane + 128) * k + j] : 0.0;
        const double lam_l = lane < k ? lamv[lane] : 0.0;
        double *pub = ws;  // P doubles (the twisted factorisations are done with their scratch)
        bool dirty = false;
        auto normalise = [&]() {
            const double inv = 1.0 / sqrt(td_wave_sum(fma(z0, z0, fma(z1, z1, z2 * z2))));
            z0 *= inv;
            z1 *= inv;
            z2 *= inv;
            dirty = false;
        };
        for (int jj = 0; jj + 1 < k; ++jj) {
            const double lam_jj = td_readlane(lam_l, jj);
            const unsigned long long later = __ballot(lane > jj && lane < k && fabs(lam_l - lam_jj) < cl_tol);
            if (later == 0ull) continue;  // (workgroup-uniform)
            if (j == jj) {
                if (dirty) normalise();
                if (lane < P) pub[lane] = z0;
                if (lane + 64 < P) pub[lane + 64] = z1;
                if (lane + 128 < P) pub[lane + 128] = z2;
            }
            __syncthreads();
            if (own && ((later >> j) & 1ull)) {
                const double u0 = lane < P ? pub[lane] : 0.0, u1 = lane + 64 < P ? pub[lane + 64] : 0.0,
                             u2 = lane + 128 < P ? pub[lane + 128] : 0.0;
                const double dot = td_wave_sum(fma(z0, u0, fma(z1, u1, z2 * u2)));
                z0 = fma(-dot, u0, z0);
                z1 = fma(-dot, u1, z1);
                z2 = fma(-dot, u2, z2);
                dirty = true;
            }
            __syncthreads();
        }
        if (dirty) normalise();
        TD_CLK(4);
        if (own) {
            for (int kk = P - 3; kk >= 0; --kk) {
                const double tk = tau[kk];
                if (tk == 0.0) continue;
                const int r0 = kk + 1;
                const double sc = scl[kk];
                // v_i for row i: 0 above r0, 1 at r0, the stored column entry times the scale below
                auto vrow = [&](int i) { return i < r0 ? 0.0 : (i == r0 ? 1.0 : (i < P ? At[td_tri(i, kk)] * sc : 0.0)); };
                const double v0 = vrow(lane), v1 = vrow(lane + 64), v2 = vrow(lane + 128);
                const double f = tk * td_wave_sum(fma(v0, z0, fma(v1, z1, v2 * z2)));
                z0 = fma(-f, v0, z0);
                z1 = fma(-f, v1, z1);
                z2 = fma(-f, v2, z2);
            }
            if (lane < P) Z[(size_t)lane * k + j] = z0;
            if (lane + 64 < P) Z[(size_t)(lane + 64) * k + j] = z1;
            if (lane + 128 < P) Z[(size_t)(lane + 128) * k + j] = z2;
        }
    } else {
        // more vectors than waves: the Gram-Schmidt by one wave on the vectors in V (global), then a wave per vector in turn
        if (wave == 0) {
            for (int j = 1; j < k; ++j) {
                bool any = false;
                for (int jj = 0; jj < j; ++jj)
                    if (fabs(lamv[jj] - lamv[j]) < cl_tol) {
                        any = true;
                        double dot = 0.0;
                        for (int i = lane; i < P; i += 64) dot = fma(Z[(size_t)i * k + j], Z[(size_t)i * k + jj], dot);
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
                        for (int i = lane; i < P; i += 64) Z[(size_t)i * k + j] -= dot * Z[(size_t)i * k + jj];
                    }
                if (any) {
                    double n2 = 0.0;
                    for (int i = lane; i < P; i += 64) n2 = fma(Z[(size_t)i * k + j], Z[(size_t)i * k + j], n2);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o);
                    const double inv = 1.0 / sqrt(n2);
                    for (int i = lane; i < P; i += 64) Z[(size_t)i * k + j] *= inv;
                }
            }
        }
        __syncthreads();
        TD_CLK(4);
        for (int j = wave; j < k; j += NW) {
            double z0 = lane < P ? Z[(size_t)lane * k + j] : 0.0;
            double z1 = lane + 64 < P ? Z[(size_t)(lane + 64) * k + j] : 0.0;
            double z2 = lane + 128 < P ? Z[(size_t)(lane + 128) * k + j] : 0.0;
            for (int kk = P - 3; kk >= 0; --kk) {
                const double tk = tau[kk];
                if (tk == 0.0) continue;
                const int r0 = kk + 1;
                const double sc = scl[kk];
                auto vrow = [&](int i) { return i < r0 ? 0.0 : (i == r0 ? 1.0 : (i < P ? At[td_tri(i, kk)] * sc : 0.0)); };
                const double v0 = vrow(lane), v1 = vrow(lane + 64), v2 = vrow(lane + 128);
                const double f = tk * td_wave_sum(fma(v0, z0, fma(v1, z1, v2 * z2)));
                z0 = fma(-f, v0, z0);
                z1 = fma(-f, v1, z1);
                z2 = fma(-f, v2, z2);
            }
            if (lane < P) Z[(size_t)lane * k + j] = z0;
            if (lane + 64 < P) Z[(size_t)(lane + 64) * k + j] = z1;
            if (lane + 128 < P) Z[(size_t)(lane + 128) * k + j] = z2;
        }
    }
    __syncthreads();
    TD_CLK(5);
#ifdef LK_PLD_DEBUG
    if (clk && tid == 0) clk[8 + 4 * (size_t)b + 1] = wall_clock64();
#endif
    if (tid < k) lam[(size_t)b * k + tid] = lamv[tid];
#undef TD_CLK
}

// ------------------------------------------------------------------------------------------------ launcher
// top-k eigenpairs of the B Gram matrices G (P x P, leading dimension ldg) -> V (B x P x k), lam (B x k), both allocated
// from ws.  mirror: G holds the upper 64 x 64 blocks only (gram_plain_launch) and is completed in place first.
static int eig_topk(lk_handle *h, double *G, int ldg, int B, int P, int k, bool products, bool mirror, double **V_out,
                    double **lam_out, hipStream_t stream, Arena &ws, const float *G32 = nullptr, double tol = -1.0) {
    if (!(tol > 0.0)) tol = h->pld_eig_tol > 0.0 ? h->pld_eig_tol : PLD_EIG_TOL;   // (lk_pld_set_eig_tolerance)
    constexpr int direct_max = PLD_DIRECT_MAX;
    // Convergence: || C r - theta r || <= eig_tol * theta_max * sqrt(k) over the k wanted pairs (PLD_EIG_TOL / PCA_EIG_TOL above).
#ifdef LK_PLD_DEBUG   // development builds: LK_PLD_TOL sweeps the stop (tools/pld_tol_sweep.py -> profiles/r05_pld_tol_sweep.txt)
    const double eig_tol = getenv("LK_PLD_TOL") ? atof(getenv("LK_PLD_TOL")) : tol;
#else
    const double eig_tol = tol;
#endif
    constexpr int npow_std = 3;  // C^3 (or the degree-3 Chebyshev filter) between two Rayleigh-Ritz steps
    // mid-size product blocks: a product with the 136 x 136 C is cheap next to the l x l Jacobi and the Cholesky-QR of a
    // Rayleigh-Ritz step, and their flat spectrum keeps C^8 R well conditioned — 8 products per step need 5 steps where 3
    // need 12.  (Pixel blocks must NOT do this: their steep spectrum makes C^5 R numerically rank-deficient, the Cholesky
    // breaks down and the iteration stalls.)
    constexpr int npow_prod = 8;
    // Chebyshev-filtered steps: bit 0 = for flat spectra only (round 2's first version), bit 1 = for every spectrum.
    // Default 3: the degree-3 filter on [0, theta_cut] needs 5.0 Rayleigh-Ritz steps where C^3 needs 7.2 on the 816-column
    // blocks (4 instead of 5 on the pixel blocks); the feared Cholesky breakdowns on steep spectra do not occur — the
    // columns are scaled to unit length first and SVQB stands behind — PLD step 83.7 -> 77.7 ms.
    constexpr int cheb_on = 3;
#ifdef LK_PLD_DEBUG   // `make DEBUG=1`: LK_PLD_ITERS=1 prints Rayleigh-Ritz step counts and per-phase clocks
    static const bool dbg_iters = getenv("LK_PLD_ITERS") && atoi(getenv("LK_PLD_ITERS")) != 0;
#else
    constexpr bool dbg_iters = false;
#endif
    {
        int rc_ = want_lds(h, reinterpret_cast<const void *>(pld_topk_eig_kernel<2>), 160 * 1024);
        if (!rc_) rc_ = want_lds(h, reinterpret_cast<const void *>(pld_topk_eig_kernel<4>), 160 * 1024);
        if (rc_) return rc_;
    }
    double *V = (double *)ws.alloc((size_t)B * P * k * 8), *lam = (double *)ws.alloc((size_t)B * k * 8);
    if (!V || !lam) {
        set_error("PLD workspace exhausted (V)");
        return LK_ENOMEM;
    }
    // Gram matrices that fit LDS (pixel / background blocks of <= 138 pixels, the 136-column 2nd-order product block): the
    // direct tridiagonal solver — ~0.5 ms per launch and matrix instead of the 2-3 ms of a subspace iteration's
    // Rayleigh-Ritz steps at this size
    // (P = 134 .. 138 with k in the 40s needs more than 160 KB: those shapes fall through to the subspace iteration)
    // LDS of the direct solver: packed triangle + six P-vectors + eigenvalues and vector norms + the twisted factorisations' scratch
    // (the eigenvectors themselves live in V).  8 factorisations at a time instead of 16 where that brings a workgroup under half
    // a CU's LDS (P <= ~122 at k = 16: the 121-column pixel / background blocks).
    auto td_lds_bytes = [&](int tdl_) {
        return ((size_t)((((P * (P + 1)) / 2) + 1) & ~1) + 6 * (size_t)P + 2 * (size_t)((k + 1) & ~1) + (size_t)tdl_ * 2 * P) * 8 + 16;
    };
    const int tdl = (td_lds_bytes(TD_LANES) > 80 * 1024 && td_lds_bytes(TD_LANES / 2) <= 80 * 1024) ? TD_LANES / 2 : TD_LANES;
    const size_t td_lds = td_lds_bytes(tdl);
    if (P >= 3 && P <= PLD_DIRECT_MAX && td_lds <= 160 * 1024) {
        const size_t lds = td_lds;
        int rc_ = want_lds(h, reinterpret_cast<const void *>(pld_tridiag_eig_kernel), 160 * 1024);
        if (rc_) return rc_;
        unsigned long long *d_clk = nullptr;
#ifdef LK_PLD_DEBUG
        if (dbg_iters) d_clk = (unsigned long long *)ws.alloc(64 + 96 * (size_t)B + 64);
#endif
        hipLaunchKernelGGL(pld_tridiag_eig_kernel, dim3(B), dim3(TD_NT), lds, stream, G, ldg, P, k, V, lam, d_clk, tdl);
#ifdef LK_PLD_DEBUG
        if (d_clk) {
            unsigned long long hc[8];
            LK_HIP_CHECK(hipMemcpyAsync(hc, d_clk, 48, hipMemcpyDeviceToHost, stream));
            LK_HIP_CHECK(hipStreamSynchronize(stream));
            fprintf(stderr, "[pld tridiag] P=%d k=%d per matrix us: tridiagonalise %.0f | eigenvalues %.0f | twisted factorisation %.0f | "
                            "Gram-Schmidt %.0f | back-transform %.0f\n", P, k, (hc[1] - hc[0]) * 0.01, (hc[2] - hc[1]) * 0.01,
                    (hc[3] - hc[2]) * 0.01, (hc[4] - hc[3]) * 0.01, (hc[5] - hc[4]) * 0.01);
            // the launch as a whole: when each workgroup started and ended, and where it ran
            std::vector<unsigned long long> tl(4 * (size_t)B);
            LK_HIP_CHECK(hipMemcpyAsync(tl.data(), d_clk + 8, 32 * (size_t)B, hipMemcpyDeviceToHost, stream));
            LK_HIP_CHECK(hipStreamSynchronize(stream));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int i = 0; i < B; ++i) {
                t0 = std::min(t0, tl[4 * i]);
                t1 = std::max(t1, tl[4 * i + 1]);
            }
            std::vector<double> st(B), du(B);
            std::map<unsigned, int> per_cu;
            for (int i = 0; i < B; ++i) {
                st[i] = (tl[4 * i] - t0) * 0.01;
                du[i] = (tl[4 * i + 1] - tl[4 * i]) * 0.01;
                const unsigned hw = (unsigned)tl[4 * i + 2], xcc = (unsigned)(tl[4 * i + 2] >> 32) & 0xf;
                per_cu[(xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)]++;
            }
            std::sort(st.begin(), st.end());
            std::sort(du.begin(), du.end());
            {   // phases of the fast and of the slow workgroups
                std::vector<unsigned long long> ph(8 * (size_t)B);
                LK_HIP_CHECK(hipMemcpyAsync(ph.data(), d_clk + 8 + 4 * (size_t)B, 64 * (size_t)B, hipMemcpyDeviceToHost, stream));
                LK_HIP_CHECK(hipStreamSynchronize(stream));
                const double cut = 1.25 * du[B / 2];
                double acc[2][6] = {{0}};
                int cnt[2] = {0, 0};
                for (int i = 0; i < B; ++i) {
                    const int gsl = (tl[4 * i + 1] - tl[4 * i]) * 0.01 > cut ? 1 : 0;
                    cnt[gsl]++;
                    acc[gsl][0] += (ph[8 * i] - tl[4 * i]) * 0.01;
                    for (int q = 1; q < 6; ++q) acc[gsl][q] += (ph[8 * i + q] - ph[8 * i + q - 1]) * 0.01;
                }
                for (int gsl = 0; gsl < 2; ++gsl)
                    if (cnt[gsl])
                        fprintf(stderr, "[pld tridiag timeline] %s workgroups (%d): load %.0f | tridiagonalise %.0f | eigenvalues %.0f | twisted %.0f | "
                                        "Gram-Schmidt %.0f | back-transform %.0f\n", gsl ? "slow" : "fast", cnt[gsl], acc[gsl][0] / cnt[gsl],
                                acc[gsl][1] / cnt[gsl], acc[gsl][2] / cnt[gsl], acc[gsl][3] / cnt[gsl], acc[gsl][4] / cnt[gsl], acc[gsl][5] / cnt[gsl]);
                std::string slow_ids;
                for (int i = 0; i < B && slow_ids.size() < 200; ++i)
                    if ((tl[4 * i + 1] - tl[4 * i]) * 0.01 > cut) slow_ids += std::to_string(i) + " ";
                fprintf(stderr, "[pld tridiag timeline] slow ids: %s\n", slow_ids.c_str());
            }
            int mx = 0;
            for (auto &kv : per_cu) mx = std::max(mx, kv.second);
            int late = 0;
            for (int i = 0; i < B; ++i) late += st[i] > 50.0 ? 1 : 0;
            fprintf(stderr, "[pld tridiag timeline] span %.0f us | workgroup duration min %.0f median %.0f p90 %.0f max %.0f | starts: median %.0f "
                            "p90 %.0f max %.0f, %d of %d later than 50 us | %zu CUs used, at most %d workgroups on one\n",
                    (t1 - t0) * 0.01, du[0], du[B / 2], du[(B * 9) / 10], du[B - 1], st[B / 2], st[(B * 9) / 10], st[B - 1], late, B,
                    per_cu.size(), mx);
        }
#endif
        *V_out = V;
        *lam_out = lam;
        return LK_OK;
    }
    // Mid-size blocks (PLD_LMAX < P <= PLD_DIRECT_MAX) get two passes: a SHORT subspace iteration (a pixel block with a
    // few dominant stars converges in ~5 Rayleigh-Ritz steps), then the direct Jacobi on C itself for the matrices that
    // did not converge (2nd-order product blocks decay slowly: ~45 steps of the iteration vs one ~15 ms Jacobi).
    // mid-size product blocks (2nd-order: 136 columns): subspace iteration with 8 products per Rayleigh-Ritz step (5 steps,
    // 4.4 ms per matrix) instead of the direct Jacobi on C (135 rotation rounds x ~10 sweeps, 16 ms per matrix).
    constexpr bool prod_direct = false;
    constexpr int prod_l = 0;  // basis width of the product blocks: the default k + 16
    const bool wide_sub = products && !prod_direct && P > PLD_LMAX && P <= PLD_DIRECT_MAX;
    const bool two_pass = !wide_sub && P > PLD_LMAX && P <= std::min(direct_max, PLD_DIRECT_MAX);
    const int npow = wide_sub ? npow_prod : npow_std;
    int *status = nullptr;
    if (two_pass) {
        status = (int *)ws.alloc((size_t)B * 4);
        if (!status) {
            set_error("PLD workspace exhausted (status)");
            return LK_ENOMEM;
        }
    }
    // product blocks are known to need the Jacobi: skip their short subspace pass (status stays all zero)
    const bool direct_only = two_pass && products;
    if (direct_only) LK_HIP_CHECK(hipMemsetAsync(status, 0, (size_t)B * 4, stream));
    if (P > PLD_LMAX && !direct_only) {  // subspace iteration
        const int l = (wide_sub && prod_l > 0) ? std::min(PLD_LMAX, prod_l & ~1) : std::min(PLD_LMAX, (k + 16 + 1) & ~1), ld = l + 1;
        double *scr = (double *)ws.alloc((size_t)B * 4 * P * l * 8);
        if (!scr) {
            set_error("PLD workspace exhausted (subspace)");
            return LK_ENOMEM;
        }
        // ---- phase-split form (see pld_eigs_*): the blocks beyond the direct solver's reach with the standard three products
        // per step.  A fixed number of steps is queued (converged matrices drop out of each launch by their flag); the
        // one-kernel iteration below then runs for whatever is left, skipping the converged ones.
        EigsState *d_state = nullptr;
#ifdef LK_PLD_DEBUG
        if (getenv("LK_PLD_SPLIT")) h->pld_eig_split = atoi(getenv("LK_PLD_SPLIT"));  // 0: one-kernel form, 1: phase-split
        const bool split_on = h->pld_eig_split != 0;
#else
        const bool split_on = h->pld_eig_split != 0;
#endif
        const bool split = !two_pass && !wide_sub && npow == 3 && split_on;
        if (split) {
            d_state = (EigsState *)ws.alloc((size_t)B * sizeof(EigsState));
            if (!d_state) {
                set_error("PLD workspace exhausted (state)");
                return LK_ENOMEM;
            }
            const int have32 = G32 != nullptr ? 1 : 0;
            const int kc_s = l <= 32 ? 32 : 64;                       // eig_xty's partial tiles: (512 / 64) or 16 x 256 doubles
            const int kc_p = 128;                                     // rows per float64 stage of a product: 66 KB
            const int kc32_p = (int)(((size_t)kc_p * PLD_QS * 8) / ((size_t)pld_qs32(l <= 32 ? 2 : 4) * 4)) / 24 * 24;  // (a multiple of 8 x eig_cq32_pass's PF)
            const size_t lds_s = eigs_small_lds_bytes(l, 512, kc_s), lds_p = (size_t)kc_p * PLD_QS * 8 + 64;
            constexpr int nsteps_split = 6;
            int rc_ = 0;
#define LK_EIGS_WANT(NA_)                                                                                              \
    do {                                                                                                               \
        if (!rc_) rc_ = want_lds(h, reinterpret_cast<const void *>(pld_eigs_init_kernel<NA_>), 160 * 1024);             \
        if (!rc_) rc_ = want_lds(h, reinterpret_cast<const void *>(pld_eigs_prod_kernel<NA_>), 160 * 1024);             \
        if (!rc_) rc_ = want_lds(h, reinterpret_cast<const void *>(pld_eigs_rr_kernel<NA_>), 160 * 1024);               \
        if (!rc_) rc_ = want_lds(h, reinterpret_cast<const void *>(pld_eigs_orth_kernel<NA_>), 160 * 1024);             \
    } while (0)
            // matrices [b0_, b0_ + nb_) on stream st_: every array is indexed by matrix, so a part of the batch is an offset
#define LK_EIGS_RUN(NA_, b0_, nb_, st_)                                                                                   \
    do {                                                                                                               \
        double *G_ = G + (size_t)(b0_) * ldg * ldg, *scr_ = scr + (size_t)(b0_) * 4 * P * l;                            \
        const float *G32_ = G32 ? G32 + (size_t)(b0_) * ldg * ldg : nullptr;                                            \
        EigsState *state_ = d_state + (b0_);                                                                           \
        double *V_ = V + (size_t)(b0_) * P * k, *lam_ = lam + (size_t)(b0_) * k;                                        \
        hipLaunchKernelGGL(pld_eigs_init_kernel<NA_>, dim3(nb_), dim3(512), lds_s, st_, G_, ldg, P, k, l, scr_, state_, \
                           kc_s, mirror ? 1 : 0, have32);                                                              \
        for (int it_ = 0; it_ < nsteps_split; ++it_) {                                                                 \
            const int last_ = it_ + 1 == nsteps_split ? 1 : 0;                                                         \
            hipLaunchKernelGGL(pld_eigs_prod_kernel<NA_>, dim3(nb_), dim3(512), lds_p, st_, G_, G32_, ldg, P, l, scr_,  \
                               state_, 0, kc_p, kc32_p);                                                               \
            hipLaunchKernelGGL(pld_eigs_rr_kernel<NA_>, dim3(nb_), dim3(512), lds_s, st_, P, k, l, scr_, state_, V_,    \
                               lam_, kc_s, eig_tol, cheb_on, last_, have32);                                           \
            if (last_) break;                                                                                          \
            hipLaunchKernelGGL(pld_eigs_prod_kernel<NA_>, dim3(nb_), dim3(512), lds_p, st_, G_, G32_, ldg, P, l, scr_,  \
                               state_, 1, kc_p, kc32_p);                                                               \
            hipLaunchKernelGGL(pld_eigs_prod_kernel<NA_>, dim3(nb_), dim3(512), lds_p, st_, G_, G32_, ldg, P, l, scr_,  \
                               state_, 2, kc_p, kc32_p);                                                               \
            hipLaunchKernelGGL(pld_eigs_orth_kernel<NA_>, dim3(nb_), dim3(512), lds_s, st_, P, k, l, scr_, state_,      \
                               kc_s);                                                                                  \
        }                                                                                                              \
    } while (0)
            // (Measured and dropped: the batch in two halves on two streams, the second started one product late so that one half's
            // serial phases meet the other half's products — profiles/r06_pld_eig_modes_ab.txt: +1 % on the PLD step over one
            // stream; a half-batch product takes as long as a full one while a small-phase launch runs beside it.)
            if (l <= 32)
                LK_EIGS_WANT(2);
            else
                LK_EIGS_WANT(4);
            if (rc_) return rc_;
            if (l <= 32)
                LK_EIGS_RUN(2, 0, B, stream);
            else
                LK_EIGS_RUN(4, 0, B, stream);
#undef LK_EIGS_WANT
#undef LK_EIGS_RUN
#ifdef LK_PLD_DEBUG
            if (dbg_iters) {
                std::vector<EigsState> hs((size_t)B);
                LK_HIP_CHECK(hipMemcpyAsync(hs.data(), d_state, (size_t)B * sizeof(EigsState), hipMemcpyDeviceToHost, stream));
                LK_HIP_CHECK(hipStreamSynchronize(stream));
                long long sum = 0, mx = 0, nconv = 0;
                double rmax = 0.0;
                for (int b2 = 0; b2 < B; ++b2) {
                    sum += hs[b2].it;
                    mx = std::max<long long>(mx, hs[b2].it);
                    nconv += hs[b2].converged;
                    rmax = std::max(rmax, hs[b2].res / std::max(hs[b2].th0, 1e-300));
                }
                fprintf(stderr, "[pld eig split] P=%d k=%d l=%d: Rayleigh-Ritz steps mean %.1f max %lld, %lld of %d converged in the queued "
                                "steps, largest final relative residual %.2e\n", P, k, l, (double)sum / B, mx, nconv, B, rmax);
            }
#endif
        }
        // 512-thread workgroups, two per CU (LDS 57 KB each): one matrix's serial stretches (l x l Jacobi, Cholesky on one
        // wave, the random start) overlap the other's stream through C — 97.2 -> 86.6 ms per PLD step against one
        // 1024-thread workgroup per CU
        constexpr int nt_env = 512;
        int nt_sub = (nt_env == 256 || nt_env == 512 || nt_env == 1024) ? nt_env : 512;
        if (nt_sub == 256 && l > 32) nt_sub = 512;  // 16 partial tiles of eig_xty need the 64-row stage
        const int kc = nt_sub == 256 ? 32 : PLD_KC;  // LDS stage rows: the stage also holds eig_xty's (nt / 64) x 2 KB of partial tiles
        const size_t lds = ((size_t)2 * l * ld + 2 * l + nt_sub + 2 * l + (l + 1) / 2 + 1 + kc * PLD_QS) * 8 + 64;
        long long *d_it = dbg_iters ? (long long *)ws.alloc((size_t)B * 64) : nullptr;
        if (l <= 32)
            hipLaunchKernelGGL(pld_topk_eig_kernel<2>, dim3(B), dim3(nt_sub), lds, stream, G, ldg, P, k, l, npow, scr, V, lam,
                               d_it, two_pass ? 8 : 400, status, cheb_on, kc, (mirror && !split) ? 1 : 0, eig_tol, G32, d_state);
        else
            hipLaunchKernelGGL(pld_topk_eig_kernel<4>, dim3(B), dim3(nt_sub), lds, stream, G, ldg, P, k, l, npow, scr, V, lam,
                               d_it, two_pass ? 8 : 400, status, cheb_on, kc, (mirror && !split) ? 1 : 0, eig_tol, G32, d_state);
        if (d_it) {
            std::vector<long long> hit((size_t)B * 8);
            LK_HIP_CHECK(hipMemcpyAsync(hit.data(), d_it, (size_t)B * 64, hipMemcpyDeviceToHost, stream));
            LK_HIP_CHECK(hipStreamSynchronize(stream));
            long long sum = 0, mx = 0;
            double ph[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int b2 = 0; b2 < B; ++b2) {
                sum += hit[(size_t)b2 * 8];
                mx = std::max(mx, hit[(size_t)b2 * 8]);
                for (int s2 = 0; s2 < 7; ++s2) ph[s2] += (double)hit[(size_t)b2 * 8 + 1 + s2] / B * 0.01;  // 100 MHz ticks -> us
            }
            fprintf(stderr,
                    "[pld eig] P=%d k=%d l=%d opt=%d: Rayleigh-Ritz steps mean %.1f max %lld over %d matrices; per matrix us: "
                    "init %.0f | C*Q %.0f | Q^T Z %.0f | Jacobi %.0f | Ritz+resid %.0f | power products %.0f | CholQR %.0f\n",
                    P, k, l, cheb_on, (double)sum / B, mx, B, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6]);
            if (B > 256) {  // the first workgroup of every CU against the one that came second onto it
                for (int grp = 0; grp < 2; ++grp) {
                    const int b_lo = grp ? 256 : 0, b_hi = grp ? std::min(B, 512) : 256;
                    double pg[7] = {0, 0, 0, 0, 0, 0, 0};
                    for (int b2 = b_lo; b2 < b_hi; ++b2)
                        for (int s2 = 0; s2 < 7; ++s2) pg[s2] += (double)hit[(size_t)b2 * 8 + 1 + s2] / (b_hi - b_lo) * 0.01;
                    fprintf(stderr, "[pld eig] matrices %d-%d: init %.0f | C*Q %.0f | Q^T Z %.0f | Jacobi %.0f | Ritz+resid %.0f | power products %.0f | "
                                    "CholQR %.0f | sum %.0f\n", b_lo, b_hi - 1, pg[0], pg[1], pg[2], pg[3], pg[4], pg[5], pg[6],
                            pg[0] + pg[1] + pg[2] + pg[3] + pg[4] + pg[5] + pg[6]);
                }
            }
        }
    }
    if (P <= PLD_LMAX || two_pass) {  // direct Jacobi on C
        const int l = (P + 1) & ~1, ld = l + 1;
        double *scr = nullptr;
        if (two_pass) {
            scr = (double *)ws.alloc((size_t)B * l * l * 8);
            if (!scr) {
                set_error("PLD workspace exhausted (eigenvectors)");
                return LK_ENOMEM;
            }
        }
        const int nt_eig = two_pass ? 512 : 1024;
        const size_t lds = two_pass ? ((size_t)l * ld + 2 * l + nt_eig + 2 * l + (l + 1) / 2 + 1) * 8 + 64
                                    : ((size_t)2 * l * ld + 2 * l + 1024 + 2 * l + (l + 1) / 2 + 1) * 8 + 64;
        hipLaunchKernelGGL(pld_topk_eig_kernel<2>, dim3(B), dim3(nt_eig), lds, stream, G, ldg, P, k, l, npow, scr, V, lam,
                           (long long *)nullptr, 400, status, cheb_on, PLD_KC, mirror ? 1 : 0, eig_tol);
    }
    *V_out = V;
    *lam_out = lam;
    return LK_OK;
}

// PCA of B centred matrices A (N x P) -> top-k left singular vectors into X[:, col0:col0+k]
struct PcaF32Source {  // the block as float32 pixels: A = (double)(mode == 0 ? pix : pix / div[row]) - mean[col], never written out
    const float *pix, *div;
    const double *mean;
    int mode;
};
#ifndef PLD_F32_SOURCE
#define PLD_F32_SOURCE 1
#endif
// (can the narrow Gram kernel and the float32 projection take this block?  P >= 4 columns, at most 9 column tiles, k <= 48)
static bool pca_f32_ok(int P, int k) { return PLD_F32_SOURCE && P >= 4 && (P + 15) / 16 <= 9 && k <= 48; }
static int pca_block(lk_handle *h, double *A, int B, int N, int P, int k, const int64_t *d_off, double *X, int ldx,
                     int col0, hipStream_t stream, Arena &ws, bool centred = false, bool products = false,
                     double tol = -1.0, PcaF32Source fs = PcaF32Source{nullptr, nullptr, nullptr, 0}) {
    const bool f32src = fs.pix != nullptr;
    if (!centred && !f32src) hipLaunchKernelGGL(pld_center_kernel, dim3((P + 31) / 32, B), dim3(256), 0, stream, A, N, P);
    const int KB = (P + 63) / 64, ldg = KB * 64;
    double *G = (double *)ws.alloc((size_t)B * ldg * ldg * 8);
    if (!G) {
        set_error("PLD workspace exhausted (Gram)");
        return LK_ENOMEM;
    }
    if (f32src) {
        if (gram_plain_f32_launch(fs.pix, fs.div, fs.mean, fs.mode, d_off, B, P, G, stream, h) == 0) {
            set_error("PLD: the float32-source Gram kernel refused a %d-column block", P);
            return LK_EHIP;
        }
    } else {
        gram_plain_launch(A, d_off, B, P, G, stream, h);
    }
    double *V = nullptr, *lam = nullptr;
    const int rc = eig_topk(h, G, ldg, B, P, k, products, true, &V, &lam, stream, ws, nullptr, tol);
    if (rc) return rc;
    {
        const dim3 grid((N + 63) / 64, B), blk(256);
        const int kt = (k + 15) / 16;
        const bool v4 = (P & 3) == 0 && P >= 4;
        if (f32src) {
#define LK_PROJF(KT) hipLaunchKernelGGL((pld_project_f32_kernel<KT>), grid, blk, 0, stream, fs.pix, fs.div, fs.mean, fs.mode, V, lam, N, P, k, ldx, col0, X)
            if (kt <= 1) LK_PROJF(1); else if (kt == 2) LK_PROJF(2); else LK_PROJF(3);
#undef LK_PROJF
            return LK_OK;
        }
#define LK_PROJ(KT, V4) hipLaunchKernelGGL((pld_project_kernel<KT, V4>), grid, blk, 0, stream, A, V, lam, N, P, k, ldx, col0, X)
        if (kt <= 1) {
            if (v4) LK_PROJ(1, true); else LK_PROJ(1, false);
        } else if (kt == 2) {
            if (v4) LK_PROJ(2, true); else LK_PROJ(2, false);
        } else if (kt == 3) {
            if (v4) LK_PROJ(3, true); else LK_PROJ(3, false);
        } else {
            if (v4) LK_PROJ(4, true); else LK_PROJ(4, false);
        }
#undef LK_PROJ
    }
    return LK_OK;
}

// Host-built tables of the moment-form Gram (see pld_moment_gram_kernel), cached per (device, k, order).
struct MomentPlan {
    int device, k, order, Pc, ldm, nwt, nfull;  // nfull: leading wave tiles whose 16 tiles are all needed
    uint8_t *d_comb, *d_rcomb;  // factor tuples in natural order / in row order (sorted by largest factor)
    int4 *d_wt;                 // wave tiles: first row (row order), first column, mask of the 16 x 16 tiles to compute
    uint32_t *d_src;            // [Pc][Pc]: where the moment of columns (i, j) sits in the canonical array
    uint32_t *d_packed;         // natural-order tuples, one dword each (projection kernel)
    int *d_rperm;               // row order -> natural index
};

static const MomentPlan *moment_plan(lk_handle *h, int k, int o, const std::vector<uint8_t> &comb, int Pc) {
    // one plan per (device, k, order) for the life of the process (a few MB of index tables each); the cache is shared by
    // the handles of a process, so two handles driven from two threads must not build / look up concurrently
    static std::vector<MomentPlan *> cache;
    static std::mutex cache_mutex;
    std::lock_guard<std::mutex> guard(cache_mutex);
    for (const MomentPlan *pl : cache)
        if (pl->device == h->device && pl->k == k && pl->order == o) return pl;
    auto tup = [&](int idx, int pos) { return (int)comb[(size_t)idx * o + pos]; };
    // row order: stable sort by the largest (= last) factor
    std::vector<int> rperm(Pc), rowpos(Pc);
    for (int i = 0; i < Pc; ++i) rperm[i] = i;
    std::stable_sort(rperm.begin(), rperm.end(), [&](int a, int b2) { return tup(a, o - 1) < tup(b2, o - 1); });
    for (int r = 0; r < Pc; ++r) rowpos[rperm[r]] = r;
    std::vector<int> colstart(k + 2, 0);  // number of tuples whose smallest factor is < m
    for (int m = 0; m <= k + 1; ++m) {
        int c = 0;
        while (c < Pc && tup(c, 0) < m) ++c;
        colstart[m] = c;
    }
    // rank of a sorted tuple in natural order
    size_t kp = 1;
    for (int pos = 0; pos < o; ++pos) kp *= (size_t)k;
    std::vector<int> rank(kp, -1);
    auto key = [&](const int *a) {
        size_t q = 0;
        for (int pos = 0; pos < o; ++pos) q = q * k + a[pos];
        return q;
    };
    for (int i = 0; i < Pc; ++i) {
        int a[4];
        for (int pos = 0; pos < o; ++pos) a[pos] = tup(i, pos);
        rank[key(a)] = i;
    }
    const int ldm = ((Pc + 63) / 64) * 64;
    std::vector<uint32_t> src((size_t)Pc * Pc), packed(Pc);
    for (int i = 0; i < Pc; ++i) {
        uint32_t pk = 0;
        for (int pos = 0; pos < o; ++pos) pk |= (uint32_t)tup(i, pos) << (8 * pos);
        packed[i] = pk;
        for (int j = 0; j < Pc; ++j) {
            int z[8], a = 0, b2 = 0;
            for (int q = 0; q < 2 * o; ++q)  // merge of two sorted tuples
                z[q] = (b2 >= o || (a < o && tup(i, a) <= tup(j, b2))) ? tup(i, a++) : tup(j, b2++);
            src[(size_t)i * Pc + j] = (uint32_t)rowpos[rank[key(z)]] * (uint32_t)ldm + (uint32_t)rank[key(z + o)];
        }
    }
    std::vector<int4> wt;
    for (int r0 = 0; r0 < Pc; r0 += 64) {
        int rt[4], mn = k;
        for (int i = 0; i < 4; ++i) {
            rt[i] = -1;  // smallest "largest factor" among the rows of tile i
            for (int r = r0 + 16 * i; r < std::min(Pc, r0 + 16 * i + 16); ++r)
                rt[i] = rt[i] < 0 ? tup(rperm[r], o - 1) : std::min(rt[i], tup(rperm[r], o - 1));
            if (rt[i] >= 0) mn = std::min(mn, rt[i]);
        }
        for (int cg = (colstart[mn] / 16) * 16; cg < Pc; cg += 64) {
            int mask = 0;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    if (rt[i] >= 0 && cg + 16 * j < Pc && cg + 16 * j + 16 > colstart[rt[i]]) mask |= 1 << (4 * i + j);
            if (mask) wt.push_back(int4{r0, cg, mask, 0});
        }
    }
    // waves of a workgroup wait for each other at every stage: put tiles of similar size together
    std::stable_sort(wt.begin(), wt.end(), [](const int4 &a, const int4 &b2) { return __builtin_popcount(a.z) > __builtin_popcount(b2.z); });
    {   // column means: one wave tile that generates row tile (r0, i) also sums it — a masked one where there is a choice
        // (the FULL body has no registers to spare), and a full one that has to do it moves to the masked launch
        std::vector<char> done((size_t)(Pc + 15) / 16, 0);
        for (int pass = 0; pass < 2; ++pass)
            for (int4 &w4 : wt)
                for (int i = 0; i < 4; ++i) {
                    const int rt_ = w4.x / 16 + i;
                    if ((pass == 1 || w4.z != 0xffff) && (w4.z & (0xf << (4 * i))) && rt_ < (int)done.size() && !done[rt_]) {
                        done[rt_] = 1;
                        w4.w |= 1 << i;
                    }
                }
        std::stable_partition(wt.begin(), wt.end(), [](const int4 &a) { return a.z == 0xffff && a.w == 0; });
    }
    std::vector<uint8_t> rcomb((size_t)Pc * o);
    for (int r = 0; r < Pc; ++r)
        for (int pos = 0; pos < o; ++pos) rcomb[(size_t)r * o + pos] = (uint8_t)tup(rperm[r], pos);
    int nfull = 0;
    while (nfull < (int)wt.size() && wt[nfull].z == 0xffff && wt[nfull].w == 0) ++nfull;
    MomentPlan *pl = new MomentPlan{h->device, k, o, Pc, ldm, (int)wt.size(), nfull, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    auto up = [&](void **d, const void *src_, size_t bytes) {
        if (hipMalloc(d, bytes) != hipSuccess) return false;
        return hipMemcpy(*d, src_, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    const bool ok = up((void **)&pl->d_comb, comb.data(), comb.size()) && up((void **)&pl->d_rcomb, rcomb.data(), rcomb.size()) &&
                    up((void **)&pl->d_wt, wt.data(), wt.size() * sizeof(int4)) && up((void **)&pl->d_src, src.data(), src.size() * 4) &&
                    up((void **)&pl->d_packed, packed.data(), packed.size() * 4) && up((void **)&pl->d_rperm, rperm.data(), rperm.size() * 4);
    if (!ok) {
        set_error("PLD: could not allocate the moment-Gram tables");
        delete pl;
        return nullptr;
    }
    cache.push_back(pl);
    return pl;
}

// PCA of an order-o product block without materialising it: moment-form Gram -> eigenpairs -> projection generated on
// the fly.  Mcan: scratch of B * ldm * ldm doubles.
static int pca_products_moment(lk_handle *h, const MomentPlan &pl, int B, int N, int k1, int ko, double *X, int K, int col1,
                               int col0, double *d_mean, double *Mcan, hipStream_t stream, Arena &ws) {
    const int Pc = pl.Pc, o = pl.order, ldg = ((Pc + 63) / 64) * 64;
    double *G = (double *)ws.alloc((size_t)B * ldg * ldg * 8);
    if (!G) {
        set_error("PLD workspace exhausted (Gram)");
        return LK_ENOMEM;
    }
    {
        const size_t lds = (size_t)2 * MG_CH * (k1 | 1) * 8;
#define LK_MG(O, PRE, KS)                                                                                                     \
    do {                                                                                                                \
        if (pl.nfull > 0)                                                                                               \
            hipLaunchKernelGGL((pld_moment_gram_kernel<O, true, PRE, KS>), dim3((pl.nfull + 3) / 4, B), dim3(256), lds, stream, X, K, \
                               col1, k1, N, Pc, pl.ldm, pl.d_rcomb, pl.d_comb, pl.d_wt, pl.nfull, Mcan, pl.d_rperm, d_mean); \
        if (pl.nwt > pl.nfull)                                                                                          \
            hipLaunchKernelGGL((pld_moment_gram_kernel<O, false, PRE, KS>), dim3((pl.nwt - pl.nfull + 3) / 4, B), dim3(256), lds,    \
                               stream, X, K, col1, k1, N, Pc, pl.ldm, pl.d_rcomb, pl.d_comb, pl.d_wt + pl.nfull,        \
                               pl.nwt - pl.nfull, Mcan, pl.d_rperm, d_mean);                                           \
    } while (0)
        if (k1 <= 16) {
            if (o == 2) LK_MG(2, 4, 0); else if (o == 3) LK_MG(3, 4, 0); else LK_MG(4, 4, 0);
        } else {
            if (o == 2) LK_MG(2, 12, 0); else if (o == 3) LK_MG(3, 12, 0); else LK_MG(4, 12, 0);
        }
#undef LK_MG
    }
    // blocks that go to the subspace iteration (too wide for the direct solver) also get a float32 copy of C: the
    // early steps' filter products stream it instead (half the bytes); optional — without workspace nothing changes
    // (+ 4 KB: the product kernels' unconditional loads of the last row may run a few columns past the last matrix)
    float *G32 = (Pc > PLD_DIRECT_MAX && (Pc & 3) == 0) ? (float *)ws.alloc((size_t)B * ldg * ldg * 4 + 4096) : nullptr;
    if (MEXP_SYM && (Pc & 3) == 0 && Pc >= 128) {
        const int T = (Pc + 63) / 64;
        hipLaunchKernelGGL(pld_moment_expand_sym_kernel, dim3(T * (T + 1) / 2, B), dim3(256), 0, stream, Mcan, (size_t)pl.ldm * pl.ldm,
                           pl.d_src, d_mean, Pc, ldg, (double)N, G, G32, T);
    } else {
        hipLaunchKernelGGL(pld_moment_expand_kernel, dim3((Pc * ((Pc + 3) / 4) + 256 * MEXP_U - 1) / (256 * MEXP_U), B), dim3(256), 0, stream, Mcan,
                           (size_t)pl.ldm * pl.ldm, pl.d_src, d_mean, Pc, ldg, (double)N, G, G32);
    }
    double *V = nullptr, *lam = nullptr;
    const int rc = eig_topk(h, G, ldg, B, Pc, ko, true, false, &V, &lam, stream, ws, G32);
    if (rc) return rc;
    {
        const int kt = (ko + 15) / 16;
        const int wr = kt <= 1 ? (o >= 3 ? 4 : 2) : 1;  // row tiles per wave (the wider bases keep one: their accumulators are the registers)
        const size_t lds = (size_t)(64 * wr * (k1 | 1)) * 8 + (size_t)Pc * 4;
        const dim3 grid((N + 64 * wr - 1) / (64 * wr), B);
        double *d_mv = (double *)ws.alloc((size_t)B * ko * 8);
        if (!d_mv) {
            set_error("PLD workspace exhausted (mean projection)");
            return LK_ENOMEM;
        }
        hipLaunchKernelGGL(pld_mean_proj_kernel, dim3(B), dim3(256), 0, stream, d_mean, V, Pc, ko, d_mv);
#define LK_PP(O, KT, WR) hipLaunchKernelGGL((pld_project_products_kernel<O, KT, WR>), grid, dim3(256), lds, stream, X, K, col1, k1, N, Pc, \
                                            pl.d_packed, d_mv, V, lam, ko, col0, X)
#define LK_PPO(O) do { if (kt <= 1) LK_PP(O, 1, (O >= 3 ? 4 : 2)); else if (kt == 2) LK_PP(O, 2, 1); else LK_PP(O, 3, 1); } while (0)
        if (o == 2) LK_PPO(2); else if (o == 3) LK_PPO(3); else LK_PPO(4);
#undef LK_PPO
#undef LK_PP
    }
    return LK_OK;
}

int pld_design_launch(lk_handle *h, int B, int N, int P, int Pb, const float *pld_pix, const float *bkg_pix,
                      const float *lc_flux, const double *time, const double *knots, int n_inner, int pld_order,
                      int pca_components, int n_knots, int spline_degree, int normalize_bkg, int K, double *X,
                      double *prior_sigma, hipStream_t stream) {
    LK_REQUIRE(B >= 1 && N >= 2, "need B >= 1 cutouts with N >= 2 cadences");
    LK_REQUIRE(pca_components >= 1 && pca_components <= 48, "pca_components must be between 1 and 48 on the HIP path");
    LK_REQUIRE(pld_order >= 0 && pld_order <= 4, "pld_order outside 0..4");
    LK_REQUIRE(Pb >= 1 && bkg_pix, "at least one background pixel is required");
    LK_REQUIRE(spline_degree >= 0 && spline_degree <= 7, "spline_degree outside 0..7");
    LK_REQUIRE(n_knots == n_inner + spline_degree + 1 && n_inner >= 0, "n_knots must equal n_inner + degree + 1");
    LK_REQUIRE(K == pld_design_width(P, Pb, pld_order, pca_components, n_knots), "K does not match the design width");
    LK_REQUIRE(lc_flux && time && knots && X && prior_sigma, "NULL buffer");
    const int k1 = (P > 0 && pld_order > 0) ? std::min(pca_components, P) : 0;
    int pmax = std::max(P, Pb);
    for (int o = 2; o <= pld_order && k1 > 0; ++o) pmax = std::max(pmax, ncombos(k1, o));
    LK_REQUIRE(pmax <= 4096, "a design block has %d columns before PCA; the HIP path supports up to 4096", pmax);
    const int lmax = PLD_LMAX;
    const int ldgmax = ((pmax + 63) / 64) * 64;
    h->ws.reset();
    const size_t per = (size_t)N * pmax * 8 + (size_t)2 * ldgmax * ldgmax * 8 + (size_t)4 * pmax * lmax * 8 +
                       (size_t)pmax * pca_components * 8 + (size_t)ldgmax * ldgmax * 4 + 4096;  // 2 x ldgmax^2: Gram + the moment form's canonical array
    int rc = h->ws.reserve((size_t)B * per * 2 + (size_t)(B + 1) * 8 + 65536);
    if (rc) return rc;
    std::vector<int64_t> off((size_t)B + 1);
    for (int b = 0; b <= B; ++b) off[b] = (int64_t)b * N;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    LK_HIP_CHECK(hipMemcpyAsync(d_off, off.data(), (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));
    double *A = (double *)h->ws.alloc((size_t)B * N * pmax * 8);
    LK_REQUIRE(A != nullptr, "PLD workspace exhausted");
    int col = 0;
    const size_t mark = h->ws.used;
    if (k1 > 0) {
        LK_REQUIRE(pld_pix != nullptr, "pld_pix is NULL");
        double *d_cm = (double *)h->ws.alloc((size_t)B * P * 8);
        LK_REQUIRE(d_cm != nullptr, "PLD workspace exhausted (column means)");
        hipLaunchKernelGGL(pld_colmean_kernel, dim3(B), dim3(1024), 0, stream, pld_pix, lc_flux, 1, N, P, d_cm);
        if (pca_f32_ok(P, k1)) {  // the Gram kernel and the projection form pix / flux - mean themselves
            rc = pca_block(h, nullptr, B, N, P, k1, d_off, X, K, col, stream, h->ws, true, false, -1.0, PcaF32Source{pld_pix, lc_flux, d_cm, 1});
        } else {
            hipLaunchKernelGGL(pld_ratio_kernel, dim3((N + 31) / 32, B), dim3(256), 0, stream, pld_pix, lc_flux, 1, N, P, A, d_cm, (const float *)nullptr);
            rc = pca_block(h, A, B, N, P, k1, d_off, X, K, col, stream, h->ws, true);
        }
        if (rc) return rc;
        const int col1 = col;
        col += k1;
        for (int o = 2; o <= pld_order; ++o) {
            h->ws.used = mark;  // the previous block's Gram / subspace scratch is dead once its kernels are enqueued
            const int Pc = ncombos(k1, o), ko = std::min(pca_components, Pc);
            // combination table (itertools.combinations_with_replacement order)
            std::vector<uint8_t> comb((size_t)Pc * o);
            {
                std::vector<int> cur((size_t)o, 0);
                for (int idx = 0; idx < Pc; ++idx) {
                    for (int pos = 0; pos < o; ++pos) comb[(size_t)idx * o + pos] = (uint8_t)cur[pos];
                    int pos = o - 1;
                    while (pos >= 0 && cur[pos] == k1 - 1) --pos;
                    if (pos >= 0) {
                        const int v = cur[pos] + 1;
                        for (int q = pos; q < o; ++q) cur[q] = v;
                    }
                }
            }
            uint8_t *d_comb = (uint8_t *)h->ws.alloc(comb.size());
            double *d_mean = (double *)h->ws.alloc((size_t)B * Pc * 8);
            if (!d_comb || !d_mean) {
                set_error("PLD workspace exhausted (products)");
                return LK_ENOMEM;
            }
            LK_HIP_CHECK(hipMemcpyAsync(d_comb, comb.data(), comb.size(), hipMemcpyHostToDevice, stream));
            LK_HIP_CHECK(hipStreamSynchronize(stream));  // comb dies at the end of this iteration
            // wide product blocks: moment-form Gram, the products are never materialised (A serves as its scratch) and their
            // column means come out of the Gram kernel
            const int ldm = ((Pc + 63) / 64) * 64;
            const bool moment = Pc >= kMomentMinCols && k1 <= 48;
            if (!moment)
                hipLaunchKernelGGL(pld_products_mean_kernel, dim3((Pc + 255) / 256, B), dim3(1024), 0, stream, X, K, col1, k1, o,
                                   N, Pc, d_comb, d_mean);
            if (moment) {
                const MomentPlan *pl = moment_plan(h, k1, o, comb, Pc);
                if (!pl) return LK_ENOMEM;
                // scratch for the canonical moments: A is free during a product block — when it is large enough
                double *Mcan = (size_t)ldm * ldm <= (size_t)N * pmax ? A : (double *)h->ws.alloc((size_t)B * ldm * ldm * 8);
                if (!Mcan) {
                    set_error("PLD workspace exhausted (moments)");
                    return LK_ENOMEM;
                }
                rc = pca_products_moment(h, *pl, B, N, k1, ko, X, K, col1, col, d_mean, Mcan, stream, h->ws);
                if (rc) return rc;
            } else {
                hipLaunchKernelGGL(pld_products_kernel, dim3((N + PP_ROWS - 1) / PP_ROWS, B), dim3(256), 0, stream, X, K, col1, k1, o,
                                   N, Pc, d_comb, d_mean, A);
                rc = pca_block(h, A, B, N, Pc, ko, d_off, X, K, col, stream, h->ws, true, true);
                if (rc) return rc;
            }
            col += ko;
        }
    }
    const int n_pld_cols = col;
    h->ws.used = mark;
    bool bkg_f32 = false;
    PcaF32Source bkg_src{nullptr, nullptr, nullptr, 0};
    {
        double *d_cm = (double *)h->ws.alloc((size_t)B * Pb * 8);
        float *d_div = normalize_bkg ? (float *)h->ws.alloc((size_t)B * N * 4) : nullptr;
        LK_REQUIRE(d_cm != nullptr && (!normalize_bkg || d_div != nullptr), "PLD workspace exhausted (column means)");
        if (normalize_bkg) hipLaunchKernelGGL(pld_rowdiv_kernel, dim3((N + 31) / 32, B), dim3(256), 0, stream, bkg_pix, N, Pb, d_div);
        hipLaunchKernelGGL(pld_colmean_kernel, dim3(B), dim3(1024), 0, stream, bkg_pix, d_div, normalize_bkg ? 2 : 0, N, Pb, d_cm);
        bkg_f32 = pca_f32_ok(Pb, std::min(pca_components, Pb));
        if (bkg_f32)
            bkg_src = PcaF32Source{bkg_pix, d_div, d_cm, normalize_bkg ? 2 : 0};
        else
            hipLaunchKernelGGL(pld_ratio_kernel, dim3((N + 31) / 32, B), dim3(256), 0, stream, bkg_pix, lc_flux,
                               normalize_bkg ? 2 : 0, N, Pb, A, d_cm, d_div);
    }
    const int kb = std::min(pca_components, Pb);
    if (bkg_f32)
        rc = pca_block(h, nullptr, B, N, Pb, kb, d_off, X, K, col, stream, h->ws, true, false, -1.0, bkg_src);
    else
        rc = pca_block(h, A, B, N, Pb, kb, d_off, X, K, col, stream, h->ws, true);
    if (rc) return rc;
    col += kb;
    hipLaunchKernelGGL(pld_spline_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, time, knots, n_inner,
                       spline_degree, N, K, col, X, 1);
    hipLaunchKernelGGL(pld_prior_kernel, dim3(B), dim3(256), 0, stream, lc_flux, N, K, n_pld_cols, pca_components,
                       prior_sigma);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ---------------------------------------------------------------------------------------- standalone design-matrix operations
// DesignMatrix.pca (correctors/designmatrix.py:252-282: fbpca.pca(values, nterms) -> U): the first k left singular vectors of
// the column-centred matrix, for B same-shaped matrices.  Same machinery as the PLD blocks: Gram on the fp64 matrix cores,
// top-k eigenpairs by subspace iteration, U = A V diag(lambda)^(-1/2).  A is not modified.
int dm_pca_launch(lk_handle *h, int B, int N, int P, int k, const double *A_in, double *U, hipStream_t stream) {
    LK_REQUIRE(B >= 1 && N >= 2 && P >= 1, "need B >= 1 matrices of N >= 2 rows and P >= 1 columns");
    LK_REQUIRE(k >= 1 && k <= 48 && k <= P, "nterms must be between 1 and min(48, columns) on the HIP path (got %d for %d columns)", k, P);
    LK_REQUIRE(P <= 4096, "the HIP path supports up to 4096 columns (got %d)", P);
    LK_REQUIRE(A_in && U, "NULL buffer");
    const int lmax = PLD_LMAX, ldg = ((P + 63) / 64) * 64;
    h->ws.reset();
    const size_t per = (size_t)N * P * 8 + (size_t)2 * ldg * ldg * 8 + (size_t)4 * P * lmax * 8 + (size_t)P * k * 8 + 4096;
    int rc = h->ws.reserve((size_t)B * per * 2 + (size_t)(B + 1) * 8 + 65536);
    if (rc) return rc;
    std::vector<int64_t> off((size_t)B + 1);
    for (int b = 0; b <= B; ++b) off[b] = (int64_t)b * N;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    LK_HIP_CHECK(hipMemcpyAsync(d_off, off.data(), (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));  // `off` leaves scope
    double *A = (double *)h->ws.alloc((size_t)B * N * P * 8);
    LK_REQUIRE(A != nullptr, "workspace exhausted");
    LK_HIP_CHECK(hipMemcpyAsync(A, A_in, (size_t)B * N * P * 8, hipMemcpyDeviceToDevice, stream));
    rc = pca_block(h, A, B, N, P, k, d_off, U, k, 0, stream, h->ws, false, false, PCA_EIG_TOL);
    if (rc) return rc;
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// create_spline_matrix (correctors/designmatrix.py:952-997: patsy bs(x, ..., include_intercept=True) - 1): the clamped
// B-spline basis of x on knots [lo, interior..., hi] -> out[B][N][n_inner + degree + 1]
int dm_spline_launch(lk_handle *h, int B, int N, const double *x, const double *knots, int n_inner, int degree, double *out,
                     hipStream_t stream) {
    LK_REQUIRE(B >= 1 && N >= 1 && n_inner >= 0, "bad shapes");
    LK_REQUIRE(B <= 65535, "at most 65535 sample vectors per call (grid.y; got %d): split the batch", B);
    LK_REQUIRE(degree >= 0 && degree <= 7, "spline degree outside 0..7");
    LK_REQUIRE(x && knots && out, "NULL buffer");
    (void)h;
    const int nb = n_inner + degree + 1;
    hipLaunchKernelGGL(pld_spline_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, x, knots, n_inner, degree, N, nb, 0, out, 0);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
