// fold.hip — LightCurve.fold for a ragged batch on gfx950.
//
// Reference: src/lightkurve/lightcurve.py:1089-1214 over astropy TimeSeries.fold (astropy@4.3.1
// timeseries/sampled.py:230-233): phase = ((t - epoch) + epoch_phase + (P - wrap)) % P - (P - wrap)  (numpy's `%`:
// the result takes the divisor's sign), then the table is sorted by phase (a stable sort: ties keep cadence order).
// One workgroup per target: phases with the same operation order as the reference (fmod is exact, so they are
// bit-identical), then a bitonic sort of (sortable phase key, cadence index) pairs — the index makes every key unique,
// which turns the network into a stable sort — in an L2-resident scratch slab; stages whose stride fits a 4096-pair
// LDS tile run there.  Outputs: sorted phase, the permutation, and any number of gathered value columns.
#include <algorithm>
#include <cmath>
#include <vector>

#include "block_select.hpp"  // f64_sortable
#include "lk_common.hpp"

namespace lk {

__device__ __forceinline__ double np_mod(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

struct FoldKey {
    unsigned long long k;  // order-preserving bits of the phase (NaN sorts last, like numpy)
    unsigned int i;        // cadence index (tie-break => stable)
    unsigned int pad;
};

__device__ __forceinline__ bool fold_less(const FoldKey &a, const FoldKey &b) {
    return a.k < b.k || (a.k == b.k && a.i < b.i);
}

constexpr int FOLD_TILE = 4096;

__global__ __launch_bounds__(1024) void fold_kernel(const double *__restrict__ t, const int64_t *__restrict__ n_off,
                                                     const double *__restrict__ period,
                                                     const double *__restrict__ epoch_time, double epoch_phase,
                                                     const double *__restrict__ wrap_phase, int normalize_phase,
                                                     FoldKey *__restrict__ scratch,
                                                     const int64_t *__restrict__ scratch_off,
                                                     double *__restrict__ phase_out, int64_t *__restrict__ order_out) {
    __shared__ FoldKey tile[FOLD_TILE];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int64_t lo = n_off[b];
    const int n = (int)(n_off[b + 1] - lo);
    if (n <= 0) return;
    t += lo;
    FoldKey *keys = scratch + scratch_off[b];
    int npad = 1;
    while (npad < n) npad <<= 1;
    const double P = period[b], ep = epoch_time[b];
    // astropy: wrap_phase defaults to P/2 (or 0.5 when normalising); in phase units it is multiplied by P
    const double wrap = normalize_phase ? wrap_phase[b] * P : wrap_phase[b];
    const double eph = normalize_phase ? epoch_phase * P : epoch_phase;
    const double shift = P - wrap;
    for (int i = tid; i < npad; i += nt) {
        FoldKey kk;
        if (i < n) {
            const double rel = ((t[i] - ep) + eph) + shift;
            double ph = np_mod(rel, P) - shift;
            if (normalize_phase) ph = ph / P;
            phase_out[lo + i] = ph;  // cadence order for now; permuted below
            kk.k = isnan(ph) ? 0xfffffffffffffffeull : f64_sortable(ph == 0.0 ? 0.0 : ph);  // -0.0 == +0.0 for the sort
            kk.i = (unsigned int)i;
        } else {
            kk.k = 0xffffffffffffffffull;  // padding sorts after everything
            kk.i = (unsigned int)i;
        }
        kk.pad = 0;
        keys[i] = kk;
    }
    __syncthreads();
    // bitonic network over npad pairs; (size, stride) stages with stride < FOLD_TILE / 2 ... run tile-wise in LDS
    for (int size = 2; size <= npad; size <<= 1) {
        int stride = size >> 1;
        // global stages while the partner is outside a tile
        for (; stride >= FOLD_TILE; stride >>= 1) {
            for (int q = tid; q < (npad >> 1); q += nt) {
                const int i = ((q / stride) * stride << 1) + (q % stride), j = i + stride;
                const bool up = ((i & size) == 0);
                const FoldKey a = keys[i], c = keys[j];
                if (fold_less(c, a) == up) {
                    keys[i] = c;
                    keys[j] = a;
                }
            }
            __syncthreads();
        }
        // remaining strides of this size: whole tiles through LDS
        for (int t0 = 0; t0 < npad; t0 += FOLD_TILE) {
            const int tn = min(FOLD_TILE, npad - t0);
            for (int i = tid; i < tn; i += nt) tile[i] = keys[t0 + i];
            __syncthreads();
            for (int s2 = min(stride, tn >> 1); s2 >= 1; s2 >>= 1) {
                for (int q = tid; q < (tn >> 1); q += nt) {
                    const int i = ((q / s2) * s2 << 1) + (q % s2), j = i + s2;
                    const bool up = (((t0 + i) & size) == 0);
                    const FoldKey a = tile[i], c = tile[j];
                    if (fold_less(c, a) == up) {
                        tile[i] = c;
                        tile[j] = a;
                    }
                }
                __syncthreads();
            }
            for (int i = tid; i < tn; i += nt) keys[t0 + i] = tile[i];
            __syncthreads();
        }
    }
    // permutation out; phases permuted through the scratch (keys[].k no longer needed once read)
    for (int i = tid; i < n; i += nt) order_out[lo + i] = (int64_t)keys[i].i;
    __syncthreads();
    double *tmp = reinterpret_cast<double *>(keys);  // overlay: element i of tmp lies inside keys[i/2]
    for (int i = tid; i < n; i += nt) {
        const unsigned int src = (unsigned int)order_out[lo + i];
        const double ph = phase_out[lo + src];
        // stash in the upper half of the slab (npad pairs * 16 B >= 2 * n * 8 B): disjoint from nothing we still read
        tmp[npad + i] = ph;
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) phase_out[lo + i] = tmp[npad + i];
}

// out[lo + i] = in[lo + order[lo + i]]
__global__ __launch_bounds__(256) void fold_gather_kernel(const double *__restrict__ in,
                                                           const int64_t *__restrict__ order,
                                                           const int64_t *__restrict__ n_off,
                                                           double *__restrict__ out) {
    const int b = blockIdx.y;
    const int64_t lo = n_off[b], n = n_off[b + 1] - lo;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[lo + i] = in[lo + order[lo + i]];
}

int fold_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *period_host,
                const double *epoch_time_host, double epoch_phase, const double *wrap_phase_host, int normalize_phase,
                int ncols, const double *const *cols_in, double *const *cols_out, double *phase, int64_t *order,
                hipStream_t stream) {
    LK_REQUIRE(B >= 0 && n_off_host != nullptr, "bad batch description");
    LK_REQUIRE(B <= 65535, "at most 65535 targets per call on this path (got %d): split the batch", B);
    if (B == 0) return LK_OK;
    LK_REQUIRE(t && period_host && epoch_time_host && wrap_phase_host && phase && order, "NULL buffer");
    LK_REQUIRE(n_off_host[0] == 0, "n_off[0] must be 0");
    LK_REQUIRE(ncols >= 0 && (ncols == 0 || (cols_in && cols_out)), "bad column list");
    std::vector<int64_t> soff((size_t)B + 1, 0);
    int64_t nmax = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t n = n_off_host[b + 1] - n_off_host[b];
        LK_REQUIRE(n >= 0 && n < ((int64_t)1 << 30), "target %d has %lld cadences", b, (long long)n);
        LK_REQUIRE(std::isfinite(period_host[b]) && period_host[b] != 0.0, "target %d: period must be finite and non-zero", b);
        int64_t npad = 1;
        while (npad < n) npad <<= 1;
        soff[b + 1] = soff[b] + npad;
        nmax = std::max(nmax, n);
    }
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 16 + (size_t)B * 24 + (size_t)soff[B] * sizeof(FoldKey) + 4096);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8), *d_soff = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    double *d_par = (double *)h->ws.alloc((size_t)B * 24);
    FoldKey *d_keys = (FoldKey *)h->ws.alloc((size_t)soff[B] * sizeof(FoldKey));
    std::vector<double> par((size_t)3 * B);
    for (int b = 0; b < B; ++b) {
        par[b] = period_host[b];
        par[B + b] = epoch_time_host[b];
        par[2 * B + b] = wrap_phase_host[b];
    }
    LK_HIP_CHECK(hipMemcpyAsync(d_off, n_off_host, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_soff, soff.data(), (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_par, par.data(), (size_t)B * 24, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));
    hipLaunchKernelGGL(fold_kernel, dim3(B), dim3(1024), 0, stream, t, d_off, d_par, d_par + B, epoch_phase, d_par + 2 * B,
                       normalize_phase, d_keys, d_soff, phase, order);
    for (int c = 0; c < ncols; ++c) {
        LK_REQUIRE(cols_in[c] && cols_out[c] && cols_in[c] != cols_out[c], "column %d: NULL or aliased buffers", c);
        hipLaunchKernelGGL(fold_gather_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)B), dim3(256), 0, stream,
                           cols_in[c], order, d_off, cols_out[c]);
    }
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
