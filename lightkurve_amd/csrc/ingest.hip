// ingest.hip — the steps BEFORE the hot path for a ragged batch of light curves on gfx950: what a
// pipeline over a LightCurveCollection does per target in Python/astropy today, done for all targets in a few launches so
// that batches reach the periodogram / flatten / regression kernels without a host round trip per light curve.
//
//   lk_ingest_batch         LightCurve.remove_nans (src/lightkurve/lightcurve.py:1300-1327) + .normalize (:1216-1292):
//                           drop the cadences whose flux is NaN, order preserved, repack the batch contiguously, divide
//                           flux and flux_err by nanmedian(flux)
//   lk_transit_mask_batch   LightCurve.create_transit_mask (:2967-3037): |((t - t0 + P/2) % P) - P/2| < duration / 2 for
//                           any of the planets (numpy `%`: result carries the divisor's sign)
//   lk_bin_batch            LightCurve.bin (:1558-1763) over astropy aggregate_downsample (astropy@4.3.1
//                           timeseries/downsample.py:12-125): equal-width bins from time_bin_start, nanmean of the flux,
//                           root-mean-square of flux_err (or nanstd of the flux when there are no errors)
#include <cmath>
#include <vector>

#include "block_select.hpp"
#include "lk_common.hpp"

namespace lk {

__device__ __forceinline__ double np_mod_ingest(double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

// ------------------------------------------------------------------------------------------------ remove_nans + normalize
__global__ __launch_bounds__(256) void ingest_count_kernel(const double *__restrict__ flux, const int64_t *__restrict__ n_off,
                                                            int64_t *__restrict__ kept) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t lo = n_off[b], n = n_off[b + 1] - lo;
    long long c = 0;
    for (int64_t i = tid; i < n; i += 256) c += isnan(flux[lo + i]) ? 0 : 1;
    __shared__ long long sh[8];
    const long long tot = block_count_fast(c, sh);
    if (tid == 0) kept[b] = tot;
}

__global__ __launch_bounds__(1024) void ingest_scan_kernel(const int64_t *__restrict__ kept, int B, int64_t *__restrict__ new_off) {
    // exclusive prefix sum of B counts by one workgroup: per-thread chunks + wave scans
    __shared__ long long sh[32];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int chunk = (B + nt - 1) / nt;
    const int lo = min(tid * chunk, B), hi = min(lo + chunk, B);
    long long c = 0;
    for (int i = lo; i < hi; ++i) c += kept[i];
    long long inc = c;
    const int lane = tid & 63, nw = nt >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const long long v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) sh[tid >> 6] = inc;
    __syncthreads();
    long long base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += sh[w];
    long long run = base + inc - c;
    for (int i = lo; i < hi; ++i) {
        new_off[i] = run;
        run += kept[i];
    }
    if (tid == nt - 1) {
        long long tot = 0;
        for (int w = 0; w < nw; ++w) tot += sh[w];
        new_off[B] = tot;
    }
}

__global__ __launch_bounds__(512) void ingest_pack_kernel(const double *__restrict__ t, const double *__restrict__ flux,
                                                           const double *__restrict__ err, const int64_t *__restrict__ n_off,
                                                           const int64_t *__restrict__ new_off, int normalize,
                                                           double *__restrict__ t_out, double *__restrict__ f_out,
                                                           double *__restrict__ e_out, double *__restrict__ median_out,
                                                           int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long ing_lds[];
    unsigned long long *sh = ing_lds;                               // 512 words
    double *cand = reinterpret_cast<double *>(ing_lds + 512);       // cap doubles
    int *shi = reinterpret_cast<int *>(cand + cap);
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int64_t lo = n_off[b];
    const int n = (int)(n_off[b + 1] - lo);
    const int64_t olo = new_off[b];
    const int nk = (int)(new_off[b + 1] - olo);
    t += lo;
    flux += lo;
    if (err) err += lo;
    t_out += olo;
    f_out += olo;
    if (e_out) e_out += olo;
    // order-preserving compaction: every wave owns a contiguous strip, positions from ballot prefixes
    const int nw = nt >> 6, wv = tid >> 6, lane = tid & 63;
    const int strip = ((n + nw - 1) / nw + 63) & ~63;
    const int k_lo = min(wv * strip, n), k_hi = min(k_lo + strip, n);
    int c = 0;
    for (int k = k_lo + lane; k < k_hi; k += 64) c += isnan(flux[k]) ? 0 : 1;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) shi[wv] = c;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += shi[w];
    for (int k0 = k_lo; k0 < k_hi; k0 += 64) {
        const int k = k0 + lane;
        const bool in = k < k_hi;
        const double f = in ? flux[k] : 0.0;
        const bool m = in && !isnan(f);
        const unsigned long long bal = __ballot(m);
        if (m) {
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            t_out[pos] = t[k];
            f_out[pos] = f;
            if (e_out) e_out[pos] = err ? err[k] : __longlong_as_double(0x7ff8000000000000ll);
        }
        base += __popcll(bal);
    }
    __syncthreads();
    // nanmedian of the flux = median of the kept values; then flux /= median, flux_err /= median (:1283-1284)
    auto val = [&](int i) { return f_out[i]; };
    auto keep = [&](int) { return true; };
    const double med = block_median_sampled(nk, (long long)nk, val, keep, sh, cand, cap);
    if (tid == 0 && median_out) median_out[b] = med;
    if (normalize) {
        for (int i = tid; i < nk; i += nt) {
            f_out[i] = f_out[i] / med;
            if (e_out) e_out[i] = e_out[i] / med;
        }
    }
}

int ingest_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux, const double *err,
                  int normalize, double *t_out, double *f_out, double *e_out, int64_t *new_off_host, double *median_out,
                  hipStream_t stream) {
    LK_REQUIRE(B >= 0 && n_off_host != nullptr && new_off_host != nullptr, "bad batch description");
    if (B == 0) {
        new_off_host[0] = 0;
        return LK_OK;
    }
    LK_REQUIRE(t && flux && t_out && f_out, "NULL buffer");
    LK_REQUIRE(n_off_host[0] == 0, "n_off[0] must be 0");
    for (int b = 0; b < B; ++b) {
        const int64_t n = n_off_host[b + 1] - n_off_host[b];
        LK_REQUIRE(n >= 0 && n < ((int64_t)1 << 30), "target %d has %lld cadences", b, (long long)n);
    }
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 8 * 3 + 4096);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    int64_t *d_kept = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    int64_t *d_new = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    rc = h->stage.copy(d_off, n_off_host, (size_t)(B + 1) * 8, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(ingest_count_kernel, dim3(B), dim3(256), 0, stream, flux, d_off, d_kept);
    hipLaunchKernelGGL(ingest_scan_kernel, dim3(1), dim3(1024), 0, stream, d_kept, B, d_new);
    constexpr int cap = 4096;
    const size_t lds = 512 * 8 + (size_t)cap * 8 + 64 * 4;
    static bool attr = false;
    if (!attr) {
        LK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ingest_pack_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(ingest_pack_kernel, dim3(B), dim3(512), lds, stream, t, flux, err, d_off, d_new, normalize, t_out,
                       f_out, e_out, median_out, cap);
    LK_HIP_CHECK(hipMemcpyAsync(new_off_host, d_new, (size_t)(B + 1) * 8, hipMemcpyDeviceToHost, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));  // the caller needs the new offsets to address the packed batch
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ create_transit_mask
__global__ __launch_bounds__(256) void transit_mask_kernel(const double *__restrict__ t, int64_t ntot,
                                                            const int64_t *__restrict__ n_off, int B,
                                                            const double *__restrict__ period,
                                                            const double *__restrict__ duration,
                                                            const double *__restrict__ transit_time,
                                                            const int *__restrict__ p_off, uint8_t *__restrict__ mask) {
    // one workgroup per (target, 256-cadence tile): blockIdx.y = target
    const int b = blockIdx.y;
    const int64_t lo = n_off[b], n = n_off[b + 1] - lo;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double tv = t[lo + i];
    bool in_transit = false;
    for (int p = p_off[b]; p < p_off[b + 1]; ++p) {
        const double per = period[p], hp = per / 2.0;
        in_transit |= fabs(np_mod_ingest(tv - transit_time[p] + hp, per) - hp) < 0.5 * duration[p];
    }
    mask[lo + i] = in_transit ? 1 : 0;
}

int transit_mask_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const int *p_off_host,
                        const double *period_host, const double *duration_host, const double *transit_time_host,
                        uint8_t *mask, hipStream_t stream) {
    LK_REQUIRE(B >= 0 && n_off_host && p_off_host, "bad batch description");
    if (B == 0) return LK_OK;
    LK_REQUIRE(t && mask, "NULL buffer");
    LK_REQUIRE(B <= 65535, "at most 65535 targets per call");
    const int np_tot = p_off_host[B];
    LK_REQUIRE(p_off_host[0] == 0 && np_tot >= 0, "p_off must be prefix offsets starting at 0");
    LK_REQUIRE(np_tot == 0 || (period_host && duration_host && transit_time_host), "NULL planet parameters");
    int64_t nmax = 0;
    for (int b = 0; b < B; ++b) {
        LK_REQUIRE(p_off_host[b + 1] >= p_off_host[b], "p_off must be non-decreasing");
        nmax = std::max(nmax, n_off_host[b + 1] - n_off_host[b]);
    }
    for (int p = 0; p < np_tot; ++p) LK_REQUIRE(period_host[p] != 0.0, "period must be non-zero");
    if (nmax == 0) return LK_OK;
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 12 + (size_t)np_tot * 24 + 8 * 256 + 4096);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    int *d_poff = (int *)h->ws.alloc((size_t)(B + 1) * 4);
    double *d_par = (double *)h->ws.alloc((size_t)std::max(np_tot, 1) * 24);
    std::vector<double> par((size_t)std::max(np_tot, 1) * 3, 1.0);
    for (int p = 0; p < np_tot; ++p) {
        par[p] = period_host[p];
        par[(size_t)np_tot + p] = duration_host[p];
        par[2 * (size_t)np_tot + p] = transit_time_host[p];
    }
    LK_HIP_CHECK(hipMemcpyAsync(d_off, n_off_host, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_poff, p_off_host, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_par, par.data(), par.size() * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));  // host staging buffers go out of scope
    hipLaunchKernelGGL(transit_mask_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)B), dim3(256), 0, stream, t,
                       n_off_host[B], d_off, B, d_par, d_par + np_tot, d_par + 2 * (size_t)np_tot, d_poff, mask);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ bin
// One thread per (target, bin).  edges[k] = the running sum 0, s, s + s, ... in seconds exactly as numpy's cumsum forms it
// (host-built, shared by all targets); a cadence with relative time r [s] belongs to bin k when edges[k] < r <= edges[k+1]
// (r == 0 -> bin 0) and r < edges[n_bins] (downsample.py:85-97).  Times are sorted, so a bin is a contiguous cadence range,
// found by two binary searches.
__global__ __launch_bounds__(256) void bin_kernel(const double *__restrict__ t, const double *__restrict__ flux,
                                                   const double *__restrict__ err, const int64_t *__restrict__ n_off,
                                                   const int64_t *__restrict__ bin_off, int B,
                                                   const double *__restrict__ start, const double *__restrict__ edges,
                                                   double bin_size_sec, const uint8_t *__restrict__ has_err,
                                                   double *__restrict__ t_out, double *__restrict__ f_out,
                                                   double *__restrict__ e_out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= bin_off[B]) return;
    int b_lo = 0, b_hi = B;  // the target that owns global bin g: last b with bin_off[b] <= g
    while (b_hi - b_lo > 1) {
        const int mid = (b_lo + b_hi) >> 1;
        if (bin_off[mid] <= g)
            b_lo = mid;
        else
            b_hi = mid;
    }
    const int b = b_lo;
    const int k = (int)(g - bin_off[b]), nb = (int)(bin_off[b + 1] - bin_off[b]);
    const int64_t lo = n_off[b];
    const int n = (int)(n_off[b + 1] - lo);
    const double t0 = start[b];
    auto rel = [&](int i) { return (t[lo + i] - t0) * 86400.0; };
    // first cadence with rel > x (strict) / rel >= x
    auto first_gt = [&](double x) {
        int a = 0, c = n;
        while (a < c) {
            const int mid = (a + c) >> 1;
            if (rel(mid) > x)
                c = mid;
            else
                a = mid + 1;
        }
        return a;
    };
    auto first_ge = [&](double x) {
        int a = 0, c = n;
        while (a < c) {
            const int mid = (a + c) >> 1;
            if (rel(mid) >= x)
                c = mid;
            else
                a = mid + 1;
        }
        return a;
    };
    const int i0 = k == 0 ? first_ge(edges[0]) : first_gt(edges[k]);
    int i1 = first_gt(edges[k + 1]);
    i1 = min(i1, first_ge(edges[nb]));  // keep: rel < last edge
    double s = 0.0, s2 = 0.0;
    int cf = 0, ce = 0;
    for (int i = i0; i < i1; ++i) {
        const double f = flux[lo + i];
        if (!isnan(f)) {
            s += f;
            ++cf;
        }
        if (err && has_err[b]) {
            const double e = err[lo + i];
            if (isfinite(e)) {  // rmse: sqrt(nansum(x^2) / count(isfinite(x)))  (lightcurve.py:167-172)
                ++ce;
            }
            if (!isnan(e)) s2 = fma(e, e, s2);
        }
    }
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double mean = cf ? s / (double)cf : qnan;
    double eo;
    if (err && has_err[b]) {
        eo = ce ? sqrt(s2 / (double)ce) : qnan;
    } else {  // no usable errors anywhere in this light curve: nanstd of the flux in the bin
        double v = 0.0;
        for (int i = i0; i < i1; ++i) {
            const double f = flux[lo + i];
            if (!isnan(f)) {
                const double d = f - mean;
                v = fma(d, d, v);
            }
        }
        eo = cf ? sqrt(v / (double)cf) : qnan;
    }
    if (i1 <= i0) eo = qnan;
    t_out[g] = t0 + (edges[k] + bin_size_sec / 2.0) / 86400.0;
    f_out[g] = (i1 > i0) ? mean : qnan;
    e_out[g] = eo;
}

int bin_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux, const double *err,
               const int64_t *bin_off_host, const double *start_host, const double *edges_host, int64_t n_edges,
               double bin_size_sec, const uint8_t *has_err_host, double *t_out, double *f_out, double *e_out,
               hipStream_t stream) {
    LK_REQUIRE(B >= 0 && n_off_host && bin_off_host, "bad batch description");
    if (B == 0) return LK_OK;
    LK_REQUIRE(t && flux && start_host && edges_host && has_err_host && t_out && f_out && e_out, "NULL buffer");
    LK_REQUIRE(bin_size_sec > 0.0, "time_bin_size must be positive");
    const int64_t nbins = bin_off_host[B];
    LK_REQUIRE(bin_off_host[0] == 0 && nbins >= 0, "bin_off must be prefix offsets starting at 0");
    for (int b = 0; b < B; ++b)
        LK_REQUIRE(bin_off_host[b + 1] >= bin_off_host[b] && bin_off_host[b + 1] - bin_off_host[b] < n_edges,
                   "target %d needs more bin edges than were passed", b);
    if (nbins == 0) return LK_OK;
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(B + 1) * 16 + (size_t)B * 9 + (size_t)n_edges * 8 + 6 * 256 + 4096);
    if (rc) return rc;
    int64_t *d_off = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8), *d_boff = (int64_t *)h->ws.alloc((size_t)(B + 1) * 8);
    double *d_start = (double *)h->ws.alloc((size_t)B * 8), *d_edges = (double *)h->ws.alloc((size_t)n_edges * 8);
    uint8_t *d_he = (uint8_t *)h->ws.alloc((size_t)B);
    LK_HIP_CHECK(hipMemcpyAsync(d_off, n_off_host, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_boff, bin_off_host, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_start, start_host, (size_t)B * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_edges, edges_host, (size_t)n_edges * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_he, has_err_host, (size_t)B, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));
    hipLaunchKernelGGL(bin_kernel, dim3((unsigned)((nbins + 255) / 256)), dim3(256), 0, stream, t, flux, err, d_off, d_boff,
                       B, d_start, d_edges, bin_size_sec, d_he, t_out, f_out, e_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
