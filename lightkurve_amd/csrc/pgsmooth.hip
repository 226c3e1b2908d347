// pgsmooth.hip — Periodogram.smooth / Periodogram.flatten on gfx950.
//
// Reference: src/lightkurve/periodogram.py:182-284 (smooth: 'boxkernel' = astropy.convolution.convolve with a
// Box1DKernel, 'logmedian' = moving nanmedian in log10(frequency) windows) and :381-429 (flatten = power / smooth).
// The reference's logmedian is a Python while-loop with one np.nanmedian per window; here every (window, target) pair
// is one workgroup running the radix select of block_select.hpp, and a second kernel averages, for every frequency,
// the medians of the windows that contain it — in window order, like the reference's `bkg[m] += ...`.
// The window bookkeeping (which frequencies fall in window k, which windows contain frequency j) depends only on the
// frequency grid and is prepared by the caller exactly as the reference does it (numpy log10, the same running sum
// for the window centres): see lightkurve_amd/periodogram.py::_logmedian_windows.
// Compiled with -ffp-contract=off: products and sums round separately, as in astropy's C convolution loop.
#include "block_select.hpp"
#include "lk_common.hpp"

namespace lk {

// one workgroup per (window k, target b): med[b][k] = nanmedian(power[b][lo_k : hi_k]) / corr
__global__ __launch_bounds__(256) void pg_window_median_kernel(const double *__restrict__ power, int64_t M,
                                                                const int *__restrict__ win_lo,
                                                                const int *__restrict__ win_hi, int K, double corr,
                                                                double *__restrict__ med) {
    __shared__ unsigned long long sh[264];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int lo = win_lo[k], n = win_hi[k] - lo;
    const double *row = power + (size_t)b * (size_t)M + lo;
    auto val = [&](int i) { return row[i]; };
    auto keep = [&](int i) { return !isnan(row[i]); };
    long long c = 0;
    for (int i = tid; i < n; i += 256) c += keep(i) ? 1 : 0;
    const long long cnt = block_count_dyn(c, reinterpret_cast<long long *>(sh));
    const double m = block_median(n, cnt, val, keep, sh);  // NaN if nothing is kept (np.nanmedian of all-NaN)
    if (tid == 0) med[(size_t)b * K + k] = m / corr;
}

// out[b][j] = (sum over the windows klo_j..khi_j that contain j, in window order) / (number of those windows)
__global__ __launch_bounds__(256) void pg_window_average_kernel(const double *__restrict__ med, int K,
                                                                 const int *__restrict__ klo,
                                                                 const int *__restrict__ khi, int64_t M,
                                                                 double *__restrict__ out) {
    const int b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const int a = klo[j], z = khi[j];
    double s = 0.0;
    for (int k = a; k <= z; ++k) s += med[(size_t)b * K + k];
    out[(size_t)b * (size_t)M + j] = s / (double)(z - a + 1);  // no window: 0/0 = NaN like the reference
}

// per target: does the row hold a NaN?  (astropy: nan_interpolate = isnan(array.sum()))
__global__ __launch_bounds__(256) void pg_has_nan_kernel(const double *__restrict__ power, int64_t M,
                                                          int *__restrict__ flag) {
    const int b = blockIdx.x;
    const double *row = power + (size_t)b * (size_t)M;
    int f = 0;
    for (int64_t j = threadIdx.x; j < M; j += 256) f |= isnan(row[j]) ? 1 : 0;
    f = __syncthreads_or(f);
    if (threadIdx.x == 0) flag[b] = f;
}

// astropy.convolution.convolve(power, kernel): boundary='fill' (zeros), normalize_kernel=True,
// nan_treatment='interpolate'.  taps = the (already flipped) kernel, ksum = its sum.
__global__ __launch_bounds__(256) void pg_boxsmooth_kernel(const double *__restrict__ power, int64_t M,
                                                            const double *__restrict__ taps, int nk, double ksum,
                                                            const int *__restrict__ has_nan,
                                                            double *__restrict__ out) {
    const int b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const double *row = power + (size_t)b * (size_t)M;
    const int half = nk / 2;
    double top = 0.0;
    if (!has_nan[b]) {
        for (int ii = 0; ii < nk; ++ii) {
            const int64_t idx = j + ii - half;
            const double v = (idx >= 0 && idx < M) ? row[idx] : 0.0;
            top += v * taps[ii];
        }
        out[(size_t)b * (size_t)M + j] = top / ksum;
    } else {
        double bot = 0.0;
        for (int ii = 0; ii < nk; ++ii) {
            const int64_t idx = j + ii - half;
            const double v = (idx >= 0 && idx < M) ? row[idx] : 0.0;
            if (!isnan(v)) {
                top += v * taps[ii];
                bot += taps[ii];
            }
        }
        out[(size_t)b * (size_t)M + j] = bot != 0.0 ? top / bot : __longlong_as_double(0x7ff8000000000000ll);
    }
}

int pg_logmedian_launch(lk_handle *h, int B, int64_t M, const double *power, int K, const int *win_lo_host,
                        const int *win_hi_host, const int *klo_host, const int *khi_host, double corr, double *out,
                        hipStream_t stream) {
    LK_REQUIRE(B >= 0 && M >= 1 && K >= 0, "need B >= 0, M >= 1, K >= 0");
    if (B == 0) return LK_OK;
    LK_REQUIRE(power && out && klo_host && khi_host, "NULL buffer");
    LK_REQUIRE(K == 0 || (win_lo_host && win_hi_host), "NULL window table");
    LK_REQUIRE(M < ((int64_t)1 << 31), "M too large");
    LK_REQUIRE(B <= 65535, "at most 65535 periodograms per call (got %d)", B);
    for (int k = 0; k < K; ++k)
        LK_REQUIRE(win_lo_host[k] >= 0 && win_lo_host[k] <= win_hi_host[k] && win_hi_host[k] <= M,
                   "window %d = [%d, %d) outside [0, M]", k, win_lo_host[k], win_hi_host[k]);
    for (int64_t j = 0; j < M; ++j)
        LK_REQUIRE(klo_host[j] >= 0 && khi_host[j] < K + (K == 0) && klo_host[j] <= khi_host[j] + 1,
                   "frequency %lld lists windows [%d, %d] outside [0, K)", (long long)j, klo_host[j], khi_host[j]);
    h->ws.reset();
    int rc = h->ws.reserve((size_t)(2 * K + 2 * M) * 4 + (size_t)B * (size_t)(K + 1) * 8 + 4096);
    if (rc) return rc;
    int *d_lo = (int *)h->ws.alloc((size_t)(K + 1) * 4), *d_hi = (int *)h->ws.alloc((size_t)(K + 1) * 4);
    int *d_klo = (int *)h->ws.alloc((size_t)M * 4), *d_khi = (int *)h->ws.alloc((size_t)M * 4);
    double *d_med = (double *)h->ws.alloc((size_t)B * (size_t)(K + 1) * 8);
    if (K) {
        LK_HIP_CHECK(hipMemcpyAsync(d_lo, win_lo_host, (size_t)K * 4, hipMemcpyHostToDevice, stream));
        LK_HIP_CHECK(hipMemcpyAsync(d_hi, win_hi_host, (size_t)K * 4, hipMemcpyHostToDevice, stream));
    }
    LK_HIP_CHECK(hipMemcpyAsync(d_klo, klo_host, (size_t)M * 4, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipMemcpyAsync(d_khi, khi_host, (size_t)M * 4, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));  // the host tables belong to the caller
    if (K)
        hipLaunchKernelGGL(pg_window_median_kernel, dim3((unsigned)K, (unsigned)B), dim3(256), 0, stream, power, M, d_lo,
                           d_hi, K, corr, d_med);
    hipLaunchKernelGGL(pg_window_average_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)B), dim3(256), 0, stream,
                       d_med, K, d_klo, d_khi, M, out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

int pg_boxsmooth_launch(lk_handle *h, int B, int64_t M, const double *power, const double *taps_host, int nk,
                        double *out, hipStream_t stream) {
    LK_REQUIRE(B >= 0 && M >= 1, "need B >= 0 and M >= 1");
    if (B == 0) return LK_OK;
    LK_REQUIRE(power && out && taps_host, "NULL buffer");
    LK_REQUIRE(nk >= 1 && nk % 2 == 1, "the kernel must have an odd number of taps");
    LK_REQUIRE(B <= 65535, "at most 65535 periodograms per call (got %d)", B);
    double ksum = 0.0;
    for (int i = 0; i < nk; ++i) ksum += taps_host[i];
    LK_REQUIRE(ksum > 1e-8, "The kernel can't be normalized, because its sum is close to zero.");
    h->ws.reset();
    int rc = h->ws.reserve((size_t)nk * 8 + (size_t)B * 4 + 4096);
    if (rc) return rc;
    double *d_taps = (double *)h->ws.alloc((size_t)nk * 8);
    int *d_flag = (int *)h->ws.alloc((size_t)B * 4);
    LK_HIP_CHECK(hipMemcpyAsync(d_taps, taps_host, (size_t)nk * 8, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));
    hipLaunchKernelGGL(pg_has_nan_kernel, dim3(B), dim3(256), 0, stream, power, M, d_flag);
    hipLaunchKernelGGL(pg_boxsmooth_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)B), dim3(256), 0, stream, power,
                       M, d_taps, nk, ksum, d_flag, out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ seismology 2-D ACF
// estimate_numax_acf2d (reference src/lightkurve/seismology/numax_estimators.py:15-205) slides a window of W = 2 spread
// samples over the spectrum and, for every central frequency, takes the full autocorrelation of the mean-subtracted
// window (seismology/utils.py:106-158: p_sel -= nanmean(p_sel); C = np.correlate(p_sel, p_sel, "full")[W - 1:]) and the
// "mean collapsed correlation" (sum |C| - 1) / W.  One workgroup per (window, periodogram): the window lives in LDS, a
// thread owns four consecutive lags and slides a four-value register window over the samples (one broadcast read and one
// new sample per four FMAs).  W^2 / 2 MACs per window, everything on chip; HBM traffic is W in and W out per window.
__global__ __launch_bounds__(256) void pg_acf2d_kernel(const double *__restrict__ power, int64_t M,
                                                        const int *__restrict__ win_start, int n_win, int W,
                                                        double *__restrict__ acf2d, double *__restrict__ metric) {
    extern __shared__ __attribute__((aligned(16))) double acf_lds[];  // W + 8 samples | 8 doubles of reduction scratch
    double *p = acf_lds, *red = acf_lds + W + 8;
    const int w = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const double *src = power + (size_t)b * (size_t)M + win_start[w];
    double s = 0.0;
    long long c = 0;
    for (int i = tid; i < W; i += 256) {
        const double v = src[i];
        p[i] = v;
        if (!isnan(v)) {
            s += v;
            ++c;
        }
    }
    if (tid < 8) p[W + tid] = 0.0;  // the register window runs up to three samples past the end
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        c += __shfl_xor(c, o);
    }
    __syncthreads();
    if ((tid & 63) == 0) {
        red[tid >> 6] = s;
        red[4 + (tid >> 6)] = (double)c;
    }
    __syncthreads();
    const double mean = (red[0] + red[1] + red[2] + red[3]) / (red[4] + red[5] + red[6] + red[7]);
    __syncthreads();
    for (int i = tid; i < W; i += 256) p[i] -= mean;  // NaN samples stay NaN (and poison every lag, as in numpy)
    __syncthreads();
    double *out = acf2d + ((size_t)b * n_win + w) * (size_t)W;
    double msum = 0.0;
    // lags l0 .. l0 + 3; groups are dealt from both ends (short and long lags alternate) to balance the triangle
    const int ngrp = (W + 3) / 4;
    for (int g = tid; g < ngrp; g += 256) {
        const int gg = (g & 1) ? (ngrp - 1 - (g >> 1)) : (g >> 1);
        const int l0 = 4 * gg;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        double q0 = p[l0], q1 = p[l0 + 1], q2 = p[l0 + 2], q3 = p[l0 + 3];
        // C[l] = sum_{i < W - l} p[i] p[i + l]; terms past the window's end multiply the zero pad
        const int n_i = W - l0;
        int i = 0;
        for (; i < n_i - 3; ++i) {  // all four lags have a partner inside the window
            const double x = p[i];
            a0 = fma(x, q0, a0);
            a1 = fma(x, q1, a1);
            a2 = fma(x, q2, a2);
            a3 = fma(x, q3, a3);
            q0 = q1;
            q1 = q2;
            q2 = q3;
            q3 = p[i + l0 + 4];
        }
        for (; i < n_i; ++i) {  // last three samples: lag l0 + r only pairs samples i < W - l0 - r (no 0 * NaN terms)
            const double x = p[i];
            a0 = fma(x, q0, a0);
            if (i < n_i - 1) a1 = fma(x, q1, a1);
            if (i < n_i - 2) a2 = fma(x, q2, a2);
            q0 = q1;
            q1 = q2;
            q2 = q3;
            q3 = 0.0;
        }
        if (l0 < W) { out[l0] = a0; msum += fabs(a0); }
        if (l0 + 1 < W) { out[l0 + 1] = a1; msum += fabs(a1); }
        if (l0 + 2 < W) { out[l0 + 2] = a2; msum += fabs(a2); }
        if (l0 + 3 < W) { out[l0 + 3] = a3; msum += fabs(a3); }
    }
    for (int o = 32; o > 0; o >>= 1) msum += __shfl_xor(msum, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = msum;
    __syncthreads();
    if (tid == 0) metric[(size_t)b * n_win + w] = ((red[0] + red[1] + red[2] + red[3]) - 1.0) / (double)W;
}

int pg_acf2d_launch(lk_handle *h, int B, int64_t M, const double *power, int n_win, const int *win_start_host, int W,
                    double *acf2d, double *metric, hipStream_t stream) {
    LK_REQUIRE(B >= 0 && M >= 1 && n_win >= 0, "need B >= 0, M >= 1, n_win >= 0");
    if (B == 0 || n_win == 0) return LK_OK;
    LK_REQUIRE(power && win_start_host && acf2d && metric, "NULL buffer");
    LK_REQUIRE(W >= 1 && W <= 16384, "window of %d samples outside 1..16384", W);
    LK_REQUIRE(B <= 65535, "at most 65535 periodograms per call");
    for (int k = 0; k < n_win; ++k)
        LK_REQUIRE(win_start_host[k] >= 0 && (int64_t)win_start_host[k] + W <= M, "window %d = [%d, %d) outside [0, M)", k,
                   win_start_host[k], win_start_host[k] + W);
    h->ws.reset();
    int rc = h->ws.reserve((size_t)n_win * 4 + 4096);
    if (rc) return rc;
    int *d_start = (int *)h->ws.alloc((size_t)n_win * 4);
    LK_HIP_CHECK(hipMemcpyAsync(d_start, win_start_host, (size_t)n_win * 4, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipStreamSynchronize(stream));
    {
        const int rc_ = want_lds(h, reinterpret_cast<const void *>(pg_acf2d_kernel), 160 * 1024);
        if (rc_) return rc_;
    }
    hipLaunchKernelGGL(pg_acf2d_kernel, dim3((unsigned)n_win, (unsigned)B), dim3(256), (size_t)(W + 16) * 8, stream, power,
                       M, d_start, n_win, W, acf2d, metric);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
