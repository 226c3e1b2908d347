// capi.hip — extern "C" boundary of liblkhip.so (declared in include/lkhip.h).
// Host-pointer entry points stage caller buffers into HBM, call the device-pointer entry points and copy
// the results back; the device-pointer entry points only validate, carve scratch and enqueue kernels.
#include "lk_common.hpp"

namespace lk {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

int Arena::reserve(size_t bytes) {
    if (bytes <= cap) return LK_OK;
    if (used != 0) {
        set_error("Arena::reserve while sub-allocations are live");
        return LK_EHIP;
    }
    if (base) (void)hipFree(base);
    base = nullptr;
    cap = 0;
    size_t want = bytes + (bytes >> 3) + 4096;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&base), want);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        base = nullptr;
        return e == hipErrorOutOfMemory ? LK_ENOMEM : LK_EHIP;
    }
    cap = want;
    return LK_OK;
}

int HostStage::copy(void *dst, const void *src, size_t bytes, hipStream_t stream) {
    const int s = next;
    next = (next + 1) % SLOTS;
    if (ev[s]) LK_HIP_CHECK(hipEventSynchronize(ev[s]));  // the copy that last used this slot has finished
    if (cap[s] < bytes) {
        if (buf[s]) (void)hipHostFree(buf[s]);
        buf[s] = nullptr;
        cap[s] = 0;
        LK_HIP_CHECK(hipHostMalloc(&buf[s], bytes + 4096, hipHostMallocDefault));
        cap[s] = bytes + 4096;
    }
    if (!ev[s]) LK_HIP_CHECK(hipEventCreateWithFlags(&ev[s], hipEventDisableTiming));
    memcpy(buf[s], src, bytes);
    LK_HIP_CHECK(hipMemcpyAsync(dst, buf[s], bytes, hipMemcpyHostToDevice, stream));
    LK_HIP_CHECK(hipEventRecord(ev[s], stream));
    return LK_OK;
}

void HostStage::release() {
    for (int s = 0; s < SLOTS; ++s) {
        if (ev[s]) {
            (void)hipEventSynchronize(ev[s]);
            (void)hipEventDestroy(ev[s]);
        }
        if (buf[s]) (void)hipHostFree(buf[s]);
        buf[s] = nullptr;
        ev[s] = nullptr;
        cap[s] = 0;
    }
}

void Arena::release() {
    if (base) (void)hipFree(base);
    base = nullptr;
    cap = used = 0;
}

}  // namespace lk

using lk::set_error;

extern "C" {

int lk_version(void) { return 100; }

const char *lk_last_error(void) { return lk::g_err.c_str(); }

int lk_device_count(int *count) {
    LK_REQUIRE(count != nullptr, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return LK_EHIP;
    }
    *count = n;
    return LK_OK;
}

int lk_init(int device_id, lk_handle **out) {
    LK_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    int n = 0;
    LK_HIP_CHECK(hipGetDeviceCount(&n));
    LK_REQUIRE(device_id >= 0 && device_id < n, "device_id %d out of range (have %d devices)", device_id, n);
    LK_HIP_CHECK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    LK_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    lk_handle *h = new (std::nothrow) lk_handle();
    if (!h) return LK_ENOMEM;
    h->device = device_id;
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // the library's ONE environment knob, read once per handle: the chunk size of the pinned host pipeline
    // (lk_set_host_chunk_mb changes it afterwards)
    if (const char *e = getenv("LK_HOST_CHUNK_MB")) h->host_chunk_mb = std::max(1, atoi(e));
    *out = h;
    return LK_OK;
}

int lk_set_host_chunk_mb(lk_handle *h, int mb) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(mb >= 1, "chunk size must be >= 1 MiB");
    h->host_chunk_mb = mb;
    return LK_OK;
}

int lk_bls_set_ordered_histogram(lk_handle *h, int on) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    h->bls_force_serial_hist = on ? 1 : 0;
    return LK_OK;
}

int lk_pld_set_eig_mode(lk_handle *h, int mode) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (one kernel per matrix) or 1 (phase-split launches)");
    h->pld_eig_split = mode;
    return LK_OK;
}

int lk_pld_set_eig_tolerance(lk_handle *h, double tol) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(tol >= 0.0 && tol < 1.0, "tolerance must be in [0, 1) (0 = the default)");
    h->pld_eig_tol = tol;
    return LK_OK;
}

void lk_destroy(lk_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    h->stage.release();
    for (hipStream_t st : {h->s_in, h->s_comp, h->s_out})
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t *arr : {h->ev_in, h->ev_comp, h->ev_out})
        for (int i = 0; i < 2; ++i)
            if (arr[i]) (void)hipEventDestroy(arr[i]);
    if (h->h_plan) (void)hipHostFree(h->h_plan);
    if (h->s_probe) (void)hipStreamDestroy(h->s_probe);
    for (int a = 0; a < 3; ++a) {
        if (h->s_ls_aux[a]) (void)hipStreamDestroy(h->s_ls_aux[a]);
        if (h->ev_ls_join[a]) (void)hipEventDestroy(h->ev_ls_join[a]);
    }
    if (h->ev_ls_fork) (void)hipEventDestroy(h->ev_ls_fork);
    if (h->clk_buf) (void)hipHostFree(h->clk_buf);
    if (h->flat_tab_dev) (void)hipFree(h->flat_tab_dev);
    h->ws.release();
    h->staging.release();
    delete h;
}

int lk_synchronize(lk_handle *h) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    LK_HIP_CHECK(hipDeviceSynchronize());
    return LK_OK;
}

int64_t lk_workspace_bytes(const lk_handle *h) { return h ? (int64_t)(h->ws.cap + h->staging.cap) : 0; }

// ------------------------------------------------------------------------------------------------ LS
int lk_ls_chi2_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                         const double *dy, const double *freq, double f0, double df, int64_t M, int nterms,
                         int fit_mean, int center_data, int normalization, const double *scale, double *power,
                         void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::ls_chi2_launch(h, B, n_off_host, t, y, dy, freq, f0, df, M, nterms, fit_mean, center_data,
                              normalization, scale, power, static_cast<hipStream_t>(stream));
}

int lk_ls_power_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                          const double *dy, const double *freq, double f0, double df, int64_t M, int fit_mean,
                          int center_data, int normalization, const double *scale, double *power, void *stream) {
    return lk_ls_chi2_batch_dev(h, B, n_off_host, t, y, dy, freq, f0, df, M, 1, fit_mean, center_data, normalization,
                                scale, power, stream);
}

int lk_ls_chi2_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y, const double *dy,
                     const double *freq, double f0, double df, int64_t M, int nterms, int fit_mean, int center_data,
                     int normalization, const double *scale, double *power) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && n_off != nullptr, "bad batch description");
    LK_REQUIRE(M >= 0, "M must be >= 0");
    if (B == 0 || M == 0) return LK_OK;
    LK_REQUIRE(t && y && power, "t, y, power must be non-NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ntot = (size_t)n_off[B];
    const size_t nb = ntot * sizeof(double), fb = freq ? (size_t)M * sizeof(double) : 0;
    const size_t pb = (size_t)B * (size_t)M * sizeof(double), sb = scale ? (size_t)B * sizeof(double) : 0;
    h->staging.reset();
    int rc = h->staging.reserve(3 * (nb + 256) + fb + pb + sb + 4096);
    if (rc) return rc;
    double *dt = (double *)h->staging.alloc(nb), *dyv = (double *)h->staging.alloc(nb);
    double *ddy = dy ? (double *)h->staging.alloc(nb) : nullptr;
    double *dfreq = freq ? (double *)h->staging.alloc(fb) : nullptr;
    double *dscale = scale ? (double *)h->staging.alloc(sb) : nullptr;
    double *dpow = (double *)h->staging.alloc(pb);
    LK_HIP_CHECK(hipMemcpy(dt, t, nb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dyv, y, nb, hipMemcpyHostToDevice));
    if (dy) LK_HIP_CHECK(hipMemcpy(ddy, dy, nb, hipMemcpyHostToDevice));
    if (freq) LK_HIP_CHECK(hipMemcpy(dfreq, freq, fb, hipMemcpyHostToDevice));
    if (scale) LK_HIP_CHECK(hipMemcpy(dscale, scale, sb, hipMemcpyHostToDevice));
    rc = lk::ls_chi2_launch(h, B, n_off, dt, dyv, ddy, dfreq, f0, df, M, nterms, fit_mean, center_data, normalization,
                            dscale, dpow, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(power, dpow, pb, hipMemcpyDeviceToHost));  // null-stream copy orders after the kernels
    return LK_OK;
}

int lk_ls_power_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y, const double *dy,
                      const double *freq, double f0, double df, int64_t M, int fit_mean, int center_data,
                      int normalization, const double *scale, double *power) {
    return lk_ls_chi2_batch(h, B, n_off, t, y, dy, freq, f0, df, M, 1, fit_mean, center_data, normalization, scale,
                            power);
}

// ------------------------------------------------------------------------------------------------ fold
int lk_fold_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *period,
                      const double *epoch_time, double epoch_phase, const double *wrap_phase, int normalize_phase,
                      int ncols, const double *const *cols_in, double *const *cols_out, double *phase, int64_t *order,
                      void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::fold_launch(h, B, n_off_host, t, period, epoch_time, epoch_phase, wrap_phase, normalize_phase, ncols,
                           cols_in, cols_out, phase, order, static_cast<hipStream_t>(stream));
}

int lk_fold_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *period,
                  const double *epoch_time, double epoch_phase, const double *wrap_phase, int normalize_phase,
                  int ncols, const double *const *cols_in, double *const *cols_out, double *phase, int64_t *order) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && n_off != nullptr, "bad batch description");
    if (B == 0) return LK_OK;
    LK_REQUIRE(t && phase && order, "NULL buffer");
    LK_REQUIRE(ncols >= 0 && ncols <= 16 && (ncols == 0 || (cols_in && cols_out)), "bad column list (at most 16 columns)");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ntot = (size_t)n_off[B], nb = ntot * 8;
    h->staging.reset();
    int rc = h->staging.reserve((size_t)(3 + 2 * ncols) * (nb + 256) + 4096);
    if (rc) return rc;
    double *dt = (double *)h->staging.alloc(nb), *dph = (double *)h->staging.alloc(nb);
    int64_t *dord = (int64_t *)h->staging.alloc(nb);
    const double *din[16];
    double *dout[16];
    LK_HIP_CHECK(hipMemcpy(dt, t, nb, hipMemcpyHostToDevice));
    for (int c = 0; c < ncols; ++c) {
        LK_REQUIRE(cols_in[c] && cols_out[c], "column %d is NULL", c);
        double *a = (double *)h->staging.alloc(nb);
        dout[c] = (double *)h->staging.alloc(nb);
        LK_HIP_CHECK(hipMemcpy(a, cols_in[c], nb, hipMemcpyHostToDevice));
        din[c] = a;
    }
    rc = lk::fold_launch(h, B, n_off, dt, period, epoch_time, epoch_phase, wrap_phase, normalize_phase, ncols, din, dout,
                         dph, dord, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(phase, dph, nb, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(order, dord, nb, hipMemcpyDeviceToHost));
    for (int c = 0; c < ncols; ++c) LK_HIP_CHECK(hipMemcpy(cols_out[c], dout[c], nb, hipMemcpyDeviceToHost));
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ Periodogram.smooth
int lk_pg_logmedian_batch_dev(lk_handle *h, int B, int64_t M, const double *power, int K, const int32_t *win_lo,
                              const int32_t *win_hi, const int32_t *klo, const int32_t *khi, double corr, double *out,
                              void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::pg_logmedian_launch(h, B, M, power, K, win_lo, win_hi, klo, khi, corr, out,
                                   static_cast<hipStream_t>(stream));
}

int lk_pg_boxsmooth_batch_dev(lk_handle *h, int B, int64_t M, const double *power, const double *taps, int nk,
                              double *out, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::pg_boxsmooth_launch(h, B, M, power, taps, nk, out, static_cast<hipStream_t>(stream));
}

// host-pointer flavours: stage power in, run, copy the smoothed rows back
extern "C++" {
template <class Launch>
static int pg_host_stage(lk_handle *h, int B, int64_t M, const double *power, double *out, Launch launch) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && M >= 1, "need B >= 0 and M >= 1");
    if (B == 0) return LK_OK;
    LK_REQUIRE(power && out, "NULL buffer");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t pb = (size_t)B * (size_t)M * sizeof(double);
    h->staging.reset();
    int rc = h->staging.reserve(2 * (pb + 256) + 4096);
    if (rc) return rc;
    double *dp = (double *)h->staging.alloc(pb), *dout = (double *)h->staging.alloc(pb);
    LK_HIP_CHECK(hipMemcpy(dp, power, pb, hipMemcpyHostToDevice));
    rc = launch(dp, dout);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(out, dout, pb, hipMemcpyDeviceToHost));
    return LK_OK;
}
}  // extern "C++"

int lk_pg_logmedian_batch(lk_handle *h, int B, int64_t M, const double *power, int K, const int32_t *win_lo,
                          const int32_t *win_hi, const int32_t *klo, const int32_t *khi, double corr, double *out) {
    return pg_host_stage(h, B, M, power, out, [&](const double *dp, double *dout) {
        return lk::pg_logmedian_launch(h, B, M, dp, K, win_lo, win_hi, klo, khi, corr, dout, nullptr);
    });
}

int lk_pg_boxsmooth_batch(lk_handle *h, int B, int64_t M, const double *power, const double *taps, int nk,
                          double *out) {
    return pg_host_stage(h, B, M, power, out, [&](const double *dp, double *dout) {
        return lk::pg_boxsmooth_launch(h, B, M, dp, taps, nk, dout, nullptr);
    });
}

int lk_pg_acf2d_batch_dev(lk_handle *h, int B, int64_t M, const double *power, int n_win, const int32_t *win_start,
                          int W, double *acf2d, double *metric, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::pg_acf2d_launch(h, B, M, power, n_win, win_start, W, acf2d, metric, static_cast<hipStream_t>(stream));
}

int lk_pg_acf2d_batch(lk_handle *h, int B, int64_t M, const double *power, int n_win, const int32_t *win_start, int W,
                      double *acf2d, double *metric) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && M >= 1 && n_win >= 0 && W >= 1, "bad shapes");
    if (B == 0 || n_win == 0) return LK_OK;
    LK_REQUIRE(power && acf2d && metric, "NULL buffer");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t pb = (size_t)B * (size_t)M * 8, ab = (size_t)B * n_win * (size_t)W * 8, mb = (size_t)B * n_win * 8;
    h->staging.reset();
    int rc = h->staging.reserve(pb + ab + mb + 3 * 256 + 4096);
    if (rc) return rc;
    double *dp = (double *)h->staging.alloc(pb), *da = (double *)h->staging.alloc(ab), *dm = (double *)h->staging.alloc(mb);
    LK_HIP_CHECK(hipMemcpy(dp, power, pb, hipMemcpyHostToDevice));
    rc = lk::pg_acf2d_launch(h, B, M, dp, n_win, win_start, W, da, dm, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(acf2d, da, ab, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(metric, dm, mb, hipMemcpyDeviceToHost));
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ LS 'fast'
int lk_ls_fast_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                         const double *dy, double f0, double df, int64_t M, int fit_mean, int center_data,
                         int normalization, const double *scale, int oversampling, double *power, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::lsfast_launch(h, B, n_off_host, t, y, dy, f0, df, M, fit_mean, center_data, normalization, scale,
                             oversampling, power, static_cast<hipStream_t>(stream));
}

int lk_ls_fastchi2_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                             const double *dy, double f0, double df, int64_t M, int nterms, int fit_mean,
                             int center_data, int normalization, const double *scale, int oversampling, double *power,
                             void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::lsfastchi2_launch(h, B, n_off_host, t, y, dy, f0, df, M, nterms, fit_mean, center_data, normalization,
                                 scale, oversampling, power, static_cast<hipStream_t>(stream));
}

int lk_ls_fastchi2_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y, const double *dy,
                         double f0, double df, int64_t M, int nterms, int fit_mean, int center_data, int normalization,
                         const double *scale, int oversampling, double *power) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && n_off != nullptr, "bad batch description");
    LK_REQUIRE(M >= 0, "M must be >= 0");
    if (B == 0 || M == 0) return LK_OK;
    LK_REQUIRE(t && y && power, "t, y, power must be non-NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ntot = (size_t)n_off[B], nb = ntot * 8;
    const size_t pb = (size_t)B * (size_t)M * 8, sb = scale ? (size_t)B * 8 : 0;
    h->staging.reset();
    int rc = h->staging.reserve(3 * (nb + 256) + pb + sb + 4096);
    if (rc) return rc;
    double *dt = (double *)h->staging.alloc(nb), *dyv = (double *)h->staging.alloc(nb);
    double *ddy = dy ? (double *)h->staging.alloc(nb) : nullptr;
    double *dscale = scale ? (double *)h->staging.alloc(sb) : nullptr;
    double *dpow = (double *)h->staging.alloc(pb);
    LK_HIP_CHECK(hipMemcpy(dt, t, nb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dyv, y, nb, hipMemcpyHostToDevice));
    if (dy) LK_HIP_CHECK(hipMemcpy(ddy, dy, nb, hipMemcpyHostToDevice));
    if (scale) LK_HIP_CHECK(hipMemcpy(dscale, scale, sb, hipMemcpyHostToDevice));
    rc = lk::lsfastchi2_launch(h, B, n_off, dt, dyv, ddy, f0, df, M, nterms, fit_mean, center_data, normalization, dscale,
                               oversampling, dpow, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(power, dpow, pb, hipMemcpyDeviceToHost));
    return LK_OK;
}

int lk_ls_fast_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y, const double *dy,
                     double f0, double df, int64_t M, int fit_mean, int center_data, int normalization,
                     const double *scale, int oversampling, double *power) {
    // chunked, double-buffered staging (lk_ls_fast_peaks_batch below); the multi-term path keeps the simple one
    return lk_ls_fast_peaks_batch(h, B, n_off, t, y, dy, f0, df, M, fit_mean, center_data, normalization, scale,
                                  oversampling, power, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------ argmax
int lk_argmax_batch_dev(lk_handle *h, int B, int64_t M, const double *x, double *max_out, int64_t *argmax_out,
                        void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::argmax_launch(h, B, M, x, max_out, argmax_out, static_cast<hipStream_t>(stream));
}

int lk_argmax_batch(lk_handle *h, int B, int64_t M, const double *x, double *max_out, int64_t *argmax_out) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && M >= 1, "need B >= 0 and M >= 1");
    if (B == 0) return LK_OK;
    LK_REQUIRE(x && max_out && argmax_out, "NULL buffer");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t xb = (size_t)B * (size_t)M * sizeof(double);
    h->staging.reset();
    int rc = h->staging.reserve(xb + (size_t)B * 16 + 4096);
    if (rc) return rc;
    double *dx = (double *)h->staging.alloc(xb), *dm = (double *)h->staging.alloc((size_t)B * 8);
    int64_t *da = (int64_t *)h->staging.alloc((size_t)B * 8);
    LK_HIP_CHECK(hipMemcpy(dx, x, xb, hipMemcpyHostToDevice));
    rc = lk::argmax_launch(h, B, M, dx, dm, da, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(max_out, dm, (size_t)B * 8, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(argmax_out, da, (size_t)B * 8, hipMemcpyDeviceToHost));
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ BLS
int lk_bls_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                     const double *ivar, const double *period_host, const double *period_dev, int64_t nP,
                     const double *duration_host, int nD, int oversample, int use_likelihood, double *out7,
                     void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::bls_launch(h, B, n_off_host, t, y, ivar, period_host, period_dev, nP, duration_host, nD, oversample,
                          use_likelihood, out7, static_cast<hipStream_t>(stream));
}

int lk_bls_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y, const double *ivar,
                 const double *period, int64_t nP, const double *duration, int nD, int oversample,
                 int use_likelihood, double *out7) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && n_off != nullptr, "bad batch description");
    LK_REQUIRE(nP >= 0 && nD >= 1, "need nP >= 0 and nD >= 1");
    if (B == 0 || nP == 0) return LK_OK;
    LK_REQUIRE(t && y && ivar && period && duration && out7, "NULL buffer");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ntot = (size_t)n_off[B], nb = ntot * sizeof(double);
    const size_t pb = (size_t)nP * sizeof(double), ob = 7 * (size_t)B * (size_t)nP * sizeof(double);
    h->staging.reset();
    int rc = h->staging.reserve(3 * (nb + 256) + pb + ob + 4096);
    if (rc) return rc;
    double *dt = (double *)h->staging.alloc(nb), *dyv = (double *)h->staging.alloc(nb);
    double *div = (double *)h->staging.alloc(nb), *dper = (double *)h->staging.alloc(pb);
    double *dout = (double *)h->staging.alloc(ob);
    LK_HIP_CHECK(hipMemcpy(dt, t, nb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dyv, y, nb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(div, ivar, nb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dper, period, pb, hipMemcpyHostToDevice));
    rc = lk::bls_launch(h, B, n_off, dt, dyv, div, period, dper, nP, duration, nD, oversample, use_likelihood, dout,
                        nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(out7, dout, ob, hipMemcpyDeviceToHost));
    return LK_OK;
}

int lk_bls_max_period(const double *duration, int nD, int oversample, double *max_period) {
    return lk::bls_max_period_host(duration, nD, oversample, max_period);
}

// ------------------------------------------------------------------------------------------------ regression
int lk_regress_cov_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, int K, const double *X, const double *y,
                             const double *err, const uint8_t *cadence_mask, const double *prior_mu,
                             const double *prior_sigma, double clip_sigma, int niters, double *w, double *model,
                             uint8_t *outlier, double *w_cov, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::regress_launch(h, B, n_off_host, K, X, y, err, cadence_mask, prior_mu, prior_sigma, clip_sigma,
                              niters, w, model, outlier, static_cast<hipStream_t>(stream), w_cov);
}

int lk_regress_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, int K, const double *X, const double *y,
                         const double *err, const uint8_t *cadence_mask, const double *prior_mu,
                         const double *prior_sigma, double clip_sigma, int niters, double *w, double *model,
                         uint8_t *outlier, void *stream) {
    return lk_regress_cov_batch_dev(h, B, n_off_host, K, X, y, err, cadence_mask, prior_mu, prior_sigma, clip_sigma,
                                    niters, w, model, outlier, nullptr, stream);
}

int lk_regress_batch(lk_handle *h, int B, const int64_t *n_off, int K, const double *X, const double *y,
                     const double *err, const uint8_t *cadence_mask, const double *prior_mu,
                     const double *prior_sigma, double clip_sigma, int niters, double *w, double *model,
                     uint8_t *outlier) {
    return lk_regress_cov_batch(h, B, n_off, K, X, y, err, cadence_mask, prior_mu, prior_sigma, clip_sigma, niters, w,
                                model, outlier, nullptr);
}

int lk_regress_cov_batch(lk_handle *h, int B, const int64_t *n_off, int K, const double *X, const double *y,
                         const double *err, const uint8_t *cadence_mask, const double *prior_mu,
                         const double *prior_sigma, double clip_sigma, int niters, double *w, double *model,
                         uint8_t *outlier, double *w_cov) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && n_off != nullptr, "bad batch description");
    if (B == 0) return LK_OK;
    LK_REQUIRE(K >= 1, "K must be >= 1");
    LK_REQUIRE(X && y && w && model && outlier, "NULL buffer");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ntot = (size_t)n_off[B];
    const size_t xb = ntot * (size_t)K * 8, nb = ntot * 8, kb = (size_t)B * K * 8;
    h->staging.reset();
    const size_t cb = w_cov ? (size_t)B * K * K * 8 : 0;
    int rc = h->staging.reserve(xb + 3 * (nb + 256) + 2 * (ntot + 256) + 3 * (kb + 256) + cb + 4096);
    if (rc) return rc;
    double *dX = (double *)h->staging.alloc(xb), *dy = (double *)h->staging.alloc(nb);
    double *derr = err ? (double *)h->staging.alloc(nb) : nullptr;
    double *dmodel = (double *)h->staging.alloc(nb);
    uint8_t *dcm = cadence_mask ? (uint8_t *)h->staging.alloc(ntot) : nullptr;
    uint8_t *dout = (uint8_t *)h->staging.alloc(ntot);
    double *dmu = prior_mu ? (double *)h->staging.alloc(kb) : nullptr;
    double *dsg = prior_sigma ? (double *)h->staging.alloc(kb) : nullptr;
    double *dw = (double *)h->staging.alloc(kb);
    double *dcov = w_cov ? (double *)h->staging.alloc(cb) : nullptr;
    LK_HIP_CHECK(hipMemcpy(dX, X, xb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dy, y, nb, hipMemcpyHostToDevice));
    if (err) LK_HIP_CHECK(hipMemcpy(derr, err, nb, hipMemcpyHostToDevice));
    if (cadence_mask) LK_HIP_CHECK(hipMemcpy(dcm, cadence_mask, ntot, hipMemcpyHostToDevice));
    if (prior_mu) LK_HIP_CHECK(hipMemcpy(dmu, prior_mu, kb, hipMemcpyHostToDevice));
    if (prior_sigma) LK_HIP_CHECK(hipMemcpy(dsg, prior_sigma, kb, hipMemcpyHostToDevice));
    rc = lk::regress_launch(h, B, n_off, K, dX, dy, derr, dcm, dmu, dsg, clip_sigma, niters, dw, dmodel, dout,
                            nullptr, dcov);
    if (rc) return rc;
    if (w_cov) LK_HIP_CHECK(hipMemcpy(w_cov, dcov, cb, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(w, dw, kb, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(model, dmodel, nb, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(outlier, dout, ntot, hipMemcpyDeviceToHost));
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ flatten
int lk_savgol_design(int window, int polyorder, double *coeffs, double *edge) {
    return lk::savgol_design_host(window, polyorder, coeffs, edge);
}

int lk_savgol_trend_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux,
                              const uint8_t *mask, int window, int polyorder, double break_tol, int niters,
                              double sigma, double *trend, uint8_t *fit_mask, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::flatten_launch(h, B, n_off_host, t, flux, mask, window, polyorder, break_tol, niters, sigma, trend,
                              fit_mask, static_cast<hipStream_t>(stream));
}

int lk_savgol_trend_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *flux,
                          const uint8_t *mask, int window, int polyorder, double break_tol, int niters, double sigma,
                          double *trend, uint8_t *fit_mask) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && n_off != nullptr, "bad batch description");
    if (B == 0) return LK_OK;
    LK_REQUIRE(t && flux && trend, "NULL buffer");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ntot = (size_t)n_off[B], nb = ntot * 8;
    h->staging.reset();
    int rc = h->staging.reserve(3 * (nb + 256) + 2 * (ntot + 256) + 4096);
    if (rc) return rc;
    double *dt = (double *)h->staging.alloc(nb), *df = (double *)h->staging.alloc(nb);
    double *dtr = (double *)h->staging.alloc(nb);
    uint8_t *dm = mask ? (uint8_t *)h->staging.alloc(ntot) : nullptr;
    uint8_t *dfm = fit_mask ? (uint8_t *)h->staging.alloc(ntot) : nullptr;
    LK_HIP_CHECK(hipMemcpy(dt, t, nb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(df, flux, nb, hipMemcpyHostToDevice));
    if (mask) LK_HIP_CHECK(hipMemcpy(dm, mask, ntot, hipMemcpyHostToDevice));
    rc = lk::flatten_launch(h, B, n_off, dt, df, dm, window, polyorder, break_tol, niters, sigma, dtr, dfm, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(trend, dtr, nb, hipMemcpyDeviceToHost));
    if (fit_mask) LK_HIP_CHECK(hipMemcpy(fit_mask, dfm, ntot, hipMemcpyDeviceToHost));
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ PLD design matrix
int lk_pld_design_width(int P, int Pb, int pld_order, int pca_components, int n_knots) {
    return lk::pld_design_width(P, Pb, pld_order, pca_components, n_knots);
}

int lk_pld_design_batch_dev(lk_handle *h, int B, int N, int P, int Pb, const float *pld_pix, const float *bkg_pix,
                            const float *lc_flux, const double *time, const double *knots, int n_inner, int pld_order,
                            int pca_components, int n_knots, int spline_degree, int normalize_bkg, int K, double *X,
                            double *prior_sigma, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::pld_design_launch(h, B, N, P, Pb, pld_pix, bkg_pix, lc_flux, time, knots, n_inner, pld_order,
                                 pca_components, n_knots, spline_degree, normalize_bkg, K, X, prior_sigma,
                                 static_cast<hipStream_t>(stream));
}

int lk_pld_design_batch(lk_handle *h, int B, int N, int P, int Pb, const float *pld_pix, const float *bkg_pix,
                        const float *lc_flux, const double *time, const double *knots, int n_inner, int pld_order,
                        int pca_components, int n_knots, int spline_degree, int normalize_bkg, int K, double *X,
                        double *prior_sigma) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 1 && N >= 2 && P >= 0 && Pb >= 1 && K >= 1, "bad shapes");
    LK_REQUIRE(bkg_pix && lc_flux && time && knots && X && prior_sigma, "NULL buffer");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t bn = (size_t)B * N;
    const size_t pb = bn * P * 4, bb = bn * Pb * 4, lb = bn * 4, tb = bn * 8, kb = (size_t)B * (n_inner + 2) * 8;
    const size_t xb = bn * K * 8, sb = (size_t)B * K * 8;
    h->staging.reset();
    int rc = h->staging.reserve(pb + bb + lb + tb + kb + xb + sb + 8 * 256 + 4096);
    if (rc) return rc;
    float *dp = (P > 0 && pld_pix) ? (float *)h->staging.alloc(pb) : nullptr;
    float *db = (float *)h->staging.alloc(bb), *dl = (float *)h->staging.alloc(lb);
    double *dt = (double *)h->staging.alloc(tb), *dk = (double *)h->staging.alloc(kb);
    double *dX = (double *)h->staging.alloc(xb), *ds = (double *)h->staging.alloc(sb);
    if (dp) LK_HIP_CHECK(hipMemcpy(dp, pld_pix, pb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(db, bkg_pix, bb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dl, lc_flux, lb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dt, time, tb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dk, knots, kb, hipMemcpyHostToDevice));
    rc = lk::pld_design_launch(h, B, N, dp ? P : 0, Pb, dp, db, dl, dt, dk, n_inner, pld_order, pca_components, n_knots,
                               spline_degree, normalize_bkg, K, dX, ds, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(X, dX, xb, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(prior_sigma, ds, sb, hipMemcpyDeviceToHost));
    return LK_OK;
}

// PLDCorrector.correct for B same-shaped cutouts, host pointers in and out: design matrices, regression + clip loop and the
// spline block's share of the model in one call — X (B x N x K doubles, 1.7 GB per 500 K2 cutouts) never leaves HBM.
int lk_pld_correct_batch(lk_handle *h, int B, int N, int P, int Pb, const float *pld_pix, const float *bkg_pix,
                         const float *lc_flux, const double *time, const double *knots, int n_inner, int pld_order,
                         int pca_components, int n_knots, int spline_degree, int normalize_bkg, int K, const double *y,
                         const double *err, const uint8_t *cadence_mask, double clip_sigma, int niters, double *w,
                         double *model, uint8_t *outlier, double *spline_part) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 1 && N >= 2 && P >= 0 && Pb >= 1 && K >= 1, "bad shapes");
    LK_REQUIRE(bkg_pix && lc_flux && time && knots && y && w && model && outlier, "NULL buffer");
    LK_REQUIRE(n_knots + 1 <= K, "K=%d is narrower than the spline block (%d columns)", K, n_knots + 1);
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t bn = (size_t)B * N;
    const bool has_pld = P > 0 && pld_pix != nullptr;
    const bool shared = has_pld && pld_pix == bkg_pix && P == Pb;  // one aperture for both blocks: one upload
    const size_t pb = bn * P * 4, bb = bn * Pb * 4, lb = bn * 4, nb = bn * 8, kb = (size_t)B * (n_inner + 2) * 8;
    const size_t xb = bn * K * 8, sb = (size_t)B * K * 8;
    h->staging.reset();
    int rc = h->staging.reserve(pb + bb + lb + kb + xb + 3 * (sb + 256) + 5 * (nb + 256) + 2 * (bn + 256) + 8 * 256 + 4096);
    if (rc) return rc;
    float *dp = (has_pld && !shared) ? (float *)h->staging.alloc(pb) : nullptr;
    float *db = (float *)h->staging.alloc(bb), *dl = (float *)h->staging.alloc(lb);
    double *dt = (double *)h->staging.alloc(nb), *dk = (double *)h->staging.alloc(kb);
    double *dX = (double *)h->staging.alloc(xb), *ds = (double *)h->staging.alloc(sb), *dmu = (double *)h->staging.alloc(sb);
    double *dw = (double *)h->staging.alloc(sb), *dy = (double *)h->staging.alloc(nb);
    double *derr = err ? (double *)h->staging.alloc(nb) : nullptr;
    double *dmodel = (double *)h->staging.alloc(nb), *dsp = spline_part ? (double *)h->staging.alloc(nb) : nullptr;
    uint8_t *dcm = cadence_mask ? (uint8_t *)h->staging.alloc(bn) : nullptr, *dout = (uint8_t *)h->staging.alloc(bn);
    LK_REQUIRE(dout != nullptr, "staging arena exhausted");
    if (dp) LK_HIP_CHECK(hipMemcpy(dp, pld_pix, pb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(db, bkg_pix, bb, hipMemcpyHostToDevice));
    if (shared) dp = db;
    LK_HIP_CHECK(hipMemcpy(dl, lc_flux, lb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dt, time, nb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dk, knots, kb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dy, y, nb, hipMemcpyHostToDevice));
    if (err) LK_HIP_CHECK(hipMemcpy(derr, err, nb, hipMemcpyHostToDevice));
    if (cadence_mask) LK_HIP_CHECK(hipMemcpy(dcm, cadence_mask, bn, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemsetAsync(dmu, 0, sb, nullptr));   // prior_mu = 0 for every PLD column (pldcorrector.py:240-287)
    rc = lk::pld_design_launch(h, B, N, dp ? P : 0, Pb, dp, db, dl, dt, dk, n_inner, pld_order, pca_components, n_knots,
                               spline_degree, normalize_bkg, K, dX, ds, nullptr);
    if (rc) return rc;
    std::vector<int64_t> off((size_t)B + 1);
    for (int b = 0; b <= B; ++b) off[b] = (int64_t)b * N;
    rc = lk::regress_launch(h, B, off.data(), K, dX, dy, derr, dcm, dmu, ds, clip_sigma, niters, dw, dmodel, dout, nullptr);
    if (rc) return rc;
    if (dsp) {
        rc = lk::model_part_launch(h, B, N, K, K - (n_knots + 1), K, dX, dw, dsp, nullptr);
        if (rc) return rc;
    }
    LK_HIP_CHECK(hipMemcpy(w, dw, sb, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(model, dmodel, nb, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(outlier, dout, bn, hipMemcpyDeviceToHost));
    if (dsp) LK_HIP_CHECK(hipMemcpy(spline_part, dsp, nb, hipMemcpyDeviceToHost));
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ design-matrix operations
int lk_pca_batch_dev(lk_handle *h, int B, int N, int P, int k, const double *A, double *U, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::dm_pca_launch(h, B, N, P, k, A, U, static_cast<hipStream_t>(stream));
}

int lk_pca_batch(lk_handle *h, int B, int N, int P, int k, const double *A, double *U) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 1 && N >= 1 && P >= 1 && k >= 1 && A && U, "bad arguments");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ab = (size_t)B * N * P * 8, ub = (size_t)B * N * k * 8;
    h->staging.reset();
    int rc = h->staging.reserve(ab + ub + 1024);
    if (rc) return rc;
    double *dA = (double *)h->staging.alloc(ab), *dU = (double *)h->staging.alloc(ub);
    LK_HIP_CHECK(hipMemcpy(dA, A, ab, hipMemcpyHostToDevice));
    rc = lk::dm_pca_launch(h, B, N, P, k, dA, dU, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(U, dU, ub, hipMemcpyDeviceToHost));
    return LK_OK;
}

int lk_spline_basis_batch_dev(lk_handle *h, int B, int N, const double *x, const double *knots, int n_inner, int degree,
                              double *out, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::dm_spline_launch(h, B, N, x, knots, n_inner, degree, out, static_cast<hipStream_t>(stream));
}

int lk_spline_basis_batch(lk_handle *h, int B, int N, const double *x, const double *knots, int n_inner, int degree,
                          double *out) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 1 && N >= 1 && n_inner >= 0 && degree >= 0 && x && knots && out, "bad arguments");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t xb = (size_t)B * N * 8, kb = (size_t)B * (n_inner + 2) * 8, ob = (size_t)B * N * (n_inner + degree + 1) * 8;
    h->staging.reset();
    int rc = h->staging.reserve(xb + kb + ob + 1024);
    if (rc) return rc;
    double *dx = (double *)h->staging.alloc(xb), *dk = (double *)h->staging.alloc(kb), *dout = (double *)h->staging.alloc(ob);
    LK_HIP_CHECK(hipMemcpy(dx, x, xb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(dk, knots, kb, hipMemcpyHostToDevice));
    rc = lk::dm_spline_launch(h, B, N, dx, dk, n_inner, degree, dout, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(out, dout, ob, hipMemcpyDeviceToHost));
    return LK_OK;
}

int lk_standardize_batch_dev(lk_handle *h, int B, int N, int P, const double *A, double *out, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::dm_standardize_launch(h, B, N, P, A, out, static_cast<hipStream_t>(stream));
}

int lk_standardize_batch(lk_handle *h, int B, int N, int P, const double *A, double *out) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 1 && N >= 1 && P >= 1 && A && out, "bad arguments");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ab = (size_t)B * N * P * 8;
    h->staging.reset();
    int rc = h->staging.reserve(2 * ab + 1024);
    if (rc) return rc;
    double *dA = (double *)h->staging.alloc(ab), *dO = (double *)h->staging.alloc(ab);
    LK_HIP_CHECK(hipMemcpy(dA, A, ab, hipMemcpyHostToDevice));
    rc = lk::dm_standardize_launch(h, B, N, P, dA, dO, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(out, dO, ab, hipMemcpyDeviceToHost));
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ pinned host memory
int lk_host_alloc(void **ptr, size_t bytes) {
    LK_REQUIRE(ptr != nullptr, "ptr is NULL");
    *ptr = nullptr;
    LK_HIP_CHECK(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return LK_OK;
}

int lk_host_free(void *ptr) {
    if (ptr) LK_HIP_CHECK(hipHostFree(ptr));
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ LS 'fast' + peaks
int lk_ls_fast_peaks_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                               const double *dy, double f0, double df, int64_t M, int fit_mean, int center_data,
                               int normalization, const double *scale, int oversampling, double *power,
                               double *max_power, int64_t *argmax, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(power != nullptr, "power must be non-NULL for the device flavour (the spectra stay in HBM anyway)");
    LK_HIP_CHECK(hipSetDevice(h->device));
    // the per-target (max power, argmax) come out of the fused FFT step itself (no second pass over the spectra)
    return lk::lsfast_launch(h, B, n_off_host, t, y, dy, f0, df, M, fit_mean, center_data, normalization, scale,
                             oversampling, power, static_cast<hipStream_t>(stream), max_power, argmax);
}

extern "C++" {
static int pipeline_init(lk_handle *h) {
    if (h->s_in) return LK_OK;
    LK_HIP_CHECK(hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
    LK_HIP_CHECK(hipStreamCreateWithFlags(&h->s_comp, hipStreamNonBlocking));
    LK_HIP_CHECK(hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        LK_HIP_CHECK(hipEventCreateWithFlags(&h->ev_in[i], hipEventDisableTiming));
        LK_HIP_CHECK(hipEventCreateWithFlags(&h->ev_comp[i], hipEventDisableTiming));
        LK_HIP_CHECK(hipEventCreateWithFlags(&h->ev_out[i], hipEventDisableTiming));
    }
    return LK_OK;
}
}  // extern "C++"

// Host-pointer flavour, software-pipelined over chunks of targets: the H2D copy of chunk k+1 (stream s_in), the
// kernels of chunk k (s_comp) and the D2H copy of chunk k-1 (s_out) overlap; both device buffers are double
// buffered and ordered by events.  Caller buffers that are pinned (lk_host_alloc / hipHostMalloc / hipHostRegister)
// are DMA'd directly and the whole loop is asynchronous; pageable buffers go through the runtime's own staging
// (the host blocks inside each copy, the kernels of the next chunk are already queued).  power may be NULL
// (peaks only: the 8 B x B x M of spectra never cross PCIe); max_power / argmax may both be NULL.
extern "C++" {
// astropy hands LombScargle times relative to the first cadence of each light curve (lombscargle/core.py:119-126: trel =
// t - t[0]); a packed batch holds absolute times.  One workgroup per target: every lane reads t[lo] BEFORE the barrier, so
// the in-place subtraction is the same fp64 `t - t[0]` numpy does, 40 us for 160 MB instead of a host pass.
__global__ __launch_bounds__(1024) void lc_rebase_kernel(double *__restrict__ t, const int64_t *__restrict__ off,
                                                         int64_t base) {
    const int64_t lo = off[blockIdx.x] - base, n = off[blockIdx.x + 1] - off[blockIdx.x];
    if (n <= 0) return;
    const double t0 = t[lo];
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += 1024) t[lo + i] -= t0;
}

static int ls_fast_peaks_host(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y,
                              const double *dy, double f0, double df, int64_t M, int fit_mean, int center_data,
                              int normalization, const double *scale, int oversampling, double *power,
                              double *max_power, int64_t *argmax, bool rebase);
}  // extern "C++"

int lk_ls_fast_peaks_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y,
                           const double *dy, double f0, double df, int64_t M, int fit_mean, int center_data,
                           int normalization, const double *scale, int oversampling, double *power,
                           double *max_power, int64_t *argmax) {
    return ls_fast_peaks_host(h, B, n_off, t, y, dy, f0, df, M, fit_mean, center_data, normalization, scale, oversampling,
                              power, max_power, argmax, false);
}

int lk_ls_fast_peaks_lc_batch(lk_handle *h, int B, const int64_t *n_off, const double *time, const double *flux,
                              const double *dy, double f0, double df, int64_t M, int fit_mean, int center_data,
                              int normalization, const double *scale, int oversampling, double *power,
                              double *max_power, int64_t *argmax) {
    return ls_fast_peaks_host(h, B, n_off, time, flux, dy, f0, df, M, fit_mean, center_data, normalization, scale,
                              oversampling, power, max_power, argmax, true);
}

extern "C++" {
static int ls_fast_peaks_host(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y,
                              const double *dy, double f0, double df, int64_t M, int fit_mean, int center_data,
                              int normalization, const double *scale, int oversampling, double *power,
                              double *max_power, int64_t *argmax, bool rebase) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && n_off != nullptr, "bad batch description");
    LK_REQUIRE(M >= 0, "M must be >= 0");
    if (B == 0 || M == 0) return LK_OK;
    LK_REQUIRE(t && y, "t, y must be non-NULL");
    LK_REQUIRE(power || (max_power && argmax), "nothing to compute: power, max_power and argmax are all NULL");
    LK_REQUIRE((max_power == nullptr) == (argmax == nullptr), "max_power and argmax must both be given (or both NULL)");
    LK_REQUIRE(n_off[0] == 0, "n_off[0] must be 0");
    for (int b = 0; b < B; ++b) LK_REQUIRE(n_off[b + 1] >= n_off[b], "n_off must be non-decreasing");
    LK_HIP_CHECK(hipSetDevice(h->device));
    int rc = pipeline_init(h);
    if (rc) return rc;
    // chunk = as many targets as give ~64 MiB of spectra (large enough to amortise launches, small enough to pipeline)
    const size_t chunk_mb = (size_t)std::max(1, h->host_chunk_mb);
    const int C = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, (chunk_mb << 20) / ((size_t)M * 8)));
    const int nchunks = (B + C - 1) / C;
    size_t in_max = 0;
    for (int k = 0; k < nchunks; ++k) {
        const int b0 = k * C, b1 = std::min(B, b0 + C);
        in_max = std::max(in_max, (size_t)(n_off[b1] - n_off[b0]));
    }
    const int narr = dy ? 3 : 2;
    const size_t in_bytes = in_max * 8, pow_bytes = (size_t)C * (size_t)M * 8;
    h->staging.reset();
    rc = h->staging.reserve(2 * narr * (in_bytes + 256) + 2 * (pow_bytes + 256) + 4 * ((size_t)(B + 1) * 8 + 256) + 4096);
    if (rc) return rc;
    double *d_in[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    double *d_pow[2];
    for (int s = 0; s < 2; ++s) {
        for (int a = 0; a < narr; ++a) d_in[s][a] = (double *)h->staging.alloc(in_bytes);
        d_pow[s] = (double *)h->staging.alloc(pow_bytes);
    }
    double *d_scale = scale ? (double *)h->staging.alloc((size_t)B * 8) : nullptr;
    double *d_max = max_power ? (double *)h->staging.alloc((size_t)B * 8) : nullptr;
    int64_t *d_arg = argmax ? (int64_t *)h->staging.alloc((size_t)B * 8) : nullptr;
    int64_t *d_off_all = rebase ? (int64_t *)h->staging.alloc((size_t)(B + 1) * 8) : nullptr;
    if (scale) LK_HIP_CHECK(hipMemcpyAsync(d_scale, scale, (size_t)B * 8, hipMemcpyHostToDevice, h->s_in));
    if (rebase) LK_HIP_CHECK(hipMemcpyAsync(d_off_all, n_off, (size_t)(B + 1) * 8, hipMemcpyHostToDevice, h->s_in));
    std::vector<int64_t> offc((size_t)C + 1);

    auto enqueue_in = [&](int k) -> int {
        const int s = k & 1, b0 = k * C, b1 = std::min(B, b0 + C);
        const size_t lo = (size_t)n_off[b0], nbytes = (size_t)(n_off[b1] - n_off[b0]) * 8;
        if (k >= 2) LK_HIP_CHECK(hipStreamWaitEvent(h->s_in, h->ev_comp[s], 0));  // kernels of chunk k-2 are done with d_in[s]
        if (nbytes) {
            LK_HIP_CHECK(hipMemcpyAsync(d_in[s][0], t + lo, nbytes, hipMemcpyHostToDevice, h->s_in));
            LK_HIP_CHECK(hipMemcpyAsync(d_in[s][1], y + lo, nbytes, hipMemcpyHostToDevice, h->s_in));
            if (dy) LK_HIP_CHECK(hipMemcpyAsync(d_in[s][2], dy + lo, nbytes, hipMemcpyHostToDevice, h->s_in));
        }
        LK_HIP_CHECK(hipEventRecord(h->ev_in[s], h->s_in));
        return LK_OK;
    };
    auto enqueue_comp = [&](int k) -> int {
        const int s = k & 1, b0 = k * C, b1 = std::min(B, b0 + C), nb = b1 - b0;
        LK_HIP_CHECK(hipStreamWaitEvent(h->s_comp, h->ev_in[s], 0));
        if (k >= 2 && power) LK_HIP_CHECK(hipStreamWaitEvent(h->s_comp, h->ev_out[s], 0));  // D2H of chunk k-2 has left d_pow[s]
        for (int b = 0; b <= nb; ++b) offc[b] = n_off[b0 + b] - n_off[b0];
        if (rebase)  // ordered behind the copy of d_off_all by ev_in (both on s_in)
            hipLaunchKernelGGL(lc_rebase_kernel, dim3(nb), dim3(1024), 0, h->s_comp, d_in[s][0], d_off_all + b0, n_off[b0]);
        int r = lk::lsfast_launch(h, nb, offc.data(), d_in[s][0], d_in[s][1], dy ? d_in[s][2] : nullptr, f0, df, M,
                                  fit_mean, center_data, normalization, d_scale ? d_scale + b0 : nullptr, oversampling,
                                  d_pow[s], h->s_comp);
        if (r) return r;
        if (d_max) {
            r = lk::argmax_launch(h, nb, M, d_pow[s], d_max + b0, d_arg + b0, h->s_comp);
            if (r) return r;
        }
        LK_HIP_CHECK(hipEventRecord(h->ev_comp[s], h->s_comp));
        return LK_OK;
    };
    auto enqueue_out = [&](int k) -> int {
        if (!power) return LK_OK;
        const int s = k & 1, b0 = k * C, b1 = std::min(B, b0 + C);
        LK_HIP_CHECK(hipStreamWaitEvent(h->s_out, h->ev_comp[s], 0));
        LK_HIP_CHECK(hipMemcpyAsync(power + (size_t)b0 * (size_t)M, d_pow[s], (size_t)(b1 - b0) * (size_t)M * 8,
                                    hipMemcpyDeviceToHost, h->s_out));
        LK_HIP_CHECK(hipEventRecord(h->ev_out[s], h->s_out));
        return LK_OK;
    };
    // the pipeline: chunk k+1 is copied in and its kernels are queued BEFORE the copy-out of chunk k is issued, so a
    // blocking (pageable) copy-out never leaves the GPU without queued work
    if ((rc = enqueue_in(0)) || (rc = enqueue_comp(0))) goto fail;
    for (int k = 0; k < nchunks; ++k) {
        if (k + 1 < nchunks && ((rc = enqueue_in(k + 1)) || (rc = enqueue_comp(k + 1)))) goto fail;
        if ((rc = enqueue_out(k))) goto fail;
    }
    if (d_max) {
        LK_HIP_CHECK(hipMemcpyAsync(max_power, d_max, (size_t)B * 8, hipMemcpyDeviceToHost, h->s_comp));
        LK_HIP_CHECK(hipMemcpyAsync(argmax, d_arg, (size_t)B * 8, hipMemcpyDeviceToHost, h->s_comp));
    }
    LK_HIP_CHECK(hipStreamSynchronize(h->s_in));
    LK_HIP_CHECK(hipStreamSynchronize(h->s_comp));
    LK_HIP_CHECK(hipStreamSynchronize(h->s_out));
    return LK_OK;
fail:
    (void)hipStreamSynchronize(h->s_in);
    (void)hipStreamSynchronize(h->s_comp);
    (void)hipStreamSynchronize(h->s_out);
    return rc;
}
}  // extern "C++"

// ------------------------------------------------------------------------------------------------ sigma clip
int lk_sigma_clip_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *y, double sigma, int maxiters,
                            uint8_t *outlier, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::sigma_clip_launch(h, B, n_off_host, y, sigma, maxiters, outlier, static_cast<hipStream_t>(stream));
}

int lk_sigma_clip_batch(lk_handle *h, int B, const int64_t *n_off, const double *y, double sigma, int maxiters,
                        uint8_t *outlier) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && n_off != nullptr, "bad batch description");
    if (B == 0 || n_off[B] == 0) return LK_OK;
    LK_REQUIRE(y && outlier, "NULL buffer");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ntot = (size_t)n_off[B];
    h->staging.reset();
    int rc = h->staging.reserve(ntot * 9 + 2 * 256 + 4096);
    if (rc) return rc;
    double *dy = (double *)h->staging.alloc(ntot * 8);
    uint8_t *dm = (uint8_t *)h->staging.alloc(ntot);
    LK_HIP_CHECK(hipMemcpy(dy, y, ntot * 8, hipMemcpyHostToDevice));
    rc = lk::sigma_clip_launch(h, B, n_off, dy, sigma, maxiters, dm, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(outlier, dm, ntot, hipMemcpyDeviceToHost));
    return LK_OK;
}

// ------------------------------------------------------------------------------------------------ batch ingest (N4)
int lk_ingest_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux,
                        const double *flux_err, int normalize, double *t_out, double *flux_out, double *flux_err_out,
                        int64_t *new_off_host, double *median_out, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::ingest_launch(h, B, n_off_host, t, flux, flux_err, normalize, t_out, flux_out, flux_err_out, new_off_host,
                             median_out, static_cast<hipStream_t>(stream));
}

int lk_ingest_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *flux, const double *flux_err,
                    int normalize, double *t_out, double *flux_out, double *flux_err_out, int64_t *new_off,
                    double *median_out) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && n_off != nullptr && new_off != nullptr, "bad batch description");
    if (B == 0) {
        new_off[0] = 0;
        return LK_OK;
    }
    LK_REQUIRE(t && flux && t_out && flux_out, "NULL buffer");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ntot = (size_t)n_off[B], nb = ntot * 8;
    h->staging.reset();
    int rc = h->staging.reserve(6 * (nb + 256) + (size_t)B * 8 + 4096);
    if (rc) return rc;
    double *dt = (double *)h->staging.alloc(nb), *df = (double *)h->staging.alloc(nb);
    double *de = flux_err ? (double *)h->staging.alloc(nb) : nullptr;
    double *dto = (double *)h->staging.alloc(nb), *dfo = (double *)h->staging.alloc(nb);
    double *deo = flux_err_out ? (double *)h->staging.alloc(nb) : nullptr;
    double *dmed = median_out ? (double *)h->staging.alloc((size_t)B * 8) : nullptr;
    LK_HIP_CHECK(hipMemcpy(dt, t, nb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(df, flux, nb, hipMemcpyHostToDevice));
    if (flux_err) LK_HIP_CHECK(hipMemcpy(de, flux_err, nb, hipMemcpyHostToDevice));
    rc = lk::ingest_launch(h, B, n_off, dt, df, de, normalize, dto, dfo, deo, new_off, dmed, nullptr);
    if (rc) return rc;
    const size_t kb = (size_t)new_off[B] * 8;
    LK_HIP_CHECK(hipMemcpy(t_out, dto, kb, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(flux_out, dfo, kb, hipMemcpyDeviceToHost));
    if (flux_err_out) LK_HIP_CHECK(hipMemcpy(flux_err_out, deo, kb, hipMemcpyDeviceToHost));
    if (median_out) LK_HIP_CHECK(hipMemcpy(median_out, dmed, (size_t)B * 8, hipMemcpyDeviceToHost));
    return LK_OK;
}

int lk_fits_unpack_batch_dev(lk_handle *h, int B, const uint8_t *raw, const int64_t *raw_off_host, const int32_t *desc_host,
                             const int64_t *bitmask_host, double *t_out, double *flux_out, double *flux_err_out,
                             int32_t *quality_out, int64_t *new_off_host, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::fits_unpack_launch(h, B, raw, raw_off_host, desc_host, bitmask_host, t_out, flux_out, flux_err_out,
                                  quality_out, new_off_host, static_cast<hipStream_t>(stream));
}

int lk_fits_unpack_batch(lk_handle *h, int B, const uint8_t *raw, const int64_t *raw_off, const int32_t *desc,
                         const int64_t *bitmask, double *t_out, double *flux_out, double *flux_err_out,
                         int32_t *quality_out, int64_t *new_off) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && raw_off != nullptr && desc != nullptr && new_off != nullptr, "bad batch description");
    if (B == 0) {
        new_off[0] = 0;
        return LK_OK;
    }
    LK_REQUIRE(raw && bitmask && t_out && flux_out, "NULL buffer");
    LK_REQUIRE(raw_off[0] == 0, "raw_off[0] must be 0");
    LK_HIP_CHECK(hipSetDevice(h->device));
    size_t rows = 0;
    for (int b = 0; b < B; ++b) rows += (size_t)std::max(0, desc[(size_t)b * 10 + 1]);
    const size_t nraw = (size_t)raw_off[B], nb = rows * 8;
    h->staging.reset();
    int rc = h->staging.reserve(nraw + 3 * (nb + 256) + rows * 4 + 4096);
    if (rc) return rc;
    uint8_t *draw = (uint8_t *)h->staging.alloc(nraw + 16);
    double *dto = (double *)h->staging.alloc(nb), *dfo = (double *)h->staging.alloc(nb);
    double *deo = flux_err_out ? (double *)h->staging.alloc(nb) : nullptr;
    int32_t *dqo = quality_out ? (int32_t *)h->staging.alloc(rows * 4) : nullptr;
    LK_HIP_CHECK(hipMemcpy(draw, raw, nraw, hipMemcpyHostToDevice));
    rc = lk::fits_unpack_launch(h, B, draw, raw_off, desc, bitmask, dto, dfo, deo, dqo, new_off, nullptr);
    if (rc) return rc;
    const size_t kept = (size_t)new_off[B];
    LK_HIP_CHECK(hipMemcpy(t_out, dto, kept * 8, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(flux_out, dfo, kept * 8, hipMemcpyDeviceToHost));
    if (flux_err_out) LK_HIP_CHECK(hipMemcpy(flux_err_out, deo, kept * 8, hipMemcpyDeviceToHost));
    if (quality_out) LK_HIP_CHECK(hipMemcpy(quality_out, dqo, kept * 4, hipMemcpyDeviceToHost));
    return LK_OK;
}

int lk_fits_unpack_cube_dev(lk_handle *h, const uint8_t *raw, int row_bytes, int n_rows, int off_time, int code_time,
                            int off_quality, int code_quality, int64_t bitmask, int keep_nan_time, int ncols,
                            const int32_t *col_off_host, int npix, double *t_out, int32_t *quality_out, float *cubes_out,
                            int64_t *kept_host, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::fits_cube_launch(h, raw, row_bytes, n_rows, off_time, code_time, off_quality, code_quality, bitmask,
                                keep_nan_time, ncols, col_off_host, npix, t_out, quality_out, cubes_out, kept_host,
                                static_cast<hipStream_t>(stream));
}

int lk_fits_unpack_cube(lk_handle *h, const uint8_t *raw, int row_bytes, int n_rows, int off_time, int code_time,
                        int off_quality, int code_quality, int64_t bitmask, int keep_nan_time, int ncols,
                        const int32_t *col_off, int npix, double *t_out, int32_t *quality_out, float *cubes_out,
                        int64_t *kept) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(raw && col_off && t_out && cubes_out && kept, "NULL buffer");
    LK_REQUIRE(row_bytes >= 1 && n_rows >= 0 && ncols >= 1 && ncols <= 4 && npix >= 1, "bad table description");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t nraw = (size_t)row_bytes * n_rows, ncube = (size_t)ncols * n_rows * npix * 4;
    h->staging.reset();
    int rc = h->staging.reserve(nraw + ncube + (size_t)n_rows * 12 + 4096);
    if (rc) return rc;
    uint8_t *draw = (uint8_t *)h->staging.alloc(nraw + 16);
    double *dt = (double *)h->staging.alloc((size_t)n_rows * 8 + 8);
    int32_t *dq = quality_out ? (int32_t *)h->staging.alloc((size_t)n_rows * 4 + 8) : nullptr;
    float *dc = (float *)h->staging.alloc(ncube + 16);
    LK_HIP_CHECK(hipMemcpy(draw, raw, nraw, hipMemcpyHostToDevice));
    rc = lk::fits_cube_launch(h, draw, row_bytes, n_rows, off_time, code_time, off_quality, code_quality, bitmask,
                              keep_nan_time, ncols, col_off, npix, dt, dq, dc, kept, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipDeviceSynchronize());
    const size_t k = (size_t)*kept;
    LK_HIP_CHECK(hipMemcpy(t_out, dt, k * 8, hipMemcpyDeviceToHost));
    if (quality_out) LK_HIP_CHECK(hipMemcpy(quality_out, dq, k * 4, hipMemcpyDeviceToHost));
    for (int c = 0; c < ncols; ++c)  // host layout: [column][kept cadence][pixel], columns n_rows * npix apart like the device's
        LK_HIP_CHECK(hipMemcpy(cubes_out + (size_t)c * n_rows * npix, dc + (size_t)c * n_rows * npix, k * npix * 4,
                               hipMemcpyDeviceToHost));
    return LK_OK;
}

int lk_transit_mask_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const int32_t *planet_off,
                              const double *period, const double *duration, const double *transit_time, uint8_t *mask,
                              void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::transit_mask_launch(h, B, n_off_host, t, planet_off, period, duration, transit_time, mask,
                                   static_cast<hipStream_t>(stream));
}

int lk_transit_mask_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const int32_t *planet_off,
                          const double *period, const double *duration, const double *transit_time, uint8_t *mask) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && n_off != nullptr, "bad batch description");
    if (B == 0 || n_off[B] == 0) return LK_OK;
    LK_REQUIRE(t && mask, "NULL buffer");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ntot = (size_t)n_off[B];
    h->staging.reset();
    int rc = h->staging.reserve(ntot * 9 + 2 * 256 + 4096);
    if (rc) return rc;
    double *dt = (double *)h->staging.alloc(ntot * 8);
    uint8_t *dm = (uint8_t *)h->staging.alloc(ntot);
    LK_HIP_CHECK(hipMemcpy(dt, t, ntot * 8, hipMemcpyHostToDevice));
    rc = lk::transit_mask_launch(h, B, n_off, dt, planet_off, period, duration, transit_time, dm, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(mask, dm, ntot, hipMemcpyDeviceToHost));
    return LK_OK;
}

int lk_bin_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux,
                     const double *flux_err, const int64_t *bin_off, const double *time_bin_start, const double *edges_sec,
                     int64_t n_edges, double bin_size_sec, const uint8_t *has_err, double *t_out, double *flux_out,
                     double *flux_err_out, void *stream) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_HIP_CHECK(hipSetDevice(h->device));
    return lk::bin_launch(h, B, n_off_host, t, flux, flux_err, bin_off, time_bin_start, edges_sec, n_edges, bin_size_sec,
                          has_err, t_out, flux_out, flux_err_out, static_cast<hipStream_t>(stream));
}

int lk_bin_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *flux, const double *flux_err,
                 const int64_t *bin_off, const double *time_bin_start, const double *edges_sec, int64_t n_edges,
                 double bin_size_sec, const uint8_t *has_err, double *t_out, double *flux_out, double *flux_err_out) {
    LK_REQUIRE(h != nullptr, "handle is NULL");
    LK_REQUIRE(B >= 0 && n_off != nullptr && bin_off != nullptr, "bad batch description");
    if (B == 0 || bin_off[B] == 0) return LK_OK;
    LK_REQUIRE(t && flux && t_out && flux_out && flux_err_out, "NULL buffer");
    LK_HIP_CHECK(hipSetDevice(h->device));
    const size_t ntot = (size_t)n_off[B], nb = ntot * 8, ob = (size_t)bin_off[B] * 8;
    h->staging.reset();
    int rc = h->staging.reserve(3 * (nb + 256) + 3 * (ob + 256) + 4096);
    if (rc) return rc;
    double *dt = (double *)h->staging.alloc(nb), *df = (double *)h->staging.alloc(nb);
    double *de = flux_err ? (double *)h->staging.alloc(nb) : nullptr;
    double *dto = (double *)h->staging.alloc(ob), *dfo = (double *)h->staging.alloc(ob), *deo = (double *)h->staging.alloc(ob);
    LK_HIP_CHECK(hipMemcpy(dt, t, nb, hipMemcpyHostToDevice));
    LK_HIP_CHECK(hipMemcpy(df, flux, nb, hipMemcpyHostToDevice));
    if (flux_err) LK_HIP_CHECK(hipMemcpy(de, flux_err, nb, hipMemcpyHostToDevice));
    rc = lk::bin_launch(h, B, n_off, dt, df, de, bin_off, time_bin_start, edges_sec, n_edges, bin_size_sec, has_err, dto, dfo,
                        deo, nullptr);
    if (rc) return rc;
    LK_HIP_CHECK(hipMemcpy(t_out, dto, ob, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(flux_out, dfo, ob, hipMemcpyDeviceToHost));
    LK_HIP_CHECK(hipMemcpy(flux_err_out, deo, ob, hipMemcpyDeviceToHost));
    return LK_OK;
}

}  // extern "C"
