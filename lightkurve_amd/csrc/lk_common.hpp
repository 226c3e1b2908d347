// lk_common.hpp — handle, error plumbing and scratch workspace shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lkhip.h"

namespace lk {

void set_error(const char *fmt, ...);

#define LK_HIP_CHECK(expr)                                                                    \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            lk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,    \
                          __LINE__);                                                          \
            return e_ == hipErrorOutOfMemory ? LK_ENOMEM : LK_EHIP;                           \
        }                                                                                     \
    } while (0)

#define LK_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            lk::set_error(__VA_ARGS__);  \
            return LK_EINVAL;            \
        }                                \
    } while (0)

// Grow-only device scratch arena.  Sub-allocations are 256-byte aligned and valid until reset().
struct Arena {
    char *base = nullptr;
    size_t cap = 0, used = 0;
    int reserve(size_t bytes);  // ensure capacity (may reallocate: only call before any alloc())
    void *alloc(size_t bytes) {
        size_t a = (used + 255) & ~size_t(255);
        if (a + bytes > cap) return nullptr;
        used = a + bytes;
        return base + a;
    }
    void reset() { used = 0; }
    void release();
};

// Ring of pinned host buffers for small asynchronous host->device copies (batch offsets): the caller's buffer is
// captured before the entry point returns, and the device copy stays ordered on the caller's stream.
struct HostStage {
    static constexpr int SLOTS = 4;
    void *buf[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[SLOTS] = {0, 0, 0, 0};
    hipEvent_t ev[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    int next = 0;
    int copy(void *dst, const void *src, size_t bytes, hipStream_t stream);
    void release();
};
}  // namespace lk

struct lk_handle {
    int device = 0;
    int num_cu = 256;
    int host_chunk_mb = 64;  // MiB of spectra per chunk of the pinned host pipeline (LK_HOST_CHUNK_MB at lk_init, lk_set_host_chunk_mb)
    lk::HostStage stage;
    lk::Arena ws;        // kernel scratch (prepped per-cadence records, per-target stats)
    lk::Arena staging;   // device mirrors of host buffers for the *_batch (host pointer) entry points
    // host-pointer pipeline (lk_ls_fast_peaks_batch): copy-in, compute and copy-out streams + the events that order
    // the two halves of the double buffers; created on first use
    hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    int *h_plan = nullptr;         // 64 B of pinned host memory: device -> host plan words (lsfast.hip)
    std::vector<const void *> lds_attr_done;  // kernels whose dynamic-LDS attribute has been raised on this device
    int lds_attr_rc = 0;                      // first failure of want_lds inside a void launch helper (take_lds_error)
    int bls_attr_set = 0;          // bls.hip: kernel attributes set and the LDS-atomic order self-test run on this device
    int bls_serial_hist = 0;       // ... which found ds_add_f64 NOT lane-ordered: the histogram runs its atomic-free form
    int bls_force_serial_hist = 0; // lk_bls_set_ordered_histogram: the caller asked for that form (tests, diagnosis)
    double pld_eig_tol = 0.0;      // lk_pld_set_eig_tolerance: stop of the PLD blocks' subspace iteration (0: the built-in 1e-7)
    int pld_eig_split = 0;         // lk_pld_set_eig_mode: 0 = the one-kernel subspace iteration (default: the faster, profiles/r06_pld_eig_modes_ab.txt), 1 = its phase-split form (pld_eigs_* kernels) in front of it
    // side streams of the handle + fork / join events, created on first use: lsfast.hip runs its chunk loop on two streams, bls.hip
    // spreads the period groups of a small job over four; every call joins them back into the caller's stream before it returns
    hipStream_t s_ls_aux[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_ls_fork = nullptr, ev_ls_join[3] = {nullptr, nullptr, nullptr};
    // flatten.hip: the batch's two offset tables (cadence offsets | scratch-slab offsets) stay on the device between calls; a
    // call whose tables equal the previous call's (equal-shaped batches, the usual pipeline) skips the copy
    int64_t *flat_tab_dev = nullptr;
    size_t flat_tab_cap = 0;                // entries
    std::vector<int64_t> flat_tab_host;     // what flat_tab_dev holds (empty: nothing valid)
    hipStream_t flat_tab_stream = nullptr;  // the stream the table's upload was queued on (a call on another stream re-uploads)
    // device.hip: lk_shader_clock_mhz's own stream and 16 pinned bytes (kept: freeing either would synchronise the device)
    hipStream_t s_probe = nullptr;
    unsigned long long *clk_buf = nullptr;
};

namespace lk {
// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize once per DEVICE (a
// process may drive several GPUs, one handle each): remembered in the handle, not in a function-local static.
inline int want_lds_checked(lk_handle *h, const void *fn, int bytes) {
    LK_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    h->lds_attr_done.push_back(fn);
    return LK_OK;
}
// A failure is returned AND remembered in the handle (lds_attr_rc): launch helpers that return void cannot pass it up, the
// launcher that called them checks take_lds_error() before it reports success.
inline int want_lds(lk_handle *h, const void *fn, int bytes) {
    for (const void *f : h->lds_attr_done)
        if (f == fn) return LK_OK;
    const int rc = want_lds_checked(h, fn, bytes);
    if (rc) h->lds_attr_rc = rc;
    return rc;
}
inline int take_lds_error(lk_handle *h) {
    const int rc = h->lds_attr_rc;
    h->lds_attr_rc = 0;
    return rc;
}
}  // namespace lk

// launchers implemented in the .hip files (device pointers, enqueue on stream, no sync)
namespace lk {
int ls_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *dy,
              const double *freq, double f0, double df, int64_t M, int fit_mean, int center_data,
              int normalization, const double *scale, double *power, hipStream_t stream);
int ls_chi2_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *dy,
                   const double *freq, double f0, double df, int64_t M, int nterms, int fit_mean, int center_data,
                   int normalization, const double *scale, double *power, hipStream_t stream);
int argmax_launch(lk_handle *h, int B, int64_t M, const double *x, double *max_out, int64_t *argmax_out,
                  hipStream_t stream);
int bls_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *ivar,
               const double *period_host, const double *period_dev, int64_t nP, const double *duration_host,
               int nD, int oversample, int use_likelihood, double *out7, hipStream_t stream);
int bls_max_period_host(const double *duration_host, int nD, int oversample, double *max_period);
int regress_launch(lk_handle *h, int B, const int64_t *n_off_host, int K, const double *X, const double *y,
                   const double *err, const uint8_t *cmask, const double *prior_mu, const double *prior_sigma,
                   double clip_sigma, int niters, double *w, double *model, uint8_t *outl, hipStream_t stream,
                   double *w_cov = nullptr);
int model_part_launch(lk_handle *h, int B, int N, int K, int c0, int c1, const double *X, const double *w, double *out,
                      hipStream_t stream);
int flatten_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux,
                   const uint8_t *user_mask, int window, int polyorder, double break_tol, int niters, double sigma,
                   double *trend, uint8_t *final_mask, hipStream_t stream);
int savgol_design_host(int window, int polyorder, double *coeffs, double *edge);
int gram_plain_launch(const double *A, const int64_t *d_off, int B, int K, double *G, hipStream_t stream,
                      lk_handle *h = nullptr);
int gram_plain_f32_launch(const float *pix, const float *div, const double *mean, int mode, const int64_t *d_off, int B, int K,
                          double *G, hipStream_t stream, lk_handle *h);
int pld_design_launch(lk_handle *h, int B, int N, int P, int Pb, const float *pld_pix, const float *bkg_pix,
                      const float *lc_flux, const double *time, const double *knots, int n_inner, int pld_order,
                      int pca_components, int n_knots, int spline_degree, int normalize_bkg, int K, double *X,
                      double *prior_sigma, hipStream_t stream);
int pld_design_width(int P, int Pb, int pld_order, int pca_components, int n_knots);
int dm_pca_launch(lk_handle *h, int B, int N, int P, int k, const double *A, double *U, hipStream_t stream);
int dm_spline_launch(lk_handle *h, int B, int N, const double *x, const double *knots, int n_inner, int degree, double *out,
                     hipStream_t stream);
int dm_standardize_launch(lk_handle *h, int B, int N, int P, const double *A, double *out, hipStream_t stream);
int fold_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *period_host,
                const double *epoch_time_host, double epoch_phase, const double *wrap_phase_host, int normalize_phase,
                int ncols, const double *const *cols_in, double *const *cols_out, double *phase, int64_t *order,
                hipStream_t stream);
int pg_logmedian_launch(lk_handle *h, int B, int64_t M, const double *power, int K, const int *win_lo_host,
                        const int *win_hi_host, const int *klo_host, const int *khi_host, double corr, double *out,
                        hipStream_t stream);
int pg_boxsmooth_launch(lk_handle *h, int B, int64_t M, const double *power, const double *taps_host, int nk,
                        double *out, hipStream_t stream);
int sigma_clip_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *y, double sigma, int maxiters,
                      uint8_t *outlier, hipStream_t stream);
int ingest_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux, const double *err,
                  int normalize, double *t_out, double *f_out, double *e_out, int64_t *new_off_host, double *median_out,
                  hipStream_t stream);
int fits_unpack_launch(lk_handle *h, int B, const uint8_t *raw, const int64_t *raw_off_host, const int32_t *desc_host,
                       const int64_t *bitmask_host, double *t_out, double *f_out, double *e_out, int32_t *q_out,
                       int64_t *new_off_host, hipStream_t stream);
int fits_cube_launch(lk_handle *h, const uint8_t *raw, int row_bytes, int n_rows, int off_time, int code_time, int off_qual,
                     int code_qual, int64_t bitmask, int keep_nan_time, int ncols, const int32_t *col_off_host, int npix,
                     double *t_out, int32_t *q_out, float *cubes_out, int64_t *kept_host, hipStream_t stream);
int transit_mask_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const int *p_off_host,
                        const double *period_host, const double *duration_host, const double *transit_time_host,
                        uint8_t *mask, hipStream_t stream);
int bin_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux, const double *err,
               const int64_t *bin_off_host, const double *start_host, const double *edges_host, int64_t n_edges,
               double bin_size_sec, const uint8_t *has_err_host, double *t_out, double *f_out, double *e_out,
               hipStream_t stream);
int pg_acf2d_launch(lk_handle *h, int B, int64_t M, const double *power, int n_win, const int *win_start_host, int W,
                    double *acf2d, double *metric, hipStream_t stream);
int lsfast_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *dy,
                  double f0, double df, int64_t M, int fit_mean, int center_data, int normalization,
                  const double *scale, int oversampling, double *power, hipStream_t stream, double *max_out = nullptr,
                  int64_t *arg_out = nullptr);
int lsfastchi2_launch(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y, const double *dy,
                      double f0, double df, int64_t M, int nterms, int fit_mean, int center_data, int normalization,
                      const double *scale, int oversampling, double *power, hipStream_t stream);
}  // namespace lk
