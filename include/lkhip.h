/* lkhip.h — C ABI of liblkhip.so: the MI355X (gfx950) hot path of lightkurve's periodogram +
 * systematics-correction pipeline.  Plain pointers and sizes only; no torch / numpy types.
 *
 * lightkurve is pure Python and has no FFI of its own; each entry point below replaces the numerical
 * call the reference makes at one of its four seams:
 *
 *   lk_ls_power_batch*   <- astropy METHODS[method](t, y, dy, frequency, center_data, fit_mean, normalization)
 *                           reached from src/lightkurve/periodogram.py:961-964 (LS.power(frequency, method=ls_method)),
 *                           plus lightkurve's own normalisation lines periodogram.py:969-975 (fused).
 *   lk_argmax_batch*     <- Periodogram.max_power / frequency_at_max_power, periodogram.py:127-140 (nanmax/nanargmax).
 *   lk_bls_batch*        <- astropy methods.bls_fast(t, y, ivar, period, duration, oversample, use_likelihood)
 *                           == C run_bls(...), reached from periodogram.py:1169 (bls.power(period, duration, **kwargs)).
 *   lk_savgol_trend_batch* <- scipy.signal.savgol_filter + interp1d inside LightCurve.flatten,
 *                           src/lightkurve/lightcurve.py:996-1063.
 *   lk_regress_batch*    <- RegressionCorrector._fit_coefficients + the sigma-clip loop of .correct,
 *                           src/lightkurve/correctors/regressioncorrector.py:127-189, 243-279.
 *   lk_ls_fast_batch*    <- astropy lombscargle_fast (the DEFAULT ls_method="fast", periodogram.py:650): fast_impl.py.
 *   lk_ls_chi2_batch* / lk_ls_fastchi2_batch* <- astropy lombscargle_chi2 / lombscargle_fastchi2 (nterms > 1,
 *                           periodogram.py:948-967).
 *   lk_pld_design_batch* <- PLDCorrector.create_design_matrix, src/lightkurve/correctors/pldcorrector.py:125-287.
 *   lk_pld_correct_batch <- PLDCorrector.correct (pldcorrector.py:304-427) for a batch of cutouts: the two above fused, the
 *                           design matrices staying in device memory.
 *   lk_fold_batch*       <- LightCurve.fold, src/lightkurve/lightcurve.py:1089-1214 (astropy TimeSeries.fold + sort).
 *   lk_pg_logmedian_batch* / lk_pg_boxsmooth_batch* <- Periodogram.smooth, periodogram.py:182-284.
 *
 * Conventions
 *   - Every function returns an int status: LK_OK, LK_EINVAL (-> ValueError), LK_ENOMEM (-> MemoryError),
 *     LK_EHIP (-> RuntimeError; text from lk_last_error()).  Outputs are fully written on LK_OK.
 *   - Ragged batches: target b owns elements [n_off[b], n_off[b+1]) of the concatenated arrays.
 *   - `*_batch` takes HOST pointers (caller-owned numpy buffers; copied in/out inside the call).
 *     `*_batch_dev` takes DEVICE pointers (already resident in HBM) and enqueues on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream) without synchronising.
 *     Some `_dev` launchers synchronise `stream` internally where the host needs a device result to size
 *     the next launch (fold, flatten, BLS, Periodogram.smooth, PLD); treat "no sync" as not guaranteed.
 *   - One lk_handle drives one GPU (one process per GPU); it owns its scratch workspace and is not
 *     thread-safe.  ALL calls on one handle must use ONE stream (or be separated by a stream/device
 *     synchronisation): every launcher carves its kernel scratch from the handle's single arena, so two calls
 *     in flight on different streams would overwrite each other's live scratch.  Use one handle per stream /
 *     per thread if you need concurrency on one GPU.  (Inside a call the library may fork side streams of its own —
 *     lk_ls_fast_* runs its chunks on two, lk_bls_* spreads the period groups of a small job over four — and joins them
 *     back into `stream` by events before it returns: to the caller the call is still ordered on `stream`.)
 *     Multi-GPU = one handle per rank, targets sharded by the caller (no data-path collective).
 *   - Deviations from the ABI sketched in SURVEY.md §8(b), on purpose: lk_init takes ONE device id (one handle
 *     = one GPU = one process, the torch.distributed model; the sketch's device list would put multi-GPU fan-out
 *     inside the call); there is no `precision` argument (everything is fp64: the parity tolerance of 1e-9
 *     against the reference leaves no room for fp32/mixed variants); (max, argmax) come from lk_argmax_batch*
 *     or the fused lk_ls_fast_peaks_batch* rather than from extra outputs of lk_ls_power_batch.
 */
#ifndef LKHIP_H
#define LKHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LK_OK 0
#define LK_EINVAL 1
#define LK_ENOMEM 2
#define LK_EHIP 3

/* LS normalisations: astropy 'standard', astropy 'psd', lightkurve 'amplitude' (periodogram.py:974-975),
 * lightkurve 'psd' (periodogram.py:969-973; needs per-target scale = 2/(N*oversample*fs)). */
#define LK_NORM_STANDARD 0
#define LK_NORM_PSD 1
#define LK_NORM_LK_AMPLITUDE 2
#define LK_NORM_LK_PSD 3

typedef struct lk_handle lk_handle;

int lk_version(void);
const char *lk_last_error(void);
int lk_device_count(int *count);
int lk_init(int device_id, lk_handle **out);
/* Chunk size (MiB of spectra) of the pinned, double-buffered host pipeline behind lk_ls_fast_*_batch; default 64, or the
 * environment variable LK_HOST_CHUNK_MB read once by lk_init — the library's only environment knob. */
int lk_set_host_chunk_mb(lk_handle *h, int mb);
/* BLS histogram form.  The fast form adds a wave's 64 cadences to their phase bins with one LDS ds_add_f64 per array and
 * relies on the hardware applying same-address lanes in lane order (checked on the device, once per handle); where that
 * check fails the library switches by itself to an atomic-free form in which one lane at a time updates its bins —
 * bit-identical results, ~64 x the LDS instructions in the histogram phase.  on != 0 forces that form (tests, diagnosis of
 * a suspected ordering problem); 0 returns to the automatic choice. */
int lk_bls_set_ordered_histogram(lk_handle *h, int on);
/* Stop criterion of the subspace iteration behind the PCA blocks of lk_pld_design_batch* / lk_pld_correct_batch
 * (DesignMatrix.pca inside PLDCorrector.create_design_matrix, correctors/designmatrix.py:252-282): residual
 * ||C r - theta r|| <= tol * theta_max * sqrt(k).  Default (tol = 0) 1e-7: two decades below the first visible change of the
 * corrected flux on the reference goldens (profiles/r05_pld_tol_sweep.txt).  lk_pca_batch (whose OUTPUT is the basis) always
 * uses 1e-10. */
int lk_pld_set_eig_tolerance(lk_handle *h, double tol);
/* Form of that subspace iteration for Gram matrices beyond the direct solver (P > 138): 0 (default) = one workgroup carries a
 * matrix through the whole iteration in ONE kernel (pld_topk_eig_kernel); 1 = every phase — products with C, Rayleigh-Ritz,
 * Cholesky-QR — is its own launch over all matrices with the state in global memory (pld_eigs_* kernels: no spilled VGPRs in the small phases,
 * per-phase times in a plain kernel trace), converged matrices dropping out by flag and the one-kernel form finishing whatever is
 * left.  Same arithmetic, same results to rounding; 0 is 1-2 % faster on the PLD step (profiles/r06_pld_eig_modes_ab.txt). */
int lk_pld_set_eig_mode(lk_handle *h, int mode);
void lk_destroy(lk_handle *h);
/* Block until every kernel / copy issued through this handle's GPU has finished (hipDeviceSynchronize): for callers
 * of the *_dev entry points that do not hold a HIP runtime of their own. */
int lk_synchronize(lk_handle *h);
/* bytes of device scratch currently held by the handle */
int64_t lk_workspace_bytes(const lk_handle *h);

/* ---- Lomb-Scargle (exact floating-mean GLS, direct trig sums) --------------------------------------
 * t: times relative to the target's first cadence [d] (astropy: lombscargle/core.py:119-126);
 * y: flux; dy: per-cadence errors or NULL (uniform weights, the lightkurve default);
 * freq: M frequencies [1/d] or NULL for the regular grid f0 + df*j, j<M (the fast path);
 * scale: per-target factor for LK_NORM_LK_PSD, or NULL (=1); power: B*M row-major, float64. */
int lk_ls_power_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y,
                      const double *dy, const double *freq, double f0, double df, int64_t M,
                      int fit_mean, int center_data, int normalization, const double *scale,
                      double *power);
int lk_ls_power_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                          const double *dy, const double *freq, double f0, double df, int64_t M,
                          int fit_mean, int center_data, int normalization, const double *scale,
                          double *power, void *stream);

/* ---- Lomb-Scargle with nterms Fourier terms (lightkurve nterms > 1 with ls_method "chi2" / "fastchi2",
 * periodogram.py:948-967): astropy lombscargle_chi2 (chi2_impl.py:5-86), i.e. at every frequency the weighted
 * least-squares fit of [1,] sin(m w t), cos(m w t), m = 1..nterms; power = (X^T y)^T (X^T X)^-1 (X^T y), normalised as
 * above.  The trig sums are exact direct sums, so the result is what 'chi2' returns (which 'fastchi2' approximates by
 * extirpolation + FFT).  nterms = 1 is lk_ls_power_batch.  1 <= nterms <= LK_MAX_NTERMS. */
#define LK_MAX_NTERMS 8   /* 1..4: regular-grid and FFT kernels; 5..8: exact sums, one thread per frequency (both names) */
int lk_ls_chi2_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y, const double *dy,
                     const double *freq, double f0, double df, int64_t M, int nterms, int fit_mean, int center_data,
                     int normalization, const double *scale, double *power);
int lk_ls_chi2_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                         const double *dy, const double *freq, double f0, double df, int64_t M, int nterms,
                         int fit_mean, int center_data, int normalization, const double *scale, double *power,
                         void *stream);

/* ---- Lomb-Scargle ls_method="fastchi2" with nterms Fourier terms (periodogram.py:948-967): astropy
 * lombscargle_fastchi2 (fastchi2_impl.py:60-137) — the multi-term fit of lk_ls_chi2_batch with every trig sum taken
 * from the extirpolated FFT grids (3 nterms grids per target), regular frequency grid only.  Agrees with the
 * reference's 'fastchi2' output to ~1e-9 where the fit is well posed.  nterms = 1 is lk_ls_fast_batch. */
int lk_ls_fastchi2_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y, const double *dy,
                         double f0, double df, int64_t M, int nterms, int fit_mean, int center_data, int normalization,
                         const double *scale, int oversampling, double *power);
int lk_ls_fastchi2_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                             const double *dy, double f0, double df, int64_t M, int nterms, int fit_mean,
                             int center_data, int normalization, const double *scale, int oversampling, double *power,
                             void *stream);

/* ---- LightCurve.fold (src/lightkurve/lightcurve.py:1089-1214 over astropy TimeSeries.fold, timeseries/sampled.py:
 * 230-233) for B ragged targets: phase = ((t - epoch_time) + epoch_phase + (P - wrap)) % P - (P - wrap) (numpy `%`),
 * divided by P if normalize_phase (epoch_phase and wrap_phase are then in phase units), followed by a STABLE sort by
 * phase.  period / epoch_time / wrap_phase: one value per target (HOST arrays).  Outputs, all in sorted order:
 * phase[sum N], order[sum N] (index of the cadence, relative to its target, that lands at each sorted slot) and
 * cols_out[c][i] = cols_in[c][order[i]] for ncols value columns (flux, flux_err, ...).  The phases are bit-identical
 * to numpy's and the permutation equals np.argsort(phase, kind="stable"). */
int lk_fold_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *period,
                  const double *epoch_time, double epoch_phase, const double *wrap_phase, int normalize_phase,
                  int ncols, const double *const *cols_in, double *const *cols_out, double *phase, int64_t *order);
int lk_fold_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *period,
                      const double *epoch_time, double epoch_phase, const double *wrap_phase, int normalize_phase,
                      int ncols, const double *const *cols_in, double *const *cols_out, double *phase, int64_t *order,
                      void *stream);

/* ---- Periodogram.smooth / Periodogram.flatten (src/lightkurve/periodogram.py:182-284, 381-429) for B periodograms on
 * one shared frequency grid of M points, power row-major [B][M].
 *
 * lk_pg_logmedian_batch: method='logmedian' (:265-284).  The caller prepares the window bookkeeping from the
 * frequency grid exactly as the reference loop does (numpy log10 of the frequencies; window centres by the running
 * sum x0 += 0.5 * filter_width; membership |log10 f - x0| < filter_width): window k covers the frequency indices
 * [win_lo[k], win_hi[k]) and frequency j lies in the windows klo[j] .. khi[j] (inclusive; klo > khi = none).  Per
 * window: nanmedian(power) / corr (corr = (8/9)^3); per frequency: mean of its windows' values, summed in window
 * order.  All four tables are HOST arrays (also for the _dev flavour).
 *
 * lk_pg_boxsmooth_batch: method='boxkernel' (:236-263) = astropy.convolution.convolve(power, Box1DKernel(width)) with
 * its defaults boundary='fill' (zeros), normalize_kernel=True, nan_treatment='interpolate'.  taps = the kernel array
 * (flipped; nk odd), HOST array.  flatten = power / smooth is left to the caller (one division). */
int lk_pg_logmedian_batch(lk_handle *h, int B, int64_t M, const double *power, int K, const int32_t *win_lo,
                          const int32_t *win_hi, const int32_t *klo, const int32_t *khi, double corr, double *out);
int lk_pg_logmedian_batch_dev(lk_handle *h, int B, int64_t M, const double *power, int K, const int32_t *win_lo,
                              const int32_t *win_hi, const int32_t *klo, const int32_t *khi, double corr, double *out,
                              void *stream);
int lk_pg_boxsmooth_batch(lk_handle *h, int B, int64_t M, const double *power, const double *taps, int nk,
                          double *out);
int lk_pg_boxsmooth_batch_dev(lk_handle *h, int B, int64_t M, const double *power, const double *taps, int nk,
                              double *out, void *stream);

/* ---- seismology 2-D autocorrelation (src/lightkurve/seismology/numax_estimators.py:15-205 over seismology/utils.py:
 * 106-158): for n_win windows of W samples starting at win_start[k] (HOST array) of each of B periodograms on one grid,
 * the autocorrelation C[lag] = sum_i p[i] p[i + lag], lag < W, of the window minus its nanmean, and the mean collapsed
 * correlation metric[k] = (sum_lag |C[lag]| - 1) / W.  acf2d: B x n_win x W (window-major); metric: B x n_win. */
int lk_pg_acf2d_batch(lk_handle *h, int B, int64_t M, const double *power, int n_win, const int32_t *win_start, int W,
                      double *acf2d, double *metric);
int lk_pg_acf2d_batch_dev(lk_handle *h, int B, int64_t M, const double *power, int n_win, const int32_t *win_start,
                          int W, double *acf2d, double *metric, void *stream);

/* ---- Lomb-Scargle, lightkurve's DEFAULT method ls_method="fast" (periodogram.py:650): Press & Rybicki extirpolation
 * + FFT evaluation of the trig sums (astropy fast_impl.py / utils.py trig_sum, extirpolate), regular grid only.
 * Agrees with the reference's 'fast' output to ~1e-10 (and, like it, is ~1e-3 of the peak from the exact methods).
 * oversampling: FFT grid oversampling (astropy default 5; Nfft = bitceil(M * oversampling)). */
int lk_ls_fast_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y, const double *dy,
                     double f0, double df, int64_t M, int fit_mean, int center_data, int normalization,
                     const double *scale, int oversampling, double *power);
int lk_ls_fast_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                         const double *dy, double f0, double df, int64_t M, int fit_mean, int center_data,
                         int normalization, const double *scale, int oversampling, double *power, void *stream);

/* The same method followed by Periodogram.max_power / frequency_at_max_power (periodogram.py:127-140): per target
 * nanmax and nanargmax of the power row (first maximum wins; all-NaN row -> (nan, -1)).
 *
 * lk_ls_fast_peaks_batch (HOST pointers) is software-pipelined over chunks of targets: the H2D copy of chunk k+1,
 * the kernels of chunk k and the D2H copy of chunk k-1 run on three streams over double-buffered device memory.
 * Caller buffers that are pinned (lk_host_alloc, hipHostMalloc, hipHostRegister) are DMA'd directly; pageable
 * buffers go through the HIP runtime's staging.  power may be NULL (peaks only — the B x M spectra never cross PCIe);
 * max_power / argmax may both be NULL (spectra only: this is what lk_ls_fast_batch does).
 * lk_ls_fast_peaks_batch_dev: device pointers, power required, max_power / argmax nullable together. */
int lk_ls_fast_peaks_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y,
                           const double *dy, double f0, double df, int64_t M, int fit_mean, int center_data,
                           int normalization, const double *scale, int oversampling, double *power,
                           double *max_power, int64_t *argmax);
int lk_ls_fast_peaks_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                               const double *dy, double f0, double df, int64_t M, int fit_mean, int center_data,
                               int normalization, const double *scale, int oversampling, double *power,
                               double *max_power, int64_t *argmax, void *stream);

/* The batch form of the loop `for lc in collection: lc.to_periodogram(frequency=...)` (reference collections.py:145 over
 * lightcurve.py:2490-2535 -> periodogram.py:869-967): `time` holds the ABSOLUTE times of the packed light curves; each
 * chunk is rebased on the device to t - t[first cadence of its light curve] (astropy lombscargle/core.py:119-126 does the
 * same subtraction per object) before the kernels of lk_ls_fast_peaks_batch run on it.  Everything else as above. */
int lk_ls_fast_peaks_lc_batch(lk_handle *h, int B, const int64_t *n_off, const double *time, const double *flux,
                              const double *dy, double f0, double df, int64_t M, int fit_mean, int center_data,
                              int normalization, const double *scale, int oversampling, double *power,
                              double *max_power, int64_t *argmax);

/* Pinned (page-locked) host memory for the host-pointer entry points: numpy arrays built over it are copied by DMA
 * without a staging pass.  Any host pointer is accepted everywhere; pinned ones are simply faster. */
int lk_host_alloc(void **ptr, size_t bytes);
int lk_host_free(void *ptr);

/* ---- device-resident batches ("FITS -> ragged device arrays ... on device") ----------------------
 * The reference chains its steps per object — lk.read(...) -> lc.remove_nans().normalize() -> lc.flatten() ->
 * lc.to_periodogram() -> lc.fold() (src/lightkurve/collections.py:145 over lightcurve.py:1300-1327, 1216-1292, 943-1078,
 * 2490-2535, 1089-1214); the batch form keeps the packed arrays in HBM between the `_dev` entry points.  For callers that do
 * not hold a HIP runtime of their own (ctypes, cgo, JNI): device memory, streams and copies by plain pointers.  Copies are
 * hipMemcpyAsync on `stream` (page-locked host buffers from lk_host_alloc make them truly asynchronous; pageable ones block
 * the host inside the call); lk_stream_synchronize (or lk_synchronize) before the host touches a d2h destination.
 * lk_dev_free synchronises the device (hipFree): recycle buffers in a pipeline (lightkurve_amd/device.py does). */
int lk_dev_alloc(lk_handle *h, void **ptr, size_t bytes);
int lk_dev_free(lk_handle *h, void *ptr);
int lk_stream_create(lk_handle *h, void **stream);
int lk_stream_destroy(lk_handle *h, void *stream);
int lk_stream_synchronize(lk_handle *h, void *stream);
int lk_memcpy_h2d(lk_handle *h, void *dst_dev, const void *src_host, size_t bytes, void *stream);
int lk_memcpy_d2h(lk_handle *h, void *dst_host, const void *src_dev, size_t bytes, void *stream);
int lk_memcpy_d2d(lk_handle *h, void *dst_dev, const void *src_dev, size_t bytes, void *stream);
/* LightCurve.flatten's last lines (lightcurve.py:1064-1070): flux_out = flux / trend, flux_err_out = flux_err / trend over
 * the n packed cadences (flux_err nullable -> NaN errors; flux_err_out nullable; in-place allowed). */
int lk_flatten_apply_batch_dev(lk_handle *h, int64_t n, const double *flux, const double *flux_err, const double *trend,
                               double *flux_out, double *flux_err_out, void *stream);
/* lk_ls_fast_peaks_lc_batch with DEVICE pointers: `time` holds the light curves' own (absolute) times; they are rebased to
 * t - t[first cadence of the light curve] (astropy lombscargle/core.py:119-126) into scratch of the handle, `time` itself is
 * not modified.  power required; max_power / argmax nullable together. */
int lk_ls_fast_peaks_lc_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *time, const double *flux,
                                  const double *dy, double f0, double df, int64_t M, int fit_mean, int center_data,
                                  int normalization, const double *scale, int oversampling, double *power,
                                  double *max_power, int64_t *argmax, void *stream);
/* t_out = time - time[first cadence of its light curve] (out of place): what astropy's LombScargle hands every method,
 * for the exact entry points (lk_ls_power_batch_dev / lk_ls_chi2_batch_dev) of a device-resident batch. */
int lk_rebase_times_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *time, double *t_out,
                              void *stream);
/* Per light curve: descents_host[b] = number of cadences whose time is smaller than the previous one's (flatten, bin and
 * the bit-exact BLS need 0), finite_host[b] = number of finite values of x (LightCurve.bin's "has a finite error" test,
 * lightcurve.py:1712-1716).  time / descents_host and x / finite_host are nullable in pairs.  Synchronises `stream`. */
int lk_segment_probe_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *time, const double *x,
                               int64_t *descents_host, int64_t *finite_host, void *stream);
/* What BoxLeastSquaresPeriodogram.from_lightcurve + astropy BoxLeastSquares hand to bls_fast (periodogram.py:1093-1100,
 * 1146-1169; astropy bls/core.py:277-327), per light curve of a packed batch WITHOUT NaN flux: t_out = (t - t[0]) -
 * min(t - t[0]); y_out = flux - numpy.median(flux); ivar_out = 1 / flux_err^2 if every error of the light curve is finite,
 * else ones (flux_err NULL: ones); t_ref_out[b] (device, nullable) = min(t - t[0]) + t[0], the zero of transit_time. */
int lk_bls_prepare_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *time, const double *flux,
                             const double *flux_err, double *t_out, double *y_out, double *ivar_out, double *t_ref_out,
                             void *stream);
/* Further per-cadence columns (quality flags, centroids, ...) through lk_ingest_batch's compaction: cols_out[c][new_off[b] +
 * k] = cols_in[c][...] for the k-th cadence of light curve b whose flux is not NaN.  ncols <= 8 device pointers in two
 * HOST arrays; elem_bytes 4 or 8; n_off / new_off: the offsets lk_ingest_batch_dev was given and returned. */
int lk_compact_columns_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const int64_t *new_off_host,
                                 const double *flux, int ncols, int elem_bytes, const void *const *cols_in,
                                 void *const *cols_out, void *stream);
/* dst_host[i] = src_dev[idx_host[i]], i < n (the first / last time of every light curve for lightkurve's psd scale,
 * periodogram.py:865-868).  Synchronises `stream`. */
int lk_gather_f64_dev(lk_handle *h, int n, const int64_t *idx_host, const double *src_dev, double *dst_host, void *stream);
/* The shader clock [MHz] sustained over `spin_ms` milliseconds, measured ON the device (shader-cycle counter against the
 * constant-rate counter) by one wave on a stream of its own — i.e. under whatever load the caller has queued.  What
 * bench.py prints beside its roofline fractions: box-to-box spread of a bandwidth fraction is mostly this number. */
int lk_shader_clock_mhz(lk_handle *h, double spin_ms, double *mhz);

/* ---- batch ingest: the steps before the hot path, for B ragged light curves -------------------
 * lk_ingest_batch: LightCurve.remove_nans (src/lightkurve/lightcurve.py:1300-1327) + LightCurve.normalize (:1216-1292).
 * Cadences whose flux is NaN are dropped (order kept), the batch is repacked contiguously: new_off (B + 1, HOST, written
 * before the call returns: the call synchronises) addresses t_out / flux_out / flux_err_out (capacity: the input sizes);
 * median_out[b] = nanmedian(flux_b) (nullable); normalize != 0 divides flux and flux_err by it.  flux_err / flux_err_out
 * nullable (NaN errors are written when flux_err is NULL and flux_err_out is not). */
int lk_ingest_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *flux, const double *flux_err,
                    int normalize, double *t_out, double *flux_out, double *flux_err_out, int64_t *new_off,
                    double *median_out);
int lk_ingest_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux,
                        const double *flux_err, int normalize, double *t_out, double *flux_out, double *flux_err_out,
                        int64_t *new_off_host, double *median_out, void *stream);
/* LightCurve.remove_outliers (:1430-1556): astropy.stats.sigma_clip(y, sigma, maxiters, cenfunc=median, stdfunc=std).mask
 * for B ragged arrays: outlier[i] = 1 where the value is clipped or not finite. */
int lk_sigma_clip_batch(lk_handle *h, int B, const int64_t *n_off, const double *y, double sigma, int maxiters,
                        uint8_t *outlier);
int lk_sigma_clip_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *y, double sigma, int maxiters,
                            uint8_t *outlier, void *stream);
/* lk_fits_unpack_batch: what the reference's light-curve readers do per file through astropy — Table.read of the BINTABLE
 * (src/lightkurve/io/generic.py:21-207), drop the rows whose TIME is NaN (:98-101), drop the cadences whose quality flag
 * hits the bitmask (io/kepler.py:49-53, io/tess.py:45-48, utils.py:79-115) — for B files at once.  `raw`: the tables' bytes
 * exactly as in the files (big-endian records), file b at [raw_off[b], raw_off[b+1]) with raw_off[b] a multiple of 4 and
 * at least 3 spare bytes after its rows x record-length bytes.  desc: B x 10 int32 = {record length (<= 512), rows, byte
 * offset and TFORM code of TIME, of the flux column, of the flux-error column (offset -1: none -> NaN), of the quality
 * column (offset -1: none -> 0)}; codes 0 = D (float64), 1 = E (float32), 2 = J (int32), 3 = K (int64), 4 = I (int16),
 * 5 = B (uint8).  bitmask: B x int64.  Outputs (capacity: sum of rows): float64 time / flux / flux_err (nullable),
 * int32 quality (nullable), packed in file order; new_off (B + 1, HOST, valid when the call returns: it synchronises).
 * Header parsing stays on the host (lightkurve_amd/fitsio.py). */
int lk_fits_unpack_batch(lk_handle *h, int B, const uint8_t *raw, const int64_t *raw_off, const int32_t *desc,
                         const int64_t *bitmask, double *t_out, double *flux_out, double *flux_err_out,
                         int32_t *quality_out, int64_t *new_off);
int lk_fits_unpack_batch_dev(lk_handle *h, int B, const uint8_t *raw, const int64_t *raw_off_host, const int32_t *desc_host,
                             const int64_t *bitmask_host, double *t_out, double *flux_out, double *flux_err_out,
                             int32_t *quality_out, int64_t *new_off_host, void *stream);

/* lk_fits_unpack_cube: one target-pixel file -> what PLDCorrector reads from a TargetPixelFile: time, quality and the
 * float32 pixel cubes FLUX / FLUX_ERR / FLUX_BKG / ... of the cadences the reference keeps
 * (src/lightkurve/targetpixelfile.py:332, 372, 380, 386, 398: hdu[1].data[col][quality_mask]; quality_mask =
 * (QUALITY & bitmask) == 0, :2120-2122 for Kepler / K2; TESS additionally drops NaN times when a bitmask is set,
 * :2794-2801 — pass keep_nan_time = 1 for Kepler files and for bitmask 0; a kept cadence without a finite TIME is
 * returned as 0.0 like TargetPixelFile.time does, :333-335).  raw: the BINTABLE's bytes as in the file;
 * ncols <= 4 pixel columns of npix big-endian float32 each at byte offsets col_off[]; cubes_out: ncols x n_rows x npix
 * float32 (column c starts at c * n_rows * npix, its first *kept cadences are valid); kept: HOST scalar. */
int lk_fits_unpack_cube(lk_handle *h, const uint8_t *raw, int row_bytes, int n_rows, int off_time, int code_time,
                        int off_quality, int code_quality, int64_t bitmask, int keep_nan_time, int ncols,
                        const int32_t *col_off, int npix, double *t_out, int32_t *quality_out, float *cubes_out,
                        int64_t *kept);
int lk_fits_unpack_cube_dev(lk_handle *h, const uint8_t *raw, int row_bytes, int n_rows, int off_time, int code_time,
                            int off_quality, int code_quality, int64_t bitmask, int keep_nan_time, int ncols,
                            const int32_t *col_off_host, int npix, double *t_out, int32_t *quality_out, float *cubes_out,
                            int64_t *kept_host, void *stream);

/* LightCurve.create_transit_mask (:2967-3037): target b has planets [planet_off[b], planet_off[b+1]) of the HOST arrays
 * period / duration / transit_time [d]; mask[i] = 1 where |((t - t0 + P/2) % P) - P/2| < duration/2 for any of them. */
int lk_transit_mask_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const int32_t *planet_off,
                          const double *period, const double *duration, const double *transit_time, uint8_t *mask);
int lk_transit_mask_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const int32_t *planet_off,
                              const double *period, const double *duration, const double *transit_time, uint8_t *mask,
                              void *stream);
/* LightCurve.bin (:1558-1763) over astropy aggregate_downsample (astropy@4.3.1 timeseries/downsample.py:12-125), times
 * sorted.  Target b gets bins [bin_off[b], bin_off[b+1]) of the outputs; its bins start at time_bin_start[b] [d] and their
 * edges, in seconds relative to it, are edges_sec[0 .. n_bins_b] (HOST; numpy's cumsum of the bin size, shared by all
 * targets, n_edges entries).  A cadence at relative time r belongs to bin k when edges[k] < r <= edges[k+1] (r == 0: bin 0)
 * and r < edges[n_bins_b].  flux_out = nanmean; flux_err_out = sqrt(nansum(err^2) / #finite) where has_err[b] (the light
 * curve has at least one finite error), else nanstd of the flux in the bin; empty bins are NaN;
 * t_out = time_bin_start + (edges[k] + bin_size_sec / 2) / 86400. */
int lk_bin_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *flux, const double *flux_err,
                 const int64_t *bin_off, const double *time_bin_start, const double *edges_sec, int64_t n_edges,
                 double bin_size_sec, const uint8_t *has_err, double *t_out, double *flux_out, double *flux_err_out);
int lk_bin_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux,
                     const double *flux_err, const int64_t *bin_off, const double *time_bin_start, const double *edges_sec,
                     int64_t n_edges, double bin_size_sec, const uint8_t *has_err, double *t_out, double *flux_out,
                     double *flux_err_out, void *stream);

/* ---- nanmax / nanargmax over each row of a B x M float64 matrix (first maximum wins) ---------------- */
int lk_argmax_batch(lk_handle *h, int B, int64_t M, const double *x, double *max_out, int64_t *argmax_out);
int lk_argmax_batch_dev(lk_handle *h, int B, int64_t M, const double *x, double *max_out,
                        int64_t *argmax_out, void *stream);

/* ---- Box Least Squares (astropy run_bls semantics; bit-exact for time-sorted input) -------------------
 * t: t - min(t); y: y - median(y); ivar: 1/dy^2 (ones if no errors) — exactly what bls/core.py:304-327
 * hands to bls_fast.  period[nP], duration[nD] shared by all targets.  use_likelihood: 1 'likelihood', 0 'snr'.
 * out7: 7 x B x nP row-major: power, depth, depth_err, duration, transit_time(phase), depth_snr, log_likelihood. */
int lk_bls_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *y,
                 const double *ivar, const double *period, int64_t nP, const double *duration, int nD,
                 int oversample, int use_likelihood, double *out7);
int lk_bls_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *y,
                     const double *ivar, const double *period_host, const double *period_dev, int64_t nP,
                     const double *duration_host, int nD, int oversample, int use_likelihood, double *out7,
                     void *stream);

/* Host-only helper (no GPU): the longest period whose phase bins (period / (min duration / oversample) of them) the LDS
 * kernels hold for these durations and this oversample.  Longer periods are accepted all the same (astropy's bls_fast has no
 * limit): they run a global-memory kernel, bit-identical, ~100 x the cost per (target, period) — a multi-year baseline
 * searched with short durations. */
int lk_bls_max_period(const double *duration, int nD, int oversample, double *max_period);

/* ---- RegressionCorrector: Gaussian-prior weighted least squares, iterated with sigma clipping ----------
 * X: (sum N) x K row-major design matrix (same K for every target of the batch); y: flux; err: flux errors or
 * NULL (ones, regressioncorrector.py:157-158); cadence_mask: 1 = use the cadence, or NULL (all);
 * prior_mu / prior_sigma: B x K (sigma may be +inf) or both NULL; clip_sigma / niters as in .correct().
 * Outputs: w B x K coefficients of the last iteration; model = X w - median(X w) per target (sum N);
 * outlier: (sum N) bytes, 1 = clipped in some iteration (regressioncorrector.py:243-279). */
int lk_regress_batch(lk_handle *h, int B, const int64_t *n_off, int K, const double *X, const double *y,
                     const double *err, const uint8_t *cadence_mask, const double *prior_mu,
                     const double *prior_sigma, double clip_sigma, int niters, double *w, double *model,
                     uint8_t *outlier);
int lk_regress_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, int K, const double *X, const double *y,
                         const double *err, const uint8_t *cadence_mask, const double *prior_mu,
                         const double *prior_sigma, double clip_sigma, int niters, double *w, double *model,
                         uint8_t *outlier, void *stream);

/* The same fit with propagate_errors=True (regressioncorrector.py:183-185): w_cov (B x K x K, nullable) receives
 * inv(X^T S^-1 X + diag(1/prior_sigma^2)) of the LAST iteration's fit — RegressionCorrector.coefficients_err. */
int lk_regress_cov_batch(lk_handle *h, int B, const int64_t *n_off, int K, const double *X, const double *y,
                         const double *err, const uint8_t *cadence_mask, const double *prior_mu,
                         const double *prior_sigma, double clip_sigma, int niters, double *w, double *model,
                         uint8_t *outlier, double *w_cov);
int lk_regress_cov_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, int K, const double *X, const double *y,
                             const double *err, const uint8_t *cadence_mask, const double *prior_mu,
                             const double *prior_sigma, double clip_sigma, int niters, double *w, double *model,
                             uint8_t *outlier, double *w_cov, void *stream);

/* ---- LightCurve.flatten trend: masked, gap-segmented Savitzky-Golay + sigma-clip loop + linear re-interpolation
 * t (non-decreasing per target), flux (may hold NaN); mask: 1 = EXCLUDE the cadence from the fit (lightkurve's
 * `mask=` semantics) or NULL; window (odd), polyorder, break_tol (NaN = no gap splitting), niters, sigma as in
 * LightCurve.flatten (lightcurve.py:943).  trend: (sum N), what flatten divides flux and flux_err by;
 * fit_mask: (sum N) bytes or NULL, 1 = cadence survived every clip. */
int lk_savgol_trend_batch(lk_handle *h, int B, const int64_t *n_off, const double *t, const double *flux,
                          const uint8_t *mask, int window, int polyorder, double break_tol, int niters,
                          double sigma, double *trend, uint8_t *fit_mask);
int lk_savgol_trend_batch_dev(lk_handle *h, int B, const int64_t *n_off_host, const double *t, const double *flux,
                              const uint8_t *mask, int window, int polyorder, double break_tol, int niters,
                              double sigma, double *trend, uint8_t *fit_mask, void *stream);
/* Host-only helper (no GPU): the FIR taps (window doubles) and the two edge-refit operators
 * (2 x (window/2) x window doubles, left then right) the kernel uses == scipy savgol_coeffs / mode='interp'. */
int lk_savgol_design(int window, int polyorder, double *coeffs, double *edge);

/* ---- PLDCorrector.create_design_matrix (correctors/pldcorrector.py:186-287) for B same-shaped cutouts --------
 * pld_pix: B x N x P float32 pixel fluxes inside the PLD aperture (NULL / P = 0: no pixel block);
 * bkg_pix: B x N x Pb float32 background pixels; lc_flux: B x N float32 SAP flux; time: B x N;
 * knots: B x (n_inner + 2) = [min(t), interior knots (percentiles of t, as patsy's bs()), max(t)];
 * n_knots = n_inner + spline_degree + 1 spline columns (+1 constant).  normalize_bkg: divide background pixels
 * by their row sum.  X: B x N x K row-major with K = lk_pld_design_width(...):
 *   [ PCA(pixels/flux) | PCA(2-fold products) | ... | PCA(background) | B-splines | 1 ];  prior_sigma: B x K. */
int lk_pld_design_width(int P, int Pb, int pld_order, int pca_components, int n_knots);
int lk_pld_design_batch(lk_handle *h, int B, int N, int P, int Pb, const float *pld_pix, const float *bkg_pix,
                        const float *lc_flux, const double *time, const double *knots, int n_inner, int pld_order,
                        int pca_components, int n_knots, int spline_degree, int normalize_bkg, int K, double *X,
                        double *prior_sigma);
int lk_pld_design_batch_dev(lk_handle *h, int B, int N, int P, int Pb, const float *pld_pix, const float *bkg_pix,
                            const float *lc_flux, const double *time, const double *knots, int n_inner,
                            int pld_order, int pca_components, int n_knots, int spline_degree, int normalize_bkg,
                            int K, double *X, double *prior_sigma, void *stream);

/* ---- PLDCorrector.correct (correctors/pldcorrector.py:304-427) for B same-shaped cutouts in ONE call: the design
 * matrices above, RegressionCorrector.correct over them (prior_mu = 0, prior_sigma from the design, y / err = the SAP
 * light curve in float64, cadence_mask nullable) and — `spline_part`, nullable — the spline block's share of the model
 * X[:, K-(n_knots+1):] w[K-(n_knots+1):] that restore_trend adds back (pldcorrector.py:418-420, before its median is
 * removed).  Host pointers; X stays in device memory between the two stages.  pld_pix == bkg_pix (same pointer, P == Pb)
 * is uploaded once.  Outputs as lk_regress_batch: w B x K, model B x N (median removed), outlier B x N bytes. */
int lk_pld_correct_batch(lk_handle *h, int B, int N, int P, int Pb, const float *pld_pix, const float *bkg_pix,
                         const float *lc_flux, const double *time, const double *knots, int n_inner, int pld_order,
                         int pca_components, int n_knots, int spline_degree, int normalize_bkg, int K, const double *y,
                         const double *err, const uint8_t *cadence_mask, double clip_sigma, int niters, double *w,
                         double *model, uint8_t *outlier, double *spline_part);

/* ---- Standalone design-matrix operations (correctors/designmatrix.py) for B same-shaped matrices ----------------
 * lk_pca_batch          <- DesignMatrix.pca(nterms), designmatrix.py:252-282 (fbpca.pca(values, nterms) -> U): the first
 *                          k = nterms left singular vectors of the column-centred matrix; A: B x N x P row-major,
 *                          U: B x N x k.  A basis of the same subspace as the reference's (fbpca is a randomised range
 *                          finder: columns are defined up to sign, and up to rotation inside a degenerate cluster).
 *                          1 <= k <= min(48, P), P <= 4096.
 * lk_spline_basis_batch <- create_spline_matrix, designmatrix.py:952-997 (patsy bs(x, ..., include_intercept=True) - 1):
 *                          x: B x N; knots: B x (n_inner + 2) = [lower bound, interior knots, upper bound];
 *                          out: B x N x (n_inner + degree + 1).  (include_intercept=False drops column 0.)
 * lk_standardize_batch  <- DesignMatrix.standardize, designmatrix.py:215-250: per column, zeros are missing values;
 *                          (x - nanmedian) / nanstd of the rest, missing -> 0, constant columns unchanged. */
int lk_pca_batch(lk_handle *h, int B, int N, int P, int k, const double *A, double *U);
int lk_pca_batch_dev(lk_handle *h, int B, int N, int P, int k, const double *A, double *U, void *stream);
int lk_spline_basis_batch(lk_handle *h, int B, int N, const double *x, const double *knots, int n_inner, int degree,
                          double *out);
int lk_spline_basis_batch_dev(lk_handle *h, int B, int N, const double *x, const double *knots, int n_inner, int degree,
                              double *out, void *stream);
int lk_standardize_batch(lk_handle *h, int B, int N, int P, const double *A, double *out);
int lk_standardize_batch_dev(lk_handle *h, int B, int N, int P, const double *A, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LKHIP_H */
