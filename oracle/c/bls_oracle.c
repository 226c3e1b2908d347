/* ORACLE (test infrastructure only; never linked or called by the product path).
 *
 * CPU restatement of astropy's compiled BLS kernel `run_bls` (astropy/timeseries/periodograms/bls/bls.c,
 * reached from lightkurve at src/lightkurve/periodogram.py:1161-1169 via bls/core.py:234-330 ->
 * bls/methods.py:55-95).  The C source is NOT on disk here (only _impl.*.so); this follows the published
 * algorithm as restated in SURVEY.md Appendix B.2 and is pinned bit-for-bit against that .so by
 * oracle/gen_golden.py -> tests/golden/bls_*.npz (all 7 outputs, both objectives).
 * Build with -ffp-contract=off: every product/sum below must round separately.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static void objective_terms(double y_in, double y_out, double ivar_in, double ivar_out, int obj_flag,
                            double *objective, double *log_like, double *depth, double *depth_err,
                            double *depth_snr)
{
    if (obj_flag) {
        double arg = y_out - y_in;
        *log_like = 0.5 * ivar_in * arg * arg;
        *objective = *log_like;
    } else {
        *depth = y_out - y_in;
        *depth_err = sqrt(1.0 / ivar_in + 1.0 / ivar_out);
        *depth_snr = *depth / *depth_err;
        *objective = *depth_snr;
    }
}

/* returns 0 ok, 1 invalid period/duration (python: ValueError), 2 out of memory; out7 = 7 arrays of n_periods:
 * power, depth, depth_err, duration, transit_time(phase), depth_snr, log_likelihood */
int oracle_bls(int64_t N, const double *t, const double *y, const double *ivar,
               int64_t n_periods, const double *periods, int n_durations, const double *durations,
               int oversample, int obj_flag, double *out7)
{
    double *best_objective = out7, *best_depth = out7 + n_periods, *best_depth_err = out7 + 2 * n_periods,
           *best_duration = out7 + 3 * n_periods, *best_phase = out7 + 4 * n_periods,
           *best_depth_snr = out7 + 5 * n_periods, *best_log_like = out7 + 6 * n_periods;

    double max_period = periods[0], min_period = periods[0];
    for (int64_t k = 1; k < n_periods; ++k) {
        if (periods[k] < min_period) min_period = periods[k];
        if (periods[k] > max_period) max_period = periods[k];
    }
    if (min_period < DBL_EPSILON) return 1;
    double min_duration = durations[0], max_duration = durations[0];
    for (int k = 1; k < n_durations; ++k) {
        if (durations[k] < min_duration) min_duration = durations[k];
        if (durations[k] > max_duration) max_duration = durations[k];
    }
    if (max_duration > min_period || min_duration < DBL_EPSILON) return 1;

    double bin_duration = min_duration / ((double)oversample);
    int max_n_bins = (int)(ceil(max_period / bin_duration)) + oversample;
    double *mean_y = (double *)malloc(sizeof(double) * (size_t)(max_n_bins + 1));
    double *mean_ivar = (double *)malloc(sizeof(double) * (size_t)(max_n_bins + 1));
    if (!mean_y || !mean_ivar) { free(mean_y); free(mean_ivar); return 2; }

    double min_t = INFINITY, sum_y = 0.0, sum_ivar = 0.0;
    for (int64_t n = 0; n < N; ++n) {
        min_t = fmin(min_t, t[n]);
        sum_y += y[n] * ivar[n];
        sum_ivar += ivar[n];
    }

    for (int64_t p = 0; p < n_periods; ++p) {
        double period = periods[p];
        int n_bins = (int)(ceil(period / bin_duration)) + oversample;
        for (int n = 0; n <= n_bins; ++n) { mean_y[n] = 0.0; mean_ivar[n] = 0.0; }
        for (int64_t n = 0; n < N; ++n) {
            int ind = (int)(fabs(fmod(t[n] - min_t, period)) / bin_duration) + 1;
            mean_y[ind] += y[n] * ivar[n];
            mean_ivar[ind] += ivar[n];
        }
        for (int n = 1, ind = n_bins - oversample; n <= oversample; ++n, ++ind) {
            mean_y[ind] = mean_y[n];
            mean_ivar[ind] = mean_ivar[n];
        }
        for (int n = 1; n <= n_bins; ++n) {
            mean_y[n] += mean_y[n - 1];
            mean_ivar[n] += mean_ivar[n - 1];
        }
        double objective, log_like = 0, depth = 0, depth_err = 0, depth_snr = 0;
        best_objective[p] = -INFINITY;
        /* astropy allocates outputs with np.empty; periods with no admissible box keep garbage there.
         * We define them as 0 so the oracle is deterministic (never happens with real data). */
        best_depth[p] = best_depth_err[p] = best_duration[p] = best_phase[p] = 0.0;
        best_depth_snr[p] = best_log_like[p] = 0.0;
        for (int k = 0; k < n_durations; ++k) {
            int dur = (int)(round(durations[k] / bin_duration));
            int n_max = n_bins - dur;
            for (int n = 0; n <= n_max; ++n) {
                double y_in = mean_y[n + dur] - mean_y[n];
                double ivar_in = mean_ivar[n + dur] - mean_ivar[n];
                double y_out = sum_y - y_in;
                double ivar_out = sum_ivar - ivar_in;
                if ((ivar_in < DBL_EPSILON) || (ivar_out < DBL_EPSILON)) continue;
                y_in /= ivar_in;
                y_out /= ivar_out;
                objective_terms(y_in, y_out, ivar_in, ivar_out, obj_flag, &objective, &log_like, &depth,
                                &depth_err, &depth_snr);
                if (y_out >= y_in && objective > best_objective[p]) {
                    best_objective[p] = objective;
                    objective_terms(y_in, y_out, ivar_in, ivar_out, (obj_flag == 0), &objective, &log_like,
                                    &depth, &depth_err, &depth_snr);
                    best_depth[p] = depth;
                    best_depth_err[p] = depth_err;
                    best_depth_snr[p] = depth_snr;
                    best_log_like[p] = log_like;
                    best_duration[p] = dur * bin_duration;
                    best_phase[p] = fmod(n * bin_duration + 0.5 * best_duration[p] + min_t, period);
                }
            }
        }
    }
    free(mean_y); free(mean_ivar);
    return 0;
}
