/* ORACLE (test infrastructure only; never linked or called by the product path).
 *
 * CPU restatement of the exact generalised Lomb-Scargle periodogram that lightkurve obtains from
 * astropy: the closed-form floating-mean GLS of astropy .../lombscargle/implementations/fast_impl.py:74-131
 * with the trig sums evaluated EXACTLY (the use_fft=False branch of .../implementations/utils.py:154-156),
 * i.e. the arithmetic of astropy's 'slow' (slow_impl.py:52-118), 'cython' and 'chi2' methods, followed by
 * lightkurve's own normalisation (src/lightkurve/periodogram.py:969-975).
 *
 * Pinned against the reference itself (astropy 4.3.1 driven through /root/reference/src lightkurve) by
 * oracle/gen_golden.py -> tests/golden/ls_*.npz, checked in tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* normalization: 0 'standard', 1 astropy 'psd', 2 lightkurve 'amplitude', 3 lightkurve 'psd' (x lk_scale) */
int oracle_ls_power(int64_t n, const double *t, const double *y_in, const double *dy,
                    int64_t m, const double *freq, int fit_mean, int center_data,
                    int normalization, double lk_scale, double *power)
{
    if (n <= 0 || m < 0) return 1;
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    double *y = (double *)malloc(sizeof(double) * (size_t)n);
    if (!w || !y) { free(w); free(y); return 2; }
    double wsum = 0.0;
    for (int64_t i = 0; i < n; ++i) { w[i] = dy ? 1.0 / (dy[i] * dy[i]) : 1.0; wsum += w[i]; }
    for (int64_t i = 0; i < n; ++i) w[i] /= wsum;
    /* weighted mean taken about y[0]: algebraically np.dot(w, y) (fast_impl.py:79), but a constant light
     * curve then centres to EXACTLY zero => power exactly 0, the behaviour tests/test_periodogram.py:445-457
     * pins for the reference (whose BLAS dot happens to be exact there). */
    double ybar = 0.0;
    if (fit_mean || center_data) {
        for (int64_t i = 0; i < n; ++i) ybar += w[i] * (y_in[i] - y_in[0]);
        ybar += y_in[0];
    }
    double YY = 0.0;
    for (int64_t i = 0; i < n; ++i) { y[i] = y_in[i] - ybar; YY += w[i] * y[i] * y[i]; }

    const double twopi = 6.283185307179586476925286766559;
    for (int64_t j = 0; j < m; ++j) {
        double om = twopi * freq[j];
        double Sh = 0, Ch = 0, S = 0, C = 0, S2 = 0, C2 = 0;
        for (int64_t i = 0; i < n; ++i) {
            double ph = om * t[i];
            double s = sin(ph), c = cos(ph);
            double wy = w[i] * y[i];
            Sh += wy * s; Ch += wy * c;
            S += w[i] * s; C += w[i] * c;
            S2 += w[i] * sin(2.0 * ph); C2 += w[i] * cos(2.0 * ph);
        }
        double tan2;
        if (fit_mean) tan2 = (S2 - 2.0 * S * C) / (C2 - (C * C - S * S));
        else tan2 = S2 / C2;
        double C2w = 1.0 / sqrt(1.0 + tan2 * tan2);
        double S2w = tan2 * C2w;
        double Cw = sqrt(0.5) * sqrt(1.0 + C2w);
        double sgn = (S2w > 0) - (S2w < 0);
        double Sw = sqrt(0.5) * sgn * sqrt(1.0 - C2w);
        double YC = Ch * Cw + Sh * Sw;
        double YS = Sh * Cw - Ch * Sw;
        double CC = 0.5 * (1.0 + C2 * C2w + S2 * S2w);
        double SS = 0.5 * (1.0 - C2 * C2w - S2 * S2w);
        if (fit_mean) {
            double a = C * Cw + S * Sw, b = S * Cw - C * Sw;
            CC -= a * a; SS -= b * b;
        }
        double p = YC * YC / CC + YS * YS / SS;
        double psd_factor = dy ? 0.5 * wsum : 0.5 * (double)n;
        switch (normalization) {
            case 0: p /= YY; break;
            case 1: p *= psd_factor; break;
            case 2: p = sqrt(p * psd_factor) * sqrt(4.0 / (double)n); break;
            case 3: p = p * psd_factor * lk_scale; break;
            default: free(w); free(y); return 1;
        }
        power[j] = p;
    }
    free(w); free(y);
    return 0;
}
