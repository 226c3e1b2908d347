// ls_epilogue.hpp — closed-form floating-mean GLS power from the six trig sums
// (astropy lombscargle/implementations/fast_impl.py:93-131) + the normalisations of lkhip.h.
// Shared by the exact direct-sum kernels (ls.hip) and the FFT path (lsfast.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/lkhip.h"

namespace lk {

__device__ __forceinline__ double gls_power_sums(double Sh, double Ch, double S, double C, double S2, double C2,
                                                 int fit_mean, int norm, double YY, double psd_factor, double nN,
                                                 double scale) {
    double tan2;
    if (fit_mean)
        tan2 = (S2 - 2.0 * S * C) / (C2 - (C * C - S * S));
    else
        tan2 = S2 / C2;
    const double C2w = 1.0 / sqrt(1.0 + tan2 * tan2);
    const double S2w = tan2 * C2w;
    const double Cw = sqrt(0.5) * sqrt(1.0 + C2w);
    const double sgn = (S2w > 0.0) ? 1.0 : ((S2w < 0.0) ? -1.0 : 0.0);
    const double Sw = sqrt(0.5) * sgn * sqrt(1.0 - C2w);
    const double YC = Ch * Cw + Sh * Sw;
    const double YS = Sh * Cw - Ch * Sw;
    double CC = 0.5 * (1.0 + C2 * C2w + S2 * S2w);
    double SS = 0.5 * (1.0 - C2 * C2w - S2 * S2w);
    if (fit_mean) {
        const double a = C * Cw + S * Sw, bq = S * Cw - C * Sw;
        CC -= a * a;
        SS -= bq * bq;
    }
    double p = YC * YC / CC + YS * YS / SS;
    switch (norm) {
        case LK_NORM_STANDARD: p /= YY; break;
        case LK_NORM_PSD: p *= psd_factor; break;
        case LK_NORM_LK_AMPLITUDE: p = sqrt(p * psd_factor) * sqrt(4.0 / nN); break;
        default: p = p * psd_factor * scale; break;
    }
    return p;
}

}  // namespace lk
