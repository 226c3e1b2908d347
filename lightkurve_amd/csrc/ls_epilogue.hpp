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

// ---- multi-term least-squares periodogram from the harmonic trig sums (astropy chi2_impl.py / fastchi2_impl.py)
template <int NT>
struct Chi2Sums {
    double Sw[2 * NT], Cw[2 * NT], Sy[NT], Cy[NT];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int m = 0; m < 2 * NT; ++m) Sw[m] = Cw[m] = 0.0;
#pragma unroll
        for (int m = 0; m < NT; ++m) Sy[m] = Cy[m] = 0.0;
    }
    // fold one cadence: (a, b) = (cos, sin) of the fundamental phase, w = weight, wy = w (y - ybar)
    __device__ __forceinline__ void add(double a, double b, double w, double wy) {
        const double two_a = a + a;
        double cm = 1.0, sm = 0.0, c = a, s2 = b;
#pragma unroll
        for (int m = 0; m < 2 * NT; ++m) {
            Sw[m] = fma(w, s2, Sw[m]);
            Cw[m] = fma(w, c, Cw[m]);
            if (m < NT) {
                Sy[m] = fma(wy, s2, Sy[m]);
                Cy[m] = fma(wy, c, Cy[m]);
            }
            const double cn = fma(two_a, c, -cm), sn = fma(two_a, s2, -sm);
            cm = c;
            sm = s2;
            c = cn;
            s2 = sn;
        }
    }
    // X^T X (lower triangle) and X^T y from the sums, weights normalised to sum 1 (yws = sum w (y - ybar));
    // basis order: 0 = bias (cos 0), 2i-1 = sin i, 2i = cos i
    __device__ __forceinline__ void build(double yws, int fit_mean, double (&A)[2 * NT + 1][2 * NT + 1],
                                          double (&bv)[2 * NT + 1]) const {
        constexpr int D = 2 * NT + 1;
        auto CW = [&](int m) { return m == 0 ? 1.0 : Cw[m - 1]; };
        auto SW = [&](int m) { return m == 0 ? 0.0 : Sw[m - 1]; };
#pragma unroll
        for (int r = 0; r < D; ++r) {
            const int mr = (r + 1) >> 1;
            const bool rs = (r & 1) != 0;  // sine?
            bv[r] = r == 0 ? yws : (rs ? Sy[mr - 1] : Cy[mr - 1]);
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                const int mc = (c + 1) >> 1;
                const bool cs = (c & 1) != 0;
                const int dm = mr - mc, sm = mr + mc;  // mr >= mc because r >= c
                double v;
                if (rs && cs)
                    v = 0.5 * (CW(dm) - CW(sm));
                else if (!rs && !cs)
                    v = 0.5 * (CW(dm) + CW(sm));
                else if (rs && !cs)  // sin(mr) cos(mc): 0.5 (sign(mr - mc) Sw[|mr - mc|] + Sw[mr + mc])
                    v = 0.5 * ((dm > 0 ? SW(dm) : 0.0) + SW(sm));
                else  // cos(mr) sin(mc): 0.5 (sign(mc - mr) Sw[|mc - mr|] + Sw[mr + mc])
                    v = 0.5 * ((dm > 0 ? -SW(dm) : 0.0) + SW(sm));
                A[r][c] = v;
            }
        }
        if (!fit_mean) {  // drop the bias: identity row/column leaves the rest of the solve untouched
            A[0][0] = 1.0;
            bv[0] = 0.0;
#pragma unroll
            for (int r = 1; r < D; ++r) A[r][0] = 0.0;
        }
    }
    // (X^T y)^T (X^T X)^-1 (X^T y) by LU with partial pivoting (what np.linalg.solve does).  For the EXTIRPOLATED sums
    // of 'fastchi2' X^T X is only approximately a Gram matrix and can be slightly indefinite where the fit is ill posed
    // (f T < 1): Cholesky would return NaN there, the reference returns a finite number.
    __device__ __forceinline__ double solve_lu(double yws, int fit_mean) const {
        constexpr int D = 2 * NT + 1;
        double A[D][D], bv[D], b0[D];
        build(yws, fit_mean, A, bv);
#pragma unroll
        for (int r = 0; r < D; ++r) {
            b0[r] = bv[r];
#pragma unroll
            for (int c = r + 1; c < D; ++c) A[r][c] = A[c][r];
        }
        int perm[D];
#pragma unroll
        for (int r = 0; r < D; ++r) perm[r] = r;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            // pivot: largest |A[r][j]|, r >= j (first one wins, like LAPACK's idamax)
            int piv = j;
            double best = fabs(A[j][j]);
#pragma unroll
            for (int r = j + 1; r < D; ++r) {
                const double v = fabs(A[r][j]);
                if (v > best) {
                    best = v;
                    piv = r;
                }
            }
#pragma unroll
            for (int r = j + 1; r < D; ++r) {  // swap rows j and piv with static indices only (registers)
                const bool sw = piv == r;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    const double x = A[j][c], y2 = A[r][c];
                    A[j][c] = sw ? y2 : x;
                    A[r][c] = sw ? x : y2;
                }
                const double x = bv[j], y2 = bv[r];
                bv[j] = sw ? y2 : x;
                bv[r] = sw ? x : y2;
            }
            const double inv = 1.0 / A[j][j];
#pragma unroll
            for (int r = j + 1; r < D; ++r) {
                const double f = A[r][j] * inv;
#pragma unroll
                for (int c = j + 1; c < D; ++c) A[r][c] = fma(-f, A[j][c], A[r][c]);
                bv[r] = fma(-f, bv[j], bv[r]);
            }
        }
        (void)perm;
        // back substitution, then p = b0 . x  (b0 in the original order; x is in the original unknown order)
        double x[D];
#pragma unroll
        for (int r = D - 1; r >= 0; --r) {
            double v = bv[r];
#pragma unroll
            for (int c = r + 1; c < D; ++c) v = fma(-A[r][c], x[c], v);
            x[r] = v / A[r][r];
        }
        double p = 0.0;
#pragma unroll
        for (int r = 0; r < D; ++r) p = fma(b0[r], x[r], p);
        return p;
    }
    // the same quantity by Cholesky — for the EXACT sums X^T X is a true Gram matrix
    __device__ __forceinline__ double solve(double yws, int fit_mean) const {
        constexpr int D = 2 * NT + 1;
        double A[D][D], bv[D];
        build(yws, fit_mean, A, bv);
        // Cholesky A = L L^T (lower, in place), z = L^-1 b, power = |z|^2
        double p = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double d = A[j][j];
#pragma unroll
            for (int m = 0; m < j; ++m) d = fma(-A[j][m], A[j][m], d);
            const double inv = 1.0 / sqrt(d);
            double z = bv[j];
#pragma unroll
            for (int m = 0; m < j; ++m) z = fma(-A[j][m], bv[m], z);
            z *= inv;
            bv[j] = z;
            p = fma(z, z, p);
#pragma unroll
            for (int r = j + 1; r < D; ++r) {
                double v = A[r][j];
#pragma unroll
                for (int m = 0; m < j; ++m) v = fma(-A[r][m], A[j][m], v);
                A[r][j] = v * inv;
            }
        }
        return p;
    }
};

__device__ __forceinline__ double chi2_normalise(double p, int norm, double YY, double psd_factor, double nN,
                                                 double scale) {
    switch (norm) {
        case LK_NORM_STANDARD: return p / YY;
        case LK_NORM_PSD: return p * psd_factor;
        case LK_NORM_LK_AMPLITUDE: return sqrt(p * psd_factor) * sqrt(4.0 / nN);
        default: return p * psd_factor * scale;
    }
}

}  // namespace lk
