"""Runs under the conda interpreter that has astropy (spawned by tests/test_seams_gpu.py on the GPU box):
installs the HIP kernels behind astropy's own dispatch and compares with astropy's own CPU implementations."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from astropy.timeseries import BoxLeastSquares, LombScargle
    from astropy.timeseries.periodograms.bls import methods as bls_methods
    from lightkurve_amd import seams, synth
    installed = seams.install(lightkurve=False)
    out = {"installed": installed}
    # S1: astropy LombScargle.power(method='hip') vs astropy's exact methods, both normalisations, dy on/off
    t, y, e, _ = synth.ls_target(1, 3, 2500)
    f = synth.ls_frequency_grid(1500)
    errs = {}
    for dy in (None, e * np.linspace(0.5, 2, len(e))):
        ls = LombScargle(t - t[0], y, dy)
        for norm in ("standard", "psd", "model", "log"):        # astropy's four normalisations through the 'hip' seam
            ref = ls.power(f, method="cython", normalization=norm)
            got = ls.power(f, method="hip", normalization=norm)
            errs["%s_%s" % (norm, "dy" if dy is not None else "nody")] = float(np.max(np.abs(got - ref)) / np.max(ref))
            # the default method: GPU 'fast' vs astropy's own 'fast' (kept as 'fast_cpu')
            ref = ls.power(f, method="fast_cpu", normalization=norm)
            got = ls.power(f, method="fast", normalization=norm)
            ok = np.isfinite(ref)
            assert np.array_equal(ok, np.isfinite(got))
            errs["fast_%s_%s" % (norm, "dy" if dy is not None else "nody")] = float(
                np.max(np.abs(got[ok] - ref[ok])) / np.max(ref[ok]))
        per = np.linspace(0.1, 5, 400)     # irregular frequency grid
        errs["irregular_%s" % ("dy" if dy is not None else "nody")] = float(
            np.max(np.abs(ls.power(1 / per, method="hip") - ls.power(1 / per, method="slow"))))
    # round 6: every exact single-term NAME of astropy's registry is the HIP kernel; the originals are kept as '<name>_cpu'
    from astropy.timeseries.periodograms.lombscargle.implementations import main as ls_main0
    per = np.linspace(0.1, 5, 400)
    for name in ("slow", "cython", "scipy"):
        assert ls_main0.METHODS[name] is not ls_main0.METHODS[name + "_cpu"], name
        for norm in ("standard", "psd", "log", "model"):
            kw = dict(frequency=1 / per, normalization=norm)
            if name == "scipy":
                args = (t - t[0], y)
            else:
                args = (t - t[0], y, e)
                kw["fit_mean"] = True
            ref = ls_main0.METHODS[name + "_cpu"](*args, **kw)
            got = ls_main0.METHODS[name](*args, **kw)
            errs["%s_vs_cpu_%s" % (name, norm)] = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    ls_a = LombScargle(t - t[0], y, fit_mean=False, center_data=True)
    errs["auto_scipy_vs_cpu"] = float(np.max(np.abs(
        ls_a.power(1 / per, method="auto") - ls_main0.METHODS["scipy_cpu"](t - t[0], y, frequency=1 / per))))
    # multi-term: LombScargle(nterms=2).power(method='chi2') now runs on the GPU; astropy's own kept as 'chi2_cpu'
    ls = LombScargle(t - t[0], y, e, nterms=2)
    fm = f[f * (t[-1] - t[0]) >= 2.0]
    from astropy.timeseries.periodograms.lombscargle.implementations import main as ls_main
    # (astropy only lets the names 'chi2' / 'fastchi2' carry nterms != 1, so the kept original is called directly)
    ref = ls_main.METHODS["chi2_cpu"](t - t[0], y, e, frequency=fm, normalization="psd", nterms=2)
    got = ls.power(fm, method="chi2", normalization="psd")
    errs["chi2_nterms2"] = float(np.max(np.abs(got - ref)) / np.max(ref))
    # ... and method='fastchi2' (regular grid only), against astropy's own fastchi2
    fg = f[0] + (f[1] - f[0]) * np.arange(len(f))
    sel = fg * (t[-1] - t[0]) >= 2.0
    ref = ls_main.METHODS["fastchi2_cpu"](t - t[0], y, e, f0=fg[0], df=fg[1] - fg[0], Nf=len(fg), normalization="psd",
                                          nterms=2)
    got = ls.power(fg, method="fastchi2", normalization="psd")
    errs["fastchi2_nterms2"] = float(np.max(np.abs(got - ref)[sel]) / np.max(ref[sel]))
    out["ls_relerr"] = errs
    # S2: patched bls_fast vs astropy's compiled run_bls, bit for bit, through BoxLeastSquares.power
    t, y, e, _ = synth.bls_target(3, 7, 3000, cadence_days=10.0 / 1440.0)
    bls = BoxLeastSquares(t, y, e)
    period, duration = synth.bls_grid(300, 12, pmin=0.6, pmax=8.0)
    exact = {}
    for objective in ("likelihood", "snr"):
        got = bls.power(period, duration, objective=objective)
        bls_methods.bls_fast, saved = bls_methods._bls_fast_reference, bls_methods.bls_fast
        ref = bls.power(period, duration, objective=objective)
        bls_methods.bls_fast = saved
        exact[objective] = all(bool(np.array_equal(np.asarray(getattr(got, k)), np.asarray(getattr(ref, k))))
                               for k in ("power", "depth", "depth_err", "duration", "depth_snr", "log_likelihood"))
        exact[objective + "_tt"] = float(np.max(np.abs(np.asarray(got.transit_time) - np.asarray(ref.transit_time))))
    out["bls_bit_exact"] = exact
    # S3: the flatten seam's array-level function vs lightkurve's loop restated over LIVE scipy (savgol_filter of the conda
    # scipy, linear interp1d with extrapolation): trend to 1e-10 relative, same NaN pattern
    import scipy.signal
    from oracle import np_oracle as O
    O.savgol_filter = lambda x, w, p: scipy.signal.savgol_filter(x, w, p)
    t, y, e, _ = synth.ls_target(6, 2, 6000)
    y = y * (1 + 0.01 * np.sin(2 * np.pi * t / 5.0))
    y[[7, 3000]] = np.nan
    m = np.zeros(len(t), bool)
    m[1000:1100] = True
    s3 = {}
    for tag, kw in (("w101", dict(window_length=101)), ("w401_mask", dict(window_length=401, mask=m, niters=2, sigma=4)),
                    ("w51_nobreak", dict(window_length=51, break_tolerance=None))):
        got = seams.flatten_trend_hip(t, y, **kw)
        ref, _fm = O.flatten_trend(t, y, kw.get("window_length", 101), 2, kw.get("break_tolerance", 5), kw.get("niters", 3),
                                   kw.get("sigma", 3), mask=kw.get("mask"))
        ok = np.isfinite(ref)
        assert np.array_equal(ok, np.isfinite(got)), tag
        s3[tag] = float(np.max(np.abs(got[ok] - ref[ok]) / np.abs(ref[ok])))
    out["flatten_relerr"] = s3
    # S4: RegressionCorrector._fit_coefficients' arithmetic (one weighted ridge fit + covariance) vs numpy.linalg
    rng = np.random.default_rng(11)
    n, k = 3000, 40
    X = np.column_stack([rng.standard_normal((n, k - 1)), np.ones(n)])
    w_true = rng.normal(0, 1e-3, k)
    err = rng.uniform(0.5, 2, n) * 2e-4
    yy = 1 + X @ w_true + err * rng.standard_normal(n)
    cm = rng.random(n) > 0.1
    ps = np.where(np.arange(k) % 3 == 0, np.inf, 0.05)

    class _Fake:          # what _fit_coefficients_hip reads from a RegressionCorrector
        pass
    fake = _Fake()
    fake.lc = _Fake()
    fake.lc.flux, fake.lc.flux_err = yy, err
    fake.dmc = _Fake()
    fake.dmc.X = X
    w, cov = seams._fit_coefficients_hip(fake, cadence_mask=cm, prior_mu=np.zeros(k), prior_sigma=ps, propagate_errors=True)
    A = X[cm].T @ (X[cm] / err[cm, None] ** 2) + np.diag(1 / ps ** 2)
    wr = np.linalg.solve(A, X[cm].T @ (yy[cm] / err[cm] ** 2))
    cr = np.linalg.inv(A)
    out["fit_relerr"] = {"w": float(np.max(np.abs(w - wr)) / np.max(np.abs(wr))),
                         "cov": float(np.max(np.abs(cov - cr) / np.sqrt(np.outer(np.diag(cr), np.diag(cr)))))}
    print("SEAMS_RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
