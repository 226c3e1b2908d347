"""TEST INFRASTRUCTURE, runs under the conda interpreter with the REAL lightkurve (PYTHONPATH=<shims>:/root/reference/src):
installs all four seams into lightkurve/astropy and checks that an unmodified lightkurve returns the same objects as
without them.

    seams_lk_worker.py compare <backend>     backend = "oracle" (CPU stand-in, wiring check) or "hip" (the GPU library)
    seams_lk_worker.py reftests <backend> <reference test files ...>     run the reference's own tests with the seams active
"""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def backend(name):
    if name == "oracle":
        import oracle_backend
        return oracle_backend
    return None          # the real thing: lightkurve_amd._capi


def val(x):
    x = getattr(x, "unmasked", x)
    return np.asarray(getattr(x, "value", x), dtype=float)


def relerr(a, b):
    a, b = val(a), val(b)
    ok = np.isfinite(b)
    assert np.array_equal(ok, np.isfinite(a)), "NaN pattern differs"
    return float(np.max(np.abs(a[ok] - b[ok])) / max(np.max(np.abs(b[ok])), 1e-300))


def run_all(lk):
    """Every seam through lightkurve's public API; returns {name: object}."""
    from lightkurve.correctors import DesignMatrix, PLDCorrector, RegressionCorrector
    from lightkurve_amd import synth
    import pandas as pd
    out = {}
    t, y, e, _ = synth.ls_target(1, 5, 2500, cadence_days=10.0 / 1440.0)
    lc = lk.LightCurve(time=t + 2000.0, flux=y, flux_err=e)
    out["ls_default"] = lc.to_periodogram()                                        # S1, default method 'fast'
    out["ls_psd"] = lc.to_periodogram(normalization="psd", freq_unit="microhertz")
    out["ls_slow_period_grid"] = lc.to_periodogram(period=np.linspace(0.5, 5, 300), ls_method="slow")
    # lightkurve's own switch (periodogram.py:933-946): the DEFAULT method on a period grid is rewritten 'fast' -> 'slow'
    out["ls_period_grid_default_method"] = lc.to_periodogram(minimum_period=0.4, maximum_period=6.0)
    out["ls_cython_period_grid"] = lc.to_periodogram(period=np.linspace(0.5, 5, 300), ls_method="cython", normalization="psd")
    out["ls_chi2_nterms2"] = lc.to_periodogram(nterms=2, ls_method="chi2", oversample_factor=2)
    out["ls_chi2_nterms5"] = lc.to_periodogram(nterms=5, ls_method="chi2", oversample_factor=1)   # 5..8 terms: exact sums
    # round 5: requests the FFT kernels do not cover stay on the device as exact sums (astropy called directly, as the
    # reference's users can: use_fft=False is astropy's own exact route; Mfft / nterms > 4 converge to the same sums)
    from astropy.timeseries import LombScargle
    ls = LombScargle(t, y)
    fgrid = dict(minimum_frequency=0.02, maximum_frequency=8.0, samples_per_peak=3)
    out["raw_ls_fast_nofft"] = np.asarray(ls.power(ls.autofrequency(**fgrid), method="fast", method_kwds=dict(use_fft=False)))
    out["raw_ls_fast_mfft8"] = np.asarray(ls.power(ls.autofrequency(**fgrid), method="fast",
                                               method_kwds=dict(trig_sum_kwds=dict(Mfft=8, oversampling=10))))
    # method='auto' picks 'cython' for a short grid, 'scipy' without errors and floating mean (main.py:96-104)
    out["raw_ls_auto_short_grid"] = np.asarray(ls.power(np.linspace(0.05, 4.0, 150), method="auto"))
    out["raw_ls_auto_scipy"] = np.asarray(LombScargle(t, y, fit_mean=False).power(np.linspace(0.05, 4.0, 150), method="auto",
                                                                              normalization="psd"))
    ls5 = LombScargle(t, y, nterms=5)
    out["raw_ls_fastchi2_nterms5"] = np.asarray(ls5.power(ls5.autofrequency(**fgrid), method="fastchi2"))
    tb, yb, eb, _ = synth.bls_target(3, 9, 2500, cadence_days=10.0 / 1440.0)
    lcb = lk.LightCurve(time=tb + 2000.0, flux=yb, flux_err=eb)
    out["bls"] = lcb.to_periodogram(method="bls", period=np.linspace(0.7, 8, 500), duration=[0.05, 0.1, 0.2])   # S2
    # periods beyond what the kernels' LDS plan holds for 0.05-d durations (46 d): the global-memory kernel, same bits
    tl, yl, el, _ = synth.bls_target(3, 10, 6000, cadence_days=30.0 / 1440.0)
    out["bls_long_periods"] = lk.LightCurve(time=tl + 2000.0, flux=yl, flux_err=el).to_periodogram(
        method="bls", period=np.linspace(2.0, 60.0, 120), duration=[0.05, 0.1])
    yf = y * (1 + 0.01 * np.sin(2 * np.pi * t / 7.0))
    yf[100] = np.nan
    lcf = lk.LightCurve(time=t + 2000.0, flux=yf, flux_err=e)
    flat, trend = lcf.flatten(window_length=101, return_trend=True)               # S3
    out["flatten"], out["flatten_trend"] = flat, trend
    m = np.zeros(len(t), bool)
    m[500:560] = True
    out["flatten_mask"] = lcf.flatten(window_length=51, mask=m, niters=2, sigma=4)
    rng = np.random.default_rng(3)
    n = len(t)
    X = np.column_stack([np.sin(2 * np.pi * t / p) for p in (1.3, 2.9, 7.7)] + [t / t[-1], np.ones(n)])
    yr = 1 + X[:, :4] @ np.array([3e-3, -2e-3, 1e-3, 4e-3]) + 3e-4 * rng.standard_normal(n)
    yr[rng.integers(0, n, 12)] += 0.02
    lcr = lk.LightCurve(time=t + 2000.0, flux=yr, flux_err=np.full(n, 3e-4))
    cm = np.ones(n, bool)
    cm[300:340] = False
    rc = RegressionCorrector(lcr)
    out["regress"] = rc.correct(DesignMatrix(pd.DataFrame(X), name="X", prior_sigma=np.array([np.inf, 0.01, 0.1, 1.0, np.inf])),
                                cadence_mask=cm)                                   # S4
    out["regress_coefficients"] = rc.coefficients
    out["regress_outliers"] = rc.outlier_mask
    out["regress_diag"] = rc.diagnostic_lightcurves["X"]
    # standalone design-matrix operations (round 4): DesignMatrix.pca / .standardize reach the GPU for ANY caller
    Ad = np.column_stack([np.sin(2 * np.pi * t / p) for p in (0.7, 1.3, 2.9, 7.7, 11.0)]) @ rng.normal(size=(5, 14)) \
        + 1e-3 * rng.standard_normal((n, 14)) + rng.normal(size=14)
    dmA = DesignMatrix(pd.DataFrame(Ad), name="A")
    U = dmA.pca(4).values
    out["dm_pca_projector"] = U @ U.T @ Ad[:, :3]              # invariant under sign flips / rotations of the basis
    # the spline builders (round 5): patsy's bs() / the reference's Cox-de Boor recursion vs lk_spline_basis_batch
    from lightkurve.correctors.designmatrix import create_spline_matrix, create_sparse_spline_matrix
    out["spline_dense"] = create_spline_matrix(t, n_knots=14, degree=3).values
    out["spline_dense_knots"] = create_spline_matrix(t, knots=[2.0, 5.5, 9.0], degree=2, include_intercept=False).values
    out["spline_sparse"] = np.asarray(create_sparse_spline_matrix(t, n_knots=12, degree=3).X.todense())
    Sd = Ad.copy()
    Sd[rng.random(Sd.shape) < 0.03] = 0.0
    Sd[:, 2] = 1.5
    out["dm_standardize"] = DesignMatrix(pd.DataFrame(Sd), name="S").standardize().values
    # SURVEY 8(f) rows through the reference's own classes (round 6): Periodogram.smooth / .flatten, estimate_cdpp, the
    # over-fitting metric
    pg_psd = lc.to_periodogram(normalization="psd", minimum_frequency=5, maximum_frequency=2500)
    out["pg_smooth_box"] = pg_psd.smooth(method="boxkernel", filter_width=40.0)
    out["pg_smooth_logmedian"] = pg_psd.smooth(method="logmedian", filter_width=0.05)
    out["pg_flatten_snr"] = pg_psd.flatten()
    cd = [lcf.remove_nans().estimate_cdpp(), lcr.estimate_cdpp(transit_duration=7, savgol_window=51)]
    assert all(str(getattr(c, "unit", None)) == "ppm" for c in cd), cd          # the reference returns a Quantity in ppm
    out["cdpp"] = np.asarray([c.value for c in cd])
    from lightkurve.correctors.metrics import overfit_metric_lombscargle
    np.random.seed(42)
    lcc = lcr.copy()
    lcc.flux = lcc.flux + 2e-4 * np.sin(2 * np.pi * val(lcr.time.value) / 0.37) * lcr.flux.unit
    out["overfit_metric"] = np.asarray([overfit_metric_lombscargle(lcr, lcc, n_samples=4)])
    ref_data = os.path.join(os.environ.get("LK_REFERENCE_ROOT", "/root/reference"),
                            "tests/data/synthetic/synthetic-k2-sinusoid.targ.fits.gz")
    if os.path.exists(ref_data):
        tpf = lk.read(ref_data)
        pld = PLDCorrector(tpf)
        out["pld"] = pld.correct(pld_order=2, pca_components=8, pld_aperture_mask="all")   # S4: design matrix + regression
        out["pld_outliers"] = pld.outlier_mask
        out["pld_blocks"] = [mm.name for mm in pld.design_matrix_collection.matrices]
        plds = PLDCorrector(tpf)
        out["pld_sparse"] = plds.correct(pld_order=2, pca_components=8, pld_aperture_mask="all", sparse=True)
        out["pld_sparse_blocks"] = [type(mm).__name__ for mm in plds.design_matrix_collection.matrices]
    return out


def compare(bname):
    import lightkurve as lk
    from lightkurve_amd import seams
    if os.environ.get("LK_SEAMS_LOG"):        # show the seams' "fell back to the CPU implementation" notes
        import logging
        logging.basicConfig(level=os.environ["LK_SEAMS_LOG"], format="%(name)s %(levelname)s %(message)s")
        logging.getLogger().setLevel(logging.WARNING)
        logging.getLogger("lightkurve_amd.seams").setLevel(os.environ["LK_SEAMS_LOG"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = run_all(lk)
        installed = seams.install(backend=backend(bname))
        try:
            np.random.seed(0)
            got = run_all(lk)
        finally:
            seams.uninstall()
        again = run_all(lk)                                  # uninstall really restores the reference path
    res = {"installed": installed, "errors": {}, "types": {}}
    # ls_fast_mfft8 / ls_fastchi2_nterms5: the exact sums against astropy's own extirpolated approximation of them
    tol = {"pld": 1e-6, "pld_sparse": 1e-6, "dm_pca_projector": 1e-6, "raw_ls_fast_mfft8": 1e-5, "raw_ls_fastchi2_nterms5": 5e-3,
           "spline_dense": 1e-12, "spline_dense_knots": 1e-12, "spline_sparse": 1e-12}
    for k, r in ref.items():
        g = got[k]
        assert type(g) is type(r), (k, type(g), type(r))
        res["types"][k] = type(g).__name__
        if k.startswith("pg_"):
            assert g.frequency.unit == r.frequency.unit and g.power.unit == r.power.unit, k
            assert np.array_equal(val(g.frequency), val(r.frequency)), k
            res["errors"][k] = relerr(g.power, r.power)
        elif k.startswith("ls_") or k.startswith("bls"):
            assert g.frequency.unit == r.frequency.unit and g.power.unit == r.power.unit, k
            assert np.array_equal(val(g.frequency), val(r.frequency)), k
            res["errors"][k] = relerr(g.power, r.power)
            assert g.default_view == r.default_view
            if k.startswith("bls"):
                for attr in ("duration", "depth", "snr", "transit_time"):
                    a, b = getattr(g, attr), getattr(r, attr)
                    assert np.array_equal(val(a), val(b)), (k, attr)          # bit-exact BLS through the seam
                assert val(g.period_at_max_power) == val(r.period_at_max_power)
                st_g, st_r = g.compute_stats(), r.compute_stats()             # _BLS_object still works
                assert np.allclose(val(st_g["depth"][0]), val(st_r["depth"][0]))
            else:
                assert val(g.frequency_at_max_power) == val(r.frequency_at_max_power), k
                assert g._LS_object is not None
        elif hasattr(r, "flux"):
            assert g.flux.unit == r.flux.unit, k
            assert np.array_equal(val(g.time.value), val(r.time.value)), k
            res["errors"][k] = relerr(g.flux, r.flux)
            res["errors"][k + "_err"] = relerr(g.flux_err, r.flux_err) if np.any(np.isfinite(val(r.flux_err))) else 0.0
            assert dict(g.meta).get("NORMALIZED") == dict(r.meta).get("NORMALIZED"), k
        elif k.endswith("outliers"):
            assert np.array_equal(np.asarray(g), np.asarray(r)), k
        elif k in ("pld_blocks", "pld_sparse_blocks"):
            assert g == r
        else:
            res["errors"][k] = float(np.max(np.abs(np.asarray(g) - np.asarray(r))) / np.max(np.abs(np.asarray(r))))
    for k, v in res["errors"].items():
        assert v < tol.get(k.replace("_err", ""), 1e-8), (k, v)
    # after uninstall the reference path is bit-for-bit back
    for k in ("ls_default", "bls"):
        assert np.array_equal(val(again[k].power), val(ref[k].power)), k
    assert np.array_equal(val(again["flatten"].flux), val(ref["flatten"].flux), equal_nan=True)
    if bname == "oracle":
        import oracle_backend
        res["calls"] = sorted(set(oracle_backend.CALLS))
    else:
        from lightkurve_amd import _capi
        res["library"] = _capi.LIB_PATH
    print("SEAMS_LK_RESULT " + json.dumps(res))


def reftests(bname, files):
    import pytest
    from lightkurve_amd import seams
    seams.install(backend=backend(bname))
    rc = pytest.main(list(files) + ["-q", "-x", "-p", "no:cacheprovider", "-W", "ignore"])
    print("SEAMS_LK_REFTESTS rc=%d" % int(rc))
    sys.exit(int(rc))


if __name__ == "__main__":
    if sys.argv[1] == "compare":
        compare(sys.argv[2])
    else:
        reftests(sys.argv[2], sys.argv[3:])
