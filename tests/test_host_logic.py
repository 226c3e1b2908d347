"""CPU tests of the host-side mirror (no GPU compute): grids, method switching, validation and error messages
follow the reference (tests/test_periodogram.py:43-161, 364-442, 491-515 are the models)."""
import logging
import warnings

import numpy as np
import pytest

from lightkurve_amd import LightCurve
from lightkurve_amd import periodogram as P


def make_lc(n=1000, seed=0):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, 50, n))
    return LightCurve(time=t, flux=1 + 1e-3 * np.sin(2 * np.pi * t / 3.0) + rng.normal(0, 1e-4, n), flux_err=1e-4)


def test_default_grid_matches_reference(golden):
    g = golden("ls_c1_default")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    plan = P._ls_plan(lc)
    assert plan["freq_unit"] == "1/d" and plan["norm"] == "lk_amplitude"
    assert len(plan["frequency"]) == len(g["frequency"])
    assert np.allclose(plan["frequency"], g["frequency"], rtol=1e-14, atol=0)
    assert np.isclose(plan["nyquist"], g["nyquist"], rtol=1e-14)
    assert P.exact_grid(plan["f_day"]) is not None
    plan = P._ls_plan(lc, normalization="psd")
    assert plan["freq_unit"] == "uHz" and len(plan["frequency"]) == len(g["psd_frequency_uhz"])
    assert np.allclose(plan["frequency"], g["psd_frequency_uhz"], rtol=1e-13)
    assert np.isclose(plan["nyquist"], g["psd_nyquist"], rtol=1e-13)
    # lk psd scale = 2/(N*os*fs) == 2*T/N in 1/uHz
    T = g["time"][-1] - g["time"][0]
    assert np.isclose(plan["scale"], 2 * T / len(g["time"]) / P.UHZ_PER_CPD, rtol=1e-13)


def test_frequency_period_grid_assignment():
    lc = make_lc()
    f = np.arange(1, 100) * 0.01
    plan = P._ls_plan(lc, frequency=f)
    assert np.array_equal(plan["frequency"], f) and plan["default_view"] == "frequency"
    per = np.arange(1, 101) * 0.5
    plan = P._ls_plan(lc, period=per, ls_method="fast")
    assert np.allclose(1 / plan["frequency"], per, rtol=1e-14) and plan["default_view"] == "period"
    assert plan["ls_method"] == "slow"          # irregular in frequency: fast -> slow (reference :933-946)
    assert P.exact_grid(plan["f_day"]) is None
    plan = P._ls_plan(lc, minimum_period=2.0, maximum_period=10.0)
    assert plan["frequency"][0] == pytest.approx(0.1) and plan["frequency"][-1] < 0.5


def test_ls_errors_and_warnings():
    lc = make_lc()
    with pytest.raises(ValueError, match="both frequency and period"):
        P._ls_plan(lc, frequency=[1, 2], period=[1, 2])
    with pytest.raises(ValueError, match="minimum_frequency cannot be larger"):
        P._ls_plan(lc, minimum_frequency=2.0, maximum_frequency=1.0)
    with pytest.raises(ValueError, match="minimum_period cannot be larger"):
        P._ls_plan(lc, minimum_period=2.0, maximum_period=1.0)
    with pytest.raises(ValueError, match="not supported"):
        P._ls_plan(lc, normalization="power")
    with pytest.raises(ValueError, match="not supported"):
        P._ls_plan(lc, ls_method="nonsense")
    with pytest.warns(P.LightkurveWarning, match="nterms"):
        plan = P._ls_plan(lc, nterms=2, ls_method="slow")
    assert plan["nterms"] == 1
    with pytest.warns(P.LightkurveWarning, match="deprecated"):
        P._ls_plan(lc, min_period=1.0)
    with pytest.raises(ValueError, match="not supported"):
        lc.to_periodogram(method="unknown")


def test_nan_flux_removed_before_planning():
    lc = make_lc()
    lc.flux[[3, 50]] = np.nan
    plan = P._ls_plan(lc)
    assert len(plan["flux"]) == len(lc) - 2 and np.isfinite(plan["flux"]).all()
    assert plan["trel"][0] == 0.0


def test_bls_plan_defaults_match_reference(golden):
    g = golden("bls_default")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    plan = P._bls_plan(lc, frequency_factor=200)
    assert len(plan["period"]) == len(g["period"]) and np.allclose(plan["period"], g["period"], rtol=1e-14)
    assert np.array_equal(plan["duration"], [0.05, 0.10, 0.15, 0.20, 0.25, 0.33])
    assert plan["oversample"] == 10 and plan["objective"] == "likelihood"
    g2 = golden("bls_2500")
    lc = LightCurve(time=g2["time"], flux=g2["flux"], flux_err=g2["flux_err"])
    plan = P._bls_plan(lc, period=g2["period"], duration=g2["duration"])
    assert np.array_equal(plan["t"], g2["raw_t"]) and np.array_equal(plan["y"], g2["raw_y"])
    assert np.array_equal(plan["ivar"], g2["raw_ivar"])


def test_bls_errors(caplog):
    lc = make_lc()
    with pytest.raises(ValueError, match="period"):
        P._bls_plan(lc, period=[1, 2, 3, np.nan, 4])           # reference tests/test_periodogram.py:434-442
    with pytest.raises(ValueError, match="duration"):
        P._bls_plan(lc, duration=[0.1, np.inf])
    with pytest.raises(ValueError, match="too large to evaluate"):
        P._bls_plan(lc, frequency_factor=1e-5)
    with pytest.raises(ValueError, match="maximum transit duration"):
        P._bls_plan(lc, period=[0.2, 1.0], duration=[0.3])
    with pytest.raises(ValueError, match="oversample"):
        P._bls_plan(lc, oversample=0)
    with pytest.raises(ValueError, match="Unrecognized method"):
        P._bls_plan(lc, objective="chi2")
    with caplog.at_level(logging.WARNING):
        P._bls_plan(lc, frequency_factor=0.02)
    assert "slow to evaluate" in caplog.text
    lc2 = LightCurve(time=lc.time, flux=lc.flux)      # no errors -> ivar = 1
    assert np.array_equal(P._bls_plan(lc2)["ivar"], np.ones(len(lc2)))


def test_periodogram_container_validation():
    with pytest.raises(ValueError, match="length greater than 1"):
        P.Periodogram([1.0], [1.0])
    with pytest.raises(ValueError, match="same length"):
        P.Periodogram([1.0, 2.0], [1.0])
    with pytest.raises(ValueError, match="units of 1/time"):
        P.Periodogram([1.0, 2.0], [1.0, 2.0], frequency_unit="kg")
    pg = P.Periodogram([1.0, 2.0, 4.0], [1.0, np.nan, 3.0])
    assert pg.max_power == 3.0 and pg.frequency_at_max_power == 4.0 and pg.period_at_max_power == 0.25


def test_logmedian_windows_and_box_kernel_reproduce_the_reference_bookkeeping(golden):
    """The host side of Periodogram.smooth: window tables and Box1DKernel taps.  Emulating the two device kernels with
    numpy on these tables must give lightkurve's smoothed spectrum exactly (medians are selections)."""
    g = golden("pg_smooth")
    f, p = g["frequency"], g["power"]
    for fw in (0.01, 0.05, 0.3):
        lo, hi, klo, khi = P._logmedian_windows(f, fw)
        assert np.all(lo[1:] >= lo[:-1]) and np.all(hi[1:] >= hi[:-1]) and np.all(hi > lo)
        assert np.all(khi >= klo)                                  # every frequency sits in at least one window
        med = np.array([np.nanmedian(p[a:b]) for a, b in zip(lo, hi)]) / (8.0 / 9.0) ** 3
        out = np.array([med[a:b + 1].sum() / (b - a + 1) for a, b in zip(klo, khi)])
        assert np.max(np.abs(out - g["logmedian_%g" % fw])) <= 1e-13 * np.max(g["logmedian_%g" % fw])
    assert np.allclose(P._box1d_kernel(5), np.full(5, 0.2)) and len(P._box1d_kernel(4)) == 5
    assert np.allclose(P._box1d_kernel(4), [0.125, 0.25, 0.25, 0.25, 0.125])
    with pytest.raises(NotImplementedError):
        P._logmedian_windows(f[::-1], 0.01)
    # the window ends come from binary searches + the reference's own predicate at the edges: identical to the reference
    # loop's masks (periodogram.py:274-275) also where grid points sit exactly on window boundaries, repeat, or leave
    # windows empty
    rng = np.random.default_rng(0)
    for ff, fw in ((10 ** (0.005 * np.arange(900)), 0.01), (10 ** (0.005 * np.arange(900)), 0.005),
                   (np.array([1.0, 1.0000001, 10, 10.5, 1000]), 0.02), (np.sort(rng.uniform(0.01, 50, 5000)), 0.05),
                   (np.sort(np.r_[rng.uniform(1, 2, 300), np.full(40, 1.5)]), 0.01)):
        lf, x0, lo_ref, hi_ref = np.log10(ff), np.log10(ff[0]), [], []
        while x0 < np.log10(ff[-1]):
            m = np.flatnonzero(np.abs(lf - x0) < fw)
            if m.size:
                lo_ref.append(m[0]), hi_ref.append(m[-1] + 1)
            x0 += 0.5 * fw
        lo, hi, _klo, _khi = P._logmedian_windows(ff, fw)
        assert np.array_equal(lo, lo_ref) and np.array_equal(hi, hi_ref), fw


def test_periodogram_bin_matches_reference(golden):
    g = golden("pg_misc")
    pg = P.Periodogram(g["frequency"], g["power"])
    for meth in ("mean", "median"):
        b = pg.bin(binsize=7, method=meth)
        assert np.allclose(b.frequency, g["bin_freq_" + meth], rtol=1e-13, atol=0)
        assert np.allclose(b.power, g["bin_power_" + meth], rtol=1e-13, atol=0)
    with pytest.raises(ValueError, match="binsize"):
        pg.bin(binsize=0)


def test_cbv_collection_layout():
    """CBVCorrector builds [selected CBVs, ext, Constant] with one prior width everywhere (reference :639-778)."""
    from lightkurve_amd.correctors import CBVCorrector, DesignMatrix
    lc = make_lc(200)
    cbvs = np.random.default_rng(0).normal(size=(200, 6))
    cor = CBVCorrector(lc, cbvs)
    dmc = cor._collection(np.array([1, 3, 9]), None)               # index 9 does not exist: dropped like the reference
    assert [m.name for m in dmc.matrices] == ["SingleScale", "Constant"] and dmc.X.shape == (200, 3)
    assert np.array_equal(dmc.X[:, :2], cbvs[:, [0, 2]]) and np.all(dmc.X[:, 2] == 1.0)
    ext = DesignMatrix(np.arange(200.0), columns=["t"], name="ext")
    assert cor._collection("ALL", ext).X.shape == (200, 8)
    with pytest.raises(ValueError):
        cor._collection(None, None)
    with pytest.raises(ValueError):
        CBVCorrector(lc, cbvs[:10])


def test_minimize_scalar_bounded_matches_scipy():
    """The restated bounded Brent driver of CBVCorrector.correct visits scipy's abscissae (scipy is importable here)."""
    scipy_opt = pytest.importorskip("scipy.optimize")
    from lightkurve_amd.correctors.cbvcorrector import minimize_scalar_bounded
    for f, bounds in ((lambda x: (x - 2.0) ** 2 + 0.3 * np.sin(5 * x), (0.0, 5.0)),
                      (lambda x: -np.exp(-(np.log10(x) - 1.0) ** 2), (1e-4, 1e4)),
                      (lambda x: abs(x - 1e3) ** 0.5, (1e-4, 1e4))):
        seen_a, seen_b = [], []
        ra = minimize_scalar_bounded(lambda x: (seen_a.append(x), f(x))[1], bounds, maxiter=100)
        rb = scipy_opt.minimize_scalar(lambda x: (seen_b.append(x), f(x))[1], method="Bounded", bounds=bounds,
                                       options={"maxiter": 100})
        assert seen_a == seen_b
        assert ra["x"] == rb.x and ra["fun"] == rb.fun and ra["nfev"] == rb.nfev


def test_underfit_metric_neighbors_properties():
    """Residual-correlation goodness (reference metrics.py:141-257) with hand-made neighbours: a target that shares a
    systematic with its neighbours scores low, white noise scores ~0.95 (the calibration point of the reference)."""
    from lightkurve_amd.correctors.metrics import underfit_metric_neighbors
    from lightkurve_amd.lightcurve import LightCurve
    rng = np.random.default_rng(3)
    n = 3000
    t = np.linspace(0, 27, n)
    common = np.sin(2 * np.pi * t / 3.0)
    neigh = 1e-3 * (common[:, None] + 0.2 * rng.standard_normal((n, 30)))
    bad = LightCurve(time=t, flux=1.0 + 1e-3 * common + 2e-4 * rng.standard_normal(n), flux_err=np.full(n, 2e-4))
    good = LightCurve(time=t, flux=1.0 + 2e-4 * rng.standard_normal(n), flux_err=np.full(n, 2e-4))
    m_bad, m_good = underfit_metric_neighbors(bad, neigh), underfit_metric_neighbors(good, neigh)
    assert m_bad < 0.1 and 0.85 < m_good <= 1.0
    noise = 1e-3 * rng.standard_normal((n, 40))
    assert abs(underfit_metric_neighbors(good, noise) - 0.95) < 0.05


def test_acf2d_host_side_smoothing_and_plan(golden):
    """Host half of estimate_numax_acf2d: default numaxs / window / spacing, window index arithmetic and the Gaussian
    smoothing of the metric (astropy convolve, boundary='extend'), against the reference's diagnostics."""
    from lightkurve_amd import seismology
    from lightkurve_amd.periodogram import Periodogram
    g = golden("acf2d")
    for tag in ("rg", "ms"):
        pg = Periodogram(g[tag + "_frequency"], g[tag + "_power"], frequency_unit="uHz")
        numaxs, ww, starts, W = seismology._plan(pg, None, None, None)
        assert np.array_equal(numaxs, g[tag + "_numaxs"]) and ww == float(g[tag + "_window_width"])
        assert W == g[tag + "_acf2d"].shape[0] and starts.min() >= 0 and starts.max() + W <= len(g[tag + "_power"])
        sm = seismology._gaussian_smooth_extend(g[tag + "_metric"], np.sqrt(len(numaxs)))
        assert np.allclose(sm, g[tag + "_metric_smooth"], rtol=1e-12, atol=0)


def test_designmatrix_validate_mirrors_the_reference():
    """correctors/designmatrix.py:306-349: LightkurveWarning for a low-rank matrix, ValueError for priors of the wrong
    shape and for prior widths <= 0 (the mirror used to pass a non-positive sigma on to the kernel's 1/sigma^2);
    sparse matrices skip the rank check by default."""
    import warnings
    from lightkurve_amd.correctors.designmatrix import DesignMatrix, LightkurveWarning, SparseDesignMatrix
    rng = np.random.default_rng(0)
    low = np.outer(rng.normal(size=30), np.ones(6))
    with pytest.warns(LightkurveWarning, match="low rank"):
        DesignMatrix(low).validate()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        DesignMatrix(low).validate(rank=False)
        SparseDesignMatrix(low).validate()
        DesignMatrix(rng.normal(size=(30, 4))).validate()
    with pytest.raises(ValueError, match="smaller than or equal to zero"):
        DesignMatrix(rng.normal(size=(30, 2)), prior_sigma=[1.0, 0.0]).validate()
    with pytest.raises(ValueError, match="smaller than or equal to zero"):
        DesignMatrix(rng.normal(size=(30, 2)), prior_sigma=[1.0, -2.0]).validate()
    dm = DesignMatrix(rng.normal(size=(30, 2)))
    dm.prior_mu = np.zeros(3)
    with pytest.raises(ValueError, match="prior_mu"):
        dm.validate()
    dm = DesignMatrix(rng.normal(size=(30, 2)))
    dm.prior_sigma = np.ones(1)
    with pytest.raises(ValueError, match="prior_sigma"):
        dm.validate()
    sp = DesignMatrix(np.arange(12.0).reshape(6, 2), columns=["a", "b"], prior_mu=[1, 2], prior_sigma=[3, 4]).split([2, 4])
    assert sp.shape == (6, 6) and sp.columns == ["a 1", "b 1", "a 2", "b 2", "a 3", "b 3"]
    assert np.array_equal(sp.values[:2, :2], [[0, 1], [2, 3]]) and np.all(sp.values[:2, 2:] == 0)
    assert np.array_equal(sp.prior_mu, [1, 2] * 3) and np.array_equal(sp.prior_sigma, [3, 4] * 3)


def test_aperture_sums_fast_path_is_the_reference_formula():
    """PixelCube._aperture_sums (shared by PLDCorrector and pld_correct_batch) against the formula of
    TargetPixelFile.extract_aperture_photometry (reference targetpixelfile.py:868-923) written out plainly — float32 sums in
    numpy's order for the gathered pixels, all-NaN and all-zero cadences -> NaN — with and without non-finite pixels."""
    from lightkurve_amd import synth
    from lightkurve_amd.correctors.pldcorrector import PixelCube
    t, flux, err, _ = synth.pld_cutout(4, 3, n=300, npix=9)

    def plain(c, ap):
        with np.errstate(all="ignore"):
            f = np.asarray(np.nansum(c.flux[:, ap], axis=1))
            f[~np.any(np.isfinite(c.flux[:, ap]), axis=1)] = np.nan
            f[np.all(c.flux == 0, axis=(1, 2))] = np.nan
            e = np.nansum(c.flux_err[:, ap] ** 2, axis=1) ** 0.5
        return f, e

    for case in range(4):
        fl, er = flux.copy(), err.copy()
        if case == 1:
            fl[10] = 0.0
        if case == 2:
            fl[20] = np.nan
            fl[30, 2, 3] = np.nan
        if case == 3:
            fl[5] = 0.0
            er[7, 1, 1] = np.nan
            fl[9, 0, 0] = np.inf
        c = PixelCube(t, fl, er)
        for ap in (np.ones((9, 9), bool), np.pad(np.ones((5, 5), bool), 2)):
            got, want = c._aperture_sums(ap), plain(c, ap)
            assert got[0].dtype == np.float32
            assert np.array_equal(got[0], want[0], equal_nan=True) and np.array_equal(got[1], want[1], equal_nan=True), case
