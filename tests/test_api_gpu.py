"""GPU: the reference-shaped API (LightCurve.to_periodogram -> Periodogram objects) against golden vectors
made by the reference itself.  Written to read like the reference's tests/test_periodogram.py."""
import numpy as np
import pytest

from lightkurve_amd import BoxLeastSquaresPeriodogram, LightCurve, LombScarglePeriodogram

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b))


def test_lombscargle_default_grid_like_reference(golden):
    g = golden("ls_c1_default")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    pg = lc.to_periodogram()                       # defaults: amplitude, oversample 5, 1/d, ls_method="fast"
    assert isinstance(pg, LombScarglePeriodogram) and pg.frequency_unit == "1/d" and pg.ls_method == "fast"
    assert relmax(pg.power, g["amp_fast"]) < 1e-9           # == lightkurve's own default output
    assert abs(pg.period_at_max_power - g["true_period"]) / g["true_period"] < 0.05
    pg = lc.to_periodogram(ls_method="slow")       # exact methods
    assert relmax(pg.power, g["amp_slow"]) < 1e-9
    assert abs(pg.period_at_max_power - g["period_at_max_power"]) < 1e-12
    pg = lc.to_periodogram(normalization="psd", ls_method="slow")    # psd: uHz, oversample 1
    assert pg.frequency_unit == "uHz" and relmax(pg.power, g["psd_slow"]) < 1e-9
    # the reference's own default ('fast', an approximation) is ~1e-3 away from its exact methods
    assert 1e-6 < relmax(lc.to_periodogram().power, g["amp_slow"]) < 5e-3


def test_lombscargle_grids_nan_float32_dy(golden):
    g = golden("ls_tess3000")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    pg = lc.to_periodogram(frequency=g["frequency"], ls_method="slow")
    assert relmax(pg.power, g["amp_slow"]) < 1e-9 and pg.max_power == pytest.approx(g["max_power"], rel=1e-10)
    assert pg.frequency_at_max_power == g["frequency_at_max_power"]
    pg = lc.to_periodogram(frequency=g["frequency_uhz"], normalization="psd", ls_method="cython")
    assert relmax(pg.power, g["psd_slow"]) < 1e-9
    ok = np.isfinite(g["psd_fast"])
    pgf = lc.to_periodogram(frequency=g["frequency_uhz"], normalization="psd")      # default method
    assert np.array_equal(np.isfinite(pgf.power), ok) and relmax(pgf.power[ok], g["psd_fast"][ok]) < 1e-9
    g = golden("ls_nan_period_grid")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    pg = lc.to_periodogram(period=g["period"], ls_method="fast")
    assert pg.ls_method == "slow" == str(g["ls_method"]) and pg.default_view == "period"
    assert relmax(pg.power, g["amp"]) < 1e-9
    g = golden("ls_dy")
    lc = LightCurve(time=g["time"], flux=g["flux"])
    assert relmax(lc.to_periodogram(frequency=g["frequency"], dy=g["dy"], ls_method="slow").power, g["amp"]) < 1e-9
    assert relmax(lc.to_periodogram(frequency=g["frequency"] * 1e6 / 86400, normalization="psd", dy=g["dy"],
                                    ls_method="slow").power, g["psd"]) < 1e-9


def test_masked_nan_flux_gives_zero_power_not_nan():
    """reference tests/test_periodogram.py:445-457"""
    lc = LightCurve(time=[1, 2, 3, 4], flux=[1., np.nan, 1., 1.])
    pg = lc.to_periodogram()
    assert not np.isnan(pg.power).all()
    assert (pg.power == 0).all()


def test_bls_like_reference(golden):
    g = golden("bls_2500")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    for objective in ("likelihood", "snr"):
        pg = lc.to_periodogram(method="bls", period=g["period"], duration=g["duration"], objective=objective)
        assert isinstance(pg, BoxLeastSquaresPeriodogram) and pg.default_view == "period"
        assert np.array_equal(pg.power, g[objective + "_power"])
        assert np.array_equal(pg.depth, g[objective + "_depth"])
        assert np.array_equal(pg.duration, g[objective + "_duration"])
        assert np.array_equal(pg.snr, g[objective + "_depth_snr"])
        assert np.allclose(pg.transit_time, g[objective + "_transit_time"], rtol=0, atol=1e-9)
        assert pg.period_at_max_power == g[objective + "_period_at_max_power"]      # bit-exact best-period index
    g = golden("bls_default")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    pg = lc.to_periodogram(method="bls", frequency_factor=200)
    assert np.array_equal(pg.period, g["period"]) or np.allclose(pg.period, g["period"], rtol=1e-14)
    assert np.array_equal(pg.power, g["power"]) and pg.period_at_max_power == pytest.approx(g["period_at_max_power"])
    g = golden("bls_noerr")
    lc = LightCurve(time=g["time"], flux=g["flux"])
    pg = lc.to_periodogram(method="bls", period=g["period"], duration=[0.1, 0.2])
    assert np.array_equal(pg.power, g["power"])


def test_bls_then_fold_recovers_transit(golden):
    """folded-flux parity output: fold at the GPU's best period == fold at the reference's best period."""
    g = golden("bls_2500")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    pg = lc.to_periodogram(method="bls", period=g["period"], duration=g["duration"])
    f1 = lc.fold(period=pg.period_at_max_power, epoch_time=pg.transit_time_at_max_power)
    f2 = lc.fold(period=g["likelihood_period_at_max_power"],
                 epoch_time=g["likelihood_transit_time"][np.argmax(g["likelihood_power"])])
    assert np.array_equal(f1.flux, f2.flux) and np.allclose(f1.time, f2.time, atol=1e-9)
    in_tr = np.abs(f1.time) < 0.5 * pg.duration_at_max_power
    assert np.median(f1.flux[in_tr]) < np.median(f1.flux[~in_tr]) - 0.5 * pg.depth_at_max_power


def test_batch_entry_points_match_per_target_calls():
    from lightkurve_amd import batch, synth
    lcs = []
    for i, n in enumerate([900, 2000, 350, 1200]):
        t, y, e, _ = synth.bls_target(5, i, n, cadence_days=10.0 / 1440.0)
        lcs.append(LightCurve(time=t + 100.0 * i, flux=y, flux_err=e))
    f = 0.02 * (1 + np.arange(1500))
    for meth in ("fast", "slow"):
        P = batch.lombscargle_batch(lcs, f, ls_method=meth)
        for b, lc in enumerate(lcs):
            ref = lc.to_periodogram(frequency=f, ls_method=meth).power
            if meth == "slow":
                assert np.array_equal(P[b], ref)
            else:   # grid cells are filled by atomics: summation order (not the value) varies run to run
                assert np.allclose(P[b], ref, rtol=1e-9, atol=1e-12 * np.nanmax(ref), equal_nan=True)
    mx, am = batch.periodogram_peaks(P)
    assert np.array_equal(am, np.argmax(P, axis=1)) and np.array_equal(mx, P.max(axis=1))
    period = np.linspace(0.7, 5.0, 120)
    R = batch.bls_batch(lcs, period, [0.05, 0.1, 0.2])
    for b, lc in enumerate(lcs):
        pg = lc.to_periodogram(method="bls", period=period, duration=[0.05, 0.1, 0.2])
        assert np.array_equal(R[b, 0], pg.power) and np.array_equal(R[b, 4], pg.transit_time)


def test_empty_and_degenerate_batches():
    """Edge cases of every entry point: empty batches return empty results, one-cadence / one-frequency inputs work,
    bad arguments raise ValueError (status LK_EINVAL) instead of crashing."""
    from lightkurve_amd import _capi
    z = np.zeros(0)
    assert _capi.ls_power_batch(z, z, [0], f0=1.0, df=1.0, M=5).shape == (0, 5)
    assert _capi.ls_fast_batch(z, z, [0], f0=1.0, df=1.0, M=5).shape == (0, 5)
    assert _capi.ls_power_batch(z, z, [0], f0=1.0, df=1.0, M=5, nterms=3).shape == (0, 5)
    ph, order, cols = _capi.fold_batch(z, [0], z, z, columns=(z,))
    assert ph.size == 0 and order.size == 0 and cols[0].size == 0
    assert _capi.pg_boxsmooth_batch(np.zeros((0, 7)), np.ones(3) / 3).shape == (0, 7)
    # a single cadence / a single frequency
    p = _capi.ls_power_batch(np.array([0.0, 1.0, 2.5]), np.array([1.0, 2.0, 0.5]), [0, 3], frequency=[0.3])
    assert p.shape == (1, 1) and np.isfinite(p).all()
    ph, order, _ = _capi.fold_batch(np.array([5.0]), [0, 1], 2.0, 4.5)
    assert ph.tolist() == [0.5] and order.tolist() == [0]
    # a target with zero cadences inside a fold batch is allowed (nothing to sort)
    ph, order, _ = _capi.fold_batch(np.array([1.0, 2.0]), [0, 0, 2], [3.0, 3.0], [0.0, 0.0])
    assert ph.shape == (2,) and sorted(order.tolist()) == [0, 1]
    with pytest.raises(ValueError):
        _capi.ls_power_batch(np.arange(4.0), np.ones(4), [0, 4], f0=1.0, df=-1.0, M=5)          # negative step
    with pytest.raises(ValueError):
        _capi.ls_power_batch(np.arange(4.0), np.ones(4), [0, 2, 2, 4], f0=1.0, df=1.0, M=5)     # empty light curve
    with pytest.raises(ValueError):
        _capi.pg_logmedian_batch(np.ones((1, 4)), [0], [9], [0, 0, 0, 0], [0, 0, 0, 0])         # window outside the grid


def test_periodogram_bin_and_ls_model(golden):
    """Result-type helpers (row A15): Periodogram.bin and LombScarglePeriodogram.model against lightkurve's own outputs;
    the model's least-squares fit runs on the GPU regression path.  Tolerance 1e-9 relative."""
    from lightkurve_amd.lightcurve import LightCurve
    from lightkurve_amd.periodogram import Periodogram
    g = golden("pg_misc")
    pg = Periodogram(g["frequency"], g["power"])
    for meth in ("mean", "median"):
        b = pg.bin(binsize=7, method=meth)
        assert np.allclose(b.frequency, g["bin_freq_" + meth], rtol=1e-13, atol=0)
        assert np.allclose(b.power, g["bin_power_" + meth], rtol=1e-13, atol=0)
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    pgl = lc.to_periodogram(ls_method="slow", oversample_factor=3)
    m = pgl.model(lc.time)
    assert np.max(np.abs(m.flux - g["model_default"])) < 1e-9
    m = pgl.model(g["tfit"], frequency=float(g["model_frequency"]))
    assert np.max(np.abs(m.flux - g["model_tfit_f"])) < 1e-9
    pg2 = lc.to_periodogram(ls_method="chi2", nterms=2, oversample_factor=3)
    assert abs(pg2.frequency_at_max_power - float(g["model_nterms2_frequency"])) < 1e-12
    assert np.max(np.abs(pg2.model(lc.time).flux - g["model_nterms2"])) < 1e-9


def test_bls_transit_model_and_mask(golden):
    """get_transit_model / get_transit_mask on the GPU periodogram == lightkurve's (which asks astropy's BLS object)."""
    from lightkurve_amd.lightcurve import LightCurve
    g = golden("bls_model")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    pg = lc.to_periodogram(method="bls", period=g["period"], duration=[0.05, 0.1, 0.2, 0.3])
    assert pg.period_at_max_power == float(g["period_at_max_power"])
    assert np.max(np.abs(pg.get_transit_model().flux - g["model_default"])) < 1e-12
    assert np.array_equal(pg.get_transit_mask(), g["mask_default"])
    m = pg.get_transit_model(period=float(g["custom_period"]), duration=0.17, transit_time=float(g["custom_transit_time"]))
    assert np.max(np.abs(m.flux - g["model_custom"])) < 1e-12


def test_bls_compute_stats_vs_reference(golden):
    """BoxLeastSquaresPeriodogram.compute_stats (periodogram.py:1194-1229 -> astropy compute_stats) on the GPU periodogram:
    every entry of astropy's dict, default (at max power) and custom parameters; 1e-10 relative."""
    from lightkurve_amd.lightcurve import LightCurve
    g = golden("bls_model")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    pg = lc.to_periodogram(method="bls", period=g["period"], duration=[0.05, 0.1, 0.2, 0.3])
    for tag, kw in (("default", {}), ("custom", dict(period=float(g["custom_period"]), duration=0.17,
                                                      transit_time=float(g["custom_transit_time"])))):
        st = pg.compute_stats(**kw)
        for k, v in st.items():
            ref = g["stats_%s_%s" % (tag, k)]
            v = np.asarray(v, dtype=float)
            assert v.shape == ref.shape, (tag, k)
            assert np.allclose(v, ref, rtol=1e-10, atol=1e-12), (tag, k, v, ref)


def test_remove_outliers_matches_sigma_clip_oracle():
    """LightCurve.remove_outliers (reference lightcurve.py:1430-1556 -> astropy sigma_clip): identical masks vs the
    oracle's restatement of sigma_clip, NaN flux counted as an outlier."""
    from lightkurve_amd.lightcurve import LightCurve
    from oracle import np_oracle as O
    rng = np.random.default_rng(8)
    y = 1 + 1e-3 * rng.standard_normal(5000)
    y[rng.integers(0, 5000, 40)] += rng.choice([-1, 1], 40) * 0.02
    y[[5, 999]] = np.nan
    lc = LightCurve(time=np.arange(5000.0), flux=y)
    for sigma in (5.0, 3.0):
        clean, mask = lc.remove_outliers(sigma=sigma, return_mask=True)
        assert np.array_equal(mask, O.sigma_clip_mask(y, sigma))
        assert len(clean) == int((~mask).sum()) and not np.any(np.isnan(clean.flux))
