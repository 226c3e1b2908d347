"""Pin the oracle (oracle/np_oracle.py + oracle/c) against outputs of THE REFERENCE ITSELF
(tests/golden/*.npz, made by oracle/gen_golden.py from lightkurve@/root/reference + astropy 4.3.1)."""
import numpy as np
import pytest

from oracle import np_oracle as O


def relmax(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b))


# ------------------------------------------------------------------ Lomb-Scargle
def test_ls_exact_amplitude_and_psd(golden):
    g = golden("ls_tess3000")
    amp = O.lk_ls_periodogram(g["time"], g["flux"], g["frequency"], "amplitude")
    assert relmax(amp, g["amp_slow"]) < 1e-11
    assert relmax(amp, g["amp_cython"]) < 1e-11
    psd = O.lk_ls_periodogram(g["time"], g["flux"], g["frequency_uhz"], "psd")
    assert relmax(psd, g["psd_slow"]) < 1e-11
    assert abs(amp.max() - g["max_power"]) / g["max_power"] < 1e-11
    assert g["frequency"][np.argmax(amp)] == g["frequency_at_max_power"]


def test_ls_numpy_matches_c(golden):
    g = golden("ls_tess3000")
    t = g["time"] - g["time"][0]
    f = g["frequency"][::7]
    a = O.ls_power(t, g["flux"], None, f, normalization="lk_amplitude")
    b = O.ls_power_numpy(t, g["flux"], None, f, normalization="lk_amplitude")
    assert relmax(a, b) < 1e-11


def test_ls_fast_restatement_matches_reference_default(golden):
    g = golden("ls_tess3000")
    amp = O.lk_ls_periodogram(g["time"], g["flux"], g["frequency"], "amplitude", exact=False)
    ok = np.isfinite(g["amp_fast"])          # the reference's 'fast' path yields NaN at the very top bin here
    assert ok.sum() >= len(ok) - 1 and np.array_equal(ok, np.isfinite(amp))
    assert relmax(amp[ok], g["amp_fast"][ok]) < 1e-9
    # and the reference's own fast-vs-exact gap is ~1e-3 (SURVEY.md finding 8)
    assert 1e-6 < relmax(g["amp_fast"][ok], g["amp_slow"][ok]) < 1e-2


def test_ls_c1_default_grid(golden):
    g = golden("ls_c1_default")
    f, fs, nyq = O.lk_ls_default_grid(g["time"], "amplitude")
    assert len(f) == len(g["frequency"]) and np.allclose(f, g["frequency"], rtol=1e-14, atol=0)
    assert np.isclose(nyq, g["nyquist"], rtol=1e-14)
    amp = O.lk_ls_periodogram(g["time"], g["flux"], f, "amplitude")
    assert relmax(amp, g["amp_slow"]) < 1e-11
    f2, _, nyq2 = O.lk_ls_default_grid(g["time"], "psd")
    assert len(f2) == len(g["psd_frequency_uhz"]) and np.allclose(f2, g["psd_frequency_uhz"], rtol=1e-13)
    psd = O.lk_ls_periodogram(g["time"], g["flux"], f2, "psd")
    assert relmax(psd, g["psd_slow"]) < 1e-11
    assert abs(1 / f[np.argmax(amp)] - g["period_at_max_power"]) < 1e-12


def test_ls_nan_float32_period_grid(golden):
    g = golden("ls_nan_period_grid")
    assert str(g["ls_method"]) == "slow"
    ok = np.isfinite(g["flux"])
    amp = O.lk_ls_periodogram(g["time"][ok], g["flux"][ok].astype(float), 1.0 / g["period"], "amplitude")
    assert relmax(amp, g["amp"]) < 1e-10


def test_ls_dy_weights(golden):
    g = golden("ls_dy")
    t = g["time"] - g["time"][0]
    assert relmax(O.ls_power(t, g["flux"], g["dy"], g["frequency"], normalization="standard"),
                  g["astropy_standard"]) < 1e-10
    assert relmax(O.ls_power(t, g["flux"], g["dy"], g["frequency"], normalization="psd"),
                  g["astropy_psd"]) < 1e-10
    assert relmax(O.ls_power(t, g["flux"], g["dy"], g["frequency"], normalization="lk_amplitude"),
                  g["amp"]) < 1e-10


def test_ls_multiterm_chi2(golden):
    """nterms > 1: the restatement of astropy's lombscargle_chi2 against lightkurve / astropy outputs.  Below one
    cycle per baseline the (2 nterms + 1)-column fit is nearly singular (its answer depends on the last bits of the
    sums), so the tight comparison is made where f T >= 1 and a loose one everywhere."""
    g = golden("ls_multiterm")
    T = g["time"][-1] - g["time"][0]
    ok = g["frequency"] * T >= 1.0

    def check(a, b, m):
        assert relmax(a[m], b[m]) < 1e-9
        assert relmax(a, b) < 1e-5

    for nt in (2, 3, 4):
        amp = O.lk_ls_periodogram(g["time"], g["flux"], g["frequency"], "amplitude", nterms=nt)
        check(amp, g["amp_chi2_%d" % nt], ok)
    psd = O.lk_ls_periodogram(g["time"], g["flux"], g["frequency_uhz"], "psd", nterms=2)
    check(psd, g["psd_chi2_2"], ok)
    assert str(g["period_ls_method"]) == "chi2"  # lightkurve's irregular-grid switch
    amp = O.lk_ls_periodogram(g["time"], g["flux"], g["period_frequency"], "amplitude", nterms=2)
    check(amp, g["amp_period_chi2_2"], g["period_frequency"] * T >= 1.0)
    t = g["time"] - g["time"][0]
    for fm in (1, 0):
        for norm in ("standard", "psd"):
            p = O.ls_power_chi2(t, g["flux"], g["dy"], g["frequency"], nterms=2, fit_mean=bool(fm), normalization=norm)
            check(p, g["astropy_%s_fm%d" % (norm, fm)], ok)
    # 'fastchi2' is an approximation of 'chi2' (extirpolation): close, not equal — and it has its own restatement
    assert relmax(g["amp_fastchi2_2"][ok], g["amp_chi2_2"][ok]) < 2e-2
    f = g["frequency"]
    df = f[1] - f[0]
    for nt in (2, 3):
        a = O.ls_power_fastchi2(t, g["flux"], None, f[0], df, len(f), nterms=nt, normalization="lk_amplitude")
        check(a, g["amp_fastchi2_%d" % nt], ok)
    for fm in (1, 0):
        a = O.ls_power_fastchi2(t, g["flux"], g["dy"], f[0], df, len(f), nterms=2, fit_mean=bool(fm),
                                normalization="standard")
        check(a, g["astropy_fastchi2_standard_fm%d" % fm], ok)


def test_ls_constant_flux_zero_power(golden):
    g = golden("ls_constant")
    amp = O.lk_ls_periodogram(g["time"], g["flux"], g["frequency"], "amplitude")
    # the reference yields ~1e-29 here (its BLAS dot of w.y is off by an ulp); for the 3-cadence curve of
    # tests/test_periodogram.py:445-457 it yields exactly 0.  The oracle centres about y[0] => exactly 0 always.
    assert (g["amp"] < 1e-25).all() and (amp == 0).all()
    t = np.array([1.0, 3.0, 4.0])
    f, _, _ = O.lk_ls_default_grid(t, "amplitude")
    assert (O.lk_ls_periodogram(t, np.ones(3), f, "amplitude") == 0).all()


# ------------------------------------------------------------------ Periodogram.smooth / flatten
def test_pg_smooth_and_flatten(golden):
    g = golden("pg_smooth")
    f, p = g["frequency"], g["power"]
    for fw in (0.01, 0.05, 0.3):
        assert relmax(O.pg_smooth_logmedian(f, p, fw), g["logmedian_%g" % fw]) < 1e-13
    for fw, w in zip((3.0, 10.5, 40.0), g["box_widths"]):
        assert int(np.ceil(fw / np.mean(np.diff(f)))) == w
        assert relmax(O.pg_smooth_boxkernel(f, p, fw), g["boxkernel_%g" % fw]) < 1e-13
    bkg = O.pg_smooth_logmedian(f, p, 0.01)
    assert relmax(bkg, g["flatten_bkg"]) < 1e-13 and relmax(p / bkg, g["flatten_snr"]) < 1e-13
    a, b = O.pg_smooth_logmedian(f, g["power_nan"], 0.02), g["logmedian_nan"]
    assert np.array_equal(np.isnan(a), np.isnan(b)) and relmax(a[~np.isnan(a)], b[~np.isnan(b)]) < 1e-13
    a, b = O.pg_smooth_boxkernel(f, g["power_nan"], 10.5), g["boxkernel_nan"]
    assert np.array_equal(np.isnan(a), np.isnan(b)) and relmax(a[~np.isnan(a)], b[~np.isnan(b)]) < 1e-13


def test_overfit_metric(golden):
    g = golden("overfit_metric")
    for name in ("over", "mild", "clean", "nan"):
        np.random.seed(1234)
        m = O.overfit_metric_lombscargle(g["time"], g["flux"], g["flux_err"], g["flux_" + name], g["flux_err"],
                                         n_samples=int(g["nsamples_" + name]))
        assert abs(m - float(g["metric_" + name])) < 1e-9, name


def _fold_kw(g, k):
    kw = dict(period=float(g["period_" + k]), epoch_time=float(g["epoch_time_" + k]))
    for name in ("epoch_phase", "wrap_phase", "normalize_phase"):
        if "%s_%s" % (name, k) in g:
            kw[name] = g["%s_%s" % (name, k)].item()
    return kw


def assert_same_order_up_to_near_ties(phase_ref, a, b, tol=1e-12):
    """a == b except where neighbouring reference phases are closer than tol (their order is decided by the last bit)."""
    bad = np.flatnonzero(np.asarray(a) != np.asarray(b))
    gaps = np.abs(np.diff(phase_ref))
    for i in bad:
        near = min(gaps[i - 1] if i > 0 else np.inf, gaps[i] if i < len(gaps) else np.inf)
        assert near < tol, (i, near)
    assert len(bad) <= 0.01 * len(a)


def test_fold(golden):
    """The reference folds in seconds on astropy Time objects; the restatement works in days: phases agree to
    1e-10 d and the permutation (flux order, original times, cycles) is the same except across near-ties
    (neighbouring phases closer than 1e-12, where the last bit of either arithmetic decides)."""
    g = golden("fold")
    for k in "abcd":
        ph, order, cyc = O.fold(g["time"], **_fold_kw(g, k))
        assert np.max(np.abs(ph - g["phase_" + k])) < 1e-10
        assert_same_order_up_to_near_ties(g["phase_" + k], g["flux"][order], g["flux_" + k])
        assert_same_order_up_to_near_ties(g["phase_" + k], g["time"][order], g["time_original_" + k])
        assert_same_order_up_to_near_ties(g["phase_" + k], cyc, g["cycle_" + k])
    # the forced exact tie (time[100] == time[99]) keeps cadence order in every case
    ph, order, _ = O.fold(g["time"], **_fold_kw(g, "a"))
    i99, i100 = np.flatnonzero(order == 99)[0], np.flatnonzero(order == 100)[0]
    assert i100 == i99 + 1


# ------------------------------------------------------------------ BLS
@pytest.mark.parametrize("objective", ["likelihood", "snr"])
def test_bls_bit_exact(golden, objective):
    g = golden("bls_2500")
    t, y, ivar, t_ref = O.lk_bls_inputs(g["time"], g["flux"], g["flux_err"])
    assert np.array_equal(t, g["raw_t"]) and np.array_equal(y, g["raw_y"]) and np.array_equal(ivar, g["raw_ivar"])
    res = O.bls(t, y, ivar, g["period"], g["duration"], 10, objective == "likelihood")
    for name, arr in zip(O.BLS_FIELDS, res):
        ref = g[objective + "_" + name]
        if name == "transit_time":
            arr = arr + t_ref + g["time"][0]
            assert np.allclose(arr, ref, rtol=0, atol=1e-9)
        else:
            assert np.array_equal(arr, ref), name
    assert g["period"][np.argmax(res[0])] == g[objective + "_period_at_max_power"]


def test_bls_default_grid_and_noerr(golden):
    g = golden("bls_default")
    t, y, ivar, t_ref = O.lk_bls_inputs(g["time"], g["flux"], g["flux_err"])
    dur = np.array([0.05, 0.10, 0.15, 0.20, 0.25, 0.33])
    dt = np.median(np.diff(g["time"]))
    pmin = max(4 * dt, dur.max() + dt)
    pmax = (g["time"].max() - g["time"].min()) / 3.0
    period = O.bls_autoperiod(g["time"] - g["time"][0], dur, pmin, pmax, frequency_factor=200)
    assert len(period) == len(g["period"]) and np.allclose(period, g["period"], rtol=1e-14)
    res = O.bls(t, y, ivar, g["period"], dur)
    assert np.array_equal(res[0], g["power"]) and np.array_equal(res[1], g["depth"])
    assert np.array_equal(res[3], g["duration"])
    g = golden("bls_noerr")
    t, y, ivar, t_ref = O.lk_bls_inputs(g["time"], g["flux"], None)
    res = O.bls(t, y, ivar, g["period"], np.array([0.1, 0.2]))
    assert np.array_equal(res[0], g["power"]) and np.array_equal(res[1], g["depth"])


def test_bls_invalid_duration_raises():
    t = np.linspace(0, 10, 200)
    with pytest.raises(ValueError):
        O.bls(t, np.zeros(200), np.ones(200), np.array([0.3, 1.0]), np.array([0.5]))


# ------------------------------------------------------------------ flatten
def test_savgol_raw(golden):
    g = golden("savgol_raw")
    for key, (w, p) in dict(w101p2=(101, 2), w401p3=(401, 3), w5p4=(5, 4), w11p0=(11, 0)).items():
        assert np.max(np.abs(O.savgol_filter(g["x"], w, p) - g[key])) < 2e-11 * np.max(np.abs(g["x"]))


@pytest.mark.parametrize("name", ["flatten_w101", "flatten_w401", "flatten_w51"])
def test_flatten_trend(golden, name):
    g = golden(name)
    bt = None if np.isnan(g["break_tolerance"]) else float(g["break_tolerance"])
    trend, _ = O.flatten_trend(g["time"], g["flux"], int(g["window_length"]), int(g["polyorder"]), bt,
                               int(g["niters"]), float(g["sigma"]))
    assert np.allclose(trend, g["trend"], rtol=1e-11, atol=0, equal_nan=True)
    ok = np.isfinite(g["flux"])
    assert np.allclose((g["flux"] / trend)[ok], g["flat_flux"][ok], rtol=1e-11)


def test_flatten_user_mask(golden):
    g = golden("flatten_mask")
    trend, _ = O.flatten_trend(g["time"], g["flux"], mask=g["mask"])
    assert np.allclose(trend, g["trend"], rtol=1e-11, atol=0)


def _sha(*arrays):
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_flatten_bench_shape(golden):
    """20 000 cadences, window 401 (bench.py's flatten workload): the reference's own trend for lightkurve_amd.synth
    light curves; the fixture stores the SHA-256 of the inputs, which are regenerated here."""
    from lightkurve_amd import synth
    g = golden("flatten_20k")
    t, y, e, _ = synth.ls_target(6, 0, 20000)
    assert _sha(t, y) == str(g["sha_0"]), "synth light curve differs from the one the golden was made from"
    trend, _ = O.flatten_trend(t, y, 401, 2, 5, 3, 3)
    assert np.allclose(trend, g["trend_0"], rtol=1e-11, atol=0)


def test_flatten_long_cadence_shape(golden):
    """4 500 cadences at 30 min, the reference's default window 101 (the shape of the LDS-resident flatten kernel), NaN
    fluxes included: lightkurve's own trends for lightkurve_amd.synth light curves (inputs pinned by SHA-256)."""
    from lightkurve_amd import synth
    g = golden("flatten_4500")
    for i, idx in enumerate((0, 1, 3, 4)):
        t, y, e, _ = synth.ls_target(6, idx, 4500, cadence_days=30.0 / 1440.0)
        if i == 3:
            y = y.copy()
            y[700:705] = np.nan
        assert _sha(t, y) == str(g["sha_%d" % i]), "synth light curve differs from the one the golden was made from"
        trend, _ = O.flatten_trend(t, y, 101, 2, 5, 3, 3)
        assert np.allclose(trend, g["trend_%d" % i], rtol=1e-11, atol=0)


# ------------------------------------------------------------------ regression
def test_regression_k8(golden):
    g = golden("regress_k8")
    r = O.regression_correct(g["X"], g["flux"], g["flux_err"], g["cadence_mask"], g["prior_mu"],
                             g["prior_sigma"])
    assert np.array_equal(r["outlier_mask"], g["outlier_mask"])
    assert np.allclose(r["coefficients"], g["coefficients"], rtol=1e-9, atol=1e-13)
    assert np.allclose(r["corrected"], g["corrected"], rtol=0, atol=1e-12)


def test_regression_kat_and_noerr(golden):
    g = golden("regress_kat")
    X = np.array([[1., 1.], [1., 2.]])
    r = O.regression_correct(X, np.array([5., 10.]), np.array([1., 1.]))
    assert np.allclose(r["coefficients"], g["noprior"], atol=1e-9) and np.allclose(g["noprior"], [0, 5], atol=1e-7)
    r = O.regression_correct(X, np.array([5., 10.]), np.array([1., 1.]), None, np.array([99., 99.]),
                             np.array([1e-6, 1e-6]))
    assert np.allclose(r["coefficients"], g["tight"], atol=1e-9)
    g = golden("regress_noerr")
    r = O.regression_correct(g["X"], g["flux"], None)
    assert np.array_equal(r["outlier_mask"], g["outlier_mask"])
    assert np.allclose(r["corrected"], g["corrected"], rtol=0, atol=1e-12)


# ------------------------------------------------------------------ PLD
def _pld_kw(g):
    kw = {}
    for k in ("pld_order", "pca_components", "spline_degree"):
        if "kw_" + k in g:
            kw[k] = int(g["kw_" + k])
    if "kw_normalize_background_pixels" in g:
        kw["normalize_background_pixels"] = bool(g["kw_normalize_background_pixels"])
    return kw


def test_threshold_mask_and_spline_block(golden):
    for name in ("pld_k2sin_order3", "pld_factory11_order2"):
        g = golden(name)
        assert np.array_equal(O.threshold_mask(g["flux"]), g["threshold_mask"])
        nk = int(g["block_widths"][-1]) - 1
        deg = int(g["kw_spline_degree"]) if "kw_spline_degree" in g else 5
        B = O.bspline_basis(g["time"], nk, deg)
        assert np.allclose(B, g["X"][:, -nk - 1:-1], rtol=0, atol=1e-13)          # patsy bs(), App. B.7
        assert np.array_equal(g["X"][:, -1], np.ones(len(g["time"])))


@pytest.mark.parametrize("name", ["pld_k2sin_order3", "pld_k2sin_default", "pld_factory11_order2"])
def test_pld_corrected_flux(golden, name):
    """Parity on corrected flux (never on X columns / coefficients: PCA bases are defined up to rotations inside
    a block)."""
    g = golden(name)
    ap = g["aperture_mask"] if g["aperture_mask"].shape else np.ones(g["flux"].shape[1:], bool)
    kw = _pld_kw(g)
    if name == "pld_k2sin_default":
        kw.update(pld_order=1, pca_components=3, normalize_background_pixels=False)
    r = O.pld_correct(g["time"], g["flux"], g["flux_err"], ap, g["pld_aperture_mask"], g["background_aperture_mask"], **kw)
    assert np.allclose(r["lc_flux"], g["lc_flux"], rtol=1e-6) and np.allclose(r["lc_flux_err"], g["lc_flux_err"], rtol=1e-6)
    assert r["X"].shape == g["X"].shape and np.allclose(r["prior_sigma"], g["prior_sigma"], rtol=1e-6)
    assert np.array_equal(r["outlier_mask"], g["outlier_mask"])
    assert np.max(np.abs(r["corrected"] - g["corrected"])) / np.median(g["corrected"]) < 1e-6


def test_pld_bench_shape(golden):
    """configs[4] at its real shape (11x11 pixels x 3500 cadences, order 3, 16 components, all pixels): the reference's
    PLDCorrector output for a lightkurve_amd.synth cutout (inputs regenerated, SHA-256 checked)."""
    from lightkurve_amd import synth
    g = golden("pld_c5")
    t, flux, err, _ = synth.pld_cutout(4, 0, n=3500, npix=11)
    assert _sha(t, flux, err) == str(g["sha_0"]), "synth cutout differs from the one the golden was made from"
    allm = np.ones((11, 11), bool)
    r = O.pld_correct(g["time_0"], flux, err, allm, allm, allm, pld_order=3, pca_components=16, spline_degree=5)
    assert np.allclose(r["lc_flux"], g["lc_flux_0"], rtol=1e-6)
    assert np.array_equal(r["outlier_mask"], g["outlier_mask_0"])
    assert np.max(np.abs(r["corrected"] - g["corrected_0"])) / np.median(g["corrected_0"]) < 1e-6


def test_ingest_oracle_vs_reference(golden):
    """remove_nans + normalize, create_transit_mask and bin restatements against lightkurve / astropy outputs."""
    g = golden("ingest")
    for b in range(int(g["n"])):
        t, f, e = g["time_%d" % b], g["flux_%d" % b], g["err_%d" % b]
        ct, cf, ce, _med = O.remove_nans_normalize(t, f, e)
        assert np.array_equal(ct, g["clean_time_%d" % b])
        assert np.allclose(cf, g["clean_flux_%d" % b], rtol=1e-15, atol=0)
        assert np.allclose(ce, g["clean_err_%d" % b], rtol=1e-15, atol=0, equal_nan=True)
        m = O.transit_mask(t, g["period_%d" % b], g["duration_%d" % b], g["transit_time_%d" % b])
        assert np.array_equal(m, g["mask_%d" % b])
        bt, bf, be = O.bin_lightcurve(t, f, e, float(g["bin_size_%d" % b]))
        assert np.allclose(bt, g["bin_time_%d" % b], rtol=0, atol=1e-9)
        assert np.allclose(bf, g["bin_flux_%d" % b], rtol=1e-13, atol=0, equal_nan=True)
        assert np.allclose(be, g["bin_err_%d" % b], rtol=1e-12, atol=0, equal_nan=True)
