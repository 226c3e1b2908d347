"""GPU parity: multi-term Lomb-Scargle (nterms > 1) through the C ABI vs the reference-generated
golden vectors (lightkurve `to_periodogram(nterms=.., ls_method='chi2')`, astropy `LombScargle(nterms=2).power(method=
'chi2')`) and the oracle restatement of astropy's lombscargle_chi2.

Tolerance (stated): max |p_gpu - p_ref| <= 1e-9 * max(p_ref) where the fit is well posed (f T >= 1: at least one cycle
over the baseline); 1e-5 everywhere (below that the (2 nterms + 1)-column normal equations are nearly singular and the
reference's own answer moves with the last bits of its sums).
"""
import numpy as np
import pytest

from lightkurve_amd import _capi, synth
from lightkurve_amd.lightcurve import LightCurve
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b))


def check(a, b, ok, loose=1e-5):
    assert relmax(a[ok], b[ok]) < 1e-9
    assert np.all(np.isfinite(a)) and relmax(a, b) < loose


def test_golden_lightkurve_nterms(golden):
    g = golden("ls_multiterm")
    t = g["time"] - g["time"][0]
    f = g["frequency"]
    T = t[-1]
    ok = f * T >= 1.0
    off = [0, len(t)]
    df = (f[-1] - f[0]) / (len(f) - 1)
    for nt in (2, 3, 4):
        amp = _capi.ls_power_batch(t, g["flux"], off, f0=f[0], df=df, M=len(f), normalization="lk_amplitude", nterms=nt)[0]
        check(amp, g["amp_chi2_%d" % nt], ok)
        # arbitrary-frequency kernel on the same grid
        amp2 = _capi.ls_power_batch(t, g["flux"], off, frequency=f, normalization="lk_amplitude", nterms=nt)[0]
        check(amp2, g["amp_chi2_%d" % nt], ok)
    scale = 2.0 / (len(t) * (1.0 / (g["time"][-1] - g["time"][0])) * (1e6 / 86400.0))
    psd = _capi.ls_power_batch(t, g["flux"], off, f0=f[0], df=df, M=len(f), normalization="lk_psd", scale=[scale],
                               nterms=2)[0]
    check(psd, g["psd_chi2_2"], ok)
    # 'fastchi2' (the reference's FFT approximation of the same quantity) is close to what we return
    amp = _capi.ls_power_batch(t, g["flux"], off, frequency=f, normalization="lk_amplitude", nterms=2)[0]
    assert relmax(amp[ok], g["amp_fastchi2_2"][ok]) < 2e-2


def test_golden_astropy_dy_fit_mean(golden):
    g = golden("ls_multiterm")
    t = g["time"] - g["time"][0]
    f = g["frequency"]
    ok = f * t[-1] >= 1.0
    df = (f[-1] - f[0]) / (len(f) - 1)
    for fm in (1, 0):
        for norm in ("standard", "psd"):
            p = _capi.ls_power_batch(t, g["flux"], [0, len(t)], dy=g["dy"], f0=f[0], df=df, M=len(f), fit_mean=bool(fm),
                                     normalization=norm, nterms=2)[0]
            check(p, g["astropy_%s_fm%d" % (norm, fm)], ok)


def test_host_mirror_and_irregular_grid(golden):
    """LightCurve.to_periodogram(nterms=2, ls_method='fastchi2', period=...) -> lightkurve switches to 'chi2'."""
    g = golden("ls_multiterm")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    pg = lc.to_periodogram(period=g["period"], normalization="amplitude", ls_method="fastchi2", nterms=2)
    assert pg.ls_method == "chi2" and pg.nterms == 2
    T = g["time"][-1] - g["time"][0]
    check(np.asarray(pg.power), g["amp_period_chi2_2"], g["period_frequency"] * T >= 1.0)
    pg = lc.to_periodogram(frequency=g["frequency"], normalization="amplitude", ls_method="chi2", nterms=3)
    check(np.asarray(pg.power), g["amp_chi2_3"], g["frequency"] * T >= 1.0)


def test_ragged_batch_vs_oracle():
    """Three ragged targets, heteroscedastic errors, a grid that does not fill the last tile."""
    ts, ys, ds = [], [], []
    for k, n in enumerate((257, 1000, 640)):
        t, y, e, _ = synth.ls_target(2, k, n)
        ts.append(t - t[0])
        ys.append(y)
        ds.append(e * np.random.default_rng(k).uniform(0.5, 2.0, n))
    off = np.concatenate([[0], np.cumsum([len(t) for t in ts])])
    t, y, d = np.concatenate(ts), np.concatenate(ys), np.concatenate(ds)
    f0, df, M = 2.0, 0.25, 333
    f = f0 + df * np.arange(M)
    for nt in (2, 4):
        p = _capi.ls_power_batch(t, y, off, dy=d, f0=f0, df=df, M=M, normalization="psd", nterms=nt)
        for b in range(3):
            ref = O.ls_power_chi2(ts[b], ys[b], ds[b], f, nterms=nt, normalization="psd")
            check(p[b], ref, f * ts[b][-1] >= nt)  # nt cycles over the baseline: all harmonics resolved


def test_nterms_one_is_the_closed_form(golden):
    g = golden("ls_tess3000")
    t = g["time"] - g["time"][0]
    f = g["frequency"][:300]
    ref = O.ls_power_chi2(t, g["flux"], None, f, nterms=1, normalization="lk_amplitude")
    assert relmax(ref, g["amp_slow"][:300]) < 1e-9  # the restatement itself at nterms = 1
    with pytest.raises(ValueError):
        _capi.ls_power_batch(t, g["flux"], [0, len(t)], frequency=f, nterms=9)   # (1..8 are served since round 5)


def test_fastchi2_reproduces_the_reference_fastchi2(golden):
    """ls_method='fastchi2' with nterms > 1 on a regular grid: extirpolation + FFT trig sums, like the reference —
    its own output (not the exact 'chi2' one) is reproduced to 1e-9 where the fit is well posed (f T >= 1).  Below that
    the extirpolated normal equations are close to singular (even slightly indefinite): finite values within 1e-2 of the
    reference's, which themselves hinge on the last bits of its FFT."""
    g = golden("ls_multiterm")
    t = g["time"] - g["time"][0]
    f = g["frequency"]
    ok = f * t[-1] >= 1.0
    df = f[1] - f[0]
    off = [0, len(t)]
    for nt in (2, 3):
        amp = _capi.ls_fast_batch(t, g["flux"], off, f0=f[0], df=df, M=len(f), normalization="lk_amplitude", nterms=nt)[0]
        check(amp, g["amp_fastchi2_%d" % nt], ok, 1e-2)
    for fm in (1, 0):
        p = _capi.ls_fast_batch(t, g["flux"], off, dy=g["dy"], f0=f[0], df=df, M=len(f), fit_mean=bool(fm),
                                normalization="standard", nterms=2)[0]
        check(p, g["astropy_fastchi2_standard_fm%d" % fm], ok, 1e-2)
    # through the host mirror (lightkurve psd scaling) and the batch API
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    pg = lc.to_periodogram(frequency=g["frequency_uhz"], normalization="psd", ls_method="fastchi2", nterms=2)
    assert pg.ls_method == "fastchi2"
    check(np.asarray(pg.power), g["psd_fastchi2_2"], ok, 1e-2)
    from lightkurve_amd.batch import lombscargle_batch
    pb = lombscargle_batch([lc, lc], f, ls_method="fastchi2", nterms=3)
    check(pb[1], g["amp_fastchi2_3"], ok, 1e-2)
    # a 4-term run against the oracle restatement, ragged batch
    ts = [t, t[:500]]
    ys = [g["flux"], g["flux"][:500]]
    tt, o2 = np.concatenate(ts), [0, len(t), len(t) + 500]
    p4 = _capi.ls_fast_batch(tt, np.concatenate(ys), o2, f0=f[0], df=df, M=len(f), normalization="psd", nterms=4)
    for b in range(2):
        ref = O.ls_power_fastchi2(ts[b], ys[b], None, f[0], df, len(f), nterms=4, normalization="psd")
        check(p4[b], ref, f * ts[b][-1] >= 4.0, 1.0)


@pytest.mark.parametrize("nterms", [5, 6, 8])
def test_five_to_eight_terms_run_the_exact_kernel(nterms):
    """Round 5: nterms 5..8 stay on the device — exact sums, one thread per frequency — for the regular-grid
    form, the explicit-frequency form and the 'fastchi2' entry point alike; vs the oracle restatement of astropy's
    lombscargle_chi2 (chi2_impl.py:5-86) at the tolerances stated on top."""
    t, y, e, _ = synth.ls_target(2, nterms, 1500, cadence_days=10.0 / 1440.0)
    t = t - t[0]
    f0, df, M = 0.3, 0.011, 400          # f T >= 3 everywhere: 2 nterms + 1 = 17 columns stay well posed
    f = f0 + df * np.arange(M)
    ref = O.ls_power_chi2(t, y, e, f, nterms=nterms, normalization="standard")
    ok = np.ones(M, bool)
    off = [0, len(t)]
    a = _capi.ls_power_batch(t, y, off, dy=e, f0=f0, df=df, M=M, normalization="standard", nterms=nterms)[0]
    b = _capi.ls_power_batch(t, y, off, dy=e, frequency=f, normalization="standard", nterms=nterms)[0]
    c = _capi.ls_fast_batch(t, y, off, dy=e, f0=f0, df=df, M=M, normalization="standard", nterms=nterms)[0]
    check(a, ref, ok, loose=1e-7)
    check(b, ref, ok, loose=1e-7)
    assert np.array_equal(a, c)
    with pytest.raises(ValueError):
        _capi.ls_power_batch(t, y, off, f0=f0, df=df, M=M, nterms=9)
