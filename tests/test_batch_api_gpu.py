"""GPU: the per-batch Python front end (lightkurve_amd.batch on lists and on LightCurveBatch) against the per-object
constructors (the mirrors of reference periodogram.py:636-989, 1043-1192, lightcurve.py:943-1078) on the same light
curves — the loop over a LightCurveCollection (collections.py:145) that the batch calls replace — and the device-side
rebase of lk_ls_fast_peaks_lc_batch against the host-side one, bit for bit."""
import numpy as np
import pytest

from lightkurve_amd import BoxLeastSquaresPeriodogram, LightCurve, LombScarglePeriodogram, _capi, batch, packed, synth
from lightkurve_amd.ingest import LightCurveBatch

pytestmark = pytest.mark.gpu


def _lcs(ns=(3000, 411, 5000, 1200, 2048), nan_every=(0, 7, 0, 13, 0), seed=5):
    out = []
    for i, n in enumerate(ns):
        t, y, e, _ = synth.ls_target(seed, i, n)
        y = y.copy()
        if nan_every[i]:
            y[::nan_every[i]] = np.nan
        out.append(LightCurve(time=t + 2457000.0, flux=y, flux_err=e, meta={"TARGETID": i}))
    return out


def relmax(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b))


def test_device_rebase_is_bit_identical_to_the_host_rebase():
    lcs = [lc.remove_nans() for lc in _lcs()]
    (t, y), off = packed.pack_columns(lcs, ("time", "flux"))
    trel = packed.rebase_times(t, off)
    f0, df, M = 0.01, 0.004, 20000
    for kw in (dict(normalization="lk_amplitude"), dict(normalization="lk_psd", scale=np.linspace(1.0, 2.0, len(lcs)))):
        p_host, mx_h, am_h = _capi.ls_fast_peaks_batch(trel, y, off, f0=f0, df=df, M=M, **kw)
        p_dev, mx_d, am_d = _capi.ls_fast_peaks_batch(t, y, off, f0=f0, df=df, M=M, absolute_time=True, **kw)
        assert np.array_equal(p_host, p_dev, equal_nan=True) and np.array_equal(mx_h, mx_d) and np.array_equal(am_h, am_d)
    # chunked pipeline (several chunks of targets): every chunk rebases its own light curves
    h = _capi.Handle.get(0)
    h.set_host_chunk_mb(1)
    try:
        p_dev2 = _capi.ls_fast_peaks_batch(t, y, off, f0=f0, df=df, M=M, absolute_time=True, normalization="lk_amplitude")[0]
    finally:
        h.set_host_chunk_mb(64)
    p_ref = _capi.ls_fast_peaks_batch(trel, y, off, f0=f0, df=df, M=M, normalization="lk_amplitude")[0]
    assert relmax(np.nan_to_num(p_dev2), np.nan_to_num(p_ref)) < 1e-12
    assert np.array_equal(t, np.concatenate([lc.time for lc in lcs]))          # the caller's times are not modified


@pytest.mark.parametrize("kw", [dict(), dict(normalization="psd"), dict(ls_method="slow"), dict(ls_method="chi2", nterms=2),
                                dict(ls_method="fastchi2", nterms=2)])
def test_lombscargle_batch_equals_per_object_calls(kw):
    lcs = _lcs()
    freq = 0.02 + 0.002 * np.arange(6000)
    ref = [LombScarglePeriodogram.from_lightcurve(lc, frequency=freq, **kw).power for lc in lcs]
    got = batch.lombscargle_batch(lcs, freq, **kw)
    lb = LightCurveBatch.from_lightcurves(lcs, pinned=True)
    got_b = lb.to_periodogram_power(freq, **kw)
    assert got.shape == (len(lcs), len(freq)) and np.array_equal(got, got_b, equal_nan=True)
    for b in range(len(lcs)):
        ok = np.isfinite(ref[b])
        assert np.array_equal(np.isfinite(got[b]), ok) and relmax(got[b][ok], ref[b][ok]) < 1e-10
    if "ls_method" not in kw:
        pk = lb.to_periodogram_peaks(freq, **kw)
        assert np.array_equal(pk[:, 0], np.nanmax(got, axis=1)) and np.array_equal(pk[:, 1], np.nanargmax(got, axis=1))
        out = _capi.pinned_empty(got.shape)
        assert batch.lombscargle_batch(lcs, freq, out=out, **kw) is out and np.array_equal(out, got, equal_nan=True)


def test_bls_and_flatten_batches_equal_per_object_calls():
    lcs = []
    for i, n in enumerate((2500, 1800, 3000)):
        t, y, e, _ = synth.bls_target(11, i, n, cadence_days=10.0 / 1440.0)
        if i == 1:
            y = y.copy()
            y[::17] = np.nan
        lcs.append(LightCurve(time=t + 2457000.0, flux=y, flux_err=e))
    period = np.linspace(0.8, 5.0, 300)
    dur = [0.05, 0.1, 0.2]
    got = batch.bls_batch(lcs, period, dur)
    got_b = LightCurveBatch.from_lightcurves(lcs).bls(period, dur)
    assert np.array_equal(got, got_b)
    for b, lc in enumerate(lcs):
        ref = BoxLeastSquaresPeriodogram.from_lightcurve(lc, period=period, duration=dur)
        for i, k in enumerate(_capi.BLS_FIELDS):
            assert np.array_equal(got[b, i], ref._BLS_result[k]), k
    clean = [lc.remove_nans() for lc in lcs]
    trends = batch.flatten_batch(clean, window_length=101)
    for lc, tr in zip(clean, trends):
        ref_tr = lc.flatten(window_length=101, return_trend=True)[1].flux
        assert np.array_equal(tr, ref_tr)
    assert np.array_equal(np.concatenate(trends), LightCurveBatch.from_lightcurves(clean).flatten_trend(window_length=101))
