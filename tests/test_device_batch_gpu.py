"""GPU: the device-RESIDENT batch (lightkurve_amd/device.py) against the staged host path
(LightCurveBatch / lightkurve_amd.batch — itself pinned to the reference's goldens by the other -m gpu tests): the same
kernels run on the same numbers, so every stage must agree BIT FOR BIT (`==`, stated), and the new element-wise kernels
(flux / trend, BLS inputs, t - t[0], carried columns) against numpy's own arithmetic on the host."""
import os

import numpy as np
import pytest

from lightkurve_amd import LightCurve, LightCurveBatch, _capi, synth
from lightkurve_amd import batch as LB
from lightkurve_amd.device import DeviceBuffer, DeviceLightCurveBatch, release_device_pool
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
FDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fits")


def _lcs(n_targets=7, seed=11, with_nan=True):
    rng = np.random.default_rng(seed)
    lcs = []
    for b in range(n_targets):
        n = int(rng.integers(900, 2600))
        t, y, e, _ = synth.bls_target(3, 40 + b, n, cadence_days=10.0 / 1440.0)
        y = y * (1.0 + 0.004 * np.sin(2 * np.pi * t / (3.0 + b)))
        if with_nan:
            y[rng.integers(0, n, 9)] = np.nan
            e[rng.integers(0, n, 3)] = np.nan
        lcs.append(LightCurve(time=t + 2457000.0 - 2454833.0, flux=y * (900.0 + 50 * b), flux_err=e * 900.0))
    return lcs


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def test_upload_ingest_and_download_match_the_host_batch():
    lcs = _lcs()
    host = LightCurveBatch.from_lightcurves(lcs)
    dev = DeviceLightCurveBatch.from_lightcurves(lcs)
    back = dev.to_host()
    assert _same(back.time, host.time) and _same(back.flux, host.flux) and _same(back.flux_err, host.flux_err)
    assert np.array_equal(back.n_off, host.n_off) and dev.is_sorted is True and not dev.nan_free
    for meth in ("remove_nans", "normalize"):
        h2, d2 = getattr(host, meth)(), getattr(dev, meth)()
        b2 = d2.to_host()
        assert np.array_equal(b2.n_off, h2.n_off) and d2.nan_free
        assert _same(b2.time, h2.time) and _same(b2.flux, h2.flux) and _same(b2.flux_err, h2.flux_err)
        med = d2.median_flux.download(np.float64, len(d2))
        assert _same(med, h2.median_flux) and _same(med, [np.nanmedian(lc.flux) for lc in lcs])
    assert d2.meta[0]["NORMALIZED"] is True and "NORMALIZED" not in dev.meta[0]


def test_flatten_resident_equals_staged_flatten_and_numpy_division():
    lcs = _lcs()
    host = LightCurveBatch.from_lightcurves(lcs).remove_nans().normalize()
    dev = DeviceLightCurveBatch.from_lightcurves(lcs).remove_nans().normalize()
    m = np.zeros(host.time.size, bool)
    m[100:160] = True
    for kw in (dict(window_length=101), dict(window_length=51, polyorder=3, niters=2, sigma=4, break_tolerance=None, mask=m)):
        trend = host.flatten_trend(**kw)
        flat, tr = dev.flatten(return_trend=True, **kw)
        assert _same(tr.flux_host(), trend)
        with np.errstate(invalid="ignore", divide="ignore"):
            assert _same(flat.flux_host(), host.flux / trend)              # lightcurve.py:1066: flux / trend_signal
            assert _same(flat.flux_err_host(), host.flux_err / trend)
        assert _same(flat.time_host(), host.time) and flat.meta[0]["NORMALIZED"] is True
    # a device-resident mask (create_transit_mask(to_host=False)) feeds flatten without touching the host
    d_mask = dev.create_transit_mask(2.5, float(host.time[0]) + 0.3, 0.4, to_host=False)
    assert isinstance(d_mask, DeviceBuffer)
    hm = host.create_transit_mask(2.5, float(host.time[0]) + 0.3, 0.4)
    assert np.array_equal(dev.create_transit_mask(2.5, float(host.time[0]) + 0.3, 0.4), hm)
    assert _same(dev.flatten_trend(window_length=75, mask=d_mask).download(np.float64, dev.n_cadences),
                 host.flatten_trend(window_length=75, mask=hm))


def test_periodograms_of_a_resident_batch_equal_the_batch_api():
    lcs = _lcs(5, seed=3)
    host = LightCurveBatch.from_lightcurves(lcs).remove_nans().normalize()
    dev = DeviceLightCurveBatch.from_lightcurves(lcs).normalize()
    f = 0.05 + 0.004 * np.arange(6000)
    pk_h, pk_d = host.to_periodogram_peaks(f), dev.to_periodogram_peaks(f)
    assert np.array_equal(pk_h, pk_d)
    for kw in (dict(), dict(normalization="psd", freq_unit="1/d"), dict(ls_method="slow"),
               dict(ls_method="chi2", nterms=2), dict(ls_method="fastchi2", nterms=2)):
        assert np.array_equal(host.to_periodogram_power(f, **kw), dev.to_periodogram_power(f, **kw)), kw
    per = np.linspace(0.4, 9.0, 700)                   # irregular in frequency: lightkurve's 'fast' -> 'slow' switch
    assert np.array_equal(host.to_periodogram_power(1 / per), dev.to_periodogram_power(1 / per))
    # NaN flux is dropped inside (periodogram.py:869-872), with or without an explicit remove_nans() before
    raw_dev = DeviceLightCurveBatch.from_lightcurves(lcs)
    raw_host = LightCurveBatch.from_lightcurves(lcs)
    assert np.array_equal(raw_dev.to_periodogram_peaks(f), raw_host.to_periodogram_peaks(f))
    # spectra kept in HBM, peaks alongside
    d_pow, pk = dev.to_periodogram_power(f, to_host=False, want_peaks=True)
    assert isinstance(d_pow, DeviceBuffer) and np.array_equal(pk, pk_h)
    P = d_pow.download(np.float64, len(dev) * len(f)).reshape(len(dev), len(f))
    assert np.array_equal(P, host.to_periodogram_power(f))
    # against the oracle, so the pair is not merely self-consistent
    b = 2
    sl = slice(host.n_off[b], host.n_off[b + 1])
    ref = O.ls_power(host.time[sl] - host.time[sl][0], host.flux[sl], None, f, normalization="lk_amplitude")
    got = dev.to_periodogram_power(f, ls_method="slow")[b]
    assert np.max(np.abs(got - ref)) <= 1e-9 * np.max(ref)
    short = DeviceLightCurveBatch.from_lightcurves([LightCurve(time=[0.0], flux=[1.0])])
    with pytest.raises(ValueError, match="at least two cadences"):
        short.to_periodogram_peaks(f)


def test_bls_inputs_and_search_resident():
    lcs = _lcs(4, seed=5)
    host = LightCurveBatch.from_lightcurves(lcs)
    dev = DeviceLightCurveBatch.from_lightcurves(lcs)
    periods = np.linspace(0.6, 7.0, 900)
    durations = [0.05, 0.1, 0.2]
    for objective in ("likelihood", "snr"):
        ref = LB.bls_batch(host, periods, durations, objective=objective)       # float64[B, 7, nP], transit_time absolute
        res = dev.bls(periods, durations, objective=objective)
        got = res.to_host()
        assert got.shape == ref.shape and np.array_equal(got, ref)
    pk = res.peaks()
    am = np.argmax(ref[:, 0, :], axis=1)
    assert np.array_equal(pk["argmax"], am) and np.array_equal(pk["max_power"], ref[np.arange(len(am)), 0, am])
    assert np.array_equal(pk["period"], periods[am])
    for key, row in (("depth", 1), ("duration", 3), ("transit_time", 4)):
        assert np.array_equal(pk[key], ref[np.arange(len(am)), row, am]), key
    # a light curve with a non-finite error is searched unweighted (periodogram.py:1093-1100): exercised by _lcs's NaN errors;
    # one without errors at all likewise
    noerr = [LightCurve(time=lc.time, flux=lc.flux) for lc in lcs[:2]]
    assert np.array_equal(DeviceLightCurveBatch.from_lightcurves(noerr).bls(periods, durations).to_host(),
                          LB.bls_batch(noerr, periods, durations))
    with pytest.raises(ValueError, match="shorter than the minimum period"):
        dev.bls([0.1, 0.2], [0.3])


def test_fold_bin_and_transit_mask_resident():
    lcs = _lcs(6, seed=9, with_nan=False)
    host = LightCurveBatch.from_lightcurves(lcs)
    dev = DeviceLightCurveBatch.from_lightcurves(lcs)
    B = len(lcs)
    period = np.linspace(0.9, 3.1, B)
    epoch = np.array([lc.time[5] for lc in lcs])
    for kw in (dict(), dict(normalize_phase=True), dict(epoch_phase=0.1, wrap_phase=0.7 * period)):
        ph, order, (fl, fe) = _capi.fold_batch(host.time, host.n_off, period, epoch, columns=(host.flux, host.flux_err), **kw)
        got = dev.fold(period, epoch, **kw).to_host()
        assert _same(got["phase"], ph) and np.array_equal(got["order"], order)
        assert _same(got["flux"], fl) and _same(got["flux_err"], fe)
    # default epoch = the first time of each light curve (lightcurve.py:1145-1147)
    ph0 = _capi.fold_batch(host.time, host.n_off, 1.7, host.time[host.n_off[:-1]])[0]
    assert _same(dev.fold(1.7).to_host()["phase"], ph0)
    hb, db = host.bin(time_bin_size=0.25), dev.bin(time_bin_size=0.25).to_host()
    assert np.array_equal(db.n_off, hb.n_off)
    assert _same(db.time, hb.time) and _same(db.flux, hb.flux) and _same(db.flux_err, hb.flux_err)
    per, tt, dur = [1.3, 4.1], [host.time[0] + 0.2, host.time[0] + 1.0], [0.1, 0.3]
    assert np.array_equal(dev.create_transit_mask(per, tt, dur), host.create_transit_mask(per, tt, dur))
    unsorted = DeviceLightCurveBatch.from_arrays([0.0, 2.0, 1.0, 3.0], [1.0, 1.0, 1.0, 1.0], None, [0, 4])
    assert unsorted.is_sorted is False
    with pytest.raises(ValueError, match="sorted by time"):
        unsorted.flatten(window_length=3)


def test_from_fits_resident_and_quality_through_remove_nans():
    paths = [os.path.join(FDIR, n) for n in ("kepler_llc.fits", "tess_lc.fits")]
    host = LightCurveBatch.from_fits(paths)
    dev = DeviceLightCurveBatch.from_fits(paths)
    back = dev.to_host()
    assert np.array_equal(back.n_off, host.n_off)
    assert _same(back.time, host.time) and _same(back.flux, host.flux) and _same(back.flux_err, host.flux_err)
    assert np.array_equal(back.quality, host.quality) and dev.meta[0]["LABEL"] == host.meta[0]["LABEL"]
    assert dev.is_sorted is None and dev._sorted() is True          # checked on the device, once
    h2, d2 = host.normalize(), dev.normalize().to_host()
    assert np.array_equal(d2.n_off, h2.n_off) and _same(d2.flux, h2.flux) and np.array_equal(d2.quality, h2.quality)


def test_own_stream_pool_reuse_and_clock_probe():
    import ctypes
    h = _capi.Handle.get(0)
    st = ctypes.c_void_p()
    _capi._check(_capi._lib.lk_stream_create(h._h, ctypes.byref(st)))
    try:
        lcs = _lcs(3, seed=21)
        f = 0.05 + 0.01 * np.arange(2000)
        ref = DeviceLightCurveBatch.from_lightcurves(lcs).normalize().flatten(window_length=101).to_periodogram_peaks(f)
        got = DeviceLightCurveBatch.from_lightcurves(lcs, stream=st.value).normalize().flatten(window_length=101) \
            .to_periodogram_peaks(f)
        assert np.array_equal(ref, got)
        # a dropped buffer is handed out again instead of hipFree'd (which would synchronise the device)
        release_device_pool()
        a = DeviceBuffer(h, 1 << 20)
        p = a.ptr
        del a
        b = DeviceBuffer(h, (1 << 20) - 4096)
        assert b.ptr == p
        del b
        release_device_pool()
        mhz = ctypes.c_double(0.0)
        _capi._check(_capi._lib.lk_shader_clock_mhz(h._h, 2.0, ctypes.byref(mhz)))
        assert 300.0 < mhz.value < 3000.0, mhz.value
    finally:
        _capi._check(_capi._lib.lk_stream_destroy(h._h, st))
