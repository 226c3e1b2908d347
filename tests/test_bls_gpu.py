"""GPU parity: HIP BLS (through the C ABI) vs the reference-generated golden vectors and the C oracle.
Bar: BIT-EXACT (==) on all seven outputs for time-sorted input, hence bit-exact best-period index."""
import numpy as np
import pytest

from lightkurve_amd import _capi, synth
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("objective", ["likelihood", "snr"])
def test_golden_bit_exact(golden, objective):
    g = golden("bls_2500")
    t, y, ivar = g["raw_t"], g["raw_y"], g["raw_ivar"]
    res = _capi.bls_batch(t, y, ivar, [0, len(t)], g["period"], g["duration"], 10, objective == "likelihood")
    t_ref = g["time"][0]
    for name in _capi.BLS_FIELDS:
        got, ref = res[name][0], g[objective + "_" + name]
        if name == "transit_time":
            assert np.allclose(got + t_ref, ref, rtol=0, atol=1e-9)
        else:
            assert np.array_equal(got, ref), name
    assert g["period"][np.argmax(res["power"][0])] == g[objective + "_period_at_max_power"]


def test_golden_default_and_noerr(golden):
    g = golden("bls_default")
    t, y, ivar, _ = O.lk_bls_inputs(g["time"], g["flux"], g["flux_err"])
    dur = np.array([0.05, 0.10, 0.15, 0.20, 0.25, 0.33])
    res = _capi.bls_batch(t, y, ivar, [0, len(t)], g["period"], dur)
    assert np.array_equal(res["power"][0], g["power"]) and np.array_equal(res["depth"][0], g["depth"])
    assert np.array_equal(res["duration"][0], g["duration"]) and np.array_equal(res["depth_snr"][0], g["snr"])
    g = golden("bls_noerr")
    t, y, ivar, _ = O.lk_bls_inputs(g["time"], g["flux"], None)
    res = _capi.bls_batch(t, y, ivar, [0, len(t)], g["period"], [0.1, 0.2])
    assert np.array_equal(res["power"][0], g["power"]) and np.array_equal(res["depth"][0], g["depth"])


def test_ragged_batch_bit_exact_vs_oracle():
    ns = [400, 2500, 37, 1200, 3000, 800, 64, 1999, 2, 640]
    ts, ys, ws = [], [], []
    for i, n in enumerate(ns):
        t, y, e, _ = synth.bls_target(8, i, n, cadence_days=10.0 / 1440.0)
        rng = np.random.default_rng(i)
        ivar = 1.0 / (e * rng.uniform(0.5, 2.0, n)) ** 2
        ts.append(t - t.min()), ys.append(y - np.median(y)), ws.append(ivar)
    t, off = synth.pack_ragged(ts)
    y, _ = synth.pack_ragged(ys)
    w, _ = synth.pack_ragged(ws)
    period = np.concatenate([np.linspace(0.51, 9.0, 173), [0.7, 0.7, 3.3333]])   # unsorted + duplicates
    duration = np.array([0.05, 0.021, 0.33, 0.1, 0.1, 0.5])                      # unsorted + duplicate
    for use_like in (True, False):
        res = _capi.bls_batch(t, y, w, off, period, duration, 7, use_like)
        for b, n in enumerate(ns):
            ref = O.bls(ts[b], ys[b], ws[b], period, duration, 7, use_like)
            for name, r in zip(_capi.BLS_FIELDS, ref):
                assert np.array_equal(res[name][b], r), (b, n, name, use_like)


def test_unsorted_time_serial_path_still_exact():
    t, y, e, _ = synth.bls_target(8, 99, 5000, cadence_days=10.0 / 1440.0)
    rng = np.random.default_rng(1)
    perm = rng.permutation(len(t))          # thousands of "rounds" -> serial histogram path
    t, y = (t - t.min())[perm], (y - np.median(y))[perm]
    ivar = np.full(len(t), 1.0 / 5e-4 ** 2)
    period = np.linspace(0.6, 5.0, 40)
    res = _capi.bls_batch(t, y, ivar, [0, len(t)], period, [0.05, 0.2])
    ref = O.bls(t, y, ivar, period, [0.05, 0.2])
    for name, r in zip(_capi.BLS_FIELDS, ref):
        assert np.array_equal(res[name][0], r), name


def test_full_size_sample_bit_exact():
    """BASELINE C4 shape on one target: N=20000, 200 durations, periods sampled across 0.6..13 d
    (max n_bins = 6510): bit-exact vs the C oracle, and the injected period is recovered."""
    t, y, e, truth = synth.bls_target(3, 5, 20000)
    tt, yy, ivar, _ = O.lk_bls_inputs(t, y, e)
    period, duration = synth.bls_grid(50000, 200)
    sel = np.unique(np.r_[0:20, np.linspace(0, 49999, 150).astype(int), 49980:50000,
                          np.argmin(np.abs(period - truth["period"])) + np.arange(-10, 11)])
    res = _capi.bls_batch(tt, yy, ivar, [0, len(tt)], period[sel], duration)
    ref = O.bls(tt, yy, ivar, period[sel], duration)
    for name, r in zip(_capi.BLS_FIELDS, ref):
        assert np.array_equal(res[name][0], r), name
    best = period[sel][np.argmax(res["power"][0])]
    assert abs(best - truth["period"]) / truth["period"] < 0.01


def test_invalid_period_duration():
    t = np.linspace(0, 10, 200)
    with pytest.raises(ValueError, match="period"):
        _capi.bls_batch(t, np.zeros(200), np.ones(200), [0, 200], [0.3, 1.0], [0.5])
    with pytest.raises(ValueError, match="period"):
        _capi.bls_batch(t, np.zeros(200), np.ones(200), [0, 200], [1.0, np.nan], [0.1])


@pytest.mark.parametrize("use_like", [True, False])
def test_stress_regimes_bit_exact(use_like):
    """The skip-ahead scan must never change a result: regimes that stress its bound — pure noise (best barely above
    the crowd), very deep and very shallow transits, quantised flux (exact ties), constant flux (every objective equal),
    zero-weight cadences, wildly different weights — all seven outputs == the C oracle."""
    rng = np.random.default_rng(42)
    n = 3000
    t = np.sort(rng.uniform(0, 27.0, n))
    sig = 5e-4
    cases = {}
    cases["noise"] = (rng.normal(0, sig, n), np.full(n, sig ** -2))
    y = rng.normal(0, sig, n)
    y[((t - 0.3) % 2.75) < 0.12] -= 2e-2
    cases["deep"] = (y, np.full(n, sig ** -2))
    y = rng.normal(0, sig, n)
    y[((t - 1.1) % 4.1) < 0.2] -= 3e-4
    cases["shallow"] = (y, np.full(n, sig ** -2))
    cases["quantised"] = (np.round(rng.normal(0, sig, n) / 2.5e-4) * 2.5e-4, np.full(n, 1.0))
    cases["constant"] = (np.zeros(n), np.full(n, 1.0))
    w = np.full(n, sig ** -2)
    w[rng.random(n) < 0.2] = 0.0
    cases["zero_weights"] = (rng.normal(0, sig, n), w)
    cases["wild_weights"] = (rng.normal(0, sig, n), 10.0 ** rng.uniform(2, 9, n))
    period = np.concatenate([np.linspace(0.55, 8.9, 150), [2.75, 4.1]])
    duration = np.linspace(0.02, 0.5, 41)
    ys, ws = zip(*cases.values())
    tt, off = synth.pack_ragged([t] * len(cases))
    yy, _ = synth.pack_ragged([y - np.median(y) for y in ys])
    ww, _ = synth.pack_ragged(list(ws))
    res = _capi.bls_batch(tt, yy, ww, off, period, duration, 10, use_like)
    for b, name in enumerate(cases):
        ref = O.bls(t, ys[b] - np.median(ys[b]), ws[b], period, duration, 10, use_like)
        for field, r in zip(_capi.BLS_FIELDS, ref):
            assert np.array_equal(res[field][b], r, equal_nan=True), (name, field)


def test_randomised_configurations_bit_exact():
    """Seeded fuzz over the whole parameter space of run_bls (cadence count, baseline, period and duration grids,
    oversample, objective, weights, transit shapes): every output of every configuration == the C oracle."""
    rng = np.random.default_rng(20260925)
    n_cfg = 48
    for c in range(n_cfg):
        n = int(rng.integers(20, 2500))
        span = float(rng.uniform(3.0, 90.0))
        t = np.sort(rng.uniform(0, span, n))
        if c % 5 == 0:
            t = np.round(t / 0.02) * 0.02                   # many exactly equal phases / bin edges
            t = np.sort(t)
        sig = 10.0 ** rng.uniform(-5, -2)
        y = rng.normal(0, sig, n)
        P0 = rng.uniform(0.4, span / 2.5)
        y[((t - rng.uniform(0, P0)) % P0) < rng.uniform(0.02, 0.3)] -= rng.uniform(0.2, 30.0) * sig
        if c % 7 == 3:
            y = np.round(y / sig) * sig                     # quantised: ties between candidates
        ivar = np.full(n, sig ** -2) if c % 3 else 1.0 / (sig * rng.uniform(0.3, 3.0, n)) ** 2
        if c % 11 == 5:
            ivar[rng.random(n) < 0.3] = 0.0
        nd = int(rng.integers(1, 25))
        dmin = rng.uniform(0.01, 0.08)
        duration = np.sort(rng.uniform(dmin, dmin * rng.uniform(1.5, 12.0), nd))
        if c % 4 == 1:
            duration = rng.permutation(duration)            # caller's order is the tie-break order
        pmin = duration.max() * rng.uniform(1.05, 3.0)
        period = rng.uniform(pmin, max(pmin * 1.5, span / 1.5), int(rng.integers(1, 40)))
        oversample = int(rng.integers(1, 13))
        use_like = bool(c % 2)
        yy = y - np.median(y)
        nbins = np.ceil(period.max() / (duration.min() / oversample)) + oversample
        if nbins > 8000:
            continue
        res = _capi.bls_batch(t, yy, ivar, [0, n], period, duration, oversample, use_like)
        ref = O.bls(t, yy, ivar, period, duration, oversample, use_like)
        for field, r in zip(_capi.BLS_FIELDS, ref):
            assert np.array_equal(res[field][0], r, equal_nan=True), (c, field, n, oversample, use_like)


def test_ordered_histogram_fallback_is_bit_exact(golden):
    """on a device whose LDS ds_add_f64 is not lane-ordered (bls_selftest_kernel) the library falls back to an
    atomic-free histogram instead of refusing to run.  The fall-back is forced here (lk_bls_set_ordered_histogram) and must
    reproduce the reference golden (astropy's compiled run_bls) and the atomic form bit for bit — sorted, unsorted and
    ragged input, short and long periods (one-wave teams, multi-wave teams with ticketed histogram waves)."""
    h = _capi.Handle.get(0)
    g = golden("bls_2500")
    t, y, ivar = g["raw_t"], g["raw_y"], g["raw_ivar"]
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(t))
    ts, ys, es, _ = synth.bls_target(3, 1, 6000, cadence_days=10.0 / 1440.0)
    t2, y2, w2, _ = O.lk_bls_inputs(ts, ys, es)
    long_p = np.linspace(15.0, 39.0, 40)           # thousands of bins: the multi-wave teams
    fast = [_capi.bls_batch(t, y, ivar, [0, len(t)], g["period"], g["duration"]),
            _capi.bls_batch(np.concatenate([t[perm], t2]), np.concatenate([y[perm], y2]), np.concatenate([ivar[perm], w2]),
                            [0, len(t), len(t) + len(t2)], long_p, [0.05, 0.2])]
    h.bls_set_ordered_histogram(True)
    try:
        slow = [_capi.bls_batch(t, y, ivar, [0, len(t)], g["period"], g["duration"]),
                _capi.bls_batch(np.concatenate([t[perm], t2]), np.concatenate([y[perm], y2]), np.concatenate([ivar[perm], w2]),
                                [0, len(t), len(t) + len(t2)], long_p, [0.05, 0.2])]
    finally:
        h.bls_set_ordered_histogram(False)
    for a, b in zip(fast, slow):
        for name in _capi.BLS_FIELDS:
            assert np.array_equal(a[name], b[name]), name
    for name in _capi.BLS_FIELDS:
        if name != "transit_time":
            assert np.array_equal(slow[0][name][0], g["likelihood_" + name]), name


@pytest.mark.parametrize("use_like", [True, False])
def test_periods_beyond_the_lds_plan_bit_exact(use_like):
    """Periods whose phase bins do not fit LDS (period > lk_bls_max_period: a long baseline searched with short durations)
    run bls_wide_kernel — bins in global memory, owner-thread histogram, sequential prefix, exhaustive exact scan — and
    must equal the oracle (pinned to astropy's compiled run_bls) bit for bit (`==`, stated) like every other period,
    alone, mixed with LDS-sized periods in one call, for ragged batches and for unsorted times."""
    rng = np.random.default_rng(23)
    durations = np.array([0.02, 0.05, 0.033])                       # bin = 0.002 d at oversample 10 (caller's order kept)
    limit = _capi.bls_max_period(durations, 10)
    assert 5.0 < limit < 40.0
    periods = np.concatenate([np.linspace(limit * 1.02, 3.5 * limit, 9), np.linspace(0.7, 0.98 * limit, 11)])
    rng.shuffle(periods)
    ts, ys, ws = [], [], []
    for b, n in enumerate((2500, 1700, 3100)):
        t = np.sort(rng.uniform(0.0, 400.0, n))
        y = 1e-3 * rng.standard_normal(n)
        p0, t0 = 0.6 * limit * (2.2 + b), 3.0 + b
        y[np.abs((t - t0 + 0.5 * p0) % p0 - 0.5 * p0) < 0.04] -= 6e-3
        if b == 1:                                                  # unsorted: the histogram is order-exact all the same
            perm = rng.permutation(n)
            t, y = t[perm], y[perm]
        ts.append(t - t.min()), ys.append(y - np.median(y)), ws.append(rng.uniform(0.5, 2.0, n) * 1e6)
    t, off = synth.pack_ragged(ts)
    y, w = synth.pack_ragged(ys)[0], synth.pack_ragged(ws)[0]
    res = _capi.bls_batch(t, y, w, off, periods, durations, 10, use_like)
    for b in range(3):
        ref = O.bls(ts[b], ys[b], ws[b], periods, durations, 10, use_like)
        for k, r in zip(_capi.BLS_FIELDS, ref):
            assert np.array_equal(res[k][b], r), (b, k)
    only_wide = periods[periods > limit][:3]
    one = _capi.bls_batch(ts[0], ys[0], ws[0], [0, len(ts[0])], only_wide, durations, 10, use_like)
    ref = O.bls(ts[0], ys[0], ws[0], only_wide, durations, 10, use_like)
    assert all(np.array_equal(one[k][0], r) for k, r in zip(_capi.BLS_FIELDS, ref))
    # a period with several hundred thousand bins (1000 d at 0.002 d per bin): accepted, like astropy
    far = _capi.bls_batch(ts[2], ys[2], ws[2], [0, len(ts[2])], [1000.0], durations, 10, use_like)
    ref = O.bls(ts[2], ys[2], ws[2], np.array([1000.0]), durations, 10, use_like)
    assert all(np.array_equal(far[k][0], r) for k, r in zip(_capi.BLS_FIELDS, ref))


def test_small_job_on_four_streams_equals_the_single_stream_path():
    """A call with B x nP <= 65 536 (a seam call at B = 1) spreads its period groups over four streams of the handle (round 6); the
    same light curve inside a batch beyond that threshold runs them one after the other on the caller's stream.  Seven outputs,
    bit for bit."""
    t, y, e, truth = synth.bls_target(3, 7, 6000)
    tt, yy, ivar, _ = O.lk_bls_inputs(t, y, e)
    period = np.exp(np.linspace(np.log(0.4), np.log(11.0), 3000))
    duration = np.array([0.05, 0.1, 0.2])
    one = _capi.bls_batch(tt, yy, ivar, [0, len(tt)], period, duration)
    nb = 24                                                   # 24 x 3000 > 65 536
    T, off = synth.pack_ragged([tt] * nb)
    Y, _ = synth.pack_ragged([yy] * nb)
    W, _ = synth.pack_ragged([ivar] * nb)
    many = _capi.bls_batch(T, Y, W, off, period, duration)
    for name in _capi.BLS_FIELDS:
        assert np.array_equal(one[name][0], many[name][0]), name
        assert np.array_equal(many[name][0], many[name][-1]), name
