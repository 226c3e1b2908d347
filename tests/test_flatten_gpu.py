"""GPU parity: LightCurve.flatten trend (HIP, through the C ABI) vs golden vectors from the reference and the oracle.
Tolerance (stated): trend within 1e-10 relative (fp64; scipy's correlate1d sums symmetric taps pairwise, the kernel
sums taps in order, so last-bit differences are expected); clip decisions identical on the fixtures."""
import numpy as np
import pytest

from lightkurve_amd import LightCurve, _capi, synth
from lightkurve_amd.flatten import flatten_trend_batch
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-10


@pytest.mark.parametrize("name", ["flatten_w101", "flatten_w401", "flatten_w51"])
def test_golden_trend(golden, name):
    g = golden(name)
    bt = None if np.isnan(g["break_tolerance"]) else float(g["break_tolerance"])
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    flat, trend = lc.flatten(window_length=int(g["window_length"]), polyorder=int(g["polyorder"]), break_tolerance=bt,
                             niters=int(g["niters"]), sigma=float(g["sigma"]), return_trend=True)
    assert np.allclose(trend.flux, g["trend"], rtol=RTOL, atol=0, equal_nan=True)
    ok = np.isfinite(g["flux"])
    assert np.allclose(flat.flux[ok], g["flat_flux"][ok], rtol=RTOL) and np.isnan(flat.flux[~ok]).all()
    assert np.allclose(flat.flux_err[ok], g["flat_err"][ok], rtol=RTOL)
    assert flat.meta["NORMALIZED"] is True


def test_golden_user_mask_and_short_segment(golden):
    g = golden("flatten_mask")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    _, trend = lc.flatten(window_length=101, mask=g["mask"], return_trend=True)
    assert np.allclose(trend.flux, g["trend"], rtol=RTOL, atol=0)


def test_golden_bench_shape(golden):
    """bench.py's flatten workload (20 000 cadences, window 401) against the REFERENCE's trend (fixture generated from
    lightkurve itself; inputs regenerated from lightkurve_amd.synth and checked by SHA-256)."""
    import hashlib
    g = golden("flatten_20k")
    n = int(g["n_lc"])
    lcs = [synth.ls_target(6, i, 20000) for i in range(n)]
    for i, (t, y, e, _) in enumerate(lcs):
        assert hashlib.sha256(t.tobytes() + y.tobytes()).hexdigest() == str(g["sha_%d" % i])
    trends = flatten_trend_batch([LightCurve(time=t, flux=y) for t, y, e, _ in lcs], window_length=401, polyorder=2,
                                 break_tolerance=5, niters=3, sigma=3)
    for i in range(n):
        assert np.allclose(trends[i], g["trend_%d" % i], rtol=RTOL, atol=0), i


def test_golden_long_cadence_shape_resident_kernel(golden):
    """4 500 cadences at 30 min (a Kepler quarter), the reference's default window 101: the batch is short enough for the
    LDS-RESIDENT variant of the kernel (compacted time / flux / trend / index map in LDS).  Against lightkurve's own trend
    (fixture from the reference; inputs regenerated and checked by SHA-256), NaN fluxes included; and the same light curves
    in a batch with one 20 000-cadence light curve — which takes the streaming variant — give the same bits."""
    import hashlib
    g = golden("flatten_4500")
    n = int(g["n_lc"])
    lcs = []
    for i, idx in enumerate((0, 1, 3, 4)[:n]):
        t, y, e, _ = synth.ls_target(6, idx, 4500, cadence_days=30.0 / 1440.0)
        if i == 3:
            y = y.copy()
            y[700:705] = np.nan
        assert hashlib.sha256(t.tobytes() + y.tobytes()).hexdigest() == str(g["sha_%d" % i])
        lcs.append(LightCurve(time=t, flux=y))
    trends = flatten_trend_batch(lcs, window_length=101, polyorder=2, break_tolerance=5, niters=3, sigma=3)
    for i in range(n):
        ok = np.isfinite(g["trend_%d" % i])
        assert np.array_equal(ok, np.isfinite(trends[i])), i
        assert np.allclose(trends[i][ok], g["trend_%d" % i][ok], rtol=RTOL, atol=0), i
    t, y, e, _ = synth.ls_target(6, 9, 20000)
    mixed = flatten_trend_batch(lcs + [LightCurve(time=t, flux=y)], window_length=101, polyorder=2, break_tolerance=5,
                                niters=3, sigma=3)
    for i in range(n):
        assert np.array_equal(mixed[i], trends[i], equal_nan=True), i


def test_reference_robustness_cases():
    """reference tests/test_lightcurve.py:1284-1361: NaNs kept, linear data flattens to 1, one outlier survives."""
    lc = LightCurve(time=[1, 2, 3, 4, 5], flux=[np.nan, 1.1, 1.2, np.nan, 1.4])
    assert np.isfinite(lc.flatten(window_length=3).flux).sum() == 3
    t = np.arange(0, 100.0)
    lc = LightCurve(time=t, flux=3 + 0.5 * t, flux_err=0.1)
    for kw in (dict(window_length=3, polyorder=1), dict(window_length=11, polyorder=20), dict(window_length=7, polyorder=2,
                                                                                        break_tolerance=None)):
        flat, trend = lc.flatten(return_trend=True, **kw)
        assert np.allclose(flat.flux, 1.0, rtol=1e-10) and np.allclose(lc.flux, flat.flux * trend.flux, rtol=1e-12)
    n = 2000
    flux = np.ones(n) + 1e-6 * np.sin(np.arange(n) / 50.0)
    flux[1000] = 1.5
    lc = LightCurve(time=np.arange(n, dtype=float), flux=flux)
    flat = lc.flatten(window_length=11, niters=3, sigma=3)
    assert np.isclose(flat.flux, 1.0, rtol=1e-5).sum() == n - 1


def test_ragged_batch_vs_oracle():
    rng = np.random.default_rng(21)
    lcs, masks = [], []
    for i, n in enumerate([3000, 150, 999, 20000, 60]):
        t, y, e, _ = synth.ls_target(6, i, n)
        y = y * (1 + 0.02 * np.sin(2 * np.pi * t / 2.1) + 1e-3 * t)
        y[rng.integers(0, n, max(1, n // 300))] += 0.03
        if n > 200:
            y[rng.integers(0, n, 3)] = np.nan
        lcs.append(LightCurve(time=t, flux=y))
        mk = np.zeros(n, bool)
        mk[n // 2:n // 2 + n // 40] = i % 2 == 0
        masks.append(mk)
    # (401, 2), (257, 3): moment-form interior + moment-form edges; (101, 4), (61, 5): tap-by-tap interior, moment-form edges;
    # (75, 6): operator rows for the edges (polyorder > 5); five iterations exercise the guided dt median repeatedly
    for w, p, bt, ni, sg in [(101, 2, 5, 3, 3), (31, 3, 2, 4, 2.5), (401, 2, 5, 3, 3), (257, 3, 3, 5, 2.5), (101, 4, 5, 3, 3),
                             (61, 5, 5, 3, 3), (75, 6, 5, 2, 3)]:
        trends = flatten_trend_batch(lcs, window_length=w, polyorder=p, break_tolerance=bt, niters=ni, sigma=sg, masks=masks)
        for lc, mk, tr in zip(lcs, masks, trends):
            ref, _ = O.flatten_trend(lc.time, lc.flux, w, p, bt, ni, sg, mask=mk)
            # polyorder >= 4: scipy's savgol_coeffs (and the oracle's) solve an ill-conditioned Vandermonde system in
            # double, so the REFERENCE taps carry a conditioning error of 1e-9 (p = 4, 5) to 1e-7 (p = 6) of the trend; the
            # kernel's taps are the exact solution (long double design) rounded once.  Measured: 8.2e-10, 1.0e-10, 1.5e-7.
            rtol = RTOL if p <= 3 else (5e-9 if p <= 5 else 1e-6)
            assert np.allclose(tr, ref, rtol=rtol, atol=0, equal_nan=True), (len(lc), w, p)


def test_flatten_errors():
    lc = LightCurve(time=[3, 2, 1.0, 4, 5, 6, 7], flux=np.ones(7))
    with pytest.raises(ValueError, match="sorted"):
        lc.flatten(window_length=3)
    with pytest.raises(ValueError, match="odd"):
        LightCurve(time=np.arange(10.0), flux=np.ones(10)).flatten(window_length=4)


def test_infinite_flux_sample_and_leading_nans():
    """with a +inf flux sample numpy's nanstd is NaN (inf - inf) and every comparison of the initial clip is False:
    no cadence is kept (the reference then fails inside interp1d; the kernel returns an all-NaN trend, as its two-pass
    predecessor did) — the one-pass variance used to clamp that NaN to 0 through fmax and fit a bogus trend.  Also: the
    shift of the one-pass moments is the first FINITE sample anywhere (here the first 40 are NaN)."""
    t, y, e, _ = synth.ls_target(6, 1, 3000)
    y1 = y.copy()
    y1[1500] = np.inf
    tr = flatten_trend_batch([LightCurve(time=t, flux=y1)], window_length=101)[0]
    assert np.all(np.isnan(tr))
    y2 = y.copy() * 1e6 + 3e9                  # a large offset: a zero shift would cost ~ eps * mean^2 / var
    y2[:40] = np.nan
    got = flatten_trend_batch([LightCurve(time=t, flux=y2)], window_length=101)[0]
    from oracle import np_oracle as O
    ref = O.flatten_trend(t, y2, 101, 2, 5, 3, 3)[0]
    ok = np.isfinite(ref)
    assert np.array_equal(ok, np.isfinite(got)) and np.allclose(got[ok], ref[ok], rtol=RTOL, atol=0)


def test_segment_cuts_ride_on_the_median_pass_and_fall_back():
    """Round 5: the segment cuts are noted as candidates during the dt median's own pass (flat_dtseg_kernel) and re-tested
    against the exact threshold; a list of more than 384 candidates, a non-positive break tolerance or a median route without
    a bracket falls back to the two-sweep compaction.  Every regime against the oracle: many gaps (overflow), gaps at the
    threshold, break_tolerance 0 / negative / tiny / None, time steps with heavy ties (a regular grid)."""
    rng = np.random.default_rng(8)
    n = 12000
    base = np.arange(n) * (2.0 / 1440.0)
    cases = []
    cases.append(("479 gaps: the candidate list overflows", np.cumsum(np.r_[0.0, np.where(np.arange(1, n) % 25 == 0, 0.05, 2.0 / 1440.0)])))
    cases.append(("regular grid (all steps tie)", base))
    tj = base + rng.normal(0, 1e-5, n)
    tj.sort()
    tg = tj.copy()
    tg[4000:] += 5.0 * (2.0 / 1440.0) * 0.999999        # a gap just below / above 5 x the median step
    tg[8000:] += 5.0 * (2.0 / 1440.0) * 1.000001
    cases.append(("gaps at the threshold", tg))
    cases.append(("three plain gaps", np.r_[tj[:3000], tj[3000:7000] + 0.7, tj[7000:] + 1.9]))
    y0 = 1 + 0.01 * np.sin(2 * np.pi * base / 3.3) + 5e-4 * rng.standard_normal(n)
    y0[rng.integers(0, n, 40)] += 0.02
    for name, tt in cases:
        lc = LightCurve(time=tt, flux=y0)
        for bt in (5, 0.5, 0, -1.0, 1e-9, None):
            tr = flatten_trend_batch([lc], window_length=51, polyorder=2, break_tolerance=bt, niters=3, sigma=3)[0]
            ref, _ = O.flatten_trend(tt, y0, 51, 2, bt, 3, 3)
            assert np.allclose(tr, ref, rtol=RTOL, atol=0, equal_nan=True), (name, bt)


@pytest.mark.parametrize("n,window", [(120_000, 401), (260_000, 101)])
def test_long_light_curves_refine_the_bracket_by_histogram(n, window):
    """Stitched multi-sector light curves (> ~50 000 cadences): the sampled select's bracket holds more values than its LDS
    list, so the bins of one more counting pass narrow it (block_select.hpp) instead of the eight-pass radix select.  Gaps,
    NaN fluxes, a few outliers; the flux median, the time-step median of all three iterations and the final trend go through
    it.  Also a batch that mixes one long light curve with short ones."""
    rng = np.random.default_rng(n)
    t = np.cumsum(np.where(rng.random(n) < 2e-4, rng.uniform(0.5, 3.0, n), 2.0 / 1440.0 * (1 + 1e-9 * rng.standard_normal(n))))
    y = 1.0 + 2e-3 * np.sin(2 * np.pi * t / 3.7) + 5e-4 * rng.standard_normal(n)
    y[rng.choice(n, 40, replace=False)] += 0.02
    y[rng.choice(n, 25, replace=False)] = np.nan
    off = np.array([0, n], dtype=np.int64)
    got = _capi.savgol_trend_batch(t, y, off, window_length=window)
    ref = O.flatten_trend(t, y, window_length=window)[0]
    ok = np.isfinite(ref)
    assert np.array_equal(ok, np.isfinite(got)) and np.allclose(got[ok], ref[ok], rtol=RTOL, atol=0)
    ts, ys, _, _ = synth.ls_target(6, 3, 3000)
    t2 = np.concatenate([ts, t, ts + 1.0])
    y2 = np.concatenate([1.0 + ys, y, 1.0 + ys])
    off2 = np.array([0, 3000, 3000 + n, 6000 + n], dtype=np.int64)
    got2 = _capi.savgol_trend_batch(t2, y2, off2, window_length=window)
    assert np.array_equal(got2[3000:3000 + n], got, equal_nan=True)
    assert np.allclose(got2[:3000], O.flatten_trend(ts, 1.0 + ys, window_length=window)[0], rtol=RTOL, atol=0)


def test_interpolation_without_knot_gather_edge_cases():
    """flat_interp_kernel's scalar search paths against the oracle: outliers clipped in the LAST iteration at the very ends of
    the light curve (the interval clamps to the first / last pair of surviving knots: extrapolation), runs of clipped cadences
    (the neighbouring compacted entries are not knots), duplicated time stamps next to clipped cadences (np.searchsorted 'left'
    semantics), user-masked and NaN cadences before the first and after the last knot, and light curves left with fewer than
    two knots.  80 random draws + hand-made cases, window 11..51."""
    rng = np.random.default_rng(77)
    cases = []
    for trial in range(80):
        n = int(rng.integers(60, 900))
        t = np.cumsum(rng.uniform(0.5, 1.5, n) * 0.02)
        if trial % 3 == 0:                                   # duplicated time stamps (pairs and triples)
            d = rng.choice(n - 2, max(1, n // 15), replace=False) + 1
            t[d] = t[d - 1]
            t = np.sort(t)
        y = 1.0 + 0.01 * np.sin(t / 0.7) + 1e-3 * rng.standard_normal(n)
        k = int(rng.integers(1, 6))
        y[:k] += rng.choice([-1, 1], k) * rng.uniform(0.004, 0.2, k)           # outliers at the very start ...
        y[n - k:] += rng.choice([-1, 1], k) * rng.uniform(0.004, 0.2, k)       # ... and end, of sizes that survive until different iterations
        r0 = int(rng.integers(5, n - 12))
        y[r0:r0 + int(rng.integers(1, 5))] += 0.05                             # a run of outliers
        y[rng.choice(n, 3, replace=False)] = np.nan
        m = np.zeros(n, bool)
        if trial % 2:
            m[:int(rng.integers(0, 4))] = True
            m[n - int(rng.integers(1, 4)):] = True
        cases.append((t, y, m, int(rng.choice([11, 21, 51])), float(rng.choice([2.0, 3.0]))))
    # almost everything clipped: one or zero knots left
    t = np.arange(40) * 0.02
    y = np.where(np.arange(40) % 2 == 0, 1.0, 5.0)
    y[7] = 1.0
    cases.append((t, y, np.zeros(40, bool), 11, 0.2))
    off = np.concatenate([[0], np.cumsum([len(c[0]) for c in cases])]).astype(np.int64)
    for w in (11, 21, 51):
        for s in (2.0, 3.0, 0.2):
            sel = [i for i, c in enumerate(cases) if c[3] == w and c[4] == s]
            if not sel:
                continue
            tt = np.concatenate([cases[i][0] for i in sel])
            yy = np.concatenate([cases[i][1] for i in sel])
            mm = np.concatenate([cases[i][2] for i in sel])
            oo = np.concatenate([[0], np.cumsum([len(cases[i][0]) for i in sel])]).astype(np.int64)
            got, fit = _capi.savgol_trend_batch(tt, yy, oo, mask=mm, window_length=w, sigma=s, return_fit_mask=True)
            for j, i in enumerate(sel):
                ref, rfit = O.flatten_trend(cases[i][0], cases[i][1], window_length=w, sigma=s, mask=cases[i][2])
                g = got[oo[j]:oo[j + 1]]
                ok = np.isfinite(ref)
                assert np.array_equal(ok, np.isfinite(g)), (i, w, s)
                assert np.allclose(g[ok], ref[ok], rtol=1e-9, atol=0), (i, w, s, np.max(np.abs(g[ok] - ref[ok]) / np.abs(ref[ok])))
                assert np.array_equal(fit[oo[j]:oo[j + 1]], rfit), (i, w, s)
