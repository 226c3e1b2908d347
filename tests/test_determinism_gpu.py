"""GPU: run-twice bitwise determinism of the compute entry points (a run-twice bitwise test).  No kernel
accumulates in an order that depends on scheduling: the BLS histogram and the LS spreader use LDS atomics whose order is
fixed (same-address lanes of one ds_add_f64 in lane order, one wave per cell range), the global-atomic scatter of the
fallback paths ('fastchi2', unordered Lomb-Scargle targets) adds multiples of one quantum (exact, hence commutative),
everything else is plain reductions in a fixed tree.  Each call is made twice in one process and once more after other work has touched the
device (different scratch contents), and compared with ==."""
import numpy as np
import pytest

from lightkurve_amd import _capi, synth

pytestmark = pytest.mark.gpu


def same(a, b):
    if isinstance(a, dict):
        return all(same(a[k], b[k]) for k in a)
    if isinstance(a, (tuple, list)):
        return all(same(x, y) for x, y in zip(a, b))
    if a is None:
        return b is None
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def thrice(fn, disturb, what=""):
    a = fn()
    b = fn()
    disturb()
    c = fn()
    assert same(a, b), what + ": two consecutive calls differ"
    assert same(a, c), what + ": differs after other work ran on the device"


def _disturb():
    t, y, e, off = synth.ls_batch(21, 3, 4000)
    _capi.ls_fast_batch(t - t[0], y, off, f0=0.01, df=0.01, M=20000, normalization="psd")
    tb, yb, eb, offb = synth.bls_batch(22, 2, 3000)
    _capi.bls_batch(tb - tb.min(), yb - np.median(yb), 1 / eb ** 2, offb, np.linspace(0.7, 5, 300), [0.05, 0.2])


def test_ls_entry_points_bitwise():
    t, y, dy, off = synth.ls_batch(1, 6, 6000)
    for b in range(6):
        t[off[b]:off[b + 1]] -= t[off[b]]
    df = 0.004
    thrice(lambda: _capi.ls_fast_batch(t, y, off, dy=dy, f0=df, df=df, M=40000, normalization="lk_amplitude"), _disturb, "ls_fast dy")
    thrice(lambda: _capi.ls_fast_peaks_batch(t, y, off, f0=df, df=df, M=40000, normalization="lk_amplitude"), _disturb, "ls_fast_peaks")
    thrice(lambda: _capi.ls_power_batch(t, y, off, f0=df, df=df, M=5000, normalization="psd"), _disturb, "exact LS")
    thrice(lambda: _capi.ls_power_batch(t, y, off, f0=df, df=df, M=3000, normalization="standard", nterms=2), _disturb, "chi2")
    # The grids of the multi-term 'fastchi2' method and of targets the owner spreader cannot take (unsorted time, a
    # wrapping 2f grid) are filled by lsf_scatter*_kernel with GLOBAL atomics, whose order is not fixed — but every addend is
    # a multiple of one per-target quantum (lsfast.hip: Quantum), so the additions are exact and commute: bitwise all the same.
    thrice(lambda: _capi.ls_fast_batch(t, y, off, f0=df, df=df, M=20000, normalization="psd", nterms=2), _disturb, "fastchi2")
    tu, yu = t.copy(), y.copy()
    rng = np.random.default_rng(5)
    for b in range(0, 6, 2):  # every other target: shuffled cadences (unsorted time)
        p = rng.permutation(off[b + 1] - off[b])
        tu[off[b]:off[b + 1]] = t[off[b]:off[b + 1]][p]
        yu[off[b]:off[b + 1]] = y[off[b]:off[b + 1]][p]
    thrice(lambda: _capi.ls_fast_batch(tu, yu, off, f0=df, df=df, M=40000, normalization="lk_amplitude"), _disturb, "ls_fast unsorted")
    thrice(lambda: _capi.ls_fast_batch(t, y, off, f0=0.05, df=0.05, M=40000, normalization="psd"), _disturb, "ls_fast wrapping grid")


def test_bls_flatten_fold_bitwise():
    tb, yb, eb, offb = synth.bls_batch(3, 5, 5000)
    for b in range(5):
        s = slice(offb[b], offb[b + 1])
        tb[s] -= tb[s].min()
        yb[s] -= np.median(yb[s])
    period = np.linspace(0.6, 9.0, 1500)
    thrice(lambda: _capi.bls_batch(tb, yb, 1 / eb ** 2, offb, period, np.linspace(0.03, 0.4, 30)), _disturb)
    t, y, dy, off = synth.ls_batch(6, 5, 8000)
    thrice(lambda: _capi.savgol_trend_batch(t, y, off, window_length=201, polyorder=2, break_tolerance=5, niters=3, sigma=3),
           _disturb)
    from lightkurve_amd import LightCurve
    lc = LightCurve(time=t[:8000], flux=y[:8000], flux_err=dy[:8000])
    thrice(lambda: (lambda f: (f.time, f.flux, f.flux_err))(lc.fold(period=3.3, epoch_time=t[0])), _disturb)


def test_regression_and_pld_bitwise():
    rng = np.random.default_rng(8)
    n, K, B = 1500, 24, 3
    X = rng.standard_normal((B * n, K))
    y = np.concatenate([X[b * n:(b + 1) * n] @ rng.standard_normal(K) * 1e-3 + 1 + 1e-3 * rng.standard_normal(n) for b in range(B)])
    off = np.arange(B + 1) * n
    thrice(lambda: _capi.regress_batch(X, y, off, err=np.full(B * n, 1e-3), prior_mu=np.zeros(K), prior_sigma=np.full(K, 10.0),
                                       sigma=5.0, niters=5, return_cov=True), _disturb)
    from lightkurve_amd.correctors.pldcorrector import PixelCube, pld_correct_batch
    cubes = [PixelCube(*synth.pld_cutout(4, i, n=800, npix=9)[:3], mission="K2") for i in range(2)]
    thrice(lambda: pld_correct_batch(cubes, pld_order=2, pca_components=8), _disturb)
