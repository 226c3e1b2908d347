"""GPU: the structure of the default Lomb-Scargle method — owner-computes spreader (bitwise reproducible), pruned-input
column FFT, fused row FFT + closed form + peak partials, two-stream chunk pipeline, scatter fallback for unsorted targets,
pipelined host-pointer entry point with pinned / pageable buffers and the peaks-only flavour — against the numpy port of
astropy's fast_impl (pinned to the reference at 1e-9).  Tolerance (stated): 1e-9 of the target's maximum power, identical
NaN pattern; the same call twice: bit for bit."""
import numpy as np
import pytest

from lightkurve_amd import _capi, synth
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-9


def relmax(a, b):
    ok = np.isfinite(b)
    assert np.array_equal(ok, np.isfinite(a))
    return np.max(np.abs(a[ok] - b[ok])) / np.max(np.abs(b[ok]))


class host_chunk:
    """Chunk size (MiB of spectra) of the pinned host pipeline for the calls inside the block (lk_set_host_chunk_mb)."""

    def __init__(self, mb):
        self.mb = mb

    def __enter__(self):
        _capi.Handle.get(0).set_host_chunk_mb(self.mb)

    def __exit__(self, *a):
        _capi.Handle.get(0).set_host_chunk_mb(64)


def _batch(B, N, config=1):
    t, y, dy, off = synth.ls_batch(config, B, N)
    for b in range(B):
        t[off[b]:off[b + 1]] -= t[off[b]]
    return t, y, dy, off


def test_full_size_chunks_two_streams_and_determinism():
    """configs[1] shape (N = 20000, M = 1e5, Nfft = 2^19, P = 256 of N1 = 1024 rows) on 90 targets: more than one 85-target
    chunk, so the spreader of chunk 2 runs on the second stream under the FFT of chunk 1.  Two calls agree bit for bit
    (run-twice determinism; the spreader's LDS accumulation order is fixed)."""
    B, N, M = 90, 20000, 100000
    t, y, dy, off = _batch(B, N)
    df = 360.0 / M
    kw = dict(f0=df, df=df, M=M, normalization="lk_amplitude")
    with host_chunk(4096):               # one host chunk: the device path sees all 90 targets at once
        got = _capi.ls_fast_batch(t, y, off, **kw)
    for b in (0, 84, 85, 89):
        ref = O.ls_power_fast(t[off[b]:off[b + 1]], y[off[b]:off[b + 1]], None, df, df, M, normalization="lk_amplitude")
        assert relmax(got[b], ref) < TOL, b
    with host_chunk(4096):
        again = _capi.ls_fast_batch(t, y, off, **kw)
    assert np.array_equal(got, again, equal_nan=True)
    # chunked by the host pipeline instead (80 targets per 64-MB chunk): the same bits again
    assert np.array_equal(got, _capi.ls_fast_batch(t, y, off, **kw), equal_nan=True)


def test_unsorted_and_wrapping_targets_mixed_with_ordered_ones():
    """Targets the owner-computes spreader cannot take go through lsf_zero_kernel + lsf_scatter_kernel (global atomics):
    shuffled cadences inside a batch of ordered targets, and a grid whose time span wraps the 2f grid (span x 2 df > 1)."""
    rng = np.random.default_rng(5)
    ts, ys = [], []
    for i in range(4):
        tt, yy, ee, _ = synth.ls_target(15, i, 2500, cadence_days=10.0 / 1440.0)
        tt = tt - tt[0]
        if i in (1, 3):
            p = rng.permutation(len(tt))
            tt, yy = tt[p], yy[p]
        ts.append(tt), ys.append(yy)
    t, off = synth.pack_ragged(ts)
    y, _ = synth.pack_ragged(ys)
    for M, df in [(30000, 0.004), (3000, 0.04)]:      # 17 d x 2 x 0.04 / d > 1: every target wraps on the second grid
        P = _capi.ls_fast_batch(t, y, off, f0=df, df=df, M=M, normalization="psd")
        for b in range(4):
            ref = O.ls_power_fast(ts[b], ys[b], None, df, df, M, normalization="psd")
            fr = df * (1 + np.arange(M))
            ok = np.isfinite(ref) & (fr * (ts[b].max() - ts[b].min()) >= 1.0)
            assert np.max(np.abs(P[b][ok] - ref[ok])) / np.max(np.abs(ref[ok])) < TOL, (M, b)


def test_pruned_other_fft_sizes_and_weights():
    """Nfft = 2^16, 2^17, 2^18 (N1 = 256, 512; pruned transforms of 32..256 rows), dy weights, psd normalisation."""
    rng = np.random.default_rng(3)
    ns = [2500, 1800, 3000]
    ts, ys, es = [], [], []
    for i, n in enumerate(ns):
        tt, yy, ee, _ = synth.ls_target(12, i, n, cadence_days=(2.0 + 10 * i) / 1440.0)   # different spans -> different rows_used
        ts.append(tt - tt[0])
        ys.append(yy)
        es.append(ee * rng.uniform(0.5, 2, n))
    t, off = synth.pack_ragged(ts)
    y, _ = synth.pack_ragged(ys)
    dy, _ = synth.pack_ragged(es)
    for M, df in [(13000, 0.004), (26000, 0.003), (52000, 0.002)]:
        P = _capi.ls_fast_batch(t, y, off, dy=dy, f0=df, df=df, M=M, normalization="psd")
        for b in range(len(ns)):
            ref = O.ls_power_fast(ts[b], ys[b], es[b], df, df, M, normalization="psd")
            fr = df * (1 + np.arange(M))
            ok = np.isfinite(ref) & (fr * ts[b][-1] >= 1.0)
            d = np.max(np.abs(P[b][ok] - ref[ok])) / np.max(np.abs(ref[ok]))
            assert d < TOL, (M, b, d)


def test_host_pipeline_chunks_pinned_pageable_and_peaks():
    """lk_ls_fast_peaks_batch: many small chunks (double buffers wrap around), pinned vs pageable caller buffers, spectra
    + peaks vs peaks only; peaks == numpy nanmax / nanargmax of the spectra."""
    B, M = 23, 20000
    rng = np.random.default_rng(9)
    ns = rng.integers(800, 2600, B)
    ts, ys = [], []
    for i, n in enumerate(ns):
        tt, yy, ee, _ = synth.ls_target(13, i, int(n))
        ts.append(tt - tt[0])
        ys.append(yy)
    t, off = synth.pack_ragged(ts)
    y, _ = synth.pack_ragged(ys)
    df = 0.01
    kw = dict(f0=df, df=df, M=M, normalization="lk_amplitude")
    with host_chunk(4096):
        one = _capi.ls_fast_batch(t, y, off, **kw)                        # a single chunk
    ref = O.ls_power_fast(ts[5], ys[5], None, df, df, M, normalization="lk_amplitude")
    assert relmax(one[5], ref) < TOL
    with host_chunk(1):                                         # 1 MiB / (M * 8 B) = 6 targets per chunk -> 4 chunks
        pw, mx, am = _capi.ls_fast_peaks_batch(t, y, off, **kw)
        for b in range(B):
            assert relmax(pw[b], one[b]) < 1e-12, b
        assert np.array_equal(am, np.nanargmax(pw, axis=1))
        assert np.array_equal(mx, np.nanmax(pw, axis=1))
        # pinned caller buffers (direct DMA), output into a preallocated pinned array
        ht, hy = _capi.pinned_empty(t.shape), _capi.pinned_empty(y.shape)
        ht[:], hy[:] = t, y
        hp = _capi.pinned_empty((B, M))
        hp[:] = -1.0
        pw2, mx2, am2 = _capi.ls_fast_peaks_batch(ht, hy, off, out=hp, **kw)
        assert pw2 is hp
        for b in range(B):
            assert relmax(hp[b], one[b]) < 1e-12, b
        assert np.array_equal(am2, am)
        # peaks only: the spectra never leave the device
        pw3, mx3, am3 = _capi.ls_fast_peaks_batch(ht, hy, off, want_power=False, **kw)
        assert pw3 is None and np.array_equal(am3, am) and np.allclose(mx3, mx, rtol=1e-12, atol=0)
    with pytest.raises(ValueError):
        _capi.ls_fast_peaks_batch(t, y, off, want_power=False, want_peaks=False, **kw)


def test_peaks_dev_entry_point_matches_host_path():
    """lk_ls_fast_peaks_batch_dev with device-visible pointers.  Pinned host memory (lk_host_alloc) is mapped into the
    GPU's address space, so it serves as the "device" buffers here without needing torch in the test process."""
    B, N, M = 5, 3000, 30000
    t, y, dy, off = _batch(B, N, config=14)
    df = 0.006
    ref = _capi.ls_fast_batch(t, y, off, f0=df, df=df, M=M, normalization="lk_amplitude")
    d_t, d_y = _capi.pinned_empty(t.shape), _capi.pinned_empty(y.shape)
    d_t[:], d_y[:] = t, y
    d_p = _capi.pinned_empty((B, M))
    d_m = _capi.pinned_empty(B)
    d_a = _capi.pinned_empty(B, dtype=np.int64)
    h = _capi.Handle.get(0)
    _capi.ls_fast_peaks_batch_dev(h, B, off, d_t.ctypes.data, d_y.ctypes.data, 0, df, df, M, True, True, "lk_amplitude", 0,
                                  5, d_p.ctypes.data, d_m.ctypes.data, d_a.ctypes.data, 0)
    h.synchronize()
    for b in range(B):
        assert relmax(d_p[b], ref[b]) < 1e-12
    assert np.array_equal(d_a, np.nanargmax(ref, axis=1))
    assert np.array_equal(d_m, np.nanmax(ref, axis=1))


def test_low_amplitude_scatter_paths_keep_their_precision():
    """the scatter fallback rounds its addends to a per-target quantum so that global atomics commute.  With one
    quantum scaled by max(w, |w y|) the w (y - ybar) addends of a low-amplitude light curve kept only ~2^-30 of relative
    precision; the w grids and the w y grids now have their own quanta.  A light curve scaled by 1e-6 (so that |w y| is 1e-6
    of w) must match the reference port as closely as the unscaled one, relative to its own maximum power — through the
    unsorted 'fast' path and through fastchi2."""
    rng = np.random.default_rng(11)
    tt, yy, ee, _ = synth.ls_target(21, 0, 2500, cadence_days=10.0 / 1440.0)
    tt = tt - tt[0]
    p = rng.permutation(len(tt))
    tt, yy = tt[p], yy[p]                              # unsorted: lsf_scatter_kernel
    off = np.array([0, len(tt)], dtype=np.int64)
    M, df = 20000, 0.004
    for scale in (1.0, 1e-6):
        ys = yy * scale
        ref = O.ls_power_fast(tt, ys, None, df, df, M, normalization="psd")
        got = _capi.ls_fast_batch(tt, ys, off, f0=df, df=df, M=M, normalization="psd")[0]
        ok = np.isfinite(ref) & (df * (1 + np.arange(M)) * (tt.max() - tt.min()) >= 1.0)
        assert np.max(np.abs(got[ok] - ref[ok])) / np.max(np.abs(ref[ok])) < TOL, scale
        ref2 = O.ls_power_fastchi2(tt, ys, None, df, df, M, nterms=2, normalization="psd")
        got2 = _capi.ls_fast_batch(tt, ys, off, f0=df, df=df, M=M, nterms=2, normalization="psd")[0]
        ok2 = np.isfinite(ref2) & ok
        assert np.max(np.abs(got2[ok2] - ref2[ok2])) / np.max(np.abs(ref2[ok2])) < TOL, scale
