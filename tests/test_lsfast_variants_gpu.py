"""GPU: the round-2 structure of the default Lomb-Scargle method — pruned-input column FFT (16-column tiles, natural and
permuted row order), two-stream chunk pipeline, pipelined host-pointer entry point with pinned / pageable buffers and
the peaks-only flavour — every variant against the numpy port of astropy's fast_impl (pinned to the reference at 1e-9)
and against each other.  Tolerance (stated): 1e-9 of the target's maximum power, identical NaN pattern; variants
among themselves 1e-12 (LDS atomics accumulate in no fixed order, so not bit for bit)."""
import os

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


class env:
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _batch(B, N, config=1):
    t, y, dy, off = synth.ls_batch(config, B, N)
    for b in range(B):
        t[off[b]:off[b + 1]] -= t[off[b]]
    return t, y, dy, off


def test_pruned_permuted_and_two_stream_variants_full_size():
    """configs[1] shape (N = 20000, M = 1e5, Nfft = 2^19, P = 256 of N1 = 1024 rows) on 7 targets cut into chunks of 2."""
    B, N, M = 7, 20000, 100000
    t, y, dy, off = _batch(B, N)
    df = 360.0 / M
    kw = dict(f0=df, df=df, M=M, normalization="lk_amplitude")
    ref0 = O.ls_power_fast(t[:N], y[:N], None, df, df, M, normalization="lk_amplitude")
    ref6 = O.ls_power_fast(t[off[6]:off[7]], y[off[6]:off[7]], None, df, df, M, normalization="lk_amplitude")
    with env(LK_LSF_PRUNED=0, LK_LSF_STREAMS=0):
        base = _capi.ls_fast_batch(t, y, off, **kw)
    assert relmax(base[0], ref0) < TOL and relmax(base[6], ref6) < TOL
    for var in (dict(LK_LSF_PRUNED=1, LK_LSF_PERM=0, LK_LSF_STREAMS=0),
                dict(LK_LSF_PRUNED=1, LK_LSF_PERM=1, LK_LSF_STREAMS=0),
                dict(LK_LSF_PRUNED=1, LK_LSF_PERM=0, LK_LSF_STREAMS=1, LK_FAST_CHUNK_MB=64),
                dict(LK_LSF_PRUNED=0, LK_LSF_STREAMS=1, LK_FAST_CHUNK_MB=64),
                dict(LK_LSF_PRUNED=1, LK_LSF_PERM=1, LK_LSF_STREAMS=1, LK_FAST_CHUNK_MB=30)):
        with env(**var):
            got = _capi.ls_fast_batch(t, y, off, **kw)
        assert relmax(got[0], ref0) < TOL and relmax(got[6], ref6) < TOL, var
        for b in range(B):
            assert relmax(got[b], base[b]) < 1e-12, (var, b)


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
        for perm in (0, 1):
            with env(LK_LSF_PRUNED=1, LK_LSF_PERM=perm):
                P = _capi.ls_fast_batch(t, y, off, dy=dy, f0=df, df=df, M=M, normalization="psd")
            for b in range(len(ns)):
                ref = O.ls_power_fast(ts[b], ys[b], es[b], df, df, M, normalization="psd")
                fr = df * (1 + np.arange(M))
                ok = np.isfinite(ref) & (fr * ts[b][-1] >= 1.0)
                d = np.max(np.abs(P[b][ok] - ref[ok])) / np.max(np.abs(ref[ok]))
                assert d < TOL, (M, perm, b, d)


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
    with env(LK_HOST_CHUNK_MB=4096):
        one = _capi.ls_fast_batch(t, y, off, **kw)                        # a single chunk
    ref = O.ls_power_fast(ts[5], ys[5], None, df, df, M, normalization="lk_amplitude")
    assert relmax(one[5], ref) < TOL
    with env(LK_HOST_CHUNK_MB=1):                                         # 1 MiB / (M * 8 B) = 6 targets per chunk -> 4 chunks
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


@pytest.mark.parametrize("envkw", [dict(LK_LSF_FUSED_SPREAD="1"), dict(LK_LSF_ROWS_STREAM="1", LK_FAST_CHUNK_MB="64"),
                                   dict(LK_FFT3="1")])
def test_opt_in_variants_match_default(tmp_path, envkw):
    """The opt-in structures of the default method — LK_LSF_FUSED_SPREAD=1 (the extirpolation inside the pruned column
    kernel, search-free through the per-16-cell table), LK_LSF_ROWS_STREAM=1 (row transforms of chunk k on a second stream
    under the column transforms of chunk k + 1; 64-MB chunks = 2 targets each here), LK_FFT3=1 (three-phase row kernel) —
    against the default path.  The switches are read once per process, hence the subprocess."""
    import subprocess
    import sys
    B, N, M = 5, 20000, 100000
    t, y, dy, off = _batch(B, N)
    df = 360.0 / M
    ref = _capi.ls_fast_batch(t, y, off, dy=dy, f0=df, df=df, M=M, normalization="lk_amplitude")
    inp, outp = str(tmp_path / "in.npz"), str(tmp_path / "out.npy")
    np.savez(inp, t=t, y=y, dy=dy, off=off)
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from lightkurve_amd import _capi; d = np.load(%r); "
            "p = _capi.ls_fast_batch(d['t'], d['y'], d['off'], dy=d['dy'], f0=%r, df=%r, M=%d, normalization='lk_amplitude'); "
            "np.save(%r, p)" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), inp, df, df, M, outp))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, **envkw), timeout=600)
    var = np.load(outp)
    for b in range(B):
        assert relmax(var[b], ref[b]) < 1e-12, b
    port = O.ls_power_fast(t[off[0]:off[1]], y[off[0]:off[1]], dy[off[0]:off[1]], df, df, M, normalization="lk_amplitude")
    assert relmax(var[0], port) < TOL
