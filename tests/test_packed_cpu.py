"""CPU: the per-BATCH front end (lightkurve_amd/packed.py, batch.py, LightCurveBatch) against the per-OBJECT planning it
replaces (periodogram._ls_plan / _bls_plan, which mirror reference periodogram.py:783-967, 1093-1168).  The kernels are
stood in for by the oracle (tests/oracle_backend.py); what is under test is the host logic: packing, NaN removal, the psd
scale, the BLS inputs, the grid decisions, and that a LightCurveBatch goes to the C ABI without rebuilding objects."""
import os
import socket

import numpy as np
import pytest

from lightkurve_amd import _capi, batch, packed, synth
from lightkurve_amd.ingest import LightCurveBatch
from lightkurve_amd.lightcurve import LightCurve
from lightkurve_amd.periodogram import BoxLeastSquaresPeriodogram, LombScarglePeriodogram, _bls_plan, _ls_plan
from tests import oracle_backend as OB


def _lcs(ns=(300, 90, 500, 120), nan_every=(0, 7, 0, 13), bad_err=(False, False, True, False), seed=3):
    out = []
    for i, n in enumerate(ns):
        t, y, e, _ = synth.ls_target(seed, i, n)
        y = y.copy()
        if nan_every[i]:
            y[::nan_every[i]] = np.nan
        e = e.copy()
        if bad_err[i]:
            e[5] = np.inf
        out.append(LightCurve(time=t + 2457000.0, flux=y, flux_err=e, meta={"TARGETID": i}))
    return out


@pytest.fixture
def oracle_capi(monkeypatch):
    OB.CALLS.clear()

    def scaled(fn):
        def wrapped(t, y, n_off, *a, scale=None, normalization="psd", **kw):
            p = fn(t, y, n_off, *a, normalization=normalization, **kw)
            return p * np.asarray(scale, float)[:, None] if (scale is not None and normalization == "lk_psd") else p
        return wrapped

    monkeypatch.setattr(_capi, "ls_fast_batch", scaled(OB.ls_fast_batch))
    monkeypatch.setattr(_capi, "ls_power_batch", scaled(OB.ls_power_batch))
    monkeypatch.setattr(_capi, "ls_fast_peaks_batch", OB.ls_fast_peaks_batch)
    monkeypatch.setattr(_capi, "bls_batch", OB.bls_batch)
    monkeypatch.setattr(_capi, "argmax_batch", lambda p, device=0: (np.nanmax(p, axis=1), np.nanargmax(p, axis=1)))
    return OB


def test_pack_columns_and_nan_removal_match_the_per_object_steps():
    lcs = _lcs()
    (t, f, e), off = packed.pack_columns(lcs, ("time", "flux", "flux_err"))
    assert np.array_equal(off, np.concatenate([[0], np.cumsum([len(lc) for lc in lcs])]))
    assert np.array_equal(t, np.concatenate([lc.time for lc in lcs]))
    assert np.array_equal(f, np.concatenate([lc.flux for lc in lcs]), equal_nan=True)
    assert packed.any_nan(f) and not packed.any_nan(t)
    t2, f2, off2, e2 = packed.drop_nan_flux(t, f, off, e)
    clean = [lc.remove_nans() for lc in lcs]
    assert np.array_equal(off2, np.concatenate([[0], np.cumsum([len(lc) for lc in clean])]))
    assert np.array_equal(t2, np.concatenate([lc.time for lc in clean]))
    assert np.array_equal(f2, np.concatenate([lc.flux for lc in clean]))
    assert np.array_equal(e2, np.concatenate([lc.flux_err for lc in clean]))
    same = packed.drop_nan_flux(t2, f2, off2)
    assert same[0] is t2 and same[1] is f2                       # no NaN: the inputs themselves, no copy
    assert np.array_equal(packed.rebase_times(t2, off2), np.concatenate([lc.time - lc.time[0] for lc in clean]))
    assert packed.check_sorted(t2, off2)
    bad = t2.copy()
    bad[off2[1] + 3] = bad[off2[1] + 2] - 1.0
    assert not packed.check_sorted(bad, off2)
    assert packed.check_sorted(np.concatenate([[5.0, 6.0], [1.0, 2.0]]), np.array([0, 2, 4]))   # a boundary is not a step
    with pytest.raises(ValueError):
        packed.pack_columns([LightCurve(time=[0.0, 1.0], flux=[1.0, 1.0])] + [type("X", (), {"time": np.zeros(3), "flux": np.zeros(2)})()])


@pytest.mark.parametrize("normalization", ["amplitude", "psd"])
def test_ls_scales_and_grid_plan_are_bit_identical_to_ls_plan(normalization):
    lcs = _lcs()
    freq = 0.05 + 0.01 * np.arange(400)
    plan = packed.ls_grid_plan(freq, normalization=normalization)
    (t, f), off = packed.pack_columns(lcs, ("time", "flux"))
    t, f, off = packed.drop_nan_flux(t, f, off)
    scale = packed.ls_scales(t, off, plan)
    for b, lc in enumerate(lcs):
        ref = _ls_plan(lc, frequency=freq, normalization=normalization)
        assert ref["scale"] == scale[b] and ref["norm"] == plan.norm and ref["ls_method"] == plan.ls_method
        assert np.array_equal(ref["f_day"], plan.f_day) and ref["power_unit"] == plan.power_unit
    irregular = packed.ls_grid_plan(np.sort(np.random.default_rng(0).uniform(0.1, 5, 50)))
    assert irregular.ls_method == "slow" and irregular.exact is None
    assert packed.ls_grid_plan(freq, ls_method="auto").ls_method == "fast"
    assert packed.ls_grid_plan(freq[:100], ls_method="auto").ls_method == "cython"
    with pytest.warns(Warning):
        assert packed.ls_grid_plan(freq, nterms=3).nterms == 1
    with pytest.raises(ValueError):
        packed.ls_grid_plan(freq, normalization="bogus")


def test_bls_inputs_are_bit_identical_to_bls_plan():
    period = np.linspace(0.7, 3.0, 50)
    for ns in ((300, 90, 500, 120), (128, 128, 128)):              # ragged, and the equal-length (reshape) median route
        lcs = _lcs(ns=ns, nan_every=(0, 7, 0, 13)[: len(ns)], bad_err=(False, False, True, False)[: len(ns)])
        (t, f, e), off = packed.pack_columns(lcs, ("time", "flux", "flux_err"))
        tt, yy, ww, off2, t_ref = packed.bls_inputs(t, f, e, off)
        for b, lc in enumerate(lcs):
            ref = _bls_plan(lc, period=period, duration=[0.05, 0.1])
            s = slice(int(off2[b]), int(off2[b + 1]))
            assert np.array_equal(tt[s], ref["t"]) and np.array_equal(yy[s], ref["y"]) and np.array_equal(ww[s], ref["ivar"])
            assert t_ref[b] == ref["t_ref"]


@pytest.mark.parametrize("kw", [dict(), dict(normalization="psd"), dict(ls_method="slow"),
                                dict(ls_method="chi2", nterms=2), dict(ls_method="fastchi2", nterms=2)])
def test_lombscargle_batch_equals_the_per_object_constructor(oracle_capi, kw):
    lcs = _lcs()
    freq = 0.05 + 0.01 * np.arange(120)
    ref = np.stack([LombScarglePeriodogram.from_lightcurve(lc, frequency=freq, **kw).power for lc in lcs])
    oracle_capi.CALLS.clear()
    got_list = batch.lombscargle_batch(lcs, freq, **kw)
    lb = LightCurveBatch.from_lightcurves(lcs)
    lb.to_lightcurves = None                                        # the packed batch must not be exploded into objects
    got_batch = lb.to_periodogram_power(freq, **kw)
    eq = lambda a, b: np.array_equal(a, b, equal_nan=True)         # the FFT method's amplitude is NaN where its power dips < 0
    assert eq(got_list, ref) and eq(got_batch, ref)
    if not kw or kw == dict(normalization="psd"):
        assert oracle_capi.CALLS == ["ls_fast_peaks_lc_batch"] * 2   # the device-rebase entry, once per call
        pk = lb.to_periodogram_peaks(freq, **kw)
        assert eq(pk[:, 0], np.nanmax(ref, axis=1)) and eq(pk[:, 1], np.nanargmax(ref, axis=1))
        out = np.empty_like(ref)
        assert batch.lombscargle_batch(lcs, freq, out=out, **kw) is out and eq(out, ref)


def test_lombscargle_batch_errors_like_the_reference(oracle_capi):
    lcs = _lcs()
    with pytest.raises(ValueError, match="at least two cadences"):
        batch.lombscargle_batch(lcs + [LightCurve(time=[1.0], flux=[1.0])], 0.1 + 0.1 * np.arange(10))
    with pytest.raises(ValueError, match="regular frequency grid"):
        batch.lombscargle_peaks_batch(lcs, np.array([0.1, 0.2, 0.4, 0.5]))
    assert batch.lombscargle_batch([], 0.1 + 0.1 * np.arange(10)).shape == (0, 10)


def test_bls_batch_equals_the_per_object_constructor(oracle_capi):
    lcs = _lcs(ns=(200, 150, 260), nan_every=(0, 9, 0), bad_err=(False, False, True))
    period = np.linspace(0.7, 3.0, 24)
    dur = [0.05, 0.1]
    got = batch.bls_batch(lcs, period, dur)
    got_b = LightCurveBatch.from_lightcurves(lcs).bls(period, dur)
    for b, lc in enumerate(lcs):
        ref = BoxLeastSquaresPeriodogram.from_lightcurve(lc, period=period, duration=dur)
        for i, k in enumerate(_capi.BLS_FIELDS):
            assert np.array_equal(got[b, i], ref._BLS_result[k]), k
    assert np.array_equal(got, got_b)
    with pytest.raises(ValueError, match="shorter than the minimum period"):
        batch.bls_batch(lcs, period, [0.9])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _capi.ls_fast_peaks_batch, _capi.bls_batch = OB.ls_fast_peaks_batch, OB.bls_batch
        lcs = _lcs(ns=(300, 41, 500, 120, 77), nan_every=(0, 7, 0, 13, 0), bad_err=(False,) * 5)
        freq = 0.05 + 0.01 * np.arange(60)
        period = np.linspace(0.7, 3.0, 12)
        lb = LightCurveBatch.from_lightcurves(lcs)
        np.savez(os.path.join(outdir, "p%d.npz" % rank),
                 full=batch.lombscargle_batch(lcs, freq), full_b=lb.to_periodogram_power(freq),
                 part=batch.lombscargle_batch(lcs, freq, gather=False), peaks=lb.to_periodogram_peaks(freq),
                 bls=batch.bls_batch(lb, period, [0.05, 0.1]), calls=np.array(OB.CALLS))
    finally:
        dist.destroy_process_group()


def test_packed_batches_shard_gloo_world2(tmp_path, oracle_capi):
    """The N > 1 path of the new front end: a list is cut BEFORE packing, a LightCurveBatch by slicing its arrays; rows
    come back in input order on every rank."""
    import torch.multiprocessing as mp
    from lightkurve_amd import distributed as D
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    lcs = _lcs(ns=(300, 41, 500, 120, 77), nan_every=(0, 7, 0, 13, 0), bad_err=(False,) * 5)
    freq = 0.05 + 0.01 * np.arange(60)
    ref = batch.lombscargle_batch(lcs, freq)
    ref_bls = batch.bls_batch(lcs, np.linspace(0.7, 3.0, 12), [0.05, 0.1])
    b = D.shard_bounds(5, 2, [len(lc) for lc in lcs])
    assert 0 < b[1] < 5
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), "p%d.npz" % r))
        eq = lambda x, y: np.array_equal(x, y, equal_nan=True)
        assert eq(z["full"], ref) and eq(z["full_b"], ref)
        assert eq(z["part"], ref[b[r]:b[r + 1]])
        assert eq(z["peaks"][:, 0], np.nanmax(ref, axis=1)) and eq(z["peaks"][:, 1], np.nanargmax(ref, axis=1))
        assert np.array_equal(z["bls"], ref_bls)


def test_check_sorted_over_thread_chunks():
    """check_sorted cuts the packed times into chunks for the thread pool: a descending step is found in any chunk, a light
    curve boundary is excused wherever it falls (also exactly at a chunk edge), empty light curves in between do not matter."""
    n1, n2 = 3_000_000, 2_500_000
    off = np.array([0, n1, n1, n1 + n2], dtype=np.int64)                     # (an empty light curve in the middle)
    t = np.concatenate([np.arange(n1, dtype=np.float64), np.arange(n2, dtype=np.float64)])
    assert packed.check_sorted(t, off)                                        # the boundary step n1 - 1 descends: excused
    for i in (0, (1 << 20) - 1, 1 << 20, n1 - 2, n1, n1 + n2 - 2):
        bad = t.copy()
        bad[i + 1] = bad[i] - 1.0
        assert not packed.check_sorted(bad, off), i
    t[n1 - 1] = 1e12                                                          # a huge last time before the boundary: still sorted
    assert packed.check_sorted(t, off)


def test_result_buffers_are_recycled_only_when_released():
    """_capi.result_empty places large results in recycled page-locked blocks.  A block is handed out again only after the
    array, every view of it AND every buffer export of it (memoryview, np.frombuffer over it) has been garbage collected —
    numpy's own ownership rule (the old `sys.getrefcount` test did not see exports); idle blocks are capped
    and `release_pinned_pool()` frees them.  The pool logic runs here over malloc'd blocks (no GPU: no pinned memory)."""
    import ctypes
    import gc
    from lightkurve_amd import _capi
    libc = ctypes.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes = ctypes.c_void_p, [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    freed = []

    def free(addr):
        freed.append(addr)
        libc.free(addr)

    n = 1 << 20
    pool = _capi._ResultPool(libc.malloc, free, idle_limit_bytes=24 * n)
    saved = _capi._RESULT_POOL
    _capi._RESULT_POOL = pool
    try:
        def addr(x):
            return x.__array_interface__["data"][0]

        a = _capi.result_empty(n)
        first = addr(a)
        a[:] = 1.0
        keep = a[5:100]
        del a
        gc.collect()
        b = _capi.result_empty(n)
        assert addr(b) != first and pool.leased == 2        # `keep` still views the first block
        del keep
        gc.collect()
        c = _capi.result_empty((2, n // 2))
        assert addr(c) == first and c.shape == (2, n // 2) and c.dtype == np.float64
        # a buffer export without any ndarray reference blocks reuse too
        mv = memoryview(c)
        del c
        gc.collect()
        d = _capi.result_empty(n)
        assert addr(d) not in (first, addr(b))
        raw = np.frombuffer(mv, dtype=np.uint8)
        del mv
        gc.collect()
        assert addr(_capi.result_empty(n)) != first          # ... and so does an array built over the export
        assert first not in [a_ for _c, a_ in pool.idle]
        del raw
        gc.collect()
        assert first in [a_ for _c, a_ in pool.idle]         # only now is the block idle again
        e = _capi.result_empty(n)
        big = _capi.result_empty((4, n))                     # larger than any idle block: a new one
        assert addr(big) not in (first, addr(b), addr(d)) and big.nbytes == 32 * n
        # the idle cap: a returned block beyond it is freed at once, not kept
        del big, e, d, b
        gc.collect()
        assert pool.leased == 0 and pool.idle_bytes() <= 24 * n and len(freed) >= 1
        _capi.release_pinned_pool()
        assert pool.idle_bytes() == 0
        assert _capi.result_empty(10).base is None           # small results are plain numpy arrays
        import os
        os.environ["LK_RESULT_POOL"] = "0"
        try:
            assert _capi.result_empty(n).base is None
        finally:
            del os.environ["LK_RESULT_POOL"]
    finally:
        _capi._RESULT_POOL = saved


def test_staging_pool_is_per_thread():
    """_capi.pinned_pool keys its staging buffers by (thread, key): two threads packing concurrently never share one
    (ADVICE r5).  Without a GPU runtime the allocation itself fails — then there is nothing to share either."""
    import threading
    from lightkurve_amd import _capi
    out = {}

    def work(name):
        try:
            out[name] = _capi.pinned_pool("t:x", 1000).__array_interface__["data"][0]
        except (OSError, RuntimeError, MemoryError) as e:
            out[name] = repr(e)

    ths = [threading.Thread(target=work, args=(k,)) for k in ("a", "b")]
    [t.start() for t in ths]
    [t.join() for t in ths]
    if all(isinstance(v, int) for v in out.values()):
        assert out["a"] != out["b"]
    assert all(k[1] == "t:x" and isinstance(k[0], int) for k in _capi._POOL if isinstance(k, tuple) and k[1] == "t:x")
