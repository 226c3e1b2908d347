"""CPU (gloo, world_size 2): the N>1 sharding path — cost-balanced contiguous partition, per-rank compute on
the local block only, ragged all-gather back to every rank in input order.  The compute function injected
here is the oracle (tests may use it as a stand-in; the product wires the HIP call into the same sharded_map)."""
import os
import socket

import numpy as np
import pytest

from lightkurve_amd import distributed as D


def test_partition_by_cost_properties():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 5, 64, 1000):
            costs = rng.integers(1, 100, n).astype(float)
            b = D.partition_by_cost(costs, world)
            assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0) and len(b) == world + 1
            if n >= 8 * world:
                loads = np.array([costs[b[r]:b[r + 1]].sum() for r in range(world)])
                assert loads.max() <= costs.sum() / world + costs.max()
    assert np.array_equal(D.shard_bounds(10, 4), [0, 3, 6, 9, 10])
    assert np.array_equal(D.shard_bounds(0, 4), [0, 0, 0, 0, 0])
    with pytest.raises(ValueError):
        D.partition_by_cost([1, -1], 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outdir):
    import torch.distributed as dist
    from lightkurve_amd import synth
    from oracle import np_oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ns = [300, 50, 800, 120, 640, 90, 10]               # ragged: 7 targets over 2 ranks
        items = [synth.ls_target(7, i, n) for i, n in enumerate(ns)]
        freq = 0.05 + 0.05 * np.arange(40)
        seen = []

        def fn(local):
            seen.append(len(local))
            if not local:
                return np.zeros((0, len(freq)))
            return np.stack([O.ls_power(t - t[0], y, None, freq, normalization="lk_amplitude") for t, y, e, _ in local])

        full = D.sharded_map(items, fn, costs=[n * len(freq) for n in ns])
        part = D.sharded_map(items, fn, costs=[n * len(freq) for n in ns], gather=False)
        bounds = D.shard_bounds(len(ns), world, [n * len(freq) for n in ns])
        np.savez(os.path.join(outdir, "r%d.npz" % rank), full=full, part=part, seen=np.array(seen), bounds=bounds)
    finally:
        dist.destroy_process_group()


def test_sharded_map_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    from lightkurve_amd import synth
    from oracle import np_oracle as O
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ns = [300, 50, 800, 120, 640, 90, 10]
    freq = 0.05 + 0.05 * np.arange(40)
    ref = np.stack([O.ls_power(t - t[0], y, None, freq, normalization="lk_amplitude")
                    for t, y, e, _ in (synth.ls_target(7, i, n) for i, n in enumerate(ns))])
    r0, r1 = (np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(2))
    assert np.array_equal(r0["full"], ref) and np.array_equal(r1["full"], ref)      # everyone has everything, in order
    b = r0["bounds"]
    assert np.array_equal(r0["part"], ref[b[0]:b[1]]) and np.array_equal(r1["part"], ref[b[1]:b[2]])
    assert r0["seen"].tolist() == [b[1] - b[0]] * 2 and r1["seen"].tolist() == [b[2] - b[1]] * 2   # local work only
    assert 0 < b[1] < len(ns)


def test_bench_gpus_flag_self_launches_ranks():
    """`python bench.py --gpus 2` without WORLD_SIZE must really form a 2-rank group (the flag was dead).
    --dry runs the launch / rendezvous / barrier / max-over-ranks skeleton on gloo without GPU work."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry", "--total-targets", "11"],
                       env=env, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["dry"] is True
    assert res["targets_total"] == 11 and res["scaling"] == "strong"      # 6 + 5 targets seen by the two ranks
    assert res["gather"] == "summary" and res["gather_checked"] is True and sorted(res["shards"]) == [5, 6]


@pytest.mark.parametrize("extra", [["--gather", "spectra", "--total-targets", "11", "--chunks", "4"],
                                   ["--gather", "spectra", "--targets", "3"],
                                   ["--gather", "none", "--total-targets", "7"]])
def test_bench_dry_gathers_gloo_world2(extra):
    """The gathers of the real multi-GPU step on gloo / CPU tensors: strong scaling with unequal shards (11 targets over 2
    ranks: the smaller shard is padded), the chunked all-gather of the spectra, weak scaling, and no gather at all.  Every
    rank rebuilds the whole batch from what it received (bench.py dry_run)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry"] + extra,
                       env=env, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    res = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["dry"] is True
    if "none" in extra:
        assert res["gather_checked"] is None and res["targets_total"] == 7
    else:
        assert res["gather_checked"] is True
        assert res["targets_total"] == (11 if "--total-targets" in extra else 6)


def test_batch_entry_points_work_without_torch():
    """sharded_map / all_gather_rows imported torch.distributed unconditionally, so the batch entry points raised
    ModuleNotFoundError in the interpreter the seams live in (conda + astropy, no torch).  With torch blocked they must
    behave as a world of one rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.modules['torch'] = None; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from lightkurve_amd import distributed as D\n"
        "import lightkurve_amd.batch, lightkurve_amd.correctors.metrics\n"
        "out = D.sharded_map([1, 2, 3], lambda xs: np.array([[x, 2 * x] for x in xs]), costs=[1, 1, 1])\n"
        "assert out.tolist() == [[1, 2], [2, 4], [3, 6]]\n"
        "assert D.all_gather_rows(out, [0, 3]) is out\n"
        "print('NO_TORCH_OK')\n" % root)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=120)
    assert p.returncode == 0 and b"NO_TORCH_OK" in p.stdout, p.stderr.decode()[-1500:]


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: every path of SURVEY 8(e) shards (flatten, CDPP, regression, PLD beside LS / BLS).  The product entry points
# of lightkurve_amd.batch run here on gloo with the GPU calls of lightkurve_amd._capi replaced by the oracle (test
# stand-ins; the sharding, the ragged packing and the gathers are the code under test).
def _oracle_capi_standins():
    from lightkurve_amd import _capi
    from oracle import np_oracle as O

    def savgol_trend_batch(t, flux, n_off, mask=None, window_length=101, polyorder=2, break_tolerance=5, niters=3,
                           sigma=3, return_fit_mask=False, device=0):
        out = np.empty_like(flux)
        for b in range(len(n_off) - 1):
            a, e = int(n_off[b]), int(n_off[b + 1])
            out[a:e] = O.flatten_trend(t[a:e], flux[a:e], window_length, polyorder, break_tolerance, niters, sigma,
                                       None if mask is None else mask[a:e])[0]
        return out

    def sigma_clip_batch(y, n_off, sigma=5.0, maxiters=5, device=0):
        return np.concatenate([O.sigma_clip_mask(y[int(n_off[b]):int(n_off[b + 1])], sigma, maxiters)
                               for b in range(len(n_off) - 1)])

    def regress_batch(X, y, n_off, err=None, cadence_mask=None, prior_mu=None, prior_sigma=None, sigma=5.0, niters=5,
                      device=0, return_cov=False):
        B = len(n_off) - 1
        K = X.shape[1]
        res = dict(coefficients=np.empty((B, K)), model=np.empty(len(y)), outlier_mask=np.empty(len(y), bool))
        for b in range(B):
            a, e = int(n_off[b]), int(n_off[b + 1])
            r = O.regression_correct(X[a:e], y[a:e], None if err is None else err[a:e],
                                     None if cadence_mask is None else cadence_mask[a:e],
                                     None if prior_mu is None else prior_mu[b], None if prior_sigma is None else prior_sigma[b],
                                     sigma, niters)
            res["coefficients"][b], res["model"][a:e], res["outlier_mask"][a:e] = r["coefficients"], r["model"], r["outlier_mask"]
        return res

    _capi.savgol_trend_batch, _capi.sigma_clip_batch, _capi.regress_batch = savgol_trend_batch, sigma_clip_batch, regress_batch


def _batch_inputs():
    from lightkurve_amd import synth
    from lightkurve_amd.correctors.designmatrix import DesignMatrix
    from lightkurve_amd.lightcurve import LightCurve
    rng = np.random.default_rng(4)
    lcs, dms = [], []
    for i, n in enumerate([400, 260, 700, 310, 520]):           # ragged: 5 targets over 2 ranks
        t, y, e, _ = synth.ls_target(9, i, n)
        lcs.append(LightCurve(time=t, flux=y, flux_err=e))
        X = np.column_stack([np.ones(n), (t - t.mean()) / 10.0, rng.normal(size=n)])
        dms.append(DesignMatrix(X, name="m%d" % i))
    return lcs, dms


def _batch_worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lightkurve_amd import batch
        _oracle_capi_standins()
        lcs, dms = _batch_inputs()
        tr = batch.flatten_batch(lcs, window_length=51)
        tr_local = batch.flatten_batch(lcs, window_length=51, gather=False)
        cd = batch.estimate_cdpp_batch(lcs, savgol_window=51)
        fl, co, ou = batch.regression_correct_batch(lcs, dms)
        fl_l, co_l, ou_l = batch.regression_correct_batch(lcs, dms, gather=False)
        np.savez(os.path.join(outdir, "b%d.npz" % rank), n_tr=len(tr), n_tr_local=len(tr_local),
                 n_fl_local=len(fl_l), co_local=co_l, cd=cd, co=co,
                 **{"tr%d" % i: x for i, x in enumerate(tr)}, **{"fl%d" % i: x for i, x in enumerate(fl)},
                 **{"ou%d" % i: x for i, x in enumerate(ou)})
    finally:
        dist.destroy_process_group()


def test_flatten_cdpp_regression_batches_shard_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    from lightkurve_amd import batch
    world, port = 2, _free_port()
    mp.spawn(_batch_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    # the same entry points without a process group (one rank), same stand-ins
    _saved = {}
    from lightkurve_amd import _capi
    for k in ("savgol_trend_batch", "sigma_clip_batch", "regress_batch"):
        _saved[k] = getattr(_capi, k)
    try:
        _oracle_capi_standins()
        lcs, dms = _batch_inputs()
        tr = batch.flatten_batch(lcs, window_length=51)
        cd = batch.estimate_cdpp_batch(lcs, savgol_window=51)
        fl, co, ou = batch.regression_correct_batch(lcs, dms)
    finally:
        for k, v in _saved.items():
            setattr(_capi, k, v)
    r = [np.load(os.path.join(str(tmp_path), "b%d.npz" % k)) for k in range(2)]
    for k in range(2):
        assert int(r[k]["n_tr"]) == 5
        for i in range(5):
            assert np.array_equal(r[k]["tr%d" % i], tr[i]) and len(tr[i]) == len(lcs[i].time)   # ragged rows, unpadded
            assert np.array_equal(r[k]["fl%d" % i], fl[i]) and np.array_equal(r[k]["ou%d" % i], ou[i])
        assert np.array_equal(r[k]["cd"], cd) and np.array_equal(r[k]["co"], co)
    # gather=False: every rank keeps its own contiguous block, the blocks tile the batch
    assert int(r[0]["n_tr_local"]) + int(r[1]["n_tr_local"]) == 5 and 0 < int(r[0]["n_tr_local"]) < 5
    assert int(r[0]["n_fl_local"]) + int(r[1]["n_fl_local"]) == 5
    assert np.array_equal(np.concatenate([r[0]["co_local"], r[1]["co_local"]]), co)


def _pld_worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lightkurve_amd import batch
        import lightkurve_amd.correctors.pldcorrector as P
        seen = []

        def one_gpu(cubes, device=0, **kw):          # stand-in for the two GPU calls: a recognisable function of the cutout
            seen.append(len(cubes))
            flux = np.stack([np.full(7, float(c)) + np.arange(7) for c in cubes])
            return flux, flux > 3.5

        P.pld_correct_batch = one_gpu
        P.PLDCorrector = lambda c, aperture_mask="all": type("X", (), {"lc": np.zeros(7)})()
        flux, outl = batch.pld_correct_batch([0, 1, 2, 3, 4])
        np.savez(os.path.join(outdir, "p%d.npz" % rank), flux=flux, outl=outl, seen=np.array(seen))
    finally:
        dist.destroy_process_group()


def test_pld_batch_shards_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_pld_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    want = np.arange(5, dtype=float)[:, None] + np.arange(7)
    r = [np.load(os.path.join(str(tmp_path), "p%d.npz" % k)) for k in range(2)]
    for k in range(2):
        assert np.array_equal(r[k]["flux"], want) and np.array_equal(r[k]["outl"], want > 3.5)
    assert sorted(int(x) for x in (r[0]["seen"][0], r[1]["seen"][0])) == [2, 3]      # local work only
