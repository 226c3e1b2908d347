"""Worker of tests/test_distributed_gpu.py: runs in its own interpreter so that torch (with the ROCm runtime it bundles)
is imported BEFORE liblkhip.so is loaded — the order bench.py uses; the other way round the process ends up with two HIP
runtimes and torch sees no device."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    """argv: [rank world port] — default one rank.  With world > 1 every rank sees ALL GPUs (no HIP_VISIBLE_DEVICES) and
    drives GPU `rank`: tensors, stream and lk_handle of the device-resident gather must all belong to that one device
    (the handle used to default to device 0)."""
    import torch
    import torch.distributed as dist
    rank, world, port = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]) if len(sys.argv) > 3 else (0, 1, str(_free_port()))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from lightkurve_amd import batch, synth
    from lightkurve_amd.distributed import shard_bounds
    from lightkurve_amd.ingest import LightCurveBatch
    from lightkurve_amd.lightcurve import LightCurve
    try:
        lcs = []
        for i, n in enumerate((900, 300, 1500, 700, 1100)[: 3 if world == 1 else 5]):
            t, y, e, _ = synth.ls_target(5, i, n)
            lcs.append(LightCurve(time=t, flux=y, flux_err=e))
        f = synth.ls_frequency_grid(400, fmax=50.0)
        full = batch.lombscargle_batch(lcs, f)                       # gathered over RCCL (device-resident route)
        local = batch.lombscargle_batch(lcs, f, gather=False)        # this rank's block only (host route)
        b = shard_bounds(len(lcs), world, [len(lc) for lc in lcs])
        # the whole 'fast' path is bitwise reproducible (ordered LDS accumulation): gathered and local results are the same bits
        assert full.shape == (len(lcs), 400) and np.array_equal(full[b[rank]:b[rank + 1]], local, equal_nan=True)
        assert torch.cuda.current_device() == rank                   # the calls left the thread's device alone
        pk = batch.lombscargle_peaks_batch(LightCurveBatch.from_lightcurves(lcs), f)
        assert np.array_equal(pk[:, 0], np.nanmax(full, axis=1)) and np.array_equal(pk[:, 1], np.nanargmax(full, axis=1))
        if world > 1:
            print("RCCL_WORKER_OK rank %d" % rank)
            return
        for b, lc in enumerate(lcs):
            single = lc.to_periodogram(frequency=f)
            assert np.max(np.abs(full[b] - np.asarray(single.power))) <= 1e-11 * np.max(full[b])
        exact = batch.lombscargle_batch(lcs, f, ls_method="chi2", nterms=2)
        assert exact.shape == (3, 400) and np.all(np.isfinite(exact))
        # round 4: the other sharded paths under the RCCL group (flatten, CDPP, regression) against the per-target calls
        trends = batch.flatten_batch(lcs, window_length=51)
        for lc, tr in zip(lcs, trends):
            assert np.array_equal(tr, np.asarray(lc.flatten(window_length=51, return_trend=True)[1].flux), equal_nan=True)
        cd = batch.estimate_cdpp_batch(lcs, savgol_window=51)
        assert np.allclose(cd, [lc.estimate_cdpp(savgol_window=51) for lc in lcs], rtol=1e-12, atol=0)
        from lightkurve_amd.correctors import DesignMatrix, RegressionCorrector
        dms = [DesignMatrix(np.column_stack([np.ones(len(lc.time)), lc.time - lc.time.mean()]), name="lin") for lc in lcs]
        flux, coef, outl = batch.regression_correct_batch(lcs, dms)
        for lc, dm, fl, co in zip(lcs, dms, flux, coef):
            rc = RegressionCorrector(lc)
            one = rc.correct(dm)
            assert np.allclose(fl, one.flux, rtol=0, atol=1e-12 * np.max(np.abs(one.flux))) and np.allclose(co, rc.coefficients, rtol=1e-9)
    finally:
        dist.destroy_process_group()
    print("RCCL_WORKER_OK")


if __name__ == "__main__":
    main()
