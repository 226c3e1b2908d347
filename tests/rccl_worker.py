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
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from lightkurve_amd import batch, synth
    from lightkurve_amd.lightcurve import LightCurve
    try:
        lcs = []
        for i, n in enumerate((900, 300, 1500)):
            t, y, e, _ = synth.ls_target(5, i, n)
            lcs.append(LightCurve(time=t, flux=y, flux_err=e))
        f = synth.ls_frequency_grid(400, fmax=50.0)
        full = batch.lombscargle_batch(lcs, f)                       # gathered over RCCL
        local = batch.lombscargle_batch(lcs, f, gather=False)        # this rank's block only
        # the whole 'fast' path is bitwise reproducible (ordered LDS accumulation): gathered (device-resident route) and
        # local (host route) results are the same bits
        assert full.shape == (3, 400) and np.array_equal(full, local, equal_nan=True)
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
