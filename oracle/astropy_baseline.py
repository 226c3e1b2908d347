"""CPU baseline from the reference's REAL numerical dependency (bench infrastructure; runs under the conda
interpreter that has astropy):  astropy.timeseries.LombScargle(...).power(frequency, method=...) exactly as
lightkurve calls it (src/lightkurve/periodogram.py:961-964), or BoxLeastSquares(...).power(...) (:1161-1169),
one process per core over a bounded sample of the bench workload.  Prints one JSON line.

    LD_PRELOAD=<system libstdc++> PYTHONPATH=oracle/shims:. /opt/conda/bin/python3.9 -W ignore \\
        oracle/astropy_baseline.py ls <n_targets> <N> <M> <procs> [method]
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _ls_one(job):
    from astropy.timeseries import LombScargle
    from lightkurve_amd import synth
    config, index, n, m, method = job
    t, y, e, _ = synth.ls_target(config, index, n)
    f = synth.ls_frequency_grid(m)
    ls = LombScargle(t, y, normalization="psd")                   # lightkurve passes no dy (uniform weights)
    p = ls.power(f, method=method)
    p = np.sqrt(p) * np.sqrt(4.0 / len(t))                        # lightkurve amplitude normalisation (:974-975)
    return float(np.nanmax(p))


def _bls_one(job):
    from astropy.timeseries import BoxLeastSquares
    from lightkurve_amd import synth
    config, index, n, periods, durations = job
    t, y, e, _ = synth.bls_target(config, index, n)
    r = BoxLeastSquares(t, y, e).power(periods, durations)
    return float(np.max(r.power))


def main():
    kind = sys.argv[1]
    if kind == "ls":
        n_targets, n, m, procs = (int(a) for a in sys.argv[2:6])
        method = sys.argv[6] if len(sys.argv) > 6 else "fast"
        jobs = [(1, i, n, m, method) for i in range(n_targets)]
        fn, units = _ls_one, n_targets * m
    else:
        n_targets, n, n_periods, n_dur, procs = (int(a) for a in sys.argv[2:7])
        from lightkurve_amd import synth
        period, duration = synth.bls_grid(50000, n_dur)
        sel = period[np.linspace(0, len(period) - 1, n_periods).astype(int)]
        jobs = [(3, i, n, sel, duration) for i in range(n_targets)]
        fn, units = _bls_one, n_targets * n_periods
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(fn, jobs[:procs])                                 # warm: imports, FFT plans
        t0 = time.perf_counter()
        pool.map(fn, jobs)
        dt = time.perf_counter() - t0
    import astropy
    print("BASELINE " + json.dumps({"units_per_s": units / dt, "seconds": dt, "procs": procs,
                                    "astropy": astropy.__version__, "n_targets": n_targets}))


if __name__ == "__main__":
    main()
