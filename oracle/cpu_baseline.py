"""CPU baseline workers for bench.py (TEST/BENCH INFRASTRUCTURE — imports the oracle, never the product kernels).

Top-level functions so a spawn Pool can pickle them."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def ls_fast_one(job):
    """One target through the numpy port of the reference's default 'fast' LS (oracle.np_oracle.ls_power_fast)."""
    from lightkurve_amd import synth
    from oracle import np_oracle as O
    config, index, n, m = job
    t, y, e, _ = synth.ls_target(config, index, n)
    df = 360.0 / m
    p = O.ls_power_fast(t - t[0], y, None, df, df, m, normalization="lk_amplitude")
    return float(np.nanmax(p))


def ls_exact_rate(n, budget_s=4.0):
    """(cadence, frequency) pairs per second of the exact C oracle on one core."""
    from lightkurve_amd import synth
    from oracle import np_oracle as O
    t, y, e, _ = synth.ls_target(1, 0, n)
    m = 64
    f = (np.arange(m) + 1.0) * 0.0036
    O.ls_power(t - t[0], y, None, f[:4])
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < budget_s:
        O.ls_power(t - t[0], y, None, f)
        reps += 1
    return reps * m * n / (time.perf_counter() - t0)


def bls_rate(n, n_durations, budget_s=15.0):
    """periods*targets per second of the C oracle (restated astropy run_bls) on one core."""
    from lightkurve_amd import synth
    from oracle import np_oracle as O
    t, y, e, _ = synth.bls_target(3, 0, n)
    tt, yy, ivar, _ = O.lk_bls_inputs(t, y, e)
    period, duration = synth.bls_grid(50000, n_durations)
    sel = period[np.linspace(0, len(period) - 1, 24).astype(int)]
    O.bls(tt, yy, ivar, sel[:2], duration)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < budget_s:
        O.bls(tt, yy, ivar, sel, duration)
        reps += 1
    return reps * len(sel) / (time.perf_counter() - t0)
