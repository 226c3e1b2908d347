#!/usr/bin/env python
"""Wall time of the product's OTHER per-batch Python entry points (bench.py's `api_end_to_end` block covers the LS ones):
`batch.flatten_batch`, `batch.regression_correct_batch`, `batch.pld_correct_batch`, `batch.bls_batch`, one call each on a
list of objects, median of `reps` after one warm-up, with a cProfile of the last call (top entries by cumulative time) so
that host-side costs can be told from the GPU calls.  Usage: python tools/api_walls.py [flatten|regress|pld|bls ...]"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def wall(fn, reps=3, profile=True):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    txt = ""
    if profile:
        pr = cProfile.Profile()
        pr.enable()
        fn()
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
        txt = "\n".join(l for l in s.getvalue().splitlines() if l.strip() and "function calls" not in l and "Ordered by" not in l)
    return 1e3 * float(np.median(ts)), txt


def main():
    import torch  # noqa: F401  (before liblkhip.so)
    from lightkurve_amd import batch, synth
    from lightkurve_amd.lightcurve import LightCurve
    which = sys.argv[1:] or ["flatten", "regress", "pld", "bls"]
    if "flatten" in which:
        lcs = []
        for i in range(1000):
            t, y, e, _ = synth.ls_target(1, i, 20000)
            lcs.append(LightCurve(time=t, flux=1.0 + y, flux_err=e))
        ms, prof = wall(lambda: batch.flatten_batch(lcs, window_length=401))
        print("flatten_batch(1000 x 20000, window 401): %.1f ms per call (kernel alone 1.8 ms; 320 MB in, 160 MB out)\n%s\n" % (ms, prof))
    if "regress" in which:
        from lightkurve_amd.correctors import DesignMatrix
        rng = np.random.default_rng(2)
        lcs, dms = [], []
        for i in range(256):
            t, y, e, _ = synth.ls_target(1, i, 4000)
            X = np.column_stack([np.sin(2 * np.pi * t / p) for p in np.linspace(0.7, 12.0, 19)] + [np.ones(4000)])
            yr = 1 + X[:, :19] @ (1e-3 * rng.standard_normal(19)) + 3e-4 * rng.standard_normal(4000)
            lcs.append(LightCurve(time=t, flux=yr, flux_err=np.full(4000, 3e-4)))
            dms.append(DesignMatrix(X, name="X"))
        ms, prof = wall(lambda: batch.regression_correct_batch(lcs, dms))
        print("regression_correct_batch(256 x 4000 x K=20): %.1f ms per call\n%s\n" % (ms, prof))
    if "pld" in which:
        from lightkurve_amd.correctors.pldcorrector import PixelCube
        cubes = []
        for i in range(100):
            t, flux, err, _ = synth.pld_cutout(4, i, n=3500, npix=11)
            cubes.append(PixelCube(t, flux.astype(np.float32), err.astype(np.float32), mission="K2"))
        ms, prof = wall(lambda: batch.pld_correct_batch(cubes, pld_order=3, pca_components=16), reps=2)
        print("pld_correct_batch(100 cutouts x 3500 x 11 x 11): %.1f ms per call (device step: 6.8 ms per 100)\n%s\n" % (ms, prof))
    if "bls" in which:
        lcs = []
        for i in range(64):
            t, y, e, _ = synth.bls_target(3, i, 20000)
            lcs.append(LightCurve(time=t, flux=y, flux_err=e))
        periods = 1.0 / np.linspace(1 / 13.0, 1 / 0.6, 5000)[::-1]
        ms, prof = wall(lambda: batch.bls_batch(lcs, periods, duration=[0.05, 0.1, 0.2]))
        print("bls_batch(64 x 20000, 5000 periods x 3 durations): %.1f ms per call\n%s\n" % (ms, prof))


if __name__ == "__main__":
    main()
