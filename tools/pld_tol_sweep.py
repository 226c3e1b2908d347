#!/usr/bin/env python
"""Corrected-flux error, outlier-mask equality and PLD step time as a function of the eigen-solver's stop threshold
(VERDICT r4 #3 (i)).  Needs a development build of the library (-DLK_PLD_DEBUG on pld.hip: reads LK_PLD_TOL) passed as
LK_LIB_PATH; every threshold runs in a fresh process (the library reads the variable per call, the handle caches nothing of it).

    LK_LIB_PATH=build/ab/plddbg.so python tools/pld_tol_sweep.py [tol ...]

Per threshold: max relative error of the corrected flux against the REFERENCE's goldens (pld_c5: 3 cutouts of the bench
shape; pld_k2sin_order3; pld_factory11_order2), whether every outlier mask equals the reference's, and ms per 500-cutout step
of `bench.py --workload pld`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from tests.conftest import load_golden
from lightkurve_amd import synth
from lightkurve_amd.correctors import PixelCube, PLDCorrector, pld_correct_batch
out = {}
g = load_golden("pld_c5")
n = int(g["n_cutouts"])
cubes = [PixelCube(g["time_%%d" %% i], *synth.pld_cutout(4, i, n=3500, npix=11)[1:3], mission="K2") for i in range(n)]
corrected, outl = pld_correct_batch(cubes, pld_order=3, pca_components=16)
out["c5_err"] = max(float(np.max(np.abs(corrected[i] - g["corrected_%%d" %% i])) / np.median(g["corrected_%%d" %% i])) for i in range(n))
out["c5_masks"] = all(bool(np.array_equal(outl[i], g["outlier_mask_%%d" %% i])) for i in range(n))
g = load_golden("pld_k2sin_order3")
pld = PLDCorrector(PixelCube(g["time"], g["flux"], g["flux_err"]))
clc = pld.correct(pld_order=3, pca_components=16, pld_aperture_mask="all", normalize_background_pixels=True)
out["k2sin_err"] = float(np.max(np.abs(clc.flux - g["corrected"])) / np.median(g["corrected"]))
out["k2sin_masks"] = bool(np.array_equal(pld.outlier_mask, g["outlier_mask"]))
g = load_golden("pld_factory11_order2")
pld = PLDCorrector(PixelCube(g["time"], g["flux"], g["flux_err"], mission="K2"), aperture_mask="all")
clc = pld.correct(pld_order=2, pca_components=8, pld_aperture_mask="all", background_aperture_mask="all", spline_degree=3)
out["factory_err"] = float(np.max(np.abs(clc.flux - g["corrected"])) / np.median(g["corrected"]))
out["factory_masks"] = bool(np.array_equal(pld.outlier_mask, g["outlier_mask"]))
import json; print("RESULT " + json.dumps(out))
''' % ROOT


def main():
    tols = sys.argv[1:] or ["1e-10", "1e-9", "1e-8", "1e-7", "1e-6", "1e-5", "1e-4", "1e-3"]
    print("# tol        c5 err (3 cutouts)  masks   k2sin order 3  masks   factory order 2  masks   ms / 500 cutouts")
    for tol in tols:
        env = dict(os.environ, LK_PLD_TOL=tol)
        r = subprocess.run([sys.executable, "-c", CHECK], env=env, capture_output=True, text=True, cwd=ROOT)
        line = [x for x in r.stdout.splitlines() if x.startswith("RESULT ")]
        if not line:
            print(tol, "check failed:", r.stderr[-300:])
            continue
        o = json.loads(line[0][7:])
        b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "pld", "--no-cpu-baseline", "--steps", "5",
                            "--warmup", "2"], env=env, capture_output=True, text=True, cwd=ROOT)
        try:
            ms = json.loads(b.stdout.strip().splitlines()[-1])["ms_per_step"]
        except Exception:
            ms = float("nan")
        print("%-10s  %.3e           %-5s   %.3e      %-5s   %.3e        %-5s   %.2f" % (
            tol, o["c5_err"], o["c5_masks"], o["k2sin_err"], o["k2sin_masks"], o["factory_err"], o["factory_masks"], ms))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
