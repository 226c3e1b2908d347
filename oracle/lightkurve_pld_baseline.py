"""CPU baseline of kind "reference" for the PLD block (TEST/BENCH INFRASTRUCTURE; never imported by the product path):
lightkurve's OWN ``PLDCorrector(tpf).correct(pld_order=3, pca_components=16, pld_aperture_mask='all')``
(src/lightkurve/correctors/pldcorrector.py:304-427) on the bench's synthetic 11 x 11-pixel cutouts, one process per core.

Needs an importable lightkurve: on the GPU box that is the checkout staged by tools/stage_reference.sh (unpacked outside
the repo, LK_REFERENCE_ROOT) under the conda interpreter with oracle/shims on the path — fbpca is not installed there, the
shim's exact-SVD stand-in takes its place, everything else is the reference's code.

    LD_PRELOAD=<system libstdc++> PYTHONPATH=oracle/shims:$LK_REFERENCE_ROOT/src:. /opt/conda/bin/python3.9 -W ignore \
        oracle/lightkurve_pld_baseline.py <n_cutouts> <N> <procs>   ->  one JSON line {"n", "seconds", "cutouts_per_s", ...}
"""
import json
import multiprocessing as mp
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _one(job):
    warnings.simplefilter("ignore")
    from lightkurve.correctors import PLDCorrector
    from lightkurve.targetpixelfile import TargetPixelFileFactory
    from lightkurve_amd import synth
    index, n = job
    t, flux, err, _ = synth.pld_cutout(4, index, n=n, npix=11)
    fac = TargetPixelFileFactory(len(t), 11, 11)
    for k in range(len(t)):
        fac.add_cadence(frameno=k, flux=flux[k], flux_err=err[k],
                        header={"TSTART": 2000.0 + t[k] - 0.0102, "TSTOP": 2000.0 + t[k] + 0.0102})
    tpf = fac.get_tpf(hdu0_keywords={"TELESCOP": "Kepler", "INSTRUME": "Kepler Photometer", "MISSION": "K2",
                                     "OBSMODE": "long cadence"},
                      ext_info={"1CRV5P": 100, "2CRV5P": 200, "1CRV4P": 100, "2CRV4P": 200})
    t0 = time.perf_counter()                       # the timed part is the reference call itself, not the FITS factory
    clc = PLDCorrector(tpf, aperture_mask="all").correct(pld_order=3, pca_components=16, pld_aperture_mask="all",
                                                         background_aperture_mask="all")
    return time.perf_counter() - t0, float(np.nanmedian(clc.flux.value))


def main():
    n, N, procs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    import lightkurve
    jobs = [(i, N) for i in range(n)]
    t0 = time.perf_counter()
    if procs > 1:
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_one, jobs)
    else:
        res = [_one(j) for j in jobs]
    dt = time.perf_counter() - t0
    per_call = [r[0] for r in res]
    print(json.dumps({"n": n, "N": N, "procs": procs, "seconds": dt, "cutouts_per_s": n / dt,
                      "correct_call_seconds_median": float(np.median(per_call)), "lightkurve": lightkurve.__version__,
                      "pool_seconds_include": "TargetPixelFileFactory construction + PLDCorrector.correct"}))


if __name__ == "__main__":
    main()
