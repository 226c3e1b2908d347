#!/usr/bin/env python
"""B = 1 latency through the seams: wall time of ONE call of the reference idiom — `lc.to_periodogram()`,
`lc.to_periodogram(method="bls")`, `lc.flatten()`, `RegressionCorrector.correct`, `PLDCorrector.correct`
(src/lightkurve/lightcurve.py:2490-2535, 943-1078; correctors/regressioncorrector.py:191-309, pldcorrector.py:304-427) — on
an UNMODIFIED lightkurve, with the CPU originals and with lightkurve_amd.seams installed (hip backend), at N = 4 000 and
20 000 cadences.  Runs under the conda interpreter of the GPU box with the staged reference (tools/seams_e2e_gpu.sh sets
PYTHONPATH / LD_PRELOAD): median of `reps` calls after one warm-up call, same objects and arguments on both sides.

Where the CPU still wins is said by the table itself; why: a seam call pays the object handling lightkurve does around the
kernel (astropy Time / Quantity / Table construction, ~0.3-3 ms) plus one H2D / D2H round trip and a launch (~0.1 ms) — for
a cheap original (flatten at 4 000 cadences: a 101-tap filter over 4 000 points) that is the same order as the work itself."""
import sys
import time
import warnings

import numpy as np


def wall(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def cases(lk, N):
    from lightkurve.correctors import DesignMatrix, RegressionCorrector
    from lightkurve_amd import synth
    import pandas as pd
    t, y, e, _ = synth.ls_target(1, 5, N)
    lc = lk.LightCurve(time=t + 2000.0, flux=y, flux_err=e)
    tb, yb, eb, _ = synth.bls_target(3, 9, N)
    lcb = lk.LightCurve(time=tb + 2000.0, flux=yb, flux_err=eb)
    freq = (1 + np.arange(100000)) * (360.0 / 100000)
    periods = 1.0 / np.linspace(1 / 13.0, 1 / 0.6, 5000)[::-1]
    rng = np.random.default_rng(3)
    X = np.column_stack([np.sin(2 * np.pi * t / p) for p in np.linspace(0.7, 12.0, 19)] + [np.ones(N)])
    yr = 1 + X[:, :19] @ (1e-3 * rng.standard_normal(19)) + 3e-4 * rng.standard_normal(N)
    lcr = lk.LightCurve(time=t + 2000.0, flux=yr, flux_err=np.full(N, 3e-4))
    dm = DesignMatrix(pd.DataFrame(X), name="X")
    # SURVEY 8(f) rows (round 6): Periodogram.smooth / .flatten, estimate_cdpp, the over-fitting metric
    from lightkurve.correctors.metrics import overfit_metric_lombscargle
    pg = lc.to_periodogram(frequency=freq, normalization="psd", freq_unit="1/d")
    lcc = lcr.copy()
    lcc.flux = lcc.flux + 2e-4 * np.sin(2 * np.pi * t / 0.37) * lcr.flux.unit
    out = [
        ("lc.to_periodogram()  [default grid, ls_method='fast']", lambda: lc.to_periodogram(), 5),
        ("lc.to_periodogram(frequency=1e5 grid)", lambda: lc.to_periodogram(frequency=freq), 5),
        # lightkurve rewrites the default 'fast' to 'slow' for every grid that is not regular in frequency (periodogram.py:933-946)
        ("lc.to_periodogram(period=2000 periods)  ['fast' -> 'slow' by lightkurve]",
         lambda: lc.to_periodogram(period=np.linspace(0.5, 20.0, 2000)), 2),
        # (astropy's own 'slow' builds ~12 N x M temporaries: 1e4 frequencies x 20 000 cadences do not fit the host)
        ("lc.to_periodogram(frequency=1e4 grid, ls_method='slow')" if N <= 4000 else None,
         lambda: lc.to_periodogram(frequency=freq[::10], ls_method="slow"), 2),
        ("lc.to_periodogram(method='bls', 5000 periods x 6 durations)", lambda: lcb.to_periodogram(method="bls", period=periods), 3),
        ("lc.flatten(window_length=101)", lambda: lc.flatten(window_length=101), 5),
        ("lc.flatten(window_length=401)", lambda: lc.flatten(window_length=401), 5),
        ("RegressionCorrector(lc).correct(dm)  [K = 20]", lambda: RegressionCorrector(lcr).correct(dm), 5),
        ("pg.smooth(method='logmedian', filter_width=0.01)  [1e5 frequencies]", lambda: pg.smooth(method="logmedian", filter_width=0.01), 3),
        ("pg.smooth(method='boxkernel', filter_width=0.5)  [1e5 frequencies]", lambda: pg.smooth(method="boxkernel", filter_width=0.5), 3),
        ("pg.flatten()  [1e5 frequencies]", lambda: pg.flatten(), 3),
        ("lc.estimate_cdpp()", lambda: lc.estimate_cdpp(), 5),
        ("overfit_metric_lombscargle(lc, corrected, n_samples=10)", lambda: overfit_metric_lombscargle(lcr, lcc, n_samples=10), 2),
    ]
    return [c for c in out if c[0] is not None]


def pld_case(lk, N):
    from lightkurve.correctors import PLDCorrector
    from lightkurve.targetpixelfile import TargetPixelFileFactory
    from lightkurve_amd import synth
    t, flux, err, _ = synth.pld_cutout(4, 0, n=N, npix=11)
    fac = TargetPixelFileFactory(len(t), 11, 11)
    for k in range(len(t)):
        fac.add_cadence(frameno=k, flux=flux[k], flux_err=err[k], header={"TSTART": 2000.0 + t[k] - 0.0102, "TSTOP": 2000.0 + t[k] + 0.0102})
    tpf = fac.get_tpf(hdu0_keywords={"TELESCOP": "Kepler", "INSTRUME": "Kepler Photometer", "MISSION": "K2", "OBSMODE": "long cadence"},
                      ext_info={"1CRV5P": 100, "2CRV5P": 200, "1CRV4P": 100, "2CRV4P": 200})
    return ("PLDCorrector(tpf).correct(pld_order=3, pca_components=16)  [11 x 11 px, N = %d]" % N,
            lambda: PLDCorrector(tpf, aperture_mask="all").correct(pld_order=3, pca_components=16, pld_aperture_mask="all",
                                                                    background_aperture_mask="all"), 2)


def main():
    import lightkurve as lk
    from lightkurve_amd import seams
    warnings.simplefilter("ignore")
    print("# B = 1 latency through the seams: ms per call (median), CPU original vs HIP seam, unmodified lightkurve %s" % lk.__version__)
    print("%-74s %12s %12s %8s" % ("call", "CPU ms", "HIP seam ms", "CPU/HIP"))
    for N in (4000, 20000):
        print("## N = %d cadences" % N)
        cs = cases(lk, N) + [pld_case(lk, 3500 if N == 4000 else 20000)]
        cpu = [wall(fn, reps) for _, fn, reps in cs]
        seams.install()
        try:
            hip = [wall(fn, reps) for _, fn, reps in cs]
        finally:
            seams.uninstall()
        for (name, _, _), a, b in zip(cs, cpu, hip):
            print("%-74s %12.2f %12.2f %8.1f" % (name, a, b, a / b))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
