"""CPU baseline + accuracy reference from the reference's REAL numerical dependency (TEST/BENCH INFRASTRUCTURE; runs
under the conda interpreter that has astropy; never imported by the product path).

    astropy.timeseries.LombScargle(time, flux, normalization="psd").power(frequency, method=...) exactly as lightkurve
    calls it (src/lightkurve/periodogram.py:961-964) followed by its amplitude normalisation (:974-975), and
    BoxLeastSquares(time, flux, dy).power(period, duration) (:1161-1169), one process per core.

Two modes:

    astropy_baseline.py ls  <n_targets> <N> <M> <procs> [method]         (rate only, inputs from lightkurve_amd.synth)
    astropy_baseline.py bls <n_targets> <N> <n_periods> <n_dur> <procs>
    astropy_baseline.py suite <workdir> <procs>

``suite`` is what bench.py uses: the INPUT ARRAYS of the sampled bench targets are handed over as files
(<workdir>/ls.npz: t, y, off, f0, df, M;  <workdir>/bls.npz: t, y, e, off, period, duration;  <workdir>/flatten.npz:
t, y, off -> flatten_trends.npy, scipy's savgol_filter / interp1d inside the restated lightkurve loop) so both sides
provably see identical numbers, every job is timed (-> cpu_baseline rates), and the per-target results the
BASELINE metric's accuracy columns need are written to <workdir>/result.json: max power and argmax for LS
('fast' = the reference default; 'cython' = the exact method), and for BLS the argmax of power (best-period
index), the max power and the period/duration/depth at the maximum.

    LD_PRELOAD=<system libstdc++> PYTHONPATH=oracle/shims:. /opt/conda/bin/python3.9 -W ignore oracle/astropy_baseline.py ...
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_G = {}   # arrays shared with the fork()ed workers


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


def _suite_ls(job):
    """(max amplitude power, nanargmax) of target b with astropy method `method` on the bench grid."""
    from astropy.timeseries import LombScargle
    b, method = job
    d = _G["ls"]
    s = slice(int(d["off"][b]), int(d["off"][b + 1]))
    t, y = d["t"][s], d["y"][s]
    f = float(d["f0"]) + float(d["df"]) * np.arange(int(d["M"]))
    with np.errstate(all="ignore"):
        p = LombScargle(t, y, normalization="psd").power(f, method=method)
        p = np.sqrt(p) * np.sqrt(4.0 / len(t))
    return float(np.nanmax(p)), int(np.nanargmax(p))


def _suite_bls(b):
    from astropy.timeseries import BoxLeastSquares
    d = _G["bls"]
    s = slice(int(d["off"][b]), int(d["off"][b + 1]))
    r = BoxLeastSquares(d["t"][s], d["y"][s], d["e"][s]).power(d["period"], d["duration"])
    k = int(np.argmax(r.power))
    # the light curve folded at the best period, as lightkurve does it (lightcurve.py:1173-1212): astropy's
    # TimeSeries.fold with the transit time as epoch, then a table sort by the folded time
    import astropy.units as u
    from astropy.time import Time
    from astropy.timeseries import TimeSeries
    ts = TimeSeries(time=Time(d["t"][s], format="mjd"), data={"flux": d["y"][s]})
    folded = ts.fold(period=float(r.period[k]) * u.day, epoch_time=Time(float(r.transit_time[k]), format="mjd"))
    folded.sort("time")
    return (k, float(r.power[k]), float(r.period[k]), float(r.duration[k]), float(r.depth[k]),
            float(r.transit_time[k]), np.asarray(folded["flux"], dtype=np.float64))


def _suite_flatten(b):
    """LightCurve.flatten's trend for light curve b with the reference's own numerical calls — scipy.signal.savgol_filter
    per gap-free segment and scipy.interpolate.interp1d(fill_value="extrapolate") — inside a restatement of the loop of
    src/lightkurve/lightcurve.py:996-1063 (lightkurve itself is not installed on the GPU box; this loop is pinned to it
    by tests/golden/flatten_20k.npz).  Returns the trend."""
    from scipy.interpolate import interp1d
    from scipy.signal import savgol_filter
    d = _G["flatten"]
    s = slice(int(d["off"][b]), int(d["off"][b + 1]))
    time, flux = d["t"][s], d["y"][s]
    window_length, polyorder, break_tolerance, niters, sigma = int(d["window"]) if "window" in d else 401, 2, 5, 3, 3
    mask = np.ones(len(time), dtype=bool)
    with np.errstate(invalid="ignore"):
        extra = np.isfinite(flux)
        extra &= np.nan_to_num(np.abs(flux - np.nanmedian(flux))) <= np.nanstd(flux) * sigma
    mask &= extra
    trend = None
    for _ in range(niters):
        tm, fm = time[mask], flux[mask]
        dt = tm[1:] - tm[:-1]
        cut = np.where(dt > break_tolerance * np.nanmedian(dt))[0] + 1
        low, high = np.append([0], cut), np.append(cut, len(tm))
        tr = np.zeros(len(tm))
        for l, h in zip(low, high):
            if window_length > (h - l) or (h - l) < break_tolerance:
                tr[l:h] = np.nanmedian(fm[l:h])
            else:
                tr[l:h] = savgol_filter(x=fm[l:h], window_length=window_length, polyorder=polyorder)
        mask1 = np.nan_to_num(np.abs(fm - tr)) < (np.nanstd(fm - tr) * sigma + 1e-14)
        trend = interp1d(tm[mask1], tr[mask1], fill_value="extrapolate")(time)
        mask[mask] &= mask1
    return trend


def _warm(_):
    """Imports + one tiny call of each astropy kernel in every worker (not timed)."""
    from astropy.timeseries import BoxLeastSquares, LombScargle
    t = np.linspace(0, 10, 200)
    y = np.sin(t)
    for m in ("fast", "cython"):
        LombScargle(t, y, normalization="psd").power(0.05 + 0.01 * np.arange(300), method=m)
    BoxLeastSquares(t, y, np.ones_like(t)).power(np.linspace(1, 3, 5), 0.2)
    time.sleep(0.05)
    return 0


def _timed_map(pool, fn, jobs, procs):
    pool.map(_warm, range(4 * procs), chunksize=1)
    t0 = time.perf_counter()
    out = pool.map(fn, jobs, chunksize=1)
    return out, time.perf_counter() - t0


def suite(workdir, procs):
    import astropy
    spec = json.load(open(os.path.join(workdir, "suite.json")))
    res = {"astropy": astropy.__version__, "procs": procs}
    if "ls" in spec:
        _G["ls"] = dict(np.load(os.path.join(workdir, "ls.npz")))
    if "bls" in spec:
        _G["bls"] = dict(np.load(os.path.join(workdir, "bls.npz")))
    if "flatten" in spec:
        _G["flatten"] = dict(np.load(os.path.join(workdir, "flatten.npz")))
    with mp.get_context("fork").Pool(procs) as pool:
        if "ls" in spec:
            M = int(_G["ls"]["M"])
            for method, n in (("fast", spec["ls"].get("n_fast", 0)), ("cython", spec["ls"].get("n_exact", 0))):
                if n <= 0:
                    continue
                out, dt = _timed_map(pool, _suite_ls, [(b, method) for b in range(n)], procs)
                res["ls_" + method] = {"n_targets": n, "seconds": dt, "units_per_s": n * M / dt,
                                       "max_power": [o[0] for o in out], "argmax": [o[1] for o in out]}
        if "bls" in spec and spec["bls"].get("n", 0) > 0:
            n = spec["bls"]["n"]
            out, dt = _timed_map(pool, _suite_bls, list(range(n)), procs)
            res["bls"] = {"n_targets": n, "seconds": dt, "units_per_s": n * len(_G["bls"]["period"]) / dt,
                          "argmax": [o[0] for o in out], "max_power": [o[1] for o in out],
                          "period": [o[2] for o in out], "duration": [o[3] for o in out],
                          "depth": [o[4] for o in out], "transit_time": [o[5] for o in out]}
            np.save(os.path.join(workdir, "bls_folded.npy"), np.concatenate([o[6] for o in out]))
        if "flatten" in spec and spec["flatten"].get("n", 0) > 0:
            import scipy
            n = spec["flatten"]["n"]
            out, dt = _timed_map(pool, _suite_flatten, list(range(n)), procs)
            np.save(os.path.join(workdir, "flatten_trends.npy"), np.concatenate(out))
            res["flatten"] = {"n_targets": n, "seconds": dt, "units_per_s": float(len(_G["flatten"]["t"])) / dt,
                              "scipy": scipy.__version__}
    json.dump(res, open(os.path.join(workdir, "result.json"), "w"))
    print("BASELINE " + json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items()
                                                                             if not isinstance(vv, list)})
                                    for k, v in res.items()}))


def main():
    kind = sys.argv[1]
    if kind == "suite":
        return suite(sys.argv[2], int(sys.argv[3]))
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
