"""Batched (many targets at once) entry points of the hot path, sharded over ranks when a process group exists.

These are what a pipeline over a ``LightCurveCollection`` (reference: src/lightkurve/collections.py:145) calls
instead of looping ``lc.to_periodogram()`` / ``lc.flatten()`` per target."""
import numpy as np

from . import _capi
from .distributed import all_gather_rows, device_gather_available, shard_bounds, sharded_map, sharded_map_ragged
from .periodogram import _bls_plan, _ls_plan, exact_grid

__all__ = ["lombscargle_batch", "lombscargle_peaks_batch", "bls_batch", "periodogram_peaks", "flatten_batch",
           "estimate_cdpp_batch", "pld_correct_batch", "regression_correct_batch"]


def _pack(arrs):
    off = np.zeros(len(arrs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(a) for a in arrs])
    return (np.concatenate(arrs) if arrs else np.zeros(0)), off


def lombscargle_batch(lcs, frequency, normalization="amplitude", freq_unit=None, oversample_factor=None,
                      ls_method="fast", device=0, gather=True, nterms=1):
    """Lomb-Scargle power of every light curve on one shared frequency grid -> float64[len(lcs), M].
    Same semantics per target as ``LombScarglePeriodogram.from_lightcurve(lc, frequency=frequency, ...)``
    (``ls_method="fast"``: the reference's default FFT method; any other name: the exact kernels; ``nterms`` > 1
    with ``ls_method`` "chi2"/"fastchi2": the multi-term kernels).
    With torch.distributed initialised, rank r computes a contiguous block of targets (balanced by cadence
    count) and, if ``gather``, the spectra are all-gathered so every rank returns all rows."""
    frequency = np.asarray(frequency, dtype=np.float64)

    def compute(local):
        if not local:
            return np.zeros((0, len(frequency)))
        plans = [_ls_plan(lc, frequency=frequency, normalization=normalization, freq_unit=freq_unit,
                          oversample_factor=oversample_factor, ls_method=ls_method, nterms=nterms) for lc in local]
        t, off = _pack([p["trel"] for p in plans])
        y, _ = _pack([p["flux"] for p in plans])
        f_day = plans[0]["f_day"]
        grid = exact_grid(f_day)
        kw = dict(normalization=plans[0]["norm"], scale=[p["scale"] for p in plans], device=device)
        nt = plans[0]["nterms"]
        if nt > 1 and plans[0]["ls_method"] == "fastchi2":
            return _capi.ls_fast_batch(t, y, off, f0=float(f_day[0]), df=float(f_day[1] - f_day[0]), M=len(f_day),
                                       nterms=nt, **kw)
        if nt > 1:
            if grid is not None:
                return _capi.ls_power_batch(t, y, off, f0=grid[0], df=grid[1], M=len(f_day), nterms=nt, **kw)
            return _capi.ls_power_batch(t, y, off, frequency=f_day, nterms=nt, **kw)
        if plans[0]["ls_method"] in ("fast", "fastchi2"):
            return _capi.ls_fast_batch(t, y, off, f0=float(f_day[0]), df=float(f_day[1] - f_day[0]), M=len(f_day), **kw)
        if grid is not None:
            return _capi.ls_power_batch(t, y, off, f0=grid[0], df=grid[1], M=len(f_day), **kw)
        return _capi.ls_power_batch(t, y, off, frequency=f_day, **kw)

    lcs = list(lcs)
    if gather and nterms == 1 and ls_method == "fast" and device_gather_available() and exact_grid_ok(frequency):
        return _lombscargle_fast_gather_on_device(lcs, frequency, normalization, freq_unit, oversample_factor, device)
    return sharded_map(lcs, compute, costs=[len(lc) for lc in lcs], gather=gather)


def exact_grid_ok(frequency):
    return len(frequency) >= 2 and np.allclose(np.diff(frequency), frequency[1] - frequency[0], rtol=1e-9, atol=0)


def _lombscargle_fast_gather_on_device(lcs, frequency, normalization, freq_unit, oversample_factor, device):
    """The multi-GPU form of ``lombscargle_batch`` for the default method with a RCCL group: this rank's block goes
    host -> HBM once, ``lk_ls_fast_peaks_batch_dev`` writes the spectra into a device tensor, the all-gather runs
    HBM -> xGMI -> HBM, and the full (B, M) result crosses PCIe once on the way out.  (The numpy route stages every
    shard through the host twice more: VERDICT r3.)"""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    bounds = shard_bounds(len(lcs), world, [len(lc) for lc in lcs])
    local = lcs[int(bounds[rank]):int(bounds[rank + 1])]
    M = len(frequency)
    dev = torch.device("cuda", torch.cuda.current_device())
    d_pow = torch.empty((len(local), M), dtype=torch.float64, device=dev)
    if local:
        plans = [_ls_plan(lc, frequency=frequency, normalization=normalization, freq_unit=freq_unit,
                          oversample_factor=oversample_factor, ls_method="fast") for lc in local]
        if plans[0]["ls_method"] != "fast":
            raise ValueError("the device-resident gather needs a regular frequency grid")
        t, off = _pack([p["trel"] for p in plans])
        y, _ = _pack([p["flux"] for p in plans])
        f_day = plans[0]["f_day"]
        d_t, d_y = torch.from_numpy(t).to(dev), torch.from_numpy(y).to(dev)
        d_scale = torch.from_numpy(np.asarray([p["scale"] for p in plans], dtype=np.float64)).to(dev)
        d_max = torch.empty(len(local), dtype=torch.float64, device=dev)
        d_arg = torch.empty(len(local), dtype=torch.int64, device=dev)
        h = _capi.Handle.get(device)
        stream = torch.cuda.current_stream().cuda_stream
        _capi.ls_fast_peaks_batch_dev(h, len(local), off, d_t.data_ptr(), d_y.data_ptr(), 0, float(f_day[0]),
                                      float(f_day[1] - f_day[0]), M, True, True, plans[0]["norm"], d_scale.data_ptr(), 5,
                                      d_pow.data_ptr(), d_max.data_ptr(), d_arg.data_ptr(), stream)
    return all_gather_rows(d_pow, bounds).cpu().numpy()


def lombscargle_peaks_batch(lcs, frequency, normalization="amplitude", freq_unit=None, oversample_factor=None, device=0,
                            gather=True):
    """(max power, argmax) of the default-method (``ls_method="fast"``) periodogram of every light curve on one shared
    regular grid -> float64[len(lcs), 2] (column 1 holds the index).  The spectra never leave the GPU
    (lk_ls_fast_peaks_batch with power = NULL) and with a process group only 16 B per target cross xGMI: this is the
    collective to use when ``Periodogram.max_power`` / ``frequency_at_max_power`` is what the pipeline keeps
    (reference periodogram.py:127-140) — ``lombscargle_batch(gather=True)`` makes every rank hold all B x M powers."""
    frequency = np.asarray(frequency, dtype=np.float64)

    def compute(local):
        if not local:
            return np.zeros((0, 2))
        plans = [_ls_plan(lc, frequency=frequency, normalization=normalization, freq_unit=freq_unit,
                          oversample_factor=oversample_factor, ls_method="fast") for lc in local]
        if plans[0]["ls_method"] != "fast":
            raise ValueError("lombscargle_peaks_batch needs a regular frequency grid (the reference switches to 'slow')")
        t, off = _pack([p["trel"] for p in plans])
        y, _ = _pack([p["flux"] for p in plans])
        f_day = plans[0]["f_day"]
        _pw, mx, am = _capi.ls_fast_peaks_batch(t, y, off, f0=float(f_day[0]), df=float(f_day[1] - f_day[0]), M=len(f_day),
                                                normalization=plans[0]["norm"], scale=[p["scale"] for p in plans],
                                                device=device, want_power=False)
        return np.column_stack([mx, am.astype(np.float64)])

    return sharded_map(list(lcs), compute, costs=[len(lc) for lc in lcs], gather=gather)


def bls_batch(lcs, period, duration, objective="likelihood", oversample=10, device=0, gather=True):
    """BLS power (and the other six statistics) of every light curve on one shared period grid.
    Returns float64[len(lcs), 7, nP] ordered as ``_capi.BLS_FIELDS`` (transit_time absolute, like the reference)."""
    period = np.asarray(period, dtype=np.float64)

    def compute(local):
        if not local:
            return np.zeros((0, 7, len(period)))
        plans = [_bls_plan(lc, period=period, duration=duration, objective=objective, oversample=oversample)
                 for lc in local]
        t, off = _pack([p["t"] for p in plans])
        y, _ = _pack([p["y"] for p in plans])
        w, _ = _pack([p["ivar"] for p in plans])
        res = _capi.bls_batch(t, y, w, off, period, plans[0]["duration"], oversample, objective == "likelihood",
                              device=device)
        out = np.stack([res[k] for k in _capi.BLS_FIELDS], axis=1)
        out[:, 4, :] += np.array([p["t_ref"] for p in plans])[:, None]
        return out

    return sharded_map(list(lcs), compute, costs=[len(lc) for lc in lcs], gather=gather)


def periodogram_peaks(power, device=0):
    """(max_power[B], argmax[B]) of a B x M power matrix on the GPU (Periodogram.max_power / nanargmax)."""
    return _capi.argmax_batch(power, device=device)


def flatten_batch(lcs, window_length=101, polyorder=2, break_tolerance=5, niters=3, sigma=3, masks=None, device=0,
                  gather=True):
    """``LightCurve.flatten`` trends (reference lightcurve.py:943-1078) of every light curve, sharded over the ranks of the
    process group (every target is independent): rank r flattens a contiguous, cadence-balanced block in one
    ``lk_savgol_trend_batch`` call; with ``gather`` every rank returns the list of all trends (ragged rows travel
    NaN-padded), otherwise its own block's."""
    from .flatten import flatten_trend_batch
    lcs = list(lcs)
    masks = [None] * len(lcs) if masks is None else list(masks)
    if len(masks) != len(lcs):
        raise ValueError("masks must hold one entry (array or None) per light curve")
    pairs = list(zip(lcs, masks))

    def compute(local):
        if not local:
            return []
        return flatten_trend_batch([lc for lc, _ in local], window_length=window_length, polyorder=polyorder,
                                   break_tolerance=break_tolerance, niters=niters, sigma=sigma,
                                   masks=[m for _, m in local], device=device)

    return sharded_map_ragged(pairs, compute, [len(lc.time) for lc in lcs], gather=gather)


def estimate_cdpp_batch(lcs, transit_duration=13, savgol_window=101, savgol_polyorder=2, sigma=5.0, device=0, gather=True):
    """``LightCurve.estimate_cdpp`` (reference lightcurve.py:1764-1833; its caller loops over light curves,
    correctors/metrics.py:64-84) of every light curve -> float64[len(lcs)] in ppm, sharded over the ranks."""
    from .lightcurve import estimate_cdpp_batch as one_gpu
    lcs = list(lcs)

    def compute(local):
        if not local:
            return np.zeros((0, 1))
        return one_gpu(local, transit_duration=transit_duration, savgol_window=savgol_window,
                       savgol_polyorder=savgol_polyorder, sigma=sigma, device=device)[:, None]

    out = sharded_map(lcs, compute, costs=[len(lc.time) for lc in lcs], gather=gather)
    return out[:, 0]


def pld_correct_batch(cubes, gather=True, device=0, **kwargs):
    """``PLDCorrector(tpf).correct(...)`` (reference correctors/pldcorrector.py:203-427) for a list of same-shaped cutouts,
    sharded over the ranks: rank r runs ``correctors.pldcorrector.pld_correct_batch`` (design matrices + regression, two GPU
    calls) on its block.  Returns (corrected_flux[B, N], outlier_mask[B, N]) — all B rows on every rank with ``gather``."""
    from .correctors.pldcorrector import pld_correct_batch as one_gpu
    cubes = list(cubes)

    def compute(local):
        if not local:
            return np.zeros((0, 2, 0))
        flux, outl = one_gpu(local, device=device, **kwargs)
        return np.stack([flux, outl.astype(np.float64)], axis=1)

    dist_on = False
    try:
        import torch.distributed as dist
        dist_on = dist.is_available() and dist.is_initialized()
    except ImportError:
        pass
    if dist_on and gather:
        # an empty local block has no cadence count of its own: take it from the batch (same-shaped cutouts)
        from .correctors.pldcorrector import PLDCorrector
        n = len(PLDCorrector(cubes[0], aperture_mask=kwargs.get("aperture_mask", "all")).lc) if cubes else 0

        def compute_n(local):
            return compute(local) if local else np.zeros((0, 2, n))

        out = sharded_map(cubes, compute_n, gather=True)
    else:
        out = sharded_map(cubes, compute, gather=gather)
    return out[:, 0], out[:, 1] != 0.0


def regression_correct_batch(lcs, design_matrices, cadence_masks=None, sigma=5, niters=5, device=0, gather=True):
    """``RegressionCorrector(lc).correct(dm)`` (reference correctors/regressioncorrector.py:191-309) for many light curves
    whose design matrices share their column count K: one ``lk_regress_batch`` call per rank over its block (Gram on the
    fp64 matrix cores, solve, 5-sigma clipping loop per target).  Returns (corrected_flux list, coefficients[B, K],
    outlier_mask list); with ``gather`` for all targets on every rank."""
    from .correctors.designmatrix import DesignMatrix, DesignMatrixCollection
    lcs, dms = list(lcs), list(design_matrices)
    if len(dms) != len(lcs):
        raise ValueError("one design matrix (collection) per light curve")
    dms = [d if isinstance(d, DesignMatrixCollection) else DesignMatrixCollection([d]) for d in dms]
    for d in dms:
        if not isinstance(d, (DesignMatrix, DesignMatrixCollection)):
            raise ValueError("design_matrix_collection must be a DesignMatrix or DesignMatrixCollection")
        d.validate()
    K = dms[0].X.shape[1] if dms else 0
    if any(d.X.shape[1] != K for d in dms):
        raise ValueError("regression_correct_batch needs design matrices with the same number of columns")
    cms = [None] * len(lcs) if cadence_masks is None else list(cadence_masks)
    lens = [len(lc.time) for lc in lcs]
    trip = list(zip(lcs, dms, cms))

    def compute(local):
        # one padded row per target: [flux - model (n), outlier (n), coefficients (K)], NaN beyond n
        if not local:
            return []
        X = np.concatenate([d.X for _, d, _ in local])
        y = np.concatenate([np.asarray(lc.flux, dtype=np.float64) for lc, _, _ in local])
        ns = [len(lc.time) for lc, _, _ in local]
        off = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
        errs = [np.asarray(lc.flux_err, dtype=np.float64) for lc, _, _ in local]
        err = None if all(np.all(~np.isfinite(e)) for e in errs) else np.concatenate(errs)
        cm = np.concatenate([np.ones(n, bool) if c is None else np.asarray(c, dtype=bool) for (_, _, c), n in zip(local, ns)])
        mu = np.stack([d.prior_mu for _, d, _ in local])
        sg = np.stack([d.prior_sigma for _, d, _ in local])
        has_prior = np.any(np.isfinite(sg)) or np.any(mu != 0)
        res = _capi.regress_batch(X, y, off, err=err, cadence_mask=cm, prior_mu=mu if has_prior else None,
                                  prior_sigma=sg if has_prior else None, sigma=sigma, niters=niters, device=device)
        rows = []
        for i, n in enumerate(ns):
            a, b = int(off[i]), int(off[i + 1])
            rows.append(np.concatenate([y[a:b] - res["model"][a:b], res["outlier_mask"][a:b].astype(np.float64),
                                        res["coefficients"][i]]))
        return rows

    rows = sharded_map_ragged(trip, compute, [2 * n + K for n in lens], costs=[n * K * K for n in lens] if K else None,
                              gather=gather)
    if not gather:
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                b = shard_bounds(len(lcs), dist.get_world_size(), [n * K * K for n in lens] if K else lens)
                lens = lens[int(b[dist.get_rank()]):int(b[dist.get_rank() + 1])]
        except ImportError:
            pass
    flux = [r[:n] for r, n in zip(rows, lens)]
    outl = [r[n:2 * n] != 0.0 for r, n in zip(rows, lens)]
    coef = np.stack([r[2 * n:2 * n + K] for r, n in zip(rows, lens)]) if rows else np.zeros((0, K))
    return flux, coef, outl
