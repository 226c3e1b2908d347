"""Batched (many targets at once) entry points of the hot path, sharded over ranks when a process group exists.

These are what a pipeline over a ``LightCurveCollection`` (reference: src/lightkurve/collections.py:145) calls
instead of looping ``lc.to_periodogram()`` / ``lc.flatten()`` per target."""
import numpy as np

from . import _capi
from .distributed import all_gather_rows, device_gather_available, shard_bounds, sharded_map, sharded_map_ragged
from . import packed

__all__ = ["lombscargle_batch", "lombscargle_peaks_batch", "bls_batch", "periodogram_peaks", "flatten_batch",
           "estimate_cdpp_batch", "pld_correct_batch", "regression_correct_batch"]


def _resolve_device(device):
    """The GPU this process drives: ``device`` if given; otherwise torch's current device under a RCCL process group
    (one process per GPU: the launcher's LOCAL_RANK -> torch.cuda.set_device), else 0."""
    if device is not None:
        return int(device)
    if device_gather_available():
        import torch
        return int(torch.cuda.current_device())
    return 0


def _slice_packed(cols, n_off, b0, b1):
    a, b = int(n_off[b0]), int(n_off[b1])
    return [c[a:b] for c in cols], n_off[b0:b1 + 1] - n_off[b0]


def _local_packed(lcs, columns, prefix="pack", pinned="auto"):
    """This rank's block of the batch as packed arrays -> (columns, n_off, bounds).  ``lcs``: a ``LightCurveBatch`` (its
    arrays are used as they are; a block is a slice) or an iterable of light curves (one concatenation per column into
    the pinned staging pool).  Without a process group the block is the whole batch and ``bounds`` is None; with one the
    batch is cut into contiguous blocks balanced by cadence count (distributed.shard_bounds) BEFORE packing, so a rank
    only touches its own light curves."""
    from .distributed import _dist
    from .ingest import LightCurveBatch
    dist = _dist()
    multi = dist is not None and dist.is_initialized()
    if isinstance(lcs, LightCurveBatch):
        cols, n_off = [getattr(lcs, c) for c in columns], lcs.n_off
        if not multi:
            return cols, n_off, None
        bounds = shard_bounds(len(n_off) - 1, dist.get_world_size(), np.diff(n_off))
        r = dist.get_rank()
        c, o = _slice_packed(cols, n_off, int(bounds[r]), int(bounds[r + 1]))
        return c, o, bounds
    lcs = list(lcs)
    bounds = None
    if multi:
        bounds = shard_bounds(len(lcs), dist.get_world_size(), [len(lc.time) for lc in lcs])
        r = dist.get_rank()
        lcs = lcs[int(bounds[r]):int(bounds[r + 1])]
    cols, n_off = packed.pack_columns(lcs, columns, pinned=pinned, pool_prefix=prefix)
    return cols, n_off, bounds


def _finish(local, bounds, gather):
    return all_gather_rows(local, bounds) if (bounds is not None and gather) else local


def _ls_power_packed(time, flux, n_off, plan, device, want_power=True, want_peaks=False, out=None):
    """Lomb-Scargle of a packed batch on the plan's shared grid -> (power[B, M] or None, max[B] or None, argmax[B] or None).
    One vectorised NaN pass and one endpoint gather on the host; ``ls_method="fast"`` hands the absolute times to
    ``lk_ls_fast_peaks_lc_batch`` (rebased on the device), the exact methods rebase in one numpy pass."""
    time, flux, n_off = packed.drop_nan_flux(time, flux, n_off)
    counts = np.diff(n_off)
    B, M = len(counts), len(plan.f_day)
    if B == 0:
        return (np.zeros((0, M)) if want_power else None, np.zeros(0) if want_peaks else None,
                np.zeros(0, np.int64) if want_peaks else None)
    if counts.min() < 2:
        raise ValueError("The light curve needs at least two cadences to build a periodogram.")
    scale = packed.ls_scales(time, n_off, plan)
    f_day = plan.f_day
    kw = dict(normalization=plan.norm, scale=scale, device=device)
    if plan.nterms == 1 and plan.ls_method in ("fast", "fastchi2"):
        return _capi.ls_fast_peaks_batch(time, flux, n_off, f0=float(f_day[0]), df=float(f_day[1] - f_day[0]), M=M,
                                         want_power=want_power, want_peaks=want_peaks, out=out, absolute_time=True, **kw)
    trel = packed.rebase_times(time, n_off)
    if plan.nterms > 1 and plan.ls_method == "fastchi2":
        power = _capi.ls_fast_batch(trel, flux, n_off, f0=float(f_day[0]), df=float(f_day[1] - f_day[0]), M=M,
                                    nterms=plan.nterms, **kw)
    elif plan.exact is not None:
        power = _capi.ls_power_batch(trel, flux, n_off, f0=plan.exact[0], df=plan.exact[1], M=M, nterms=plan.nterms, **kw)
    else:
        power = _capi.ls_power_batch(trel, flux, n_off, frequency=f_day, nterms=plan.nterms, **kw)
    mx = am = None
    if want_peaks:
        mx, am = _capi.argmax_batch(power, device=device)
    if out is not None and want_power:
        out[...] = power
        power = out
    return (power if want_power else None), mx, am


def lombscargle_batch(lcs, frequency, normalization="amplitude", freq_unit=None, oversample_factor=None,
                      ls_method="fast", device=None, gather=True, nterms=1, out=None):
    """Lomb-Scargle power of every light curve on one shared frequency grid -> float64[len(lcs), M].
    Same result per target as ``LombScarglePeriodogram.from_lightcurve(lc, frequency=frequency, ...)``
    (``ls_method="fast"``: the reference's default FFT method; any other name: the exact kernels; ``nterms`` > 1
    with ``ls_method`` "chi2"/"fastchi2": the multi-term kernels), but planned per BATCH: the grid is examined once,
    the light curves are packed by one concatenation per column (``lcs``: a list of light curves — this package's or
    lightkurve's — or a ``LightCurveBatch``, whose arrays are used as they are).  ``out``: preallocated float64[B, M]
    (e.g. ``_capi.pinned_empty``) for the single-process result.
    With torch.distributed initialised, rank r computes a contiguous block of targets (balanced by cadence
    count) and, if ``gather``, the spectra are all-gathered so every rank returns all rows."""
    device = _resolve_device(device)
    plan = packed.ls_grid_plan(frequency, normalization, freq_unit, oversample_factor, ls_method, nterms)
    if gather and plan.nterms == 1 and plan.ls_method == "fast" and device_gather_available():
        return _lombscargle_fast_gather_on_device(lcs, plan, device)
    (time, flux), n_off, bounds = _local_packed(lcs, ("time", "flux"))
    power = _ls_power_packed(time, flux, n_off, plan, device, out=out if bounds is None else None)[0]
    return _finish(power, bounds, gather)


def _lombscargle_fast_gather_on_device(lcs, plan, device):
    """The multi-GPU form of ``lombscargle_batch`` for the default method with a RCCL group: this rank's block goes
    host -> HBM once, ``lk_ls_fast_peaks_batch_dev`` writes the spectra into a device tensor, the all-gather runs
    HBM -> xGMI -> HBM, and the full (B, M) result crosses PCIe once on the way out.  Tensors, stream and the lk_handle
    all belong to ONE device (``device``: the rank's own GPU), made current for the duration."""
    import torch
    (t_l, y_l), off, bounds = _local_packed(lcs, ("time", "flux"), pinned=False)
    t_l, y_l, off = packed.drop_nan_flux(t_l, y_l, off)
    if len(off) > 1 and np.diff(off).min() < 2:
        raise ValueError("The light curve needs at least two cadences to build a periodogram.")
    nb, M = len(off) - 1, len(plan.f_day)
    dev = torch.device("cuda", device)
    with torch.cuda.device(dev):
        d_pow = torch.empty((nb, M), dtype=torch.float64, device=dev)
        if nb:
            f_day = plan.f_day
            d_t = torch.from_numpy(packed.rebase_times(t_l, off)).to(dev)
            d_y = torch.from_numpy(np.ascontiguousarray(y_l)).to(dev)
            d_scale = torch.from_numpy(packed.ls_scales(t_l, off, plan)).to(dev)
            d_max = torch.empty(nb, dtype=torch.float64, device=dev)
            d_arg = torch.empty(nb, dtype=torch.int64, device=dev)
            h = _capi.Handle.get(device)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _capi.ls_fast_peaks_batch_dev(h, nb, off, d_t.data_ptr(), d_y.data_ptr(), 0, float(f_day[0]),
                                          float(f_day[1] - f_day[0]), M, True, True, plan.norm, d_scale.data_ptr(), 5,
                                          d_pow.data_ptr(), d_max.data_ptr(), d_arg.data_ptr(), stream)
        return all_gather_rows(d_pow, bounds).cpu().numpy()


def lombscargle_peaks_batch(lcs, frequency, normalization="amplitude", freq_unit=None, oversample_factor=None, device=None,
                            gather=True):
    """(max power, argmax) of the default-method (``ls_method="fast"``) periodogram of every light curve on one shared
    regular grid -> float64[len(lcs), 2] (column 1 holds the index).  The spectra never leave the GPU
    (lk_ls_fast_peaks_lc_batch with power = NULL) and with a process group only 16 B per target cross xGMI: this is the
    collective to use when ``Periodogram.max_power`` / ``frequency_at_max_power`` is what the pipeline keeps
    (reference periodogram.py:127-140) — ``lombscargle_batch(gather=True)`` makes every rank hold all B x M powers."""
    device = _resolve_device(device)
    plan = packed.ls_grid_plan(frequency, normalization, freq_unit, oversample_factor, "fast", 1)
    if plan.ls_method != "fast":
        raise ValueError("lombscargle_peaks_batch needs a regular frequency grid (the reference switches to 'slow')")
    (time, flux), n_off, bounds = _local_packed(lcs, ("time", "flux"))
    _pw, mx, am = _ls_power_packed(time, flux, n_off, plan, device, want_power=False, want_peaks=True)
    return _finish(np.column_stack([mx, am.astype(np.float64)]), bounds, gather)


def _bls_options(period, duration, objective, oversample):
    """The grid-level checks of ``BoxLeastSquaresPeriodogram.from_lightcurve`` / astropy ``BoxLeastSquares.power``
    (reference periodogram.py:1101-1112, astropy bls/core.py:277-327, 668-700), once per batch."""
    period = np.atleast_1d(np.asarray(period, dtype=np.float64))
    if duration is None:
        duration = [0.05, 0.10, 0.15, 0.20, 0.25, 0.33]
    duration = np.atleast_1d(np.asarray(duration, dtype=np.float64))
    if not np.all(np.isfinite(duration)):
        raise ValueError("`duration` parameter contains illegal nan or inf value(s)")
    if not np.all(np.isfinite(period)):
        raise ValueError("`period` parameter contains illegal nan or inf value(s)")
    if period.ndim != 1 or period.size == 0:
        raise ValueError("period must be 1-dimensional")
    if duration.ndim != 1 or duration.size == 0:
        raise ValueError("duration must be 1-dimensional")
    if np.min(period) <= np.max(duration):
        raise ValueError("The maximum transit duration must be shorter than the minimum period")
    try:
        oversample = int(oversample)
    except TypeError:
        raise ValueError("oversample must be an int, got {0}".format(oversample))
    if oversample < 1:
        raise ValueError("oversample must be greater than or equal to 1")
    objective = objective or "likelihood"
    if objective not in ["snr", "likelihood"]:
        raise ValueError("Unrecognized method '{0}'\nallowed methods are: {1}".format(objective, ["snr", "likelihood"]))
    return period, duration, objective, oversample


def bls_batch(lcs, period, duration=None, objective="likelihood", oversample=10, device=None, gather=True):
    """BLS power (and the other six statistics) of every light curve on one shared period grid.
    Returns float64[len(lcs), 7, nP] ordered as ``_capi.BLS_FIELDS`` (transit_time absolute, like the reference).
    ``lcs``: a list of light curves or a ``LightCurveBatch``; the per-target inputs (t - min t, y - median y, ivar) are
    built by vectorised passes over the packed arrays (``packed.bls_inputs``)."""
    device = _resolve_device(device)
    period, duration, objective, oversample = _bls_options(period, duration, objective, oversample)
    (time, flux, err), n_off, bounds = _local_packed(lcs, ("time", "flux", "flux_err"), prefix="bls")
    if len(n_off) == 1:
        return _finish(np.zeros((0, 7, len(period))), bounds, gather)
    t, y, w, off2, t_ref = packed.bls_inputs(time, flux, err, n_off)
    res = _capi.bls_batch(t, y, w, off2, period, duration, oversample, objective == "likelihood", device=device)
    out = np.stack([res[k] for k in _capi.BLS_FIELDS], axis=1)
    out[:, 4, :] += t_ref[:, None]
    return _finish(out, bounds, gather)


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
    for d in dms:
        if not isinstance(d, (DesignMatrix, DesignMatrixCollection)):
            raise ValueError("design_matrix_collection must be a DesignMatrix or DesignMatrixCollection")
    # shapes and priors are checked for every target; the reference's low-rank WARNING (an SVD per matrix on the host,
    # designmatrix.py:306-337 — ~50 ms per target, i.e. minutes per 1000 targets) is left to the per-object path
    dms = [d if isinstance(d, DesignMatrixCollection) else DesignMatrixCollection([d], validate_rank=False) for d in dms]
    for d in dms:
        d.validate(rank=False)
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
