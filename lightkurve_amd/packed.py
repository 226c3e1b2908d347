"""Packed (ragged, concatenated) light-curve arrays and the per-BATCH planning of the hot path.

The reference decides everything per object (``LombScarglePeriodogram.from_lightcurve``, periodogram.py:783-967;
``BoxLeastSquaresPeriodogram.from_lightcurve`` :1093-1168; ``LightCurve.flatten``, lightcurve.py:996-1063) and its
callers loop over a ``LightCurveCollection`` (collections.py:145).  With the kernels at ~10 us per target that loop is
the wall: a shared frequency grid must be examined once per batch, not once per target, and the per-target quantities
(first / last time, cadence count, NaN removal, the psd scale) come out of a handful of vectorised passes over the
packed arrays — or, for ``t - t[0]``, out of the device (``lk_ls_fast_peaks_lc_batch``).

Layout = what every ``lk_*_batch`` entry point takes: ``time`` / ``flux`` (/ ``flux_err``) float64[sum N_b] and
``n_off`` int64[B + 1] prefix offsets.
"""
import logging
import warnings

import numpy as np

from . import _capi

log = logging.getLogger(__name__)

__all__ = ["values", "any_nan", "pack_columns", "drop_nan_flux", "rebase_times", "LsGridPlan", "ls_grid_plan", "ls_scales",
           "bls_inputs", "check_sorted"]


def values(x):
    """float64 ndarray of an ndarray / astropy Quantity / Time / masked column (no copy when it already is one)."""
    x = getattr(x, "unmasked", x)
    x = getattr(x, "value", x)
    return np.asarray(x, dtype=np.float64)


_EXEC = None


def _threads():
    """Host threads for the memory-bound packing passes (numpy releases the GIL inside concatenate / dot):
    LK_HOST_THREADS, default min(8, cores)."""
    import os
    try:
        n = int(os.environ.get("LK_HOST_THREADS", "0"))
    except ValueError:
        n = 0
    return n if n > 0 else max(1, min(8, os.cpu_count() or 1))


def _pmap(fn, jobs):
    """[fn(*job) for job in jobs] over the packing thread pool (serial for one job / one thread)."""
    global _EXEC
    jobs = list(jobs)
    if len(jobs) <= 1 or _threads() == 1:
        return [fn(*j) for j in jobs]
    if _EXEC is None:
        from concurrent.futures import ThreadPoolExecutor
        _EXEC = ThreadPoolExecutor(max_workers=_threads())
    return list(_EXEC.map(lambda j: fn(*j), jobs))


def _chunks(n_items, weights_cum):
    """Cut 0..n_items into <= threads contiguous runs of near-equal weight (weights_cum: prefix sums, length n_items+1)."""
    k = min(_threads(), max(1, n_items))
    total = weights_cum[-1]
    cuts = np.searchsorted(weights_cum, total * np.arange(1, k) / k, side="left")
    b = np.unique(np.concatenate([[0], cuts, [n_items]]))
    return list(zip(b[:-1].tolist(), b[1:].tolist()))


def any_nan(x):
    """True if the float64 array holds a NaN: the minimum of a chunk is NaN iff the chunk holds one (numpy's min propagates
    NaN; a vectorised pass without a temporary and without BLAS threads of its own), chunks over the thread pool."""
    n = x.size
    if n == 0:
        return False
    step = max(1 << 20, -(-n // _threads()))
    parts = _pmap(lambda a, b: float(np.min(x[a:b])), [(a, min(n, a + step)) for a in range(0, n, step)])
    return bool(np.isnan(parts).any())


def _dest(total, pinned, key):
    """float64[total] to pack into: a slice of the pinned pool (``pinned`` True / "auto" with a GPU), a page-locked array
    of the caller's own ("own"), or a fresh pageable one."""
    if pinned == "own":
        return _capi.pinned_empty(total)
    if pinned:
        try:
            return _capi.pinned_pool(key, total)
        except (OSError, RuntimeError, MemoryError):
            if pinned != "auto":
                raise
    return np.empty(total, dtype=np.float64)


def pack_columns(lcs, columns=("time", "flux"), pinned=False, pool_prefix="pack"):
    """Concatenate the named columns of every light curve -> ([float64[sum N] per column], n_off).

    One ``np.concatenate(..., out=)`` per column (a single C loop over the B source arrays).  ``pinned``: write into the
    page-locked staging pool of ``_capi`` (the arrays are then valid until the next packing call with the same
    ``pool_prefix`` — the batch entry points consume them before they return); "auto" falls back to pageable memory when
    no GPU runtime is there; "own": page-locked arrays that belong to the caller (``LightCurveBatch``)."""
    lcs = list(lcs)
    cols = []
    for c in columns:
        arrs = []
        for lc in lcs:
            v = getattr(lc, c, None)
            arrs.append(None if v is None else values(v))
        cols.append(arrs)
    lens = np.fromiter((a.shape[0] for a in cols[0]), dtype=np.int64, count=len(lcs))
    n_off = np.zeros(len(lcs) + 1, dtype=np.int64)
    np.cumsum(lens, out=n_off[1:])
    total = int(n_off[-1])
    out = []
    for c, arrs in zip(columns, cols):
        for i, a in enumerate(arrs):
            if a is None:     # e.g. a light curve without flux_err: NaN like the reference's default column
                arrs[i] = np.full(int(lens[i]), np.nan)
            elif a.ndim == 0:
                arrs[i] = np.full(int(lens[i]), float(a))
            elif a.shape[0] != lens[i]:
                raise ValueError("light curve %d: column %r has %d entries, time has %d" % (i, c, a.shape[0], lens[i]))
        dest = _dest(total, pinned, "%s:%s" % (pool_prefix, c))
        if arrs:
            _pmap(lambda i0, i1: np.concatenate(arrs[i0:i1], out=dest[n_off[i0]:n_off[i1]]), _chunks(len(arrs), n_off))
        out.append(dest)
    return out, n_off


def drop_nan_flux(time, flux, n_off, *others):
    """``lc.remove_nans()`` (reference lightcurve.py:1300-1327: cadences with NaN FLUX go) for the whole batch.
    Returns (time, flux, n_off, *others) — the inputs themselves when there is no NaN (one pass over flux)."""
    if not any_nan(flux):
        return (time, flux, n_off) + tuple(others)
    keep = ~np.isnan(flux)
    cs = np.concatenate([[0], np.cumsum(keep, dtype=np.int64)])
    return (time[keep], flux[keep], cs[n_off]) + tuple(None if o is None else o[keep] for o in others)


def rebase_times(time, n_off):
    """t - t[first cadence of its light curve] for every cadence (astropy lombscargle/core.py:119-126 per object)."""
    counts = np.diff(n_off)
    return time - np.repeat(time[n_off[:-1][counts > 0]], counts[counts > 0])


def check_sorted(time, n_off):
    """True if every light curve's times are non-decreasing (one vectorised pass; boundaries between light curves excluded)."""
    n = time.size
    if n < 2:
        return True
    inner = n_off[1:-1]
    inner = inner[(inner > 0) & (inner < n)]            # first cadence of a later light curve: its step is a boundary

    def part(a, b):                                       # steps a .. b - 1 (step i: time[i] -> time[i + 1])
        d = time[a + 1:b + 1] < time[a:b]
        if not d.any():
            return True
        lo, hi = np.searchsorted(inner, [a + 1, b + 1])   # boundaries inner[lo:hi] sit at steps inner - 1 in [a, b)
        d[inner[lo:hi] - 1 - a] = False
        return not d.any()

    step = max(1 << 20, -(-(n - 1) // _threads()))
    return all(_pmap(part, [(a, min(n - 1, a + step)) for a in range(0, n - 1, step)]))


class LsGridPlan(object):
    """What ``LombScarglePeriodogram.from_lightcurve`` decides from its options and the frequency grid alone
    (reference periodogram.py:783-800, 917-958) — decided ONCE for a batch that shares the grid."""
    __slots__ = ("normalization", "freq_unit", "unit", "oversample_factor", "frequency", "f_day", "ls_method", "nterms",
                 "norm", "power_unit", "regular", "exact")


def ls_grid_plan(frequency, normalization="amplitude", freq_unit=None, oversample_factor=None, ls_method="fast", nterms=1):
    from .periodogram import (LightkurveWarning, _LS_METHODS, _freq_unit_factor, exact_grid, is_regular, validate_method)
    p = LsGridPlan()
    p.normalization = validate_method(normalization, ["psd", "amplitude"])
    p.freq_unit = freq_unit if freq_unit is not None else ("1/d" if p.normalization == "amplitude" else "uHz")
    p.unit = _freq_unit_factor(p.freq_unit)
    p.oversample_factor = oversample_factor if oversample_factor is not None else (5.0 if p.normalization == "amplitude" else 1.0)
    p.frequency = np.asarray(frequency, dtype=np.float64)
    if p.frequency.ndim != 1 or p.frequency.size < 2:
        raise ValueError("frequency and power must have a length greater than 1.")
    ls_method = validate_method(ls_method, list(_LS_METHODS))
    if ls_method[:9] == "fastnifty":
        old = ls_method
        ls_method = {"fastnifty": "fast", "fastnifty_chi2": "fastchi2"}[ls_method]
        log.warning("nifty_ls is not available.\nMethod has been changed from '{}' to '{}'.".format(old, ls_method))
    p.regular = is_regular(p.frequency)
    if not p.regular and ls_method in ["fastchi2", "fast"]:
        old = ls_method
        ls_method = {"fastchi2": "chi2", "fast": "slow"}[ls_method]
        log.warning("The requested periodogram is not evenly sampled in frequency.\n"
                    "Method has been changed from '{}' to '{}' to allow for this.".format(old, ls_method))
    if nterms > 1 and ls_method not in ["fastchi2", "chi2"]:
        warnings.warn(
            "Building a Lomb Scargle Periodogram using the `slow` method. "
            "`nterms` has been set to >1, however this is not supported under the `{}` method. "
            "To run with higher nterms, set `ls_method` to either 'fastchi2', 'chi2', or 'fastnifty_chi2. "
            "Please refer to the `astropy.timeseries.periodogram.LombScargle` documentation.".format(ls_method),
            LightkurveWarning)
        nterms = 1
    if ls_method == "auto":
        ls_method = "fast" if (len(p.frequency) > 200 and p.regular) else "cython"
    p.ls_method, p.nterms = ls_method, int(nterms)
    p.f_day = p.frequency / p.unit
    p.exact = exact_grid(p.f_day)
    if p.normalization == "psd":
        p.norm, p.power_unit = "lk_psd", "flux^2/" + p.freq_unit
    else:
        p.norm, p.power_unit = "lk_amplitude", "flux"
    return p


def ls_scales(time, n_off, plan):
    """Per-target factor of lightkurve's normalisation (periodogram.py:865-868, 969-975): psd -> 2 / (N os fs) with
    fs = 1 / (t[-1] - t[0]) / os in the frequency unit; amplitude -> 1.  Same fp64 operations, in the same order, as
    ``_ls_plan`` performs per object."""
    counts = np.diff(n_off)
    if plan.normalization != "psd":
        return np.ones(len(counts))
    first, last = time[n_off[:-1]], time[n_off[1:] - 1]
    fs = (1.0 / (last - first)) / plan.oversample_factor * plan.unit
    return 2.0 / (counts * plan.oversample_factor * fs)


def _segment_medians(y, n_off):
    counts = np.diff(n_off)
    if len(counts) and counts.min() == counts.max() and counts[0] > 0:
        return np.median(y.reshape(len(counts), int(counts[0])), axis=1)     # one partition per row, no Python loop
    return np.array([np.median(y[a:b]) for a, b in zip(n_off[:-1], n_off[1:])])


def bls_inputs(time, flux, flux_err, n_off):
    """What ``BoxLeastSquaresPeriodogram.from_lightcurve`` + astropy hand to ``bls_fast`` (reference periodogram.py:
    1093-1100, astropy bls/core.py:277-327), for the whole batch: NaN-flux cadences dropped, t - min(t), y - median(y),
    ivar = 1 / err^2 (ones for a light curve whose errors are not all finite).  Returns (t, y, ivar, n_off, t_ref[B])."""
    time, flux, n_off, flux_err = drop_nan_flux(time, flux, n_off, flux_err)
    counts = np.diff(n_off)
    if len(counts) and counts.min() < 1:
        raise ValueError("a light curve of the batch has no finite flux")
    rep = lambda v: np.repeat(v, counts)
    t0 = time[n_off[:-1]]
    trel = time - rep(t0)
    tmin = np.minimum.reduceat(trel, n_off[:-1]) if len(counts) else np.zeros(0)
    t = trel - rep(tmin)
    y = flux - rep(_segment_medians(flux, n_off))
    if flux_err is None:
        ivar = np.ones_like(flux)
    else:
        fin = np.isfinite(flux_err)
        all_fin = np.logical_and.reduceat(fin, n_off[:-1]) if len(counts) else np.zeros(0, bool)
        with np.errstate(divide="ignore", invalid="ignore"):
            ivar = np.where(rep(all_fin), 1.0 / flux_err ** 2, 1.0)
    return t, y, ivar, n_off, tmin + t0
