"""Batched (many targets at once) entry points of the hot path, sharded over ranks when a process group exists.

These are what a pipeline over a ``LightCurveCollection`` (reference: src/lightkurve/collections.py:145) calls
instead of looping ``lc.to_periodogram()`` / ``lc.flatten()`` per target."""
import numpy as np

from . import _capi
from .distributed import sharded_map
from .periodogram import _bls_plan, _ls_plan, exact_grid

__all__ = ["lombscargle_batch", "lombscargle_peaks_batch", "bls_batch", "periodogram_peaks"]


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

    return sharded_map(list(lcs), compute, costs=[len(lc) for lc in lcs], gather=gather)


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
