"""Batched front end of ``LightCurve.flatten`` (reference: src/lightkurve/lightcurve.py:943-1078)."""
import logging

import numpy as np

from . import _capi

log = logging.getLogger(__name__)


def flatten_trend_batch(lcs, window_length=101, polyorder=2, break_tolerance=5, niters=3, sigma=3, masks=None,
                        device=0):
    """Trend of every light curve in ``lcs`` (list of objects with .time/.flux) in one GPU call.
    ``masks``: optional list of boolean arrays, True = cadence excluded from the fit (the reference's ``mask=``).
    Returns a list of float64 arrays."""
    if polyorder >= window_length:
        polyorder = window_length - 1
        log.warning("polyorder must be smaller than window_length, using polyorder={}.".format(polyorder))
    if window_length % 2 != 1:
        raise ValueError("window_length must be odd (scipy.signal.savgol_filter with mode='interp')")
    from . import packed
    lcs = list(lcs)
    (t, f), off = packed.pack_columns(lcs, ("time", "flux"), pinned="auto", pool_prefix="flat")
    if not packed.check_sorted(t, off):
        raise ValueError("flatten needs the light curve sorted by time")
    m = None
    if masks is not None:
        if len(masks) != len(lcs):
            raise ValueError("masks must hold one entry (array or None) per light curve")
        if any(mk is not None for mk in masks):
            m = np.zeros(t.size, dtype=bool)
            for i, mk in enumerate(masks):
                if mk is None:
                    continue
                n = int(off[i + 1] - off[i])
                if np.shape(mk) != (n,):
                    raise ValueError("mask %d has shape %s, its light curve has %d cadences" % (i, np.shape(mk), n))
                m[off[i]:off[i + 1]] = mk
    trend = _capi.savgol_trend_batch(t, f, off, mask=m, window_length=window_length, polyorder=polyorder,
                                     break_tolerance=break_tolerance, niters=niters, sigma=sigma, device=device)
    return [trend[off[i]:off[i + 1]] for i in range(len(lcs))]
