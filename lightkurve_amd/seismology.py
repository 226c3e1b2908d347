"""Seismology 2-D autocorrelation on the GPU (second half).

Reference: ``estimate_numax_acf2d`` (src/lightkurve/seismology/numax_estimators.py:15-205) and ``autocorrelate``
(src/lightkurve/seismology/utils.py:106-158).  The reference loops over the central frequencies ``numaxs`` and calls
``np.correlate`` on a window of the spectrum each time (W^2 / 2 multiply-adds per window, "This function is slow", :103);
here all windows of all periodograms are ONE kernel launch (``lk_pg_acf2d_batch``: one workgroup per window, the window in
LDS).  What is left on the host is what the reference does once per periodogram on len(numaxs) numbers: the Gaussian
smoothing of the metric and the argmax.  Units are plain floats in the periodogram's own frequency unit.
"""
import numpy as np

from . import _capi
from .periodogram import UHZ_PER_CPD, _freq_unit_factor

__all__ = ["autocorrelate", "estimate_numax_acf2d", "estimate_numax_acf2d_batch", "estimate_deltanu_acf2d", "get_fwhm"]


def _to_uhz(value, unit):
    return value * (UHZ_PER_CPD / _freq_unit_factor(unit))     # the ratio first: exactly 1.0 for microhertz


def _from_uhz(value, unit):
    return value * (_freq_unit_factor(unit) / UHZ_PER_CPD)


def _window_start(frequency, numax, window_width, fs):
    """Index arithmetic of seismology/utils.py:141-151 (python ``int()`` truncation included)."""
    spread = int(window_width / 2 / fs)
    x = int(numax / fs)
    x0 = int(frequency[0] / fs)
    return (x - x0) - spread, 2 * spread


def autocorrelate(periodogram, numax, window_width=25.0, frequency_spacing=None, device=0):
    """ACF of the ``window_width`` wide region centred on ``numax`` (both in the periodogram's frequency unit; the
    reference's docstring says microhertz but its arithmetic is unit-blind, seismology/utils.py:141-158)."""
    freq = np.asarray(periodogram.frequency, dtype=np.float64)
    fs = np.median(np.diff(freq)) if frequency_spacing is None else frequency_spacing
    start, W = _window_start(freq, numax, window_width, fs)
    acf, _ = _capi.pg_acf2d_batch(np.asarray(periodogram.power, dtype=np.float64)[None, :], [start], W, device=device)
    return acf[0, 0]


def _gaussian_smooth_extend(metric, stddev):
    """astropy.convolution.convolve(metric, Gaussian1DKernel(stddev), boundary='extend') (numax_estimators.py:181-183):
    kernel of _round_up_to_odd_integer(8 stddev) taps sampled at the integers, normalised by its sum, edges replicated."""
    size = int(np.ceil(8 * stddev))
    if size % 2 == 0:
        size += 1
    half = size // 2
    xk = np.arange(-half, half + 1, dtype=np.float64)
    g = np.exp(-0.5 * (xk / stddev) ** 2) / (np.sqrt(2 * np.pi) * stddev)
    g = g / g.sum()
    padded = np.concatenate([np.full(half, metric[0]), metric, np.full(half, metric[-1])])
    return np.correlate(padded, g, mode="valid")


def _plan(periodogram, numaxs, window_width, spacing):
    unit = periodogram.frequency_unit
    freq = np.asarray(periodogram.frequency, dtype=np.float64)
    if not periodogram._is_evenly_spaced():
        raise ValueError("the ACF 2D method requires that the periodogram has a grid of uniformly spaced frequencies.")
    high = _to_uhz(freq[-1], unit) > 500.0
    if window_width is None:
        window_width = _from_uhz(250.0 if high else 25.0, unit)
    if spacing is None:
        spacing = _from_uhz(10.0 if high else 1.0, unit)
    if numaxs is None:
        numaxs = np.arange(np.ceil(np.nanmin(freq)) + window_width / 2, np.floor(np.nanmax(freq)) - window_width / 2,
                           spacing)
    numaxs = np.atleast_1d(np.asarray(numaxs, dtype=np.float64))
    fs = np.median(np.diff(freq))
    for var, label in zip([np.asarray(window_width), np.asarray(spacing)], ["window_width", "spacing"]):
        if (var < fs).any():
            raise ValueError("You can't have {} smaller than the frequency separation!".format(label))
        if (var > (freq[-1] - freq[0])).any():
            raise ValueError("You can't have {} wider than the entire power spectrum!".format(label))
        if (var < 0).any():
            raise ValueError("Please pass an entirely positive {}.".format(label))
    if any(numaxs < fs):
        raise ValueError("A custom range of numaxs can not extend below a single frequency bin.")
    if any(numaxs > np.nanmax(freq)):
        raise ValueError("A custom range of numaxs can not extend above the highest frequency value in the periodogram.")
    starts = np.array([_window_start(freq, nm, window_width, fs)[0] for nm in numaxs], dtype=np.int64)
    W = int(window_width / 2 / fs) * 2
    return numaxs, window_width, starts, W


def _finish(numaxs, window_width, acf2d, metric):
    if len(numaxs) > 10:
        metric_smooth = _gaussian_smooth_extend(metric, np.sqrt(len(numaxs)))
    else:
        metric_smooth = metric
    best = float(numaxs[np.argmax(metric_smooth)])
    return dict(numax=best, numaxs=numaxs, acf2d=acf2d.T, window_width=window_width, metric=metric,
                metric_smooth=metric_smooth)


def estimate_numax_acf2d(periodogram, numaxs=None, window_width=None, spacing=None, device=0):
    """numax by the 2-D autocorrelation method.  Returns a dict with ``numax`` and the reference's diagnostics
    (``numaxs``, ``acf2d`` [lags x numaxs], ``window_width``, ``metric``, ``metric_smooth``)."""
    numaxs, window_width, starts, W = _plan(periodogram, numaxs, window_width, spacing)
    acf, met = _capi.pg_acf2d_batch(np.asarray(periodogram.power, dtype=np.float64)[None, :], starts, W, device=device)
    return _finish(numaxs, window_width, acf[0], met[0])


def estimate_numax_acf2d_batch(periodograms, numaxs=None, window_width=None, spacing=None, device=0):
    """The same for a list of periodograms that share one frequency grid: every window of every spectrum in ONE launch."""
    if not periodograms:
        return []
    f0 = np.asarray(periodograms[0].frequency)
    for pg in periodograms[1:]:
        if not np.array_equal(np.asarray(pg.frequency), f0) or pg.frequency_unit != periodograms[0].frequency_unit:
            raise ValueError("estimate_numax_acf2d_batch needs periodograms on one shared frequency grid")
    numaxs, window_width, starts, W = _plan(periodograms[0], numaxs, window_width, spacing)
    power = np.stack([np.asarray(pg.power, dtype=np.float64) for pg in periodograms])
    acf, met = _capi.pg_acf2d_batch(power, starts, W, device=device)
    return [_finish(numaxs, window_width, acf[b], met[b]) for b in range(len(periodograms))]


# ------------------------------------------------------------------------------------------------ deltanu
def get_fwhm(periodogram, numax):
    """Width of the mode envelope (seismology/utils.py:62-104): 0.25 numax above 500 microhertz of bandwidth (main
    sequence), 0.66 numax^0.88 below (red giants); ``numax`` in the periodogram's frequency unit, used as a bare number
    exactly as the reference does."""
    freq = np.asarray(periodogram.frequency, dtype=np.float64)
    if _to_uhz(freq[-1], periodogram.frequency_unit) > 500.0:
        return 0.25 * numax
    return 0.66 * numax ** 0.88


def _local_maxima_1d(x):
    """scipy.signal._peak_finding_utils._local_maxima_1d: midpoints of the (plateau) local maxima of x."""
    mid = []
    i, i_max = 1, len(x) - 1
    while i < i_max:
        if x[i - 1] < x[i]:
            ahead = i + 1
            while ahead < i_max and x[ahead] == x[i]:
                ahead += 1
            if x[ahead] < x[i]:
                mid.append((i + ahead - 1) // 2)
                i = ahead
        i += 1
    return np.asarray(mid, dtype=np.intp)


def _select_by_peak_distance(peaks, priority, distance):
    """scipy.signal._peak_finding_utils._select_by_peak_distance: keep the highest-priority peaks, drop neighbours closer
    than ``distance`` samples."""
    keep = np.ones(len(peaks), dtype=bool)
    dist = int(np.ceil(distance))
    order = np.argsort(priority)
    for i in range(len(peaks) - 1, -1, -1):
        j = order[i]
        if not keep[j]:
            continue
        k = j - 1
        while k >= 0 and peaks[j] - peaks[k] < dist:
            keep[k] = False
            k -= 1
        k = j + 1
        while k < len(peaks) and peaks[k] - peaks[j] < dist:
            keep[k] = False
            k += 1
    return keep


def _find_peaks(x, distance):
    """scipy.signal.find_peaks(x, distance=distance)[0] (the only form deltanu_estimators.py:127 uses)."""
    if distance is not None and distance < 1:
        raise ValueError("`distance` must be greater or equal to 1")
    peaks = _local_maxima_1d(np.asarray(x, dtype=np.float64))
    if distance is not None and len(peaks):
        peaks = peaks[_select_by_peak_distance(peaks, np.asarray(x)[peaks], distance)]
    return peaks


def estimate_deltanu_acf2d(periodogram, numax, device=0):
    """Large frequency separation by the autocorrelation method (reference seismology/deltanu_estimators.py:18-153): the
    ACF of the window of one envelope FWHM either side of ``numax`` — the same GPU kernel as the numax estimator
    (lk_pg_acf2d_batch) — rescaled to the Mosser & Appourchaux noise level, and the ACF peak closest to the empirical
    0.294 numax^0.772 (in microhertz).  Returns a dict with ``deltanu`` and the reference's diagnostics (``lags``,
    ``acf``, ``peaks``, ``sel``, ``numax``, ``deltanu_emp``), everything in the periodogram's frequency unit."""
    if not periodogram._is_evenly_spaced():
        raise ValueError("the ACF 2D method requires that the periodogram has a grid of uniformly spaced frequencies.")
    unit = periodogram.frequency_unit
    freq = np.asarray(periodogram.frequency, dtype=np.float64)
    numax = float(numax)
    fs = np.median(np.diff(freq))
    if numax < fs:
        raise ValueError("The input numax can not be lower than a single frequency bin.")
    if numax > np.nanmax(freq):
        raise ValueError("The input numax can not be higher thanthe highest frequency value in the periodogram.")
    deltanu_emp = _from_uhz(0.294 * _to_uhz(numax, unit) ** 0.772, unit)
    window_width = 2 * int(np.floor(get_fwhm(periodogram, numax)))
    aacf = autocorrelate(periodogram, numax=numax, window_width=window_width, device=device)
    acf = (np.abs(aacf ** 2) / np.abs(aacf[0] ** 2)) / (3 / (2 * len(aacf)))
    lags = np.linspace(0.0, len(acf) * fs, len(acf))
    sel = (lags > deltanu_emp - 0.25 * deltanu_emp) & (lags < deltanu_emp + 0.25 * deltanu_emp)
    peaks = _find_peaks(acf[sel], distance=np.floor(deltanu_emp / 2.0 / fs))
    best = lags[sel][peaks][np.argmin(np.abs(lags[sel][peaks] - deltanu_emp))]
    return dict(deltanu=float(best), lags=lags, acf=acf, peaks=peaks, sel=sel, numax=numax, deltanu_emp=deltanu_emp)
