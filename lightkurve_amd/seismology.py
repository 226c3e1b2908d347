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

__all__ = ["autocorrelate", "estimate_numax_acf2d", "estimate_numax_acf2d_batch"]


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
