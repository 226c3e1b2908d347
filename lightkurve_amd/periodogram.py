"""Host-side mirror of lightkurve's periodogram constructors for the HIP hot path.

Mirrors (same names, argument meaning and error behaviour) of
``lightkurve.periodogram.{Periodogram, LombScarglePeriodogram, BoxLeastSquaresPeriodogram}``
(reference: src/lightkurve/periodogram.py:33-140, 636-989, 1043-1192).  astropy is not importable in
the product interpreter, so quantities are plain float64 ndarrays with the unit kept as a string
attribute (``frequency_unit``: "1/d" or "uHz"; ``power_unit``).  All arithmetic that the reference
delegates to astropy (LombScargle.power / BoxLeastSquares.power) runs in liblkhip.so on the GPU.
"""
import logging
import warnings

import math

import numpy as np

from . import _capi

log = logging.getLogger(__name__)

__all__ = ["Periodogram", "LombScarglePeriodogram", "BoxLeastSquaresPeriodogram", "LightkurveWarning"]

UHZ_PER_CPD = 1e6 / 86400.0  # microhertz per (1/day)
_FREQ_UNITS = {"1/d": 1.0, "1/day": 1.0, "d-1": 1.0, "uHz": UHZ_PER_CPD, "microhertz": UHZ_PER_CPD,
               "Hz": 1.0 / 86400.0, "hertz": 1.0 / 86400.0}
# every LS method name the reference accepts is served by the same exact HIP kernels
_LS_METHODS = ("hip", "auto", "fast", "slow", "cython", "chi2", "fastchi2", "scipy", "fastnifty", "fastnifty_chi2")


class LightkurveWarning(Warning):
    """Mirror of lightkurve.utils.LightkurveWarning (utils.py:547)."""


def validate_method(method, supported_methods):
    """lightkurve.utils.validate_method (utils.py:577-600)."""
    method = method.lower()
    if method in supported_methods:
        return method
    raise ValueError("method '{}' is not supported; must be one of {}".format(method, supported_methods))


def _freq_unit_factor(unit):
    """units-per-(1/day) of a frequency unit string."""
    try:
        return _FREQ_UNITS[unit]
    except KeyError:
        raise ValueError("Frequency must be in units of 1/time (got %r; use one of %s)." % (unit, sorted(_FREQ_UNITS)))


def is_regular(frequency):
    """astropy lombscargle/implementations/main.py:40-50 ``_is_regular``."""
    frequency = np.asarray(frequency)
    if frequency.ndim != 1:
        return False
    if len(frequency) == 1:
        return True
    diff = np.diff(frequency)
    return bool(np.allclose(diff[0], diff))


def exact_grid(frequency):
    """(f0, df) if ``frequency`` equals f0 + df*arange(M) to within 4 ulp of its largest value, else None.
    astropy's ``_is_regular`` uses rtol=1e-5 on the steps; the exact kernels must not move a frequency by
    that much, so the fast regular-grid kernel is only chosen when the grid is regular to rounding."""
    f = np.asarray(frequency, dtype=np.float64)
    if f.ndim != 1 or len(f) < 2:
        return None
    df = (f[-1] - f[0]) / (len(f) - 1)
    if not df > 0 or f[0] < 0:
        return None
    resid = np.max(np.abs(f - (f[0] + df * np.arange(len(f)))))
    return (float(f[0]), float(df)) if resid <= 4 * np.finfo(float).eps * np.max(np.abs(f)) else None


class Periodogram(object):
    """Generic power spectrum container (reference periodogram.py:33-140)."""

    def __init__(self, frequency, power, nyquist=None, label=None, targetid=None, default_view="frequency",
                 meta=None, frequency_unit="1/d", power_unit=""):
        frequency = np.asarray(frequency, dtype=np.float64)
        power = np.asarray(power, dtype=np.float64)
        _freq_unit_factor(frequency_unit)
        if frequency.ndim != 1 or frequency.shape[0] <= 1:
            raise ValueError("frequency and power must have a length greater than 1.")
        if frequency.shape != power.shape:
            raise ValueError("frequency and power must have the same length.")
        self.frequency = frequency
        self.power = power
        self.nyquist = nyquist
        self.label = label
        self.targetid = targetid
        self.default_view = self._validate_view(default_view)
        self.meta = {} if meta is None else meta
        self.frequency_unit = frequency_unit
        self.power_unit = power_unit

    def _validate_view(self, view):
        if view is None and hasattr(self, "default_view"):
            view = self.default_view
        return validate_method(view, ["frequency", "period"])

    def _is_evenly_spaced(self):
        freqdiff = np.diff(self.frequency)
        return bool(np.allclose(freqdiff[0], freqdiff))

    @property
    def period(self):
        """1/frequency (in 1/frequency_unit)."""
        return 1.0 / self.frequency

    @property
    def max_power(self):
        return np.nanmax(self.power)

    @property
    def frequency_at_max_power(self):
        return self.frequency[np.nanargmax(self.power)]

    @property
    def period_at_max_power(self):
        return 1.0 / self.frequency_at_max_power

    def __repr__(self):
        return "%s(ID: %s)" % (type(self).__name__, self.targetid)

    def copy(self):
        import copy
        return copy.deepcopy(self)

    def bin(self, binsize=10, method="mean"):
        """Bin the spectrum by an integer factor (reference periodogram.py:140-181): mean or nanmedian of consecutive
        groups of ``binsize`` points; the trailing remainder is dropped.  Pure reshaping — no kernel involved."""
        if binsize < 1:
            raise ValueError("binsize must be larger than or equal to 1")
        method = validate_method(method, ["mean", "median"])
        m = int(len(self.power) / binsize)
        f = self.frequency[: m * binsize].reshape((m, binsize))
        pw = self.power[: m * binsize].reshape((m, binsize))
        out = self.copy()
        if method == "mean":
            out.frequency, out.power = f.mean(1), pw.mean(1)
        else:
            out.frequency, out.power = np.nanmedian(f, axis=1), np.nanmedian(pw, axis=1)
        return out

    def smooth(self, method="boxkernel", filter_width=0.1, device=0):
        """Smoothed copy of the power spectrum (reference periodogram.py:182-284): ``'boxkernel'`` convolves with a
        box of ``filter_width`` (frequency units; needs an evenly spaced grid), ``'logmedian'`` is a moving median
        in log10(frequency) windows of half-width ``filter_width``, divided by (8/9)^3."""
        method = validate_method(method, ["boxkernel", "logmedian"])
        out = self.copy()
        if method == "boxkernel":
            if filter_width <= 0.0:
                raise ValueError("the `filter_width` parameter must be larger than 0 for the 'boxkernel' method.")
            if not self._is_evenly_spaced():
                raise ValueError("the 'boxkernel' method requires the periodogram to have a grid of evenly spaced "
                                 "frequencies.")
            fs = np.mean(np.diff(self.frequency))
            kernel = _box1d_kernel(math.ceil(filter_width / fs))
            out.power = _capi.pg_boxsmooth_batch(self.power, kernel, device=device)[0]
            return out
        out.power = _capi.pg_logmedian_batch(self.power, *_logmedian_windows(self.frequency, filter_width),
                                             device=device)[0]
        return out

    def flatten(self, method="logmedian", filter_width=0.01, return_trend=False, device=0):
        """Signal-to-noise spectrum: power divided by its smoothed background (reference periodogram.py:381-429)."""
        bkg = self.smooth(method=method, filter_width=filter_width, device=device)
        snr = SNRPeriodogram(self.frequency, self.power / bkg.power, nyquist=self.nyquist, targetid=self.targetid,
                             label=self.label, meta=self.meta, frequency_unit=self.frequency_unit, power_unit="")
        return (snr, bkg) if return_trend else snr


class SNRPeriodogram(Periodogram):
    """Power divided by a background estimate (reference periodogram.py:528-586); unitless power."""


def _box1d_kernel(width):
    """astropy ``Box1DKernel(width).array`` (astropy@4.3.1 convolution/kernels.py + utils.py:216-223): the box
    ``1/width`` on ``|x| <= width/2`` sampled on the half-pixel grid, neighbours averaged (so an even width gets
    half-weight end taps), normalised by its sum."""
    size = int(math.ceil(width))
    size += 1 - size % 2
    x = np.arange(-(size // 2) - 0.5, size // 2 + 1.0)
    v = np.where(np.logical_and(x >= -width / 2.0, x <= width / 2.0), 1.0 / width, 0.0)
    k = 0.5 * (v[1:] + v[:-1])
    return k / k.sum()


def _logmedian_windows(frequency, filter_width):
    """The bookkeeping of the reference's logmedian loop (periodogram.py:270-281), which depends on the frequency
    grid only: window k is centred on x0_k (the same running sum ``x0 += 0.5 * filter_width``) and holds the
    frequencies with ``|log10 f - x0_k| < filter_width``.  Returns (win_lo, win_hi, klo, khi): window k covers the
    index range [win_lo[k], win_hi[k]); frequency j belongs to windows klo[j] .. khi[j] inclusive."""
    f = np.asarray(frequency, dtype=np.float64)
    if np.any(np.diff(f) < 0):
        raise NotImplementedError("the HIP logmedian filter needs a frequency grid in ascending order")
    lf = np.log10(f)
    centres = []
    x0 = np.log10(f[0])
    top = np.log10(f[-1])
    while x0 < top:
        centres.append(x0)
        x0 += 0.5 * filter_width
    lo, hi = [], []
    n = len(f)
    # the mask |lf - c| < filter_width is a contiguous run of the sorted grid (fl(lf - c) is monotone in lf): its ends come
    # from two binary searches, then the reference's own predicate decides the few entries next to them (the searches
    # compare lf with fl(c -/+ filter_width), which may round across a boundary the predicate does not)
    a0 = np.searchsorted(lf, np.asarray(centres) - filter_width, side="left")
    b0 = np.searchsorted(lf, np.asarray(centres) + filter_width, side="right")
    for c, a, b in zip(centres, a0.tolist(), b0.tolist()):
        l0, h0 = max(a - 2, 0), min(b + 2, n)
        if h0 - l0 <= 12:                                     # short (or empty) run: the predicate on all of it
            ins = np.flatnonzero(np.abs(lf[l0:h0] - c) < filter_width)
            if ins.size:
                lo.append(l0 + int(ins[0]))
                hi.append(l0 + int(ins[-1]) + 1)
            continue
        left = np.flatnonzero(np.abs(lf[l0:a + 3] - c) < filter_width)      # lf[a + 2] is well inside
        right = np.flatnonzero(np.abs(lf[b - 3:h0] - c) < filter_width)     # so is lf[b - 3]
        lo.append(l0 + int(left[0]))
        hi.append(b - 3 + int(right[-1]) + 1)
    lo, hi = np.asarray(lo, dtype=np.int32), np.asarray(hi, dtype=np.int32)
    j = np.arange(len(f))
    klo = np.searchsorted(hi, j, side="right").astype(np.int32)      # first window with hi > j
    khi = (np.searchsorted(lo, j, side="right") - 1).astype(np.int32)  # last window with lo <= j
    return lo, hi, klo, khi


def _ls_plan(lc, minimum_frequency=None, maximum_frequency=None, minimum_period=None, maximum_period=None,
             frequency=None, period=None, nterms=1, nyquist_factor=1, oversample_factor=None, freq_unit=None,
             normalization="amplitude", ls_method="fast", **kwargs):
    """Everything LombScarglePeriodogram.from_lightcurve decides BEFORE it calls astropy
    (reference periodogram.py:783-958): validated options, cleaned arrays, frequency grid, method name.
    Returns a dict; shared by the single-curve constructor and the batched entry point."""
    normalization = validate_method(normalization, ["psd", "amplitude"])
    if np.isnan(lc.flux).any():
        lc = lc.remove_nans()
        log.debug("Lightcurve contains NaN values.These are removed before creating the periodogram.")
    if freq_unit is None:
        freq_unit = "1/d" if normalization == "amplitude" else "uHz"
    unit = _freq_unit_factor(freq_unit)
    if oversample_factor is None:
        oversample_factor = 5.0 if normalization == "amplitude" else 1.0
    for old, new in (("min_period", "minimum_period"), ("max_period", "maximum_period"),
                     ("min_frequency", "minimum_frequency"), ("max_frequency", "maximum_frequency")):
        if old in kwargs:
            warnings.warn("`{}` keyword is deprecated, please use `{}` instead.".format(old, new), LightkurveWarning)
            val = kwargs.pop(old, None)
            if new == "minimum_period":
                minimum_period = val
            elif new == "maximum_period":
                maximum_period = val
            elif new == "minimum_frequency":
                minimum_frequency = val
            else:
                maximum_frequency = val
    period_args = not all(b is None for b in [period, minimum_period, maximum_period])
    freq_args = not all(b is None for b in [frequency, minimum_frequency, maximum_frequency])
    default_view = "period" if period_args else "frequency"
    if period_args and freq_args:
        raise ValueError("You have input keyword arguments for both frequency and period. Please only use one.")

    time = np.asarray(lc.time, dtype=np.float64).copy()
    flux = np.asarray(lc.flux, dtype=np.float64).copy()
    if len(time) < 2:
        raise ValueError("The light curve needs at least two cadences to build a periodogram.")
    nyquist = 0.5 * (1.0 / np.median(np.diff(time))) * unit
    fs = (1.0 / (time[-1] - time[0])) / oversample_factor * unit

    if frequency is not None and any(a is not None for a in [minimum_frequency, maximum_frequency]):
        log.warning("You have passed both a grid of frequencies and min_frequency/maximum_frequency arguments; "
                    "the latter will be ignored.")
    if period is not None and any(a is not None for a in [minimum_period, maximum_period]):
        log.warning("You have passed a grid of periods and minimum_period/maximum_period arguments; "
                    "the latter will be ignored.")
    if maximum_period is not None:
        minimum_frequency = 1.0 / maximum_period
    if minimum_period is not None:
        maximum_frequency = 1.0 / minimum_period
    if period is not None:
        frequency = 1.0 / np.asarray(period, dtype=np.float64)
    if frequency is None:
        if minimum_frequency is not None and maximum_frequency is not None:
            if minimum_frequency > maximum_frequency:
                if default_view == "frequency":
                    raise ValueError("minimum_frequency cannot be larger than maximum_frequency")
                raise ValueError("minimum_period cannot be larger than maximum_period")
        if minimum_frequency is None:
            minimum_frequency = fs
        if maximum_frequency is None:
            maximum_frequency = nyquist * nyquist_factor
        frequency = np.arange(minimum_frequency, maximum_frequency, fs)
    frequency = np.asarray(frequency, dtype=np.float64)

    ls_method = validate_method(ls_method, list(_LS_METHODS))
    if ls_method[:9] == "fastnifty":  # optional dependency in the reference; it falls back the same way
        oldmethod = ls_method
        ls_method = {"fastnifty": "fast", "fastnifty_chi2": "fastchi2"}[ls_method]
        log.warning("nifty_ls is not available.\nMethod has been changed from '{}' to '{}'.".format(oldmethod,
                                                                                                 ls_method))
    if not is_regular(frequency) and ls_method in ["fastchi2", "fast"]:
        oldmethod = ls_method
        ls_method = {"fastchi2": "chi2", "fast": "slow"}[ls_method]
        log.warning("The requested periodogram is not evenly sampled in frequency.\n"
                    "Method has been changed from '{}' to '{}' to allow for this.".format(oldmethod, ls_method))
    if nterms > 1 and ls_method not in ["fastchi2", "chi2"]:
        warnings.warn(
            "Building a Lomb Scargle Periodogram using the `slow` method. "
            "`nterms` has been set to >1, however this is not supported under the `{}` method. "
            "To run with higher nterms, set `ls_method` to either 'fastchi2', 'chi2', or 'fastnifty_chi2. "
            "Please refer to the `astropy.timeseries.periodogram.LombScargle` documentation.".format(ls_method),
            LightkurveWarning)
        nterms = 1
    if ls_method == "auto":
        # astropy resolves 'auto' itself (lombscargle/implementations/main.py:79-108 validate_method): a regular grid
        # of more than 200 frequencies goes to the extirpolation + FFT method, anything else to an exact method
        # (nterms is 1 here: the guard above reset it for every name outside chi2/fastchi2)
        ls_method = "fast" if (len(frequency) > 200 and is_regular(frequency)) else "cython"
    if nterms > _capi.MAX_NTERMS:
        raise NotImplementedError("the HIP multi-term kernels are instantiated for nterms <= %d (got %d)"
                                  % (_capi.MAX_NTERMS, nterms))
    dy = kwargs.pop("dy", None)
    fit_mean = kwargs.pop("fit_mean", True)
    center_data = kwargs.pop("center_data", True)
    if kwargs:
        raise TypeError("unexpected keyword argument(s) for LombScargle: %s" % sorted(kwargs))
    if dy is not None:
        dy = np.broadcast_to(np.asarray(dy, dtype=np.float64), time.shape).copy()

    # what the kernel needs: times relative to the first cadence (astropy lombscargle/core.py:119-126),
    # frequencies in 1/d, and lightkurve's normalisation (periodogram.py:969-975)
    f_day = frequency / unit
    if normalization == "psd":
        norm, scale = "lk_psd", 2.0 / (len(time) * oversample_factor * fs)
        power_unit = "flux^2/" + freq_unit
    else:
        norm, scale = "lk_amplitude", 1.0
        power_unit = "flux"
    return dict(lc=lc, trel=time - time[0], t0=time[0], flux=flux, dy=dy, frequency=frequency, f_day=f_day, norm=norm, scale=scale,
                nyquist=nyquist, freq_unit=freq_unit, power_unit=power_unit, default_view=default_view,
                ls_method=ls_method, nterms=nterms, fit_mean=fit_mean, center_data=center_data,
                normalization=normalization)


class LombScarglePeriodogram(Periodogram):
    """Lomb-Scargle periodogram computed by the exact HIP kernels (reference periodogram.py:589-1018)."""

    def model(self, time, frequency=None, device=0):
        return _ls_model(self, time, frequency, device)

    def __init__(self, *args, **kwargs):
        self._LS_inputs = kwargs.pop("ls_obj", None)
        self.nterms = kwargs.pop("nterms", 1)
        self.ls_method = kwargs.pop("ls_method", "hip")
        super(LombScarglePeriodogram, self).__init__(*args, **kwargs)

    @staticmethod
    def from_lightcurve(lc, minimum_frequency=None, maximum_frequency=None, minimum_period=None,
                        maximum_period=None, frequency=None, period=None, nterms=1, nyquist_factor=1,
                        oversample_factor=None, freq_unit=None, normalization="amplitude", ls_method="fast",
                        device=0, **kwargs):
        """Same contract and default as the reference constructor.  ``ls_method="fast"`` (default) runs the
        reference's default algorithm — Press & Rybicki extirpolation + FFT — on the GPU and reproduces lightkurve's
        default output to 1e-9; like the reference it needs a regular frequency grid and otherwise switches to
        ``"slow"``.  Every other name (``"slow"``, ``"cython"``, ``"chi2"``, ``"scipy"``, ``"auto"``, ``"hip"``) runs
        the exact fp64 direct-sum kernels (== the reference's exact methods to 1e-9).  ``nterms`` > 1 (with
        ``ls_method`` ``"chi2"`` or ``"fastchi2"``, as in the reference) runs the multi-term kernels: exact sums for
        ``"chi2"``, extirpolated FFT sums for ``"fastchi2"`` — each reproduces the reference's output of that name."""
        plan = _ls_plan(lc, minimum_frequency, maximum_frequency, minimum_period, maximum_period, frequency, period,
                        nterms, nyquist_factor, oversample_factor, freq_unit, normalization, ls_method, **kwargs)
        n = len(plan["trel"])
        grid = exact_grid(plan["f_day"])
        common = dict(dy=plan["dy"], fit_mean=plan["fit_mean"], center_data=plan["center_data"],
                      normalization=plan["norm"], scale=[plan["scale"]], device=device)
        if plan["nterms"] > 1 and plan["ls_method"] == "fastchi2":
            # the reference's FFT-based multi-term method, same arithmetic (extirpolated trig sums): periodogram.py:948-967
            fd = plan["f_day"]
            f0, dfq = (float(fd[0]), float(fd[1] - fd[0])) if len(fd) > 1 else (float(fd[0]), float(fd[0]))
            power = _capi.ls_fast_batch(plan["trel"], plan["flux"], [0, n], f0=f0, df=dfq, M=len(fd),
                                        nterms=plan["nterms"], **common)[0]
        elif plan["nterms"] > 1:
            # multi-term least-squares periodogram with exact sums ('chi2'): reference periodogram.py:948-967
            if grid is not None:
                power = _capi.ls_power_batch(plan["trel"], plan["flux"], [0, n], f0=grid[0], df=grid[1],
                                             M=len(plan["f_day"]), nterms=plan["nterms"], **common)[0]
            else:
                power = _capi.ls_power_batch(plan["trel"], plan["flux"], [0, n], frequency=plan["f_day"],
                                             nterms=plan["nterms"], **common)[0]
        elif plan["ls_method"] in ("fast", "fastchi2"):
            # astropy _get_frequency_grid (main.py:53-80): f0 = frequency[0], df = frequency[1] - frequency[0]
            fd = plan["f_day"]
            f0, dfq = (float(fd[0]), float(fd[1] - fd[0])) if len(fd) > 1 else (float(fd[0]), float(fd[0]))
            power = _capi.ls_fast_batch(plan["trel"], plan["flux"], [0, n], f0=f0, df=dfq, M=len(fd), **common)[0]
        elif grid is not None:
            power = _capi.ls_power_batch(plan["trel"], plan["flux"], [0, n], f0=grid[0], df=grid[1],
                                         M=len(plan["f_day"]), **common)[0]
        else:
            power = _capi.ls_power_batch(plan["trel"], plan["flux"], [0, n], frequency=plan["f_day"], **common)[0]
        lcc = plan["lc"]
        return LombScarglePeriodogram(
            frequency=plan["frequency"], power=power, nyquist=plan["nyquist"], targetid=lcc.meta.get("TARGETID"),
            label=lcc.meta.get("LABEL"), default_view=plan["default_view"],
            ls_obj=dict(trel=plan["trel"], flux=plan["flux"], dy=plan["dy"], t0=float(plan["t0"]),
                        fit_mean=plan["fit_mean"], center_data=plan["center_data"]), nterms=plan["nterms"],
            ls_method=plan["ls_method"], meta=lcc.meta, frequency_unit=plan["freq_unit"],
            power_unit=plan["power_unit"])


def _ls_model(self, time, frequency=None, device=0):
    """Best-fit truncated Fourier series at one frequency, evaluated at ``time`` (reference periodogram.py:991-1018 ->
    astropy LombScargle.model -> mle.periodic_fit, implementations/mle.py:58-114): weighted least squares of
    [1,] sin(2 pi m f t), cos(2 pi m f t), m <= nterms, on the light curve the periodogram came from (one
    Gram + solve on the GPU regression path), plus the weighted mean that was removed; normalised like the reference."""
    from .lightcurve import LightCurve
    if self._LS_inputs is None:
        raise ValueError("No Lomb Scargle inputs are attached to this periodogram.")
    if frequency is None:
        frequency = self.frequency_at_max_power
    f_day = float(frequency) / _freq_unit_factor(self.frequency_unit)
    trel, y, dy = self._LS_inputs["trel"], self._LS_inputs["flux"], self._LS_inputs["dy"]
    t0 = self._LS_inputs.get("t0", 0.0)
    fit_mean, center = self._LS_inputs.get("fit_mean", True), self._LS_inputs.get("center_data", True)

    def design(t):
        cols = [np.ones_like(t)] if fit_mean else []
        for m in range(1, self.nterms + 1):
            cols += [np.sin(2 * np.pi * m * f_day * t), np.cos(2 * np.pi * m * f_day * t)]
        return np.column_stack(cols)

    w = np.ones_like(y) if dy is None else dy ** -2.0
    y_mean = np.dot(y, w) / w.sum() if center else 0.0
    res = _capi.regress_batch(design(trel), y - y_mean, [0, len(y)], err=dy, sigma=1e300, niters=1, device=device)
    theta = res["coefficients"][0]
    tfit = np.asarray(time, dtype=np.float64) - t0
    lc = LightCurve(time=np.asarray(time, dtype=np.float64), flux=y_mean + design(tfit).dot(theta),
                    meta={"FREQUENCY": frequency, "LABEL": "LS Model",
                          "TARGETID": "{} LS Model".format(self.targetid)})
    return lc.normalize()


def _bls_plan(lc, **kwargs):
    """Everything BoxLeastSquaresPeriodogram.from_lightcurve decides before calling astropy (reference
    periodogram.py:1093-1168 + astropy bls/core.py:113-214, 277-327)."""
    lc = lc.remove_nans()
    time = np.asarray(lc.time, dtype=np.float64)
    flux = np.asarray(lc.flux, dtype=np.float64)
    dy = np.asarray(lc.flux_err, dtype=np.float64) if np.isfinite(lc.flux_err).all() else None
    duration = kwargs.pop("duration", [0.05, 0.10, 0.15, 0.20, 0.25, 0.33])
    if duration is not None and not np.all(np.isfinite(duration)):
        raise ValueError("`duration` parameter contains illegal nan or inf value(s)")
    period = kwargs.pop("period", None)
    minimum_period = kwargs.pop("minimum_period", None)
    maximum_period = kwargs.pop("maximum_period", None)
    if period is not None and not np.all(np.isfinite(period)):
        raise ValueError("`period` parameter contains illegal nan or inf value(s)")
    dt_med = np.median(np.diff(time))
    if minimum_period is None:
        minimum_period = np.max([dt_med * 4, np.max(duration) + dt_med]) if period is None else np.min(period)
    if maximum_period is None:
        maximum_period = (np.max(time) - np.min(time)) / 3.0 if period is None else np.max(period)
    time_unit = kwargs.pop("time_unit", "day")
    if time_unit not in ("day", "d", "hour", "h", "minute", "min", "second", "s", "year", "yr"):
        raise ValueError("{} is not a valid value for `time_unit`".format(time_unit))
    frequency_factor = kwargs.pop("frequency_factor", 10)
    baseline = np.max(time) - np.min(time)
    df = frequency_factor * np.min(duration) / baseline ** 2
    npoints = int(((1 / minimum_period) - (1 / maximum_period)) / df)
    if npoints > 1e7:
        raise ValueError("`period` contains {} points.Periodogram is too large to evaluate. "
                         "Consider setting `frequency_factor` to a higher value.".format(np.round(npoints, 4)))
    elif npoints > 1e5:
        log.warning("`period` contains {} points.Periodogram is likely to be large, and slow to evaluate. "
                    "Consider setting `frequency_factor` to a higher value.".format(np.round(npoints, 4)))
    duration = np.atleast_1d(np.asarray(duration, dtype=np.float64))
    trel = time - time[0]
    if period is None:
        # astropy BoxLeastSquares.autoperiod, bls/core.py:113-214
        if minimum_period > maximum_period:
            raise ValueError("The maximum period must be larger than the minimum period")
        minimum_frequency, maximum_frequency = 1.0 / maximum_period, 1.0 / minimum_period
        nf = 1 + int(np.round((maximum_frequency - minimum_frequency) / df))
        period = 1.0 / (maximum_frequency - df * np.arange(nf))
    period = np.atleast_1d(np.asarray(period, dtype=np.float64))
    # astropy _validate_period_and_duration, bls/core.py:668-700
    if period.ndim != 1 or period.size == 0:
        raise ValueError("period must be 1-dimensional")
    if duration.ndim != 1 or duration.size == 0:
        raise ValueError("duration must be 1-dimensional")
    if np.min(period) <= np.max(duration):
        raise ValueError("The maximum transit duration must be shorter than the minimum period")
    oversample = kwargs.pop("oversample", 10)
    try:
        oversample = int(oversample)
    except TypeError:
        raise ValueError("oversample must be an int, got {0}".format(oversample))
    if oversample < 1:
        raise ValueError("oversample must be greater than or equal to 1")
    objective = kwargs.pop("objective", None) or "likelihood"
    if objective not in ["snr", "likelihood"]:
        raise ValueError("Unrecognized method '{0}'\nallowed methods are: {1}".format(objective,
                                                                                    ["snr", "likelihood"]))
    method = kwargs.pop("method", None) or "fast"
    if method not in ["fast", "slow", "hip"]:
        raise ValueError("Unrecognized method '{0}'\nallowed methods are: {1}".format(method, ["fast", "slow"]))
    if kwargs:
        raise TypeError("unexpected keyword argument(s) for BoxLeastSquares.power: %s" % sorted(kwargs))
    t_ref = np.min(trel)
    ivar = np.ones_like(flux) if dy is None else 1.0 / dy ** 2
    return dict(lc=lc, t=trel - t_ref, y=flux - np.median(flux), ivar=ivar, t_ref=t_ref + time[0], period=period,
                duration=duration, oversample=oversample, objective=objective, time_unit=time_unit)


class BoxLeastSquaresPeriodogram(Periodogram):
    """BLS periodogram computed by the bit-exact HIP kernel (reference periodogram.py:1021-1296)."""

    def __init__(self, *args, **kwargs):
        self.duration = kwargs.pop("duration", None)
        self.depth = kwargs.pop("depth", None)
        self.snr = kwargs.pop("snr", None)
        self._BLS_result = kwargs.pop("bls_result", None)
        self._BLS_inputs = kwargs.pop("bls_obj", None)
        self.transit_time = kwargs.pop("transit_time", None)
        self.time = kwargs.pop("time", None)
        self.flux = kwargs.pop("flux", None)
        self.time_unit = kwargs.pop("time_unit", None)
        super(BoxLeastSquaresPeriodogram, self).__init__(*args, **kwargs)

    @staticmethod
    def from_lightcurve(lc, device=0, **kwargs):
        """Same contract as the reference constructor: kwargs ``duration``, ``period``, ``minimum_period``,
        ``maximum_period``, ``frequency_factor``, ``time_unit``, ``objective``, ``oversample``, ``method``.
        No period limit, like the reference: periods whose phase bins (period / (shortest duration / oversample)) do not fit
        LDS — beyond ``_capi.bls_max_period(duration, oversample)``: 46 d for 0.05-d durations at oversample 10 — run the
        global-memory kernel (same bits, ~100 x the cost per period)."""
        plan = _bls_plan(lc, **kwargs)
        n = len(plan["t"])
        res = _capi.bls_batch(plan["t"], plan["y"], plan["ivar"], [0, n], plan["period"], plan["duration"],
                              plan["oversample"], plan["objective"] == "likelihood", device=device)
        result = {k: v[0] for k, v in res.items()}
        result["transit_time"] = result["transit_time"] + plan["t_ref"]   # astropy _format_results, core.py:702-745
        result["period"] = plan["period"]
        result["objective"] = plan["objective"]
        lcc = plan["lc"]
        return BoxLeastSquaresPeriodogram(
            frequency=1.0 / plan["period"], power=result["power"], default_view="period",
            label=lcc.meta.get("LABEL"), targetid=lcc.meta.get("TARGETID"), transit_time=result["transit_time"],
            duration=result["duration"], depth=result["depth"], bls_result=result, snr=result["depth_snr"],
            bls_obj=dict(t=plan["t"], y=plan["y"], ivar=plan["ivar"]), time=lcc.time, flux=lcc.flux,
            time_unit=plan["time_unit"], frequency_unit="1/d", power_unit="")

    @property
    def duration_at_max_power(self):
        return self.duration[np.nanargmax(self.power)]

    @property
    def transit_time_at_max_power(self):
        return self.transit_time[np.nanargmax(self.power)]

    @property
    def depth_at_max_power(self):
        return self.depth[np.nanargmax(self.power)]

    def get_transit_model(self, period=None, duration=None, transit_time=None):
        """Box model of the transits (reference periodogram.py:1229-1269 -> astropy BoxLeastSquares.model,
        bls/core.py:332-387): the ivar-weighted mean flux inside / outside the transit windows.  Element-wise host
        arithmetic on the light curve the periodogram came from — no search involved."""
        from .lightcurve import LightCurve
        if period is None:
            period = self.period_at_max_power
            log.warning("No period specified. Using period at max power")
        if duration is None:
            duration = self.duration_at_max_power
            log.warning("No duration specified. Using duration at max power")
        if transit_time is None:
            transit_time = self.transit_time_at_max_power
            log.warning("No transit time specified. Using transit time at max power")
        t0 = float(self.time[0])                         # astropy works on times relative to the first cadence
        t = np.asarray(self.time, dtype=np.float64) - t0
        tt = float(transit_time) - t0
        y, ivar = np.asarray(self.flux, dtype=np.float64), self._BLS_inputs["ivar"]
        hp = 0.5 * period
        m_in = np.abs((t - tt + hp) % period - hp) < 0.5 * duration
        m_out = ~m_in
        with np.errstate(invalid="ignore", divide="ignore"):
            y_in = np.sum(y[m_in] * ivar[m_in]) / np.sum(ivar[m_in])
            y_out = np.sum(y[m_out] * ivar[m_out]) / np.sum(ivar[m_out])
        model = y_out + np.zeros_like(t)
        model[m_in] = y_in
        return LightCurve(time=self.time, flux=model, label="Transit Model Flux")

    def compute_stats(self, period=None, duration=None, transit_time=None):
        """Vetting statistics of one box model (reference periodogram.py:1194-1229 -> astropy
        BoxLeastSquares.compute_stats, bls/core.py:389-570).  Same keys and values as astropy's dict (plain floats /
        ndarrays instead of Quantities; ``transit_times`` absolute like ``self.time``).  Element-wise host arithmetic over
        the one light curve the periodogram came from, as in the reference — no search involved."""
        if period is None:
            period = self.period_at_max_power
            log.warning("No period specified. Using period at max power")
        if duration is None:
            duration = self.duration_at_max_power
            log.warning("No duration specified. Using duration at max power")
        if transit_time is None:
            transit_time = self.transit_time_at_max_power
            log.warning("No transit time specified. Using transit time at max power")
        period, duration = float(period), float(duration)
        if not (period > 0 and duration > 0):
            raise ValueError("period and duration must be positive")
        if duration >= period:
            raise ValueError("The maximum transit duration must be shorter than the minimum period")
        t0 = float(self.time[0])                         # astropy works on times relative to the first cadence
        t = np.asarray(self.time, dtype=np.float64) - t0
        tt = float(transit_time) - t0
        y = np.asarray(self.flux, dtype=np.float64)
        ivar = np.asarray(self._BLS_inputs["ivar"], dtype=np.float64)

        def _compute_depth(m, y_out=None, var_out=None):
            if np.any(m) and (var_out is None or np.isfinite(var_out)):
                var_m = 1.0 / np.sum(ivar[m])
                y_m = np.sum(y[m] * ivar[m]) * var_m
                if y_out is None:
                    return y_m, var_m
                return y_out - y_m, np.sqrt(var_m + var_out)
            return 0.0, np.inf

        hp = 0.5 * period
        m_in = np.abs((t - tt + hp) % period - hp) < 0.5 * duration
        m_out = ~m_in
        m_odd = np.abs((t - tt) % (2 * period) - period) < 0.5 * duration
        m_even = np.abs((t - tt + period) % (2 * period) - period) < 0.5 * duration
        y_out, var_out = _compute_depth(m_out)
        depth = _compute_depth(m_in, y_out, var_out)
        depth_odd = _compute_depth(m_odd, y_out, var_out)
        depth_even = _compute_depth(m_even, y_out, var_out)
        y_in = y_out - depth[0]
        m_phase = np.abs((t - tt) % period - hp) < 0.5 * duration
        depth_phase = _compute_depth(m_phase, *_compute_depth((~m_phase) & m_out))
        m_half = np.abs((t - tt + 0.25 * period) % (0.5 * period) - 0.25 * period) < 0.5 * duration
        depth_half = _compute_depth(m_half, *_compute_depth(~m_half))
        transit_id = np.round((t[m_in] - tt) / period).astype(int)
        transit_times = period * np.arange(transit_id.min(), transit_id.max() + 1) + tt
        unique_ids, unique_counts = np.unique(transit_id, return_counts=True)
        unique_ids -= np.min(transit_id)
        transit_id -= np.min(transit_id)
        counts = np.zeros(np.max(transit_id) + 1, dtype=int)
        counts[unique_ids] = unique_counts
        ll = -0.5 * ivar[m_in] * ((y[m_in] - y_in) ** 2 - (y[m_in] - y_out) ** 2)
        lls = np.zeros(len(counts))
        for i in unique_ids:
            lls[i] = np.sum(ll[transit_id == i])
        full_ll = -0.5 * np.sum(ivar[m_in] * (y[m_in] - y_in) ** 2)
        full_ll -= 0.5 * np.sum(ivar[m_out] * (y[m_out] - y_out) ** 2)
        A = np.vstack((np.sin(2 * np.pi * t / period), np.cos(2 * np.pi * t / period), np.ones_like(t))).T
        w = np.linalg.solve(np.dot(A.T, A * ivar[:, None]), np.dot(A.T, y * ivar))
        sin_ll = -0.5 * np.sum((y - np.dot(A, w)) ** 2 * ivar)
        return dict(transit_times=transit_times + t0, per_transit_count=counts, per_transit_log_likelihood=lls,
                    depth=depth, depth_phased=depth_phase, depth_half=depth_half, depth_odd=depth_odd,
                    depth_even=depth_even, harmonic_amplitude=np.sqrt(np.sum(w[:2] ** 2)),
                    harmonic_delta_log_likelihood=sin_ll - full_ll)

    def get_transit_mask(self, period=None, duration=None, transit_time=None):
        """True where the box model is in transit (reference periodogram.py:1271-1292)."""
        model = self.get_transit_model(period=period, duration=duration, transit_time=transit_time)
        return model.flux != np.median(model.flux)
