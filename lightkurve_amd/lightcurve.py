"""Minimal light-curve container mirroring the slice of ``lightkurve.LightCurve`` the hot path touches.

Reference: src/lightkurve/lightcurve.py — constructor :355, ``remove_nans`` :1300-1327, ``flatten`` :943-1078,
``fold`` :1089-1214, ``to_periodogram`` :2490-2535.  Columns are plain float64 ndarrays (no astropy
Time/Quantity in the product interpreter); ``meta`` carries TARGETID / LABEL / NORMALIZED like the reference.
"""
import copy as _copy

import numpy as np

__all__ = ["LightCurve", "FoldedLightCurve", "estimate_cdpp_batch", "running_mean"]


class LightCurve(object):
    def __init__(self, time=None, flux=None, flux_err=None, meta=None, **extra):
        if time is None and flux is not None:
            time = np.arange(len(flux), dtype=np.float64)
        self.time = np.array(time, dtype=np.float64)
        if self.time.ndim != 1:
            raise ValueError("time must be one-dimensional")
        self.flux = np.array(flux, dtype=np.float64) if flux is not None else np.full(len(self.time), np.nan)
        if flux_err is None:
            self.flux_err = np.full(len(self.time), np.nan)
        else:
            self.flux_err = np.array(np.broadcast_to(np.asarray(flux_err, dtype=np.float64), self.time.shape))
        if self.flux.shape != self.time.shape:
            raise ValueError("time and flux must have the same length")
        self.meta = dict(meta or {})
        for k in ("targetid", "label"):
            if k in extra:
                self.meta[k.upper()] = extra.pop(k)
        if extra:
            raise TypeError("unexpected keyword argument(s): %s" % sorted(extra))

    # ---------------------------------------------------------------- container plumbing
    def __len__(self):
        return len(self.time)

    def __getitem__(self, key):
        if isinstance(key, str):
            return getattr(self, key)
        new = _copy.copy(self)
        new.meta = dict(self.meta)
        new.time, new.flux, new.flux_err = self.time[key].copy(), self.flux[key].copy(), self.flux_err[key].copy()
        return new

    def copy(self):
        return self[slice(None)]

    @property
    def targetid(self):
        return self.meta.get("TARGETID")

    @property
    def label(self):
        return self.meta.get("LABEL")

    def __repr__(self):
        return "<LightCurve length=%d targetid=%s>" % (len(self), self.targetid)

    def remove_nans(self, column="flux"):
        """New light curve without the cadences where ``column`` is NaN (reference :1300-1327)."""
        return self[~np.isnan(self[column])]

    def normalize(self, unit="unscaled"):
        """Flux and flux_err divided by the median flux (reference :1216-1292); ``unit`` scales by 1e2/1e3/1e6."""
        import warnings
        from .periodogram import LightkurveWarning, validate_method
        validate_method(unit, ["unscaled", "percent", "ppt", "ppm"])
        median_flux = np.nanmedian(self.flux)
        std_flux = np.nanstd(self.flux)
        if (median_flux == 0) or (np.isfinite(std_flux) and (np.abs(median_flux) < 0.5 * std_flux)):
            warnings.warn("The light curve appears to be zero-centered (median={:.2e} +/- {:.2e}); `normalize()` "
                          "will divide the light curve by a value close to zero, which is probably not what you "
                          "want.".format(median_flux, std_flux), LightkurveWarning)
        if median_flux < 0:
            warnings.warn("The light curve has a negative median flux ({:.2e}); `normalize()` will therefore divide "
                          "by a negative number and invert the light curve, which is probably not what you "
                          "want".format(median_flux), LightkurveWarning)
        lc = self.copy()
        scale = {"unscaled": 1.0, "percent": 1e2, "ppt": 1e3, "ppm": 1e6}[unit]
        with np.errstate(invalid="ignore", divide="ignore"):
            lc.flux = lc.flux / median_flux * scale
            lc.flux_err = lc.flux_err / median_flux * scale
        lc.meta["NORMALIZED"] = True
        return lc

    def remove_outliers(self, sigma=5.0, return_mask=False, maxiters=5, device=0):
        """Light curve without the cadences astropy.stats.sigma_clip flags (median centre, std width, ``maxiters``
        rounds; NaN flux counts as an outlier) — reference :1430-1556, the default ``corrector_func`` of
        ``TargetPixelFile.plot_pixels``; the clip runs on the GPU (lk_sigma_clip_batch)."""
        from . import _capi
        mask = _capi.sigma_clip_batch(self.flux, [0, len(self)], sigma=sigma, maxiters=maxiters, device=device)
        return (self[~mask], mask) if return_mask else self[~mask]

    def bin(self, time_bin_size=0.5, time_bin_start=None, device=0):
        """Equal-width time bins (reference :1558-1763, the ``time_bin_size`` [d] form): nanmean of the flux, rms of the
        errors (nanstd of the flux when there are none); one GPU call (lk_bin_batch).  Empty bins hold NaN."""
        from . import _capi
        t, f, e, _off = _capi.bin_batch(self.time, self.flux, [0, len(self)], flux_err=self.flux_err,
                                        time_bin_size=time_bin_size, time_bin_start=time_bin_start, device=device)
        return LightCurve(time=t, flux=f, flux_err=e, meta=self.meta)

    def create_transit_mask(self, period, transit_time, duration, device=0):
        """True for the cadences inside any of the transits (reference :2967-3037); lk_transit_mask_batch."""
        from . import _capi
        return _capi.transit_mask_batch(self.time, [0, len(self)], period, duration, transit_time, device=device)

    def __sub__(self, other):
        """lc - scalar shifts the flux (reference LightCurve.__add__/__sub__ :610-660, scalar case)."""
        new = self.copy()
        new.flux = new.flux - other
        return new

    def __add__(self, other):
        new = self.copy()
        new.flux = new.flux + other
        return new

    # ---------------------------------------------------------------- hot-path entry points
    def to_periodogram(self, method="lombscargle", **kwargs):
        """``method`` in {"lombscargle", "ls", "boxleastsquares", "bls"} (reference :2490-2535)."""
        from .periodogram import BoxLeastSquaresPeriodogram, LombScarglePeriodogram, validate_method
        supported = ["ls", "bls", "lombscargle", "boxleastsquares"]
        method = validate_method(method.replace(" ", ""), supported)
        if method in ["bls", "boxleastsquares"]:
            return BoxLeastSquaresPeriodogram.from_lightcurve(lc=self, **kwargs)
        return LombScarglePeriodogram.from_lightcurve(lc=self, **kwargs)

    def flatten(self, window_length=101, polyorder=2, return_trend=False, break_tolerance=5, niters=3, sigma=3,
                mask=None, device=0):
        """Savitzky-Golay detrending with gap splitting and iterative sigma clipping (reference :943-1078);
        the trend is computed on the GPU (lk_savgol_trend_batch)."""
        from .flatten import flatten_trend_batch
        trend = flatten_trend_batch([self], window_length=window_length, polyorder=polyorder,
                                    break_tolerance=break_tolerance, niters=niters, sigma=sigma,
                                    masks=None if mask is None else [mask], device=device)[0]
        flat = self.copy()
        with np.errstate(invalid="ignore", divide="ignore"):
            flat.flux = flat.flux / trend
            flat.flux_err = flat.flux_err / trend
        flat.meta["NORMALIZED"] = True
        if return_trend:
            tr = self.copy()
            tr.flux = trend
            return flat, tr
        return flat

    def estimate_cdpp(self, transit_duration=13, savgol_window=101, savgol_polyorder=2, sigma=5.0, device=0):
        """Savitzky-Golay CDPP noise metric in ppm (reference :1764-1833): flatten -> remove_outliers -> normalize("ppm")
        -> standard deviation of the ``transit_duration``-cadence running mean.  The flatten and the sigma clip run on the
        GPU; see ``estimate_cdpp_batch`` for many light curves in two launches."""
        return float(estimate_cdpp_batch([self], transit_duration=transit_duration, savgol_window=savgol_window,
                                         savgol_polyorder=savgol_polyorder, sigma=sigma, device=device)[0])

    def fold(self, period=None, epoch_time=None, epoch_phase=0, wrap_phase=None, normalize_phase=False, device=0):
        """Phase-fold (reference :1089-1214 over astropy TimeSeries.fold, timeseries/sampled.py:230-233):
        phase = ((t - epoch) + epoch_phase + (P - wrap)) % P - (P - wrap), then a stable sort by phase — both on
        the GPU (lk_fold_batch: bit-identical phases, permutation == np.argsort(phase, kind="stable"))."""
        from . import _capi
        period = float(period)
        epoch_given = epoch_time is not None
        if epoch_time is None:
            epoch_time = float(self.time[0]) if len(self) else 0.0
        if wrap_phase is None:
            wrap_phase = period / 2.0 if not normalize_phase else 0.5
        phase, order, (flux, flux_err) = _capi.fold_batch(
            self.time, [0, len(self)], period, epoch_time, epoch_phase=epoch_phase, wrap_phase=wrap_phase,
            normalize_phase=normalize_phase, columns=(self.flux, self.flux_err), device=device)
        out = FoldedLightCurve(time=phase, flux=flux, flux_err=flux_err, meta=dict(self.meta))
        out.time_original = self.time[order]
        # FoldedLightCurve.cycle (reference :3213-3229): floor((t - (epoch - P/2)) / P), first cycle = 0; without an
        # explicit epoch_time the reference takes the smallest folded time as the epoch
        cyc_epoch = epoch_time if epoch_given else (float(np.min(phase)) if len(phase) else 0.0)
        cyc = np.floor((out.time_original - (cyc_epoch - period / 2.0)) / period).astype(int)
        out.cycle = cyc - cyc.min() if len(cyc) else cyc
        out.period, out.epoch_time, out.epoch_phase, out.wrap_phase = period, epoch_time, epoch_phase, wrap_phase
        out.normalize_phase = normalize_phase
        return out


class FoldedLightCurve(LightCurve):
    """Folded light curve: ``time`` holds the phase; ``time_original`` and ``cycle`` as in the reference."""

    @property
    def phase(self):
        return self.time


def running_mean(data, window_size):
    """Moving average by differences of the cumulative sum (reference utils.py:374-386)."""
    if window_size > len(data):
        window_size = len(data)
    cumsum = np.cumsum(np.insert(data, 0, 0))
    return (cumsum[window_size:] - cumsum[:-window_size]) / float(window_size)


def estimate_cdpp_batch(lcs, transit_duration=13, savgol_window=101, savgol_polyorder=2, sigma=5.0, device=0):
    """``LightCurve.estimate_cdpp`` (reference lightcurve.py:1764-1833) for a list of light curves: ONE
    lk_savgol_trend_batch call (the flatten of every light curve) and ONE lk_sigma_clip_batch call (remove_outliers),
    then per light curve the O(N) tail the reference runs in numpy (normalize to ppm, running mean, np.std).
    Returns float64[len(lcs)] in ppm."""
    from . import _capi
    from .flatten import flatten_trend_batch
    if not isinstance(transit_duration, int):
        raise ValueError("transit_duration must be an integer in units number of cadences, got {}.".format(transit_duration))
    lcs = list(lcs)
    if not lcs:
        return np.zeros(0)
    trends = flatten_trend_batch(lcs, window_length=savgol_window, polyorder=savgol_polyorder, device=device)
    with np.errstate(invalid="ignore", divide="ignore"):
        flat = [np.asarray(lc.flux, dtype=np.float64) / tr for lc, tr in zip(lcs, trends)]
    off = np.zeros(len(lcs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(f) for f in flat])
    mask = _capi.sigma_clip_batch(np.concatenate(flat), off, sigma=sigma, maxiters=5, device=device)
    out = np.empty(len(lcs))
    for b, f in enumerate(flat):
        kept = f[~mask[off[b]:off[b + 1]]]
        ppm = kept / np.nanmedian(kept) * 1e6            # normalize("ppm") of the already-normalised flattened curve
        out[b] = np.std(running_mean(ppm, transit_duration))
    return out
