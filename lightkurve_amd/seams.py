"""Install the HIP kernels behind the reference's own seams when astropy is importable.

* S1: register ``'hip'`` (exact) in astropy's Lomb-Scargle ``METHODS`` and replace ``'fast'`` (the default of
  lightkurve's ``lc.to_periodogram()``, src/lightkurve/periodogram.py:650, 961-964) by the GPU FFT path; the CPU
  original stays reachable as ``'fast_cpu'``.
* S2: replace ``astropy.timeseries.periodograms.bls.methods.bls_fast`` (reached from periodogram.py:1169).

astropy is NOT available in the product interpreter of this image; the seams are exercised on the GPU box by
``tests/test_seams_gpu.py`` under the conda interpreter that ships astropy 4.3.1.  Nothing here falls back to CPU.
"""
import numpy as np

from . import _capi


def lombscargle_hip(t, y, dy=None, frequency=None, normalization="standard", fit_mean=True, center_data=True,
                    nterms=1, **unused):
    """Signature of astropy's METHODS entries (lombscargle/implementations/main.py:182-217)."""
    if not 1 <= nterms <= _capi.MAX_NTERMS:
        raise ValueError("the HIP kernels are instantiated for 1 <= nterms <= %d" % _capi.MAX_NTERMS)
    if normalization not in ("standard", "psd"):
        raise ValueError("normalization='{}' not recognized".format(normalization))
    t = np.asarray(t, dtype=np.float64)
    frequency = np.asarray(frequency, dtype=np.float64)
    from .periodogram import exact_grid
    grid = exact_grid(frequency)
    kw = dict(dy=dy, fit_mean=fit_mean, center_data=center_data, normalization=normalization, nterms=nterms)
    if grid is not None:
        return _capi.ls_power_batch(t, y, [0, len(t)], f0=grid[0], df=grid[1], M=len(frequency), **kw)[0]
    return _capi.ls_power_batch(t, y, [0, len(t)], frequency=frequency, **kw)[0]


def lombscargle_fast_hip(t, y, dy=None, f0=0, df=None, Nf=None, center_data=True, fit_mean=True,
                         normalization="standard", use_fft=True, trig_sum_kwds=None, nterms=1, **unused):
    """Signature of astropy's lombscargle_fast / lombscargle_fastchi2 (fast_impl.py:6, fastchi2_impl.py:8): the
    f0/df/Nf form every 'fast*' method receives; ``nterms`` > 1 arrives only under the name 'fastchi2'."""
    if not 1 <= nterms <= _capi.MAX_NTERMS:
        raise ValueError("the HIP kernels are instantiated for 1 <= nterms <= %d" % _capi.MAX_NTERMS)
    if normalization not in ("standard", "psd"):
        raise ValueError("normalization='{}' not recognized".format(normalization))
    if f0 < 0:
        raise ValueError("Frequencies must be positive")
    if df <= 0:
        raise ValueError("Frequency steps must be positive")
    if Nf <= 0:
        raise ValueError("Number of frequencies must be positive")
    kw = dict(trig_sum_kwds or {})
    t = np.asarray(t, dtype=np.float64)
    return _capi.ls_fast_batch(t, y, [0, len(t)], dy=dy, f0=float(f0), df=float(df), M=int(Nf), fit_mean=fit_mean,
                               center_data=center_data, normalization=normalization,
                               oversampling=int(kw.get("oversampling", 5)), nterms=nterms)[0]


def bls_fast_hip(t, y, ivar, period, duration, oversample, use_likelihood):
    """Signature of astropy's methods.bls_fast (bls/methods.py:55-95)."""
    res = _capi.bls_batch(t, y, ivar, [0, len(t)], period, duration, oversample, use_likelihood)
    return tuple(res[k][0] for k in _capi.BLS_FIELDS)


def install():
    """Patch astropy in place; returns the list of seams installed."""
    from astropy.timeseries.periodograms.bls import methods as bls_methods
    from astropy.timeseries.periodograms.lombscargle.implementations import main as ls_main
    ls_main.METHODS["hip"] = lombscargle_hip
    # multi-term fits (lightkurve nterms > 1 requires the name 'chi2' or 'fastchi2', periodogram.py:948-958):
    # 'chi2' receives the raw frequency array like 'hip' does, so it can be replaced one-to-one
    ls_main.METHODS.setdefault("chi2_cpu", ls_main.METHODS["chi2"])
    ls_main.METHODS["chi2"] = lombscargle_hip
    # the reference's DEFAULT method: lc.to_periodogram() reaches the GPU with no change on the caller's side
    ls_main.METHODS.setdefault("fast_cpu", ls_main.METHODS["fast"])
    ls_main.METHODS["fast"] = lombscargle_fast_hip
    ls_main.METHODS.setdefault("fastchi2_cpu", ls_main.METHODS["fastchi2"])
    ls_main.METHODS["fastchi2"] = lombscargle_fast_hip
    bls_methods._bls_fast_reference = getattr(bls_methods, "_bls_fast_reference", bls_methods.bls_fast)
    bls_methods.bls_fast = bls_fast_hip
    return ["lombscargle:METHODS['hip']", "lombscargle:METHODS['chi2']", "lombscargle:METHODS['fast']",
            "lombscargle:METHODS['fastchi2']", "bls:methods.bls_fast"]
