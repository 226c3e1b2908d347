"""Install the HIP kernels behind the reference's own seams when astropy is importable.

* S1: register ``'hip'`` in astropy's Lomb-Scargle ``METHODS`` so ``LombScargle(...).power(f, method='hip')`` and
  lightkurve's ``lc.to_periodogram(ls_method='hip')`` (src/lightkurve/periodogram.py:961-964) run on the GPU.
* S2: replace ``astropy.timeseries.periodograms.bls.methods.bls_fast`` (reached from periodogram.py:1169).

astropy is NOT available in the product interpreter of this image; the seams are exercised on the GPU box by
``tests/test_seams_gpu.py`` under the conda interpreter that ships astropy 4.3.1.  Nothing here falls back to CPU.
"""
import numpy as np

from . import _capi


def lombscargle_hip(t, y, dy=None, frequency=None, normalization="standard", fit_mean=True, center_data=True,
                    nterms=1, **unused):
    """Signature of astropy's METHODS entries (lombscargle/implementations/main.py:182-217)."""
    if nterms != 1:
        raise ValueError("nterms != 1 only supported with 'chi2' or 'fastchi2' methods")
    if normalization not in ("standard", "psd"):
        raise ValueError("normalization='{}' not recognized".format(normalization))
    t = np.asarray(t, dtype=np.float64)
    frequency = np.asarray(frequency, dtype=np.float64)
    from .periodogram import exact_grid
    grid = exact_grid(frequency)
    kw = dict(dy=dy, fit_mean=fit_mean, center_data=center_data, normalization=normalization)
    if grid is not None:
        return _capi.ls_power_batch(t, y, [0, len(t)], f0=grid[0], df=grid[1], M=len(frequency), **kw)[0]
    return _capi.ls_power_batch(t, y, [0, len(t)], frequency=frequency, **kw)[0]


def bls_fast_hip(t, y, ivar, period, duration, oversample, use_likelihood):
    """Signature of astropy's methods.bls_fast (bls/methods.py:55-95)."""
    res = _capi.bls_batch(t, y, ivar, [0, len(t)], period, duration, oversample, use_likelihood)
    return tuple(res[k][0] for k in _capi.BLS_FIELDS)


def install():
    """Patch astropy in place; returns the list of seams installed."""
    from astropy.timeseries.periodograms.bls import methods as bls_methods
    from astropy.timeseries.periodograms.lombscargle.implementations import main as ls_main
    ls_main.METHODS["hip"] = lombscargle_hip
    bls_methods._bls_fast_reference = getattr(bls_methods, "_bls_fast_reference", bls_methods.bls_fast)
    bls_methods.bls_fast = bls_fast_hip
    return ["lombscargle:METHODS['hip']", "bls:methods.bls_fast"]
