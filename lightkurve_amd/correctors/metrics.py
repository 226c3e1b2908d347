"""Over-fitting goodness metric on the GPU.

Reference: src/lightkurve/correctors/metrics.py:24-138 ``overfit_metric_lombscargle`` — per sample three default
(``ls_method="fast"``) Lomb-Scargle periodograms: the original light curve, the corrected one on the same grid, and a
white-noise light curve at the level of the corrected uncertainties; the metric compares the power the correction ADDED
with the noise power.  The first two periodograms do not depend on the sample, so they are computed once; the
``n_samples`` noise periodograms are one batched launch (``lombscargle_batch``).  The noise is drawn from the global
``np.random`` stream in the reference's order, so with the same seed the metric reproduces the reference's value.
"""
import numpy as np

from ..batch import lombscargle_batch
from ..lightcurve import LightCurve

__all__ = ["overfit_metric_lombscargle"]


def _prepared(lc):
    lc = lc.copy().remove_nans().normalize()
    return lc - 1.0


def overfit_metric_lombscargle(original_lc, corrected_lc, n_samples=10, device=0):
    """A float in [0, 1]: 0 = the correction injected broadband noise far above the uncertainties, 1 = none."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # zero-centred / zero-median light curves are expected here
        orig_lc = _prepared(original_lc)
        corrected_lc = _prepared(corrected_lc)
    if len(corrected_lc) == 0:
        return 1.0
    pg_orig = orig_lc.to_periodogram(device=device)
    pg_corr = corrected_lc.to_periodogram(frequency=pg_orig.frequency, device=device)
    n = len(orig_lc)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mean_unc = np.nanmean(corrected_lc.flux_err)
    noise = [LightCurve(time=orig_lc.time, flux=(np.random.randn(n, 1) * mean_unc).T[0], flux_err=np.zeros(n))
             for _ in range(int(n_samples))]
    # a light curve's default grid depends on its times only: the noise curves share pg_orig's grid
    noise_power = lombscargle_batch(noise, pg_orig.frequency, device=device) if noise else np.zeros((0, 1))
    change = np.asarray(pg_corr.power) - np.asarray(pg_orig.power)
    change = change[~np.isnan(change)]
    n_up = int(np.count_nonzero(change > 0.0))
    per_iter = []
    for k in range(int(n_samples)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mean_noise_power = np.nanmean(noise_power[k])
        if n_up == 0:
            per_iter.append(0.0)
        else:
            den = n_up * mean_noise_power
            per_iter.append(np.inf if den == 0 else np.sum(change[change > 0.0]) / den)
    metric = np.mean(per_iter)
    with np.errstate(over="ignore"):
        return float(2.0 / (1.0 + np.exp(np.max([metric, 0.0]))))
