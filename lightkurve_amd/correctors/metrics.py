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

__all__ = ["overfit_metric_lombscargle", "overfit_metric_lombscargle_batch", "underfit_metric_neighbors"]


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


def overfit_metric_lombscargle_batch(original_lc, corrected_lcs, n_samples=1, device=0):
    """``overfit_metric_lombscargle(original_lc, c)`` for every corrected light curve ``c`` of a list that shares the
    cadences of ``original_lc`` (e.g. one correction per ridge penalty): the periodogram of the original is computed once,
    the len(corrected_lcs) corrected periodograms and all the noise periodograms are ONE batched GPU call each.  The noise
    draws follow the loop order of calling the scalar function once per light curve, so the values are the same as
    that loop's under the same seed."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        orig_lc = _prepared(original_lc)
        cors = [_prepared(c) for c in corrected_lcs]
    if not cors:
        return np.zeros(0)
    if any(len(c) != len(orig_lc) for c in cors):
        raise ValueError("every corrected light curve must share the cadences of the original")
    pg_orig = orig_lc.to_periodogram(device=device)
    freq = pg_orig.frequency
    n = len(orig_lc)
    corr_power = lombscargle_batch(cors, freq, device=device)
    noise, mean_unc = [], []
    for c in cors:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mu = np.nanmean(c.flux_err)
        mean_unc.append(mu)
        for _ in range(int(n_samples)):
            noise.append(LightCurve(time=orig_lc.time, flux=(np.random.randn(n, 1) * mu).T[0], flux_err=np.zeros(n)))
    noise_power = lombscargle_batch(noise, freq, device=device) if noise else np.zeros((0, len(freq)))
    out = np.empty(len(cors))
    for a in range(len(cors)):
        change = corr_power[a] - np.asarray(pg_orig.power)
        change = change[~np.isnan(change)]
        n_up = int(np.count_nonzero(change > 0.0))
        per_iter = []
        for k in range(int(n_samples)):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                mnp = np.nanmean(noise_power[a * int(n_samples) + k])
            if n_up == 0:
                per_iter.append(0.0)
            else:
                den = n_up * mnp
                per_iter.append(np.inf if den == 0 else np.sum(change[change > 0.0]) / den)
        metric = np.mean(per_iter) if per_iter else 0.0
        with np.errstate(over="ignore"):
            out[a] = 2.0 / (1.0 + np.exp(np.max([metric, 0.0])))
    return out


def underfit_metric_neighbors(corrected_lc, neighbor_flux):
    """Residual-correlation goodness (reference metrics.py:141-257) given the neighbours' flux on the cadences of
    ``corrected_lc`` (n_cadences x n_neighbours; the reference downloads and aligns them from MAST, :274-412 — control
    plane).  Pearson correlation of the normalised, median-subtracted target with each neighbour, the mean of the cubed
    absolute correlations scaled so that white-noise chance correlation maps to 0.95.  A (n_neighbours + 1)^2 correlation
    matrix of a few thousand cadences: host arithmetic, like the reference's."""
    import warnings
    nf = np.asarray(neighbor_flux, dtype=np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        keep = ~np.isnan(corrected_lc.flux)
        lc = _prepared(corrected_lc)
    nf = nf[keep]
    flux_matrix = np.column_stack([nf, lc.flux])
    flux_matrix = flux_matrix[~np.any(np.isnan(flux_matrix), axis=1)]
    # _compute_correlation (metrics.py:451-475): columns scaled by their RMS (NOT mean-subtracted), then X^T X / n
    n_cad = flux_matrix.shape[0]
    rms = np.sqrt(np.sum(flux_matrix ** 2.0, axis=0) / n_cad)
    rms[rms == 0.0] = np.inf
    unit = flux_matrix / rms[None, :]
    corr = unit.T.dot(unit) / n_cad
    beta = [0.0007, 0.8083, -0.5023]
    wgn = beta[0] + beta[1] * (n_cad ** beta[2])
    bad_limit = 0.95
    scale = 1.0 / wgn * np.log((2.0 / bad_limit) - 1.0)
    corr = np.tril(corr, k=-1) + np.triu(corr, k=+1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c = scale * np.nanmean(np.abs(corr) ** 3, axis=0)[-1]
    return float(2.0 / (1.0 + np.exp(c)))
