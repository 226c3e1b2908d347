"""RegressionCorrector over the GPU normal-equations path
(reference: src/lightkurve/correctors/regressioncorrector.py:88-117, 127-189, 191-342)."""
import numpy as np

from .. import _capi
from ..lightcurve import LightCurve
from .designmatrix import DesignMatrix, DesignMatrixCollection

__all__ = ["RegressionCorrector"]


class RegressionCorrector(object):
    def __init__(self, lc):
        if np.any([~np.isfinite(lc.time), ~np.isfinite(lc.flux)]):
            raise ValueError("Input light curve has NaNs in time or flux. "
                             "Please remove NaNs before correction (e.g. using `lc = lc.remove_nans()`).")
        if np.any(~np.isfinite(lc.flux_err)) and not np.all(~np.isfinite(lc.flux_err)):
            raise ValueError("Input light curve has NaNs in `flux_err`. "
                             "Please remove NaNs before correction (e.g. using `lc = lc.remove_nans()`).")
        if np.any(lc.flux_err[np.isfinite(lc.flux_err)] <= 0):
            raise ValueError("Input light curve contains flux uncertainties smaller than or equal to zero. "
                             "Please remove these (e.g. using `lc = lc[lc.flux_err > 0]`).")
        self.lc = lc
        self.design_matrix_collection = None
        self.coefficients = None
        self.corrected_lc = None
        self.model_lc = None
        self.diagnostic_lightcurves = None

    def __repr__(self):
        return "RegressionCorrector (ID: {})".format(self.lc.targetid)

    @property
    def dmc(self):
        return self.design_matrix_collection

    def correct(self, design_matrix_collection, cadence_mask=None, sigma=5, niters=5, propagate_errors=False,
                device=0):
        """Fit and subtract the best linear combination of the regressors (Gaussian priors, iterative 5-sigma
        clipping).  The whole loop (Gram on the fp64 matrix cores, solve, residuals, clipping) runs on the GPU."""
        if not isinstance(design_matrix_collection, DesignMatrixCollection):
            if not isinstance(design_matrix_collection, DesignMatrix):
                raise ValueError("design_matrix_collection must be a DesignMatrix or DesignMatrixCollection")
            design_matrix_collection = DesignMatrixCollection([design_matrix_collection])
        design_matrix_collection.validate()
        self.design_matrix_collection = design_matrix_collection
        n = len(self.lc.time)
        if self.dmc.X.shape[0] != n:
            raise ValueError("the design matrix must have one row per cadence")
        self.cadence_mask = np.ones(n, bool) if cadence_mask is None else np.asarray(cadence_mask, dtype=bool)
        err = None if np.all(~np.isfinite(self.lc.flux_err)) else self.lc.flux_err
        has_prior = np.any(np.isfinite(self.dmc.prior_sigma)) or np.any(self.dmc.prior_mu != 0)
        res = _capi.regress_batch(
            self.dmc.X, self.lc.flux, [0, n], err=err, cadence_mask=self.cadence_mask,
            prior_mu=self.dmc.prior_mu if has_prior else None,
            prior_sigma=self.dmc.prior_sigma if has_prior else None, sigma=sigma, niters=niters, device=device,
            return_cov=bool(propagate_errors))
        self.coefficients = res["coefficients"][0]
        self.outlier_mask = res["outlier_mask"]
        if propagate_errors:
            # reference :183-185 keeps inv(X^T S^-1 X + diag(1/sigma_p^2)) of the last fit (computed on the GPU here) and
            # :280-298 turns it into a model uncertainty by drawing 100 coefficient vectors from N(w, cov) with the
            # global numpy RNG: same draws, same percentiles (host glue over K x K and N x 100 arrays, as in the reference)
            self.coefficients_err = res["coefficients_cov"][0]
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                samples = np.asarray([self.dmc.X.dot(np.random.multivariate_normal(self.coefficients, self.coefficients_err))
                                      for _ in range(100)]).T
            model_err = np.abs(np.percentile(samples, [16, 84], axis=1) - np.median(samples, axis=1)[:, None].T).mean(axis=0)
        else:
            self.coefficients_err = np.zeros(len(self.coefficients)) * np.nan
            model_err = np.zeros(n)
        self.model_lc = LightCurve(time=self.lc.time, flux=res["model"], flux_err=model_err, meta=self.lc.meta)
        self.corrected_lc = self.lc.copy()
        self.corrected_lc.flux = self.lc.flux - self.model_lc.flux
        self.corrected_lc.flux_err = (self.lc.flux_err ** 2 + self.model_lc.flux_err ** 2) ** 0.5
        self.diagnostic_lightcurves = self._create_diagnostic_lightcurves()
        return self.corrected_lc

    def _create_diagnostic_lightcurves(self):
        """One model light curve per sub-matrix (reference :311-342)."""
        if self.coefficients is None:
            raise ValueError("you need to call `correct()` first")
        lcs, first = {}, 0
        for sub in self.dmc.matrices:
            k = sub.shape[1]
            lcs[sub.name] = LightCurve(time=self.lc.time, flux=sub.X.dot(self.coefficients[first:first + k]),
                                       flux_err=np.zeros(len(self.lc.time)), label=sub.name)
            first += k
        return lcs
