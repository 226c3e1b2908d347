"""Ridge-prior CBV correction over the GPU regression path.

Reference: src/lightkurve/correctors/cbvcorrector.py:221-292 ``CBVCorrector.correct_gaussian_prior`` with
``_correct_initialization`` (:639-757) and ``_set_prior_width`` (:759-778): the design-matrix collection is
[selected cotrending basis vectors, optional extra design matrix, a constant column], every column gets the same
Gaussian prior width ``sigma = median(flux_err) / sqrt(|alpha|)`` (``alpha == 0`` -> no prior), and the fit is
``RegressionCorrector.correct``.  Reading CBV FITS files / MAST is the reference's control plane and stays there: the
basis vectors are handed in as an array interpolated to the light curve's cadences (what ``cbvs.to_designmatrix``
yields).
"""
import numpy as np

from .designmatrix import DesignMatrix, DesignMatrixCollection
from .regressioncorrector import RegressionCorrector

__all__ = ["CBVCorrector"]


class CBVCorrector(RegressionCorrector):
    def __init__(self, lc, cbvs, cbv_type="SingleScale"):
        """``cbvs``: float array (n_cadences, n_vectors), column j = basis vector number j + 1 on the cadences of lc."""
        super(CBVCorrector, self).__init__(lc)
        self.cbvs = np.asarray(cbvs, dtype=np.float64)
        if self.cbvs.ndim != 2 or self.cbvs.shape[0] != len(lc.time):
            raise ValueError("cbvs must have one row per cadence of the light curve")
        self.cbv_type = cbv_type
        self.alpha = None

    def _collection(self, cbv_indices, ext_dm):
        mats = []
        if cbv_indices is not None:
            if isinstance(cbv_indices, str) and cbv_indices == "ALL":
                cbv_indices = np.arange(1, self.cbvs.shape[1] + 1)
            idx = np.array([i for i in np.asarray(cbv_indices) if 1 <= i <= self.cbvs.shape[1]], dtype=int)  # 1-based
            mats.append(DesignMatrix(self.cbvs[:, idx - 1], columns=["VECTOR_%d" % i for i in idx], name=self.cbv_type))
        if ext_dm is not None:
            if not isinstance(ext_dm, DesignMatrix):
                raise ValueError("ext_dm must be a DesignMatrix")
            if ext_dm.shape[0] != len(self.lc.flux):
                raise ValueError("ext_dm must contain the same number of cadences as lc.flux")
            mats.append(ext_dm)
        if not mats:
            raise ValueError("nothing to fit: pass cbv_indices and/or ext_dm")
        mats.append(DesignMatrix(np.ones(len(self.lc.time)), columns=["Constant"], name="Constant"))
        return DesignMatrixCollection(mats)

    def correct_gaussian_prior(self, cbv_indices=np.arange(1, 9), alpha=1e-20, ext_dm=None, cadence_mask=None, **kwargs):
        dmc = self._collection(cbv_indices, ext_dm)
        sigma = None if alpha == 0.0 else np.median(self.lc.flux_err) / np.sqrt(np.abs(alpha))
        for dm in dmc.matrices:
            dm.prior_sigma = np.ones(dm.shape[1]) * (np.inf if sigma is None else sigma)
        self.alpha = alpha
        return self.correct(dmc, cadence_mask=cadence_mask, **kwargs)
