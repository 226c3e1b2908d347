"""Ridge-prior CBV correction over the GPU regression path.

Reference: src/lightkurve/correctors/cbvcorrector.py:221-292 ``CBVCorrector.correct_gaussian_prior`` with
``_correct_initialization`` (:639-757) and ``_set_prior_width`` (:759-778): the design-matrix collection is
[selected cotrending basis vectors, optional extra design matrix, a constant column], every column gets the same
Gaussian prior width ``sigma = median(flux_err) / sqrt(|alpha|)`` (``alpha == 0`` -> no prior), and the fit is
``RegressionCorrector.correct``.  Reading CBV FITS files / MAST is the reference's control plane and stays there: the
basis vectors are handed in as an array interpolated to the light curve's cadences (what ``cbvs.to_designmatrix``
yields).

``correct`` (:397-500) is the goodness-metric optimisation: a bounded Brent search over the ridge penalty ``alpha`` of
``_goodness_metric_obj_fun`` (:781-854), each evaluation = one ridge fit (the regression kernels) + the over-fitting
metric (``correctors/metrics.py``: batched default Lomb-Scargle periodograms on the FFT path) and, if neighbour light
curves are supplied, the under-fitting metric.  ``goodness_scan`` evaluates the same objective for MANY alphas in two
GPU calls (one batched regression, one batched periodogram) — what a coarse pre-scan or a plot of the metrics against
alpha needs.  The search driver restates scipy.optimize.minimize_scalar(method='Bounded') (scipy
optimize/_optimize.py:_minimize_scalar_bounded), which is what the reference calls.
"""
import numpy as np

from .. import _capi
from .designmatrix import DesignMatrix, DesignMatrixCollection
from .metrics import overfit_metric_lombscargle, overfit_metric_lombscargle_batch, underfit_metric_neighbors
from .regressioncorrector import RegressionCorrector

__all__ = ["CBVCorrector", "minimize_scalar_bounded"]


def minimize_scalar_bounded(func, bounds, xatol=1e-5, maxiter=500):
    """scipy.optimize.minimize_scalar(func, method='Bounded', bounds=bounds, options={'maxiter': maxiter}) restated
    (Brent's fminbound: golden-section steps with parabolic interpolation; scipy optimize/_optimize.py
    ``_minimize_scalar_bounded``) so the optimisation needs no scipy in the product interpreter.  Given the same
    objective values it visits the same abscissae.  Returns dict(x, fun, nfev, status)."""
    x1, x2 = float(bounds[0]), float(bounds[1])
    if not (np.isfinite(x1) and np.isfinite(x2)):
        raise ValueError("Optimization bounds must be finite scalars.")
    if x1 > x2:
        raise ValueError("The lower bound exceeds the upper bound.")
    flag = 0
    sqrt_eps = np.sqrt(2.2e-16)
    golden_mean = 0.5 * (3.0 - np.sqrt(5.0))
    a, b = x1, x2
    fulc = a + golden_mean * (b - a)
    nfc, xf = fulc, fulc
    rat = e = 0.0
    x = xf
    fx = func(x)
    num = 1
    fu = np.inf
    ffulc = fnfc = fx
    xm = 0.5 * (a + b)
    tol1 = sqrt_eps * np.abs(xf) + xatol / 3.0
    tol2 = 2.0 * tol1
    while np.abs(xf - xm) > (tol2 - 0.5 * (b - a)):
        golden = 1
        if np.abs(e) > tol1:                      # try a parabolic step
            golden = 0
            r = (xf - nfc) * (fx - ffulc)
            q = (xf - fulc) * (fx - fnfc)
            p = (xf - fulc) * q - (xf - nfc) * r
            q = 2.0 * (q - r)
            if q > 0.0:
                p = -p
            q = np.abs(q)
            r = e
            e = rat
            if (np.abs(p) < np.abs(0.5 * q * r)) and (p > q * (a - xf)) and (p < q * (b - xf)):
                rat = (p + 0.0) / q
                x = xf + rat
                if ((x - a) < tol2) or ((b - x) < tol2):
                    si = np.sign(xm - xf) + ((xm - xf) == 0)
                    rat = tol1 * si
            else:
                golden = 1
        if golden:
            e = a - xf if xf >= xm else b - xf
            rat = golden_mean * e
        si = np.sign(rat) + (rat == 0)
        x = xf + si * np.maximum(np.abs(rat), tol1)
        fu = func(x)
        num += 1
        if fu <= fx:
            if x >= xf:
                a = xf
            else:
                b = xf
            fulc, ffulc = nfc, fnfc
            nfc, fnfc = xf, fx
            xf, fx = x, fu
        else:
            if x < xf:
                a = x
            else:
                b = x
            if (fu <= fnfc) or (nfc == xf):
                fulc, ffulc = nfc, fnfc
                nfc, fnfc = x, fu
            elif (fu <= ffulc) or (fulc == xf) or (fulc == nfc):
                fulc, ffulc = x, fu
        xm = 0.5 * (a + b)
        tol1 = sqrt_eps * np.abs(xf) + xatol / 3.0
        tol2 = 2.0 * tol1
        if num >= maxiter:
            flag = 1
            break
    if np.isnan(xf) or np.isnan(fx) or np.isnan(fu):
        flag = 2
    return dict(x=float(xf), fun=float(fx), nfev=num, status=flag)


class CBVCorrector(RegressionCorrector):
    def __init__(self, lc, cbvs, cbv_type="SingleScale"):
        """``cbvs``: float array (n_cadences, n_vectors), column j = basis vector number j + 1 on the cadences of lc."""
        super(CBVCorrector, self).__init__(lc)
        self.cbvs = np.asarray(cbvs, dtype=np.float64)
        if self.cbvs.ndim != 2 or self.cbvs.shape[0] != len(lc.time):
            raise ValueError("cbvs must have one row per cadence of the light curve")
        self.cbv_type = cbv_type
        self.alpha = None
        self.neighbor_flux = None          # (n_cadences, n_neighbours) for the under-fitting metric, see set_neighbors
        self.optimization_params = None
        self.over_fitting_score = None
        self.under_fitting_score = None

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
        return self.correct_regressioncorrector(dmc, cadence_mask=cadence_mask, **kwargs)

    def correct_regressioncorrector(self, design_matrix_collection, **kwargs):
        """Pass-through to ``RegressionCorrector.correct`` (reference cbvcorrector.py:494-500); ``CBVCorrector.correct``
        itself is the goodness-metric optimisation below."""
        return super(CBVCorrector, self).correct(design_matrix_collection, **kwargs)

    # ------------------------------------------------------------------------------------------------ goodness metrics
    def set_neighbors(self, neighbor_flux):
        """Flux of the neighbouring targets on this light curve's cadences, shape (n_cadences, n_neighbours).  The
        reference downloads them from MAST inside ``under_fitting_metric`` (metrics.py:274-330, control plane); here the
        caller hands them over once."""
        nf = np.asarray(neighbor_flux, dtype=np.float64)
        if nf.ndim != 2 or nf.shape[0] != len(self.lc.time):
            raise ValueError("neighbor_flux must be (n_cadences, n_neighbours)")
        self.neighbor_flux = nf

    def over_fitting_metric(self, n_samples=10, device=0):
        """cbvcorrector.py:502-533 -> metrics.overfit_metric_lombscargle on the unmasked cadences."""
        if self.corrected_lc is None:
            return None
        cm = self.cadence_mask
        return overfit_metric_lombscargle(self.lc[cm], self.corrected_lc[cm], n_samples=n_samples, device=device)

    def under_fitting_metric(self):
        """cbvcorrector.py:535-637 -> metrics.underfit_metric_neighbors; needs ``set_neighbors`` first."""
        if self.corrected_lc is None:
            raise Exception("A corrected light curve does not exist, please run correct first")
        if self.neighbor_flux is None:
            raise ValueError("under_fitting_metric needs the neighbouring targets: call set_neighbors(flux) first "
                             "(the reference downloads them from MAST)")
        cm = self.cadence_mask
        return underfit_metric_neighbors(self.corrected_lc[cm], self.neighbor_flux[cm])

    def _goodness_metric_obj_fun(self, alpha, device=0):
        """cbvcorrector.py:781-854: ridge fit at ``alpha``, then -(over + under) with the leaky-ReLU easing above the
        target scores."""
        op = self.optimization_params
        sigma = np.median(self.lc.flux_err) / np.sqrt(np.abs(alpha))
        for dm in self._dmc_opt.matrices:
            dm.prior_sigma = np.ones(dm.shape[1]) * sigma
        self.correct_regressioncorrector(self._dmc_opt, cadence_mask=op["cadence_mask"], device=device)
        over = self.over_fitting_metric(n_samples=op["over_metric_nSamples"], device=device) \
            if op["target_over_score"] > 0 else 1.0
        under = self.under_fitting_metric() if op["target_under_score"] > 0 else 1.0
        return -(_leaky(over, op["target_over_score"]) + _leaky(under, op["target_under_score"]))

    def correct(self, cbv_type=None, cbv_indices=np.arange(1, 9), ext_dm=None, cadence_mask=None,
                alpha_bounds=(1e-4, 1e4), target_over_score=0.5, target_under_score=0.5, max_iter=100, device=0):
        """``CBVCorrector.correct`` (reference cbvcorrector.py:397-500): bounded Brent search for the ridge penalty alpha
        that maximises the goodness metrics, a final fit at the optimum, then the scores with n_samples=10.
        ``cbv_type`` is accepted for signature compatibility (this mirror holds ONE set of basis vectors);
        ``cbv_indices=None`` fits ``ext_dm`` only, like the reference with ``do_not_load_cbvs``.  A score with a
        non-positive target is skipped (and reported as -1), as in the reference."""
        self._dmc_opt = self._collection(cbv_indices, ext_dm)
        self.optimization_params = {"alpha_bounds": list(alpha_bounds), "target_over_score": target_over_score,
                                    "target_under_score": target_under_score, "max_iter": max_iter,
                                    "cadence_mask": cadence_mask, "over_metric_nSamples": 1}
        res = minimize_scalar_bounded(lambda a: self._goodness_metric_obj_fun(a, device=device), alpha_bounds,
                                      maxiter=max_iter)
        self._goodness_metric_obj_fun(res["x"], device=device)          # minimize_scalar does not end on the optimum
        self.over_fitting_score = self.over_fitting_metric(n_samples=10, device=device) if target_over_score > 0 else -1.0
        self.under_fitting_score = self.under_fitting_metric() if target_under_score > 0 else -1.0
        self.alpha = res["x"]
        self.optimization_result = res
        return self.corrected_lc

    def goodness_scan(self, alphas, cbv_indices=np.arange(1, 9), ext_dm=None, cadence_mask=None, n_samples=1,
                      device=0):
        """The goodness objective's ingredients for MANY ridge penalties at once: ONE batched regression (the same
        light curve and design matrix with len(alphas) different prior widths) and ONE batched periodogram call for all
        the over-fitting metrics.  Returns dict(alpha, over_fitting[len(alphas)], under_fitting (or None),
        corrected_flux[len(alphas), n_cadences]).  The noise draws follow the reference's order per alpha."""
        alphas = np.atleast_1d(np.asarray(alphas, dtype=np.float64))
        dmc = self._collection(cbv_indices, ext_dm)
        n, K, A = len(self.lc.time), dmc.X.shape[1], len(alphas)
        cm = np.ones(n, bool) if cadence_mask is None else np.asarray(cadence_mask, dtype=bool)
        sig = np.median(self.lc.flux_err) / np.sqrt(np.abs(alphas))
        err = None if np.all(~np.isfinite(self.lc.flux_err)) else np.tile(self.lc.flux_err, A)
        res = _capi.regress_batch(np.tile(dmc.X, (A, 1)), np.tile(self.lc.flux, A), np.arange(A + 1) * n, err=err,
                                  cadence_mask=np.tile(cm, A), prior_mu=np.zeros((A, K)),
                                  prior_sigma=np.repeat(sig[:, None], K, axis=1), device=device)
        corrected = self.lc.flux[None, :] - res["model"].reshape(A, n)
        orig = self.lc[cm]
        cors = []
        for a in range(A):
            c = self.lc.copy()
            c.flux = corrected[a]
            cors.append(c[cm])
        over = overfit_metric_lombscargle_batch(orig, cors, n_samples=n_samples, device=device)
        under = None
        if self.neighbor_flux is not None:
            under = np.array([underfit_metric_neighbors(c, self.neighbor_flux[cm]) for c in cors])
        return dict(alpha=alphas, over_fitting=over, under_fitting=under, corrected_flux=corrected,
                    coefficients=res["coefficients"])


def _leaky(metric, target, leak=0.01):
    """cbvcorrector.py:838-850: above the target the metric only counts with 1 % of its excess."""
    if target > 0 and metric >= target:
        return target + leak * (metric - target)
    return metric
