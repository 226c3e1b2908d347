"""Install the HIP kernels behind the reference's own seams, so that an UNMODIFIED lightkurve —
``lc.to_periodogram()``, ``lc.to_periodogram(method="bls")``, ``lc.flatten()``, ``RegressionCorrector.correct``,
``PLDCorrector.correct`` — returns its own LightCurve / Periodogram objects with the arithmetic done in liblkhip.so.

* S1  astropy Lomb-Scargle method registry (reached from src/lightkurve/periodogram.py:961-964): EVERY name of the registry
      (implementations/main.py:20-25) is replaced — ``'fast'`` (lightkurve's default, :650), ``'fastchi2'``, ``'chi2'``, and
      the exact single-term names ``'slow'`` (what lightkurve itself rewrites ``'fast'`` to for every grid that is not
      regular in frequency, i.e. every ``period=`` request: periodogram.py:933-946), ``'cython'`` and ``'scipy'`` —
      and ``'hip'`` (exact direct sums) is added; ``'auto'`` only picks among those names (main.py:79-106), so it lands
      on the GPU too.  astropy's own implementations stay reachable as ``'<name>_cpu'`` and are called only for
      ``nterms`` > 8.  A ``'fast'`` / ``'fastchi2'`` call with ``use_fft=False`` IS the exact trig sums (fast_impl.py / utils.py:154-156) and
      runs the exact kernels; so does one with ``Mfft`` != 4 or ``nterms`` 5..8 (the FFT kernels extirpolate with astropy's
      default Mfft = 4 and are instantiated for <= 4 terms): the result is then the sums astropy's approximation converges
      to, not its approximation error.
* S2  ``astropy.timeseries.periodograms.bls.methods.bls_fast`` (reached from periodogram.py:1169), every period on the device.
* S3  ``lightkurve.lightcurve.LightCurve.flatten`` (lightcurve.py:943-1078): the mask / segmentation / savgol / clip /
      interpolation loop (:996-1063) is ONE call of lk_savgol_trend_batch; the object handling around it stays lightkurve's.
      Calls the kernel does not cover (extra ``savgol_filter`` keyword arguments, unsorted time) go to the original method.
* S4  ``RegressionCorrector._fit_coefficients`` (regressioncorrector.py:127-189; the sigma-clip loop around it stays the
      reference's) and — faster, ``full_loop=True`` — ``RegressionCorrector.correct`` itself (:191-309, all iterations in one
      lk_regress_cov_batch call); ``PLDCorrector.create_design_matrix`` (pldcorrector.py:125-287) builds its PCA blocks with
      lk_pld_design_batch (``sparse=True`` included: same PCA blocks, the reference's sparse spline basis beside them);
      ``DesignMatrix.pca`` / ``.standardize`` (designmatrix.py:215-282) go to lk_pca_batch / lk_standardize_batch for every
      caller; ``create_spline_matrix`` / ``create_sparse_spline_matrix`` (designmatrix.py:896-997; PLD, SFF, user code) evaluate
      their B-spline bases with lk_spline_basis_batch instead of patsy / the Python recursion.  Sparse collections are
      densified; more than 4095 regressors, ``pca`` beyond 48 terms or a sparse collection that would not fit densified go to
      the original methods, and so does ``pca_components=0`` — there the reference itself calls ``fbpca.pca(k=0)`` for the
      background block (pldcorrector.py:223), which fbpca rejects: the seam hands the call over unchanged.
      Every call handed back to a CPU implementation logs one ``log.debug`` line saying which seam and why.
* §8(f)  ``Periodogram.smooth`` (periodogram.py:182-284, hence ``Periodogram.flatten`` :381-429) -> lk_pg_boxsmooth_batch /
      lk_pg_logmedian_batch; ``LightCurve.estimate_cdpp`` (lightcurve.py:1764-1833) -> lk_savgol_trend_batch +
      lk_sigma_clip_batch; ``overfit_metric_lombscargle`` (correctors/metrics.py:23-124; CBVCorrector's goodness scan calls it
      per trial) -> its n_samples noise periodograms in one lk_ls_fast_batch call.

``backend`` is the module that provides the compute entry points (default: ``lightkurve_amd._capi``, i.e. the GPU).  The
tests pass a stand-in to check the wiring against a real lightkurve on a machine without a GPU (tests/test_seams_cpu.py);
nothing in the package itself ever substitutes a CPU implementation.

astropy / lightkurve are NOT importable in the product interpreter of this image; on the GPU box the seams are exercised
under the conda interpreter that ships astropy 4.3.1 (tests/test_seams_gpu.py).
"""
import logging

import numpy as np

from . import _capi

log = logging.getLogger(__name__)

_BACKEND = _capi
_ORIG = {}

# limits of lk_regress_cov_batch (include/lkhip.h) and a budget for densified sparse collections
_MAX_REGRESSORS = 4095
_DENSE_BUDGET_BYTES = 8 << 30


def _fell_back(seam, why):
    """Every call a seam hands back to the original CPU implementation says so (same channel and level the reference
    uses for its own notes, e.g. regressioncorrector.py:270-273)."""
    log.debug("lightkurve_amd: %s fell back to the CPU implementation: %s", seam, why)


def _be():
    return _BACKEND


# ------------------------------------------------------------------------------------------------ S1
def _split_normalization(normalization):
    """astropy's four Lomb-Scargle normalisations (fast_impl.py:124-135, chi2_impl.py:74-84, slow_impl.py:98-109) on top of the
    two the kernels produce: with p the 'standard' power, 'log' is -log(1 - p) and 'model' is p / (1 - p) — element-wise
    maps of the B x M result, applied here (lightkurve itself only ever asks for 'psd', periodogram.py:964-967)."""
    if normalization in ("standard", "psd"):
        return normalization, (lambda p: p)
    if normalization == "log":
        return "standard", (lambda p: -np.log(1.0 - p))
    if normalization == "model":
        return "standard", (lambda p: p / (1.0 - p))
    raise ValueError("normalization='{}' not recognized".format(normalization))


def lombscargle_hip(t, y, dy=None, frequency=None, normalization="standard", fit_mean=True, center_data=True,
                    nterms=1, **unused):
    """Signature of astropy's METHODS entries (lombscargle/implementations/main.py:182-217)."""
    if nterms > _capi.MAX_NTERMS and "chi2" in _ORIG:     # beyond the instantiated kernels: astropy's own chi2
        _fell_back("lombscargle METHODS['chi2']", "nterms=%d > %d" % (nterms, _capi.MAX_NTERMS))
        return _ORIG["chi2"](t, y, dy, frequency=frequency, normalization=normalization, fit_mean=fit_mean,
                             center_data=center_data, nterms=nterms)
    if not 1 <= nterms <= _capi.MAX_NTERMS:
        raise ValueError("the HIP kernels are instantiated for 1 <= nterms <= %d" % _capi.MAX_NTERMS)
    norm, finish = _split_normalization(normalization)
    t = np.asarray(t, dtype=np.float64)
    frequency = np.asarray(frequency, dtype=np.float64)
    from .periodogram import exact_grid
    grid = exact_grid(frequency)
    kw = dict(dy=dy, fit_mean=fit_mean, center_data=center_data, normalization=norm, nterms=nterms)
    if grid is not None:
        return finish(_be().ls_power_batch(t, y, [0, len(t)], f0=grid[0], df=grid[1], M=len(frequency), **kw)[0])
    return finish(_be().ls_power_batch(t, y, [0, len(t)], frequency=frequency, **kw)[0])


def lombscargle_scipy_hip(t, y, frequency, normalization="standard", center_data=True):
    """Signature of astropy's lombscargle_scipy (scipy_impl.py:4-5): main.py:194-202 strips ``dy`` and ``fit_mean`` (which
    must be False) before it calls METHODS['scipy'].  ``scipy.signal.lombscargle`` of the optionally centred data is the
    classical periodogram — the exact kernel with unit weights and no floating mean; scipy_impl.py:59-66's four
    normalisations are the same maps of it as everywhere else."""
    return lombscargle_hip(t, y, None, frequency=frequency, normalization=normalization, fit_mean=False,
                           center_data=center_data)


def lombscargle_fast_hip(t, y, dy=None, f0=0, df=None, Nf=None, center_data=True, fit_mean=True,
                         normalization="standard", use_fft=True, trig_sum_kwds=None, nterms=1, **unused):
    """Signature of astropy's lombscargle_fast / lombscargle_fastchi2 (fast_impl.py:6, fastchi2_impl.py:8): the
    f0/df/Nf form every 'fast*' method receives; ``nterms`` > 1 arrives only under the name 'fastchi2'."""
    kw = dict(trig_sum_kwds or {})
    if nterms > _capi.MAX_NTERMS and ("fastchi2" if nterms > 1 else "fast") in _ORIG:
        name = "fastchi2" if nterms > 1 else "fast"
        _fell_back("lombscargle METHODS['%s']" % name, "nterms=%d > %d" % (nterms, _capi.MAX_NTERMS))
        return _ORIG[name](t, y, dy, f0=f0, df=df, Nf=Nf, center_data=center_data, fit_mean=fit_mean,
                           normalization=normalization, use_fft=use_fft, trig_sum_kwds=trig_sum_kwds, nterms=nterms)
    exact = (not use_fft) or int(kw.get("Mfft", 4)) != 4 or nterms > 4
    if not 1 <= nterms <= _capi.MAX_NTERMS:
        raise ValueError("the HIP kernels are instantiated for 1 <= nterms <= %d" % _capi.MAX_NTERMS)
    norm, finish = _split_normalization(normalization)
    if f0 < 0:
        raise ValueError("Frequencies must be positive")
    if df <= 0:
        raise ValueError("Frequency steps must be positive")
    if Nf <= 0:
        raise ValueError("Number of frequencies must be positive")
    t = np.asarray(t, dtype=np.float64)
    if exact:
        # use_fft=False: astropy's trig_sum then forms the exact sums (utils.py:154-156) — what ls_grid_kernel computes; a
        # request for a wider extirpolation stencil or more than four terms gets the same exact sums
        return finish(_be().ls_power_batch(t, y, [0, len(t)], dy=dy, f0=float(f0), df=float(df), M=int(Nf), fit_mean=fit_mean,
                                           center_data=center_data, normalization=norm, nterms=nterms)[0])
    return finish(_be().ls_fast_batch(t, y, [0, len(t)], dy=dy, f0=float(f0), df=float(df), M=int(Nf), fit_mean=fit_mean,
                                      center_data=center_data, normalization=norm,
                                      oversampling=int(kw.get("oversampling", 5)), nterms=nterms)[0])


# ------------------------------------------------------------------------------------------------ S2
def bls_fast_hip(t, y, ivar, period, duration, oversample, use_likelihood):
    """Signature of astropy's methods.bls_fast (bls/methods.py:55-95).  Every period runs on the device: those whose phase
    bins fit LDS in the team kernels, longer ones (a multi-year baseline searched with short durations) in the
    global-memory kernel — same bits either way (round 5 merged the long ones from astropy's CPU implementation)."""
    res = _be().bls_batch(t, y, ivar, [0, len(t)], np.ascontiguousarray(period, dtype=np.float64), duration, oversample,
                          use_likelihood)
    return tuple(res[k][0] for k in _capi.BLS_FIELDS)


# ------------------------------------------------------------------------------------------------ S3
def flatten_trend_hip(time, flux, window_length=101, polyorder=2, break_tolerance=5, niters=3, sigma=3, mask=None):
    """The trend the loop of LightCurve.flatten (lightcurve.py:996-1063) ends with, for plain arrays.
    ``mask``: True = cadence excluded from the fit (lightkurve's ``mask=`` semantics)."""
    t = np.ascontiguousarray(time, dtype=np.float64)
    f = np.ascontiguousarray(flux, dtype=np.float64)
    return _be().savgol_trend_batch(t, f, [0, len(t)], mask=mask, window_length=window_length, polyorder=polyorder,
                                    break_tolerance=break_tolerance, niters=niters, sigma=sigma)


def _plain(x):
    """ndarray of a Quantity / masked Quantity / ndarray (masked entries -> NaN)."""
    x = getattr(x, "unmasked", x)
    return np.asarray(getattr(x, "value", x), dtype=np.float64)


def _make_flatten(lk_lightcurve_mod):
    orig = lk_lightcurve_mod.LightCurve.flatten
    Quantity = lk_lightcurve_mod.Quantity

    def flatten(self, window_length=101, polyorder=2, return_trend=False, break_tolerance=5, niters=3, sigma=3,
                mask=None, **kwargs):
        time = _plain(self.time.value)
        if kwargs or (len(time) > 1 and np.any(np.diff(time) < 0)) or window_length % 2 != 1 or niters < 1:
            _fell_back("LightCurve.flatten", "extra savgol_filter keyword arguments %s" % sorted(kwargs) if kwargs else
                       ("time is not sorted" if window_length % 2 == 1 and niters >= 1 else
                        "window_length=%d, niters=%d" % (window_length, niters)))
            return orig(self, window_length=window_length, polyorder=polyorder, return_trend=return_trend,
                        break_tolerance=break_tolerance, niters=niters, sigma=sigma, mask=mask, **kwargs)
        if polyorder >= window_length:
            polyorder = window_length - 1
            lk_lightcurve_mod.log.warning("polyorder must be smaller than window_length, "
                                          "using polyorder={}.".format(polyorder))
        flux = self.flux
        fl = _plain(flux)
        if hasattr(flux, "mask"):                                  # astropy >= 5: masked entries are excluded cadences
            fl = np.where(np.asarray(flux.mask, dtype=bool), np.nan, fl)
        trend = flatten_trend_hip(time, fl, window_length=window_length, polyorder=polyorder,
                                  break_tolerance=break_tolerance, niters=niters, sigma=sigma,
                                  mask=None if mask is None else np.asarray(mask, dtype=bool))
        trend_signal = Quantity(trend, self.flux.unit)
        import warnings
        flatten_lc = self.copy()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            flatten_lc.flux = flatten_lc.flux / trend_signal
            flatten_lc.flux_err = flatten_lc.flux_err / trend_signal
        flatten_lc.meta["NORMALIZED"] = True
        if return_trend:
            trend_lc = self.copy()
            trend_lc.flux = trend_signal
            return flatten_lc, trend_lc
        return flatten_lc

    flatten.__doc__ = orig.__doc__
    return flatten


# ------------------------------------------------------------------------------------------------ S4
def _dense(X):
    return np.ascontiguousarray(X.toarray() if hasattr(X, "toarray") else X, dtype=np.float64)


def _regress(self, cadence_mask, prior_mu, prior_sigma, sigma, niters, want_cov):
    """One lk_regress_cov_batch call over this corrector's light curve and design-matrix collection."""
    X = _dense(self.dmc.X)
    n = X.shape[0]
    flux = _plain(self.lc.flux)
    ferr = _plain(self.lc.flux_err)
    err = None if np.all(~np.isfinite(ferr)) else ferr
    has_prior = prior_sigma is not None
    return _be().regress_batch(X, flux, [0, n], err=err, cadence_mask=np.asarray(cadence_mask, dtype=bool),
                               prior_mu=np.asarray(prior_mu, dtype=np.float64) if has_prior else None,
                               prior_sigma=np.asarray(prior_sigma, dtype=np.float64) if has_prior else None,
                               sigma=sigma, niters=niters, return_cov=want_cov)


def _outside_kernel_limits(dmc):
    """Why this design-matrix collection cannot go through lk_regress_cov_batch, or None."""
    n, k = dmc.X.shape
    if k > _MAX_REGRESSORS:
        return "%d regressors > %d" % (k, _MAX_REGRESSORS)
    if hasattr(dmc.X, "toarray") and n * k * 8 > _DENSE_BUDGET_BYTES:
        return "sparse collection of %d x %d would densify to %.1f GB" % (n, k, n * k * 8 / 1e9)
    return None


def _fit_coefficients_hip(self, cadence_mask=None, prior_mu=None, prior_sigma=None, propagate_errors=False):
    """RegressionCorrector._fit_coefficients (regressioncorrector.py:127-189): one weighted ridge fit on the GPU."""
    if (prior_mu is None) != (prior_sigma is None):
        raise ValueError("Please specify both `prior_mu` and `prior_sigma`")
    why = _outside_kernel_limits(self.dmc)
    if why is not None and "_fit_coefficients" in _ORIG:
        _fell_back("RegressionCorrector._fit_coefficients", why)
        return _ORIG["_fit_coefficients"](self, cadence_mask=cadence_mask, prior_mu=prior_mu, prior_sigma=prior_sigma,
                                          propagate_errors=propagate_errors)
    if cadence_mask is None:
        cadence_mask = np.ones(len(self.lc.flux), bool)
    res = _regress(self, cadence_mask, prior_mu, prior_sigma, sigma=5.0, niters=1, want_cov=bool(propagate_errors))
    w = res["coefficients"][0]
    w_err = res["coefficients_cov"][0] if propagate_errors else np.zeros(len(w)) * np.nan
    return w, w_err


def _make_correct(lk_regcorr_mod):
    LightCurve = lk_regcorr_mod.LightCurve
    u = lk_regcorr_mod.u
    DMC = lk_regcorr_mod.DesignMatrixCollection
    DM = lk_regcorr_mod.DesignMatrix
    SDM = lk_regcorr_mod.SparseDesignMatrix
    SDMC = lk_regcorr_mod.SparseDesignMatrixCollection
    orig = lk_regcorr_mod.RegressionCorrector.correct

    def correct(self, design_matrix_collection, cadence_mask=None, sigma=5, niters=5, propagate_errors=False):
        if not isinstance(design_matrix_collection, DMC):
            if isinstance(design_matrix_collection, SDM):
                design_matrix_collection = SDMC([design_matrix_collection])
            elif isinstance(design_matrix_collection, DM):
                design_matrix_collection = DMC([design_matrix_collection])
        design_matrix_collection.validate()
        why = _outside_kernel_limits(design_matrix_collection)
        if why is not None:
            # the original loop; its _fit_coefficients calls fall back by the same test
            _fell_back("RegressionCorrector.correct", why)
            return orig(self, design_matrix_collection, cadence_mask=cadence_mask, sigma=sigma, niters=niters,
                        propagate_errors=propagate_errors)
        self.design_matrix_collection = design_matrix_collection
        n = len(self.lc.time)
        self.cadence_mask = np.ones(n, bool) if cadence_mask is None else cadence_mask
        res = _regress(self, self.cadence_mask, self.dmc.prior_mu, self.dmc.prior_sigma, sigma, niters,
                       bool(propagate_errors))                    # the whole clip loop, :243-279, in one call
        self.coefficients = res["coefficients"][0]
        self.outlier_mask = res["outlier_mask"]
        model_flux = res["model"]                                 # X.w - median(X.w)
        if propagate_errors:                                      # :280-298, the reference's own sampling on our covariance
            import warnings
            self.coefficients_err = res["coefficients_cov"][0]
            X = _dense(self.dmc.X)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                samples = np.asarray([X.dot(np.random.multivariate_normal(self.coefficients, self.coefficients_err))
                                      for _ in range(100)]).T
            model_err = np.abs(np.percentile(samples, [16, 84], axis=1) - np.median(samples, axis=1)[:, None].T).mean(axis=0)
        else:
            self.coefficients_err = np.zeros(len(self.coefficients)) * np.nan
            model_err = np.zeros(n)
        unit = self.lc.flux.unit
        self.model_lc = LightCurve(time=self.lc.time, flux=u.Quantity(model_flux, unit=unit),
                                   flux_err=u.Quantity(model_err, unit=unit))
        self.corrected_lc = self.lc.copy()
        self.corrected_lc.flux = self.lc.flux - self.model_lc.flux
        self.corrected_lc.flux_err = (self.lc.flux_err ** 2 + self.model_lc.flux_err ** 2) ** 0.5
        self.diagnostic_lightcurves = self._create_diagnostic_lightcurves()
        return self.corrected_lc

    correct.__doc__ = orig.__doc__
    return correct


def _make_create_design_matrix(lk_pld_mod):
    import pandas as pd
    orig = lk_pld_mod.PLDCorrector.create_design_matrix
    DM = lk_pld_mod.DesignMatrix
    DMC = lk_pld_mod.DesignMatrixCollection
    from .correctors.pldcorrector import _finite_columns, _percentile_knots

    def create_design_matrix(self, pld_order=3, pca_components=16, pld_aperture_mask=None,
                             background_aperture_mask="background", spline_n_knots=None, spline_degree=3,
                             normalize_background_pixels=None, sparse=False):
        if not pca_components or pca_components < 1:
            _fell_back("PLDCorrector.create_design_matrix", "pca_components=%r" % (pca_components,))
            return orig(self, pld_order=pld_order, pca_components=pca_components, pld_aperture_mask=pld_aperture_mask,
                        background_aperture_mask=background_aperture_mask, spline_n_knots=spline_n_knots,
                        spline_degree=spline_degree, normalize_background_pixels=normalize_background_pixels,
                        sparse=sparse)
        # None -> all pixels, as the reference's create_design_matrix does (pldcorrector.py:203-207; the mission-dependent
        # default is resolved by correct() before it gets here)
        self.pld_aperture_mask = self.tpf._parse_aperture_mask(pld_aperture_mask)
        self.background_aperture_mask = self.tpf._parse_aperture_mask(background_aperture_mask)
        n = len(self.lc)
        if spline_n_knots is None:
            spline_n_knots = int(n / 50)
        if normalize_background_pixels is None:
            normalize_background_pixels = False
        cube = np.asarray(self.tpf.flux.value, dtype=np.float32)
        pld_pix = _finite_columns(cube[:, self.pld_aperture_mask].reshape(n, -1))
        bkg_pix = _finite_columns(cube[:, self.background_aperture_mask].reshape(n, -1))
        time = np.asarray(self.lc.time.value, dtype=np.float64)
        lcf = np.asarray(self.lc.flux.value).astype(np.float32)
        knots = _percentile_knots(time, spline_n_knots, spline_degree)
        X, ps = _be().pld_design_batch(pld_pix[None] if pld_pix.shape[1] else None, bkg_pix[None], lcf[None], time[None],
                                       knots[None], pld_order, pca_components, spline_degree, normalize_background_pixels)
        X, ps = X[0], ps[0]
        nsp = spline_n_knots + 1
        kb = min(pca_components, bkg_pix.shape[1])
        npld = X.shape[1] - nsp - kb
        mats = []
        if npld > 0:
            mats.append(DM(pd.DataFrame(X[:, :npld]), name="pixel_series", prior_sigma=ps[:npld]))
        mats.append(DM(pd.DataFrame(X[:, npld:npld + kb]), name="background", prior_sigma=ps[npld:npld + kb]))
        import warnings
        if sparse:
            # reference :194-199, 226-230: the sparse collection carries the reference's OTHER spline basis
            # (create_sparse_spline_matrix: its own knots, the same kernel once the spline seam is installed); the pixel and background blocks are the same
            # PCA'd matrices, built on the GPU above
            sp = lk_pld_mod.create_sparse_spline_matrix(time, n_knots=spline_n_knots, degree=spline_degree).append_constant()
            sp.prior_sigma = np.ones(sp.shape[1]) * ps[-1]
            mats.append(sp)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                return lk_pld_mod.SparseDesignMatrixCollection(mats)
        cols = ["knot{}".format(i + 1) for i in range(spline_n_knots)] + ["offset"]
        mats.append(DM(pd.DataFrame(X[:, npld + kb:], columns=cols), name="spline", prior_sigma=ps[npld + kb:]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return DMC(mats)

    create_design_matrix.__doc__ = orig.__doc__
    return create_design_matrix


def _make_dm_pca(lk_dm_mod):
    """``DesignMatrix.pca`` (correctors/designmatrix.py:252-282) -> lk_pca_batch.  Whoever calls it — PLD through the
    reference's own create_design_matrix, CBV ``ext_dm``, SFF, user code — gets the GPU's Gram + eigen-solver instead of
    fbpca on the CPU."""
    orig = lk_dm_mod.DesignMatrix.pca

    def pca(self, nterms=6, n_iter=10):
        if nterms > self.shape[1]:
            nterms = self.shape[1]
        if nterms < 1 or nterms > 48 or self.shape[1] > 4096 or self.shape[0] < 2:
            _fell_back("DesignMatrix.pca", "nterms=%d of %d columns is outside the kernel's range (1..48 of <= 4096)"
                       % (nterms, self.shape[1]))
            return orig(self, nterms=nterms, n_iter=n_iter)
        vals = np.ascontiguousarray(np.asarray(self.values, dtype=np.float64))
        return lk_dm_mod.DesignMatrix(_be().pca_batch(vals, nterms), name=self.name)

    pca.__doc__ = orig.__doc__
    return pca


def _make_dm_standardize(lk_dm_mod):
    import pandas as pd
    orig = lk_dm_mod.DesignMatrix.standardize

    def standardize(self, inplace=False):
        vals = _be().standardize_batch(np.ascontiguousarray(np.asarray(self.df, dtype=np.float64)))
        dm = self if inplace else self.copy()
        dm.df = pd.DataFrame(vals, columns=self.columns)
        return dm

    standardize.__doc__ = orig.__doc__
    return standardize


def _make_spline_builders(lk_dm_mod):
    """``create_spline_matrix`` (designmatrix.py:952-997: patsy ``bs(x, df | knots, degree, include_intercept) - 1``) and
    ``create_sparse_spline_matrix`` (:896-949: a Python Cox-de Boor recursion) build the SAME kind of basis — clamped
    B-splines on [min x, interior knots, max x] — and differ only in where they put the interior knots (equally spaced
    percentiles of x / mid-points between the samples that end equal chunks of the sorted x).  Knot placement stays the
    reference's arithmetic on the host; the N x n_knots basis is lk_spline_basis_batch (agrees with patsy to 1e-12 and with
    the recursion to 4e-16, tests/test_designmatrix_gpu.py, tests/seams_lk_worker.py)."""
    import pandas as pd
    orig_dense, orig_sparse = lk_dm_mod.create_spline_matrix, lk_dm_mod.create_sparse_spline_matrix

    def create_spline_matrix(x, n_knots=20, knots=None, degree=3, name="spline", include_intercept=True):
        xx = np.asarray(getattr(x, "value", x), dtype=np.float64)
        if xx.ndim != 1 or len(xx) < 2 or not np.all(np.isfinite(xx)) or not 0 <= int(degree) <= 7:
            _fell_back("create_spline_matrix", "x / degree outside the kernel's range")
            return orig_dense(x, n_knots=n_knots, knots=knots, degree=degree, name=name, include_intercept=include_intercept)
        order = int(degree) + 1
        if knots is not None:
            inner = np.sort(np.asarray(knots, dtype=np.float64))
        else:
            n_inner = int(n_knots) - order + (0 if include_intercept else 1)
            if n_inner < 0:     # patsy's own error message for this
                return orig_dense(x, n_knots=n_knots, knots=knots, degree=degree, name=name,
                                  include_intercept=include_intercept)
            inner = np.percentile(xx, np.linspace(0, 100, n_inner + 2)[1:-1]) if n_inner > 0 else np.zeros(0)
        full = np.concatenate([[np.min(xx)], inner, [np.max(xx)]])
        basis = _be().spline_basis_batch(xx, full, degree=int(degree))
        if not include_intercept:
            basis = basis[:, 1:]
        df = pd.DataFrame(basis, columns=["knot{}".format(i + 1) for i in range(basis.shape[1])])
        return lk_dm_mod.DesignMatrix(df, name=name)

    def create_sparse_spline_matrix(x, n_knots=20, knots=None, degree=3, name="spline"):
        from scipy.sparse import csr_matrix
        xx = np.asarray(x, np.float64)
        if not isinstance(n_knots, int):
            raise ValueError("`n_knots` must be an integer.")
        if n_knots - degree <= 0:
            raise ValueError("n_knots must be greater than degree.")
        if xx.ndim != 1 or len(xx) < 2 or not np.all(np.isfinite(xx)) or not 1 <= int(degree) <= 7:
            _fell_back("create_sparse_spline_matrix", "x / degree outside the kernel's range")
            return orig_sparse(x, n_knots=n_knots, knots=knots, degree=degree, name=name)
        if knots is None:
            ends = np.asarray([s_[-1] for s_ in np.array_split(np.argsort(xx), n_knots - degree)[:-1]])
            knots = [np.mean([xx[k], xx[k + 1]]) for k in ends]
        kn = np.unique(np.append(np.append(xx.min(), knots), xx.max()))
        basis = _be().spline_basis_batch(xx, kn, degree=int(degree))
        basis = basis[:, basis.sum(axis=0) != 0]                      # the reference drops all-zero basis vectors
        return lk_dm_mod.SparseDesignMatrix(csr_matrix(basis), name=name)

    create_spline_matrix.__doc__ = orig_dense.__doc__
    create_sparse_spline_matrix.__doc__ = orig_sparse.__doc__
    return create_spline_matrix, create_sparse_spline_matrix



# ------------------------------------------------------------------------------------------------ §8(f) rows: N2, N3, CDPP
def _make_pg_smooth(lk_pg_mod):
    """``Periodogram.smooth`` (periodogram.py:182-284; ``Periodogram.flatten`` :381-429 divides by its result): 'boxkernel' is
    astropy's NaN-interpolating convolution with a Box1DKernel -> lk_pg_boxsmooth_batch (the kernel ARRAY stays astropy's
    own); 'logmedian' is a Python ``while`` loop with one ``np.nanmedian`` per log-frequency window -> lk_pg_logmedian_batch
    (the window bookkeeping — which depends on the grid only — is the reference loop's, run once)."""
    import math
    from .periodogram import _logmedian_windows
    orig = lk_pg_mod.Periodogram.smooth
    u = lk_pg_mod.u

    def smooth(self, method="boxkernel", filter_width=0.1):
        method = lk_pg_mod.validate_method(method, ["boxkernel", "logmedian"])
        power = np.asarray(self.power.value, dtype=np.float64)
        if method == "boxkernel":
            if filter_width <= 0.0:
                raise ValueError("the `filter_width` parameter must be larger than 0 for the 'boxkernel' method.")
            try:
                filter_width = u.Quantity(filter_width, self.frequency.unit)
            except u.UnitConversionError:
                raise ValueError("the `filter_width` parameter must have frequency units.")
            if not self._is_evenly_spaced():
                raise ValueError("the 'boxkernel' method requires the periodogram to have a grid of evenly spaced frequencies.")
            fs = np.mean(np.diff(self.frequency))
            kernel = np.asarray(lk_pg_mod.Box1DKernel(math.ceil((filter_width / fs).value)).array, dtype=np.float64)
            if kernel.size > power.size:
                _fell_back("Periodogram.smooth", "a %d-tap box over %d frequencies" % (kernel.size, power.size))
                return orig(self, method=method, filter_width=filter_width)
            smooth_power = _be().pg_boxsmooth_batch(power, kernel)[0]
        else:
            if isinstance(filter_width, u.Quantity):
                raise ValueError("the 'logmedian' method requires a dimensionless value for `filter_width` in "
                                 "log10(frequency) space.")
            freq = np.asarray(self.frequency.value, dtype=np.float64)
            if freq.size < 2 or not np.all(np.isfinite(freq)) or freq[0] <= 0 or np.any(np.diff(freq) < 0) or not filter_width > 0:
                _fell_back("Periodogram.smooth", "frequency grid not positive / ascending, or filter_width <= 0")
                return orig(self, method=method, filter_width=filter_width)
            with np.errstate(invalid="ignore", divide="ignore"):
                smooth_power = _be().pg_logmedian_batch(power, *_logmedian_windows(freq, filter_width))[0]
        smooth_pg = self.copy()
        smooth_pg.power = u.Quantity(smooth_power, self.power.unit)
        return smooth_pg

    smooth.__doc__ = orig.__doc__
    return smooth


def _make_estimate_cdpp(lk_lightcurve_mod):
    """``LightCurve.estimate_cdpp`` (lightcurve.py:1764-1833): flatten -> remove_outliers (astropy sigma_clip) ->
    normalize("ppm") -> std of the running mean.  The flatten is lk_savgol_trend_batch, the clip lk_sigma_clip_batch; the
    O(N) tail (one division, a cumulative sum, np.std) is the reference's own numpy, as in
    ``lightkurve_amd.lightcurve.estimate_cdpp_batch``."""
    orig = lk_lightcurve_mod.LightCurve.estimate_cdpp
    running_mean = lk_lightcurve_mod.running_mean

    def estimate_cdpp(self, transit_duration=13, savgol_window=101, savgol_polyorder=2, sigma=5.0):
        if not isinstance(transit_duration, int):
            raise ValueError("transit_duration must be an integer in units number of cadences, got {}.".format(transit_duration))
        time = _plain(self.time.value)
        flux = self.flux
        fl = _plain(flux)
        if hasattr(flux, "mask"):
            fl = np.where(np.asarray(flux.mask, dtype=bool), np.nan, fl)
        if len(time) < 2 or np.any(np.diff(time) < 0) or savgol_window % 2 != 1 or not np.any(np.isfinite(fl)):
            _fell_back("LightCurve.estimate_cdpp", "unsorted time, even window or no finite flux")
            return orig(self, transit_duration=transit_duration, savgol_window=savgol_window,
                        savgol_polyorder=savgol_polyorder, sigma=sigma)
        if savgol_polyorder >= savgol_window:
            savgol_polyorder = savgol_window - 1
        trend = _be().savgol_trend_batch(np.ascontiguousarray(time), np.ascontiguousarray(fl), [0, len(time)],
                                         window_length=savgol_window, polyorder=savgol_polyorder)
        with np.errstate(invalid="ignore", divide="ignore"):
            flat = fl / trend
        clipped = _be().sigma_clip_batch(flat, [0, len(flat)], sigma=sigma, maxiters=5)
        kept = flat[~clipped]
        with np.errstate(invalid="ignore", divide="ignore"):
            ppm = kept / np.nanmedian(kept) * 1e6                       # normalize("ppm"), :1283-1292
        # the reference returns what np.std makes of its ppm-valued Quantity column: a Quantity in ppm
        return lk_lightcurve_mod.u.Quantity(np.std(running_mean(data=ppm, window_size=transit_duration)), "ppm")

    estimate_cdpp.__doc__ = orig.__doc__
    return estimate_cdpp


def _make_overfit_metric(lk_metrics_mod):
    """``overfit_metric_lombscargle`` (correctors/metrics.py:23-124): per sample three default periodograms — the original
    light curve, the corrected one on the same grid, a white-noise light curve at the level of the corrected uncertainties.
    The first two do not depend on the sample (computed once, through seam S1); the ``n_samples`` noise periodograms share
    the original's grid (a default grid is a function of the times alone) and are ONE lk_ls_fast_batch call.  The noise is
    drawn from the global ``np.random`` stream in the reference's order: same seed, same metric."""
    orig = lk_metrics_mod.overfit_metric_lombscargle

    def overfit_metric_lombscargle(original_lc, corrected_lc, n_samples=10):
        orig_lc = original_lc.copy().remove_nans().normalize()
        orig_lc -= 1.0
        corrected_lc = corrected_lc.copy().remove_nans().normalize()
        corrected_lc -= 1.0
        if len(corrected_lc) == 0:
            return 1.0
        n = len(orig_lc)
        if n < 2 or int(n_samples) < 1:
            _fell_back("overfit_metric_lombscargle", "fewer than two cadences / no samples")
            return orig(original_lc, corrected_lc, n_samples=n_samples)
        pg_orig = orig_lc.to_periodogram()
        pg_corr = corrected_lc.to_periodogram(frequency=pg_orig.frequency)
        import astropy.units as u
        f_day = np.asarray(pg_orig.frequency.to(1 / u.day).value, dtype=np.float64)
        from .periodogram import exact_grid
        grid = exact_grid(f_day) if len(f_day) > 1 else None
        mean_unc = np.nanmean(_plain(corrected_lc.flux_err))
        noise = np.stack([(np.random.randn(n, 1) * mean_unc).T[0] for _ in range(int(n_samples))])
        if grid is None or str(getattr(pg_orig, "ls_method", "fast")) != "fast":
            # (a default grid is regular; anything else keeps the reference's per-sample loop, each call through seam S1)
            Lc = type(original_lc)
            powers = [np.asarray(Lc(time=orig_lc.time, flux=noise[k], flux_err=np.zeros(n)).to_periodogram().power)
                      for k in range(int(n_samples))]
        else:
            t = _plain(orig_lc.time.value)
            trel = np.tile(t - t[0], int(n_samples))
            off = np.arange(int(n_samples) + 1, dtype=np.int64) * n
            powers = _be().ls_fast_batch(trel, noise.ravel(), off, f0=grid[0], df=grid[1], M=len(f_day),
                                         normalization="lk_amplitude")
        change = np.array(pg_corr.power) - np.array(pg_orig.power)
        change = change[~np.isnan(change)]
        n_up = len(np.nonzero(change > 0.0)[0])
        per_iter = []
        for k in range(int(n_samples)):
            mean_noise_power = np.nanmean(np.asarray(powers[k]))
            if n_up == 0:
                per_iter.append(0.0)
            else:
                den = n_up * mean_noise_power
                per_iter.append(np.inf if den == 0 else np.sum(change[change > 0.0]) / den)
        metric = np.mean(per_iter)
        with np.errstate(over="ignore"):
            return 2.0 / (1 + np.exp(np.max([metric, 0.0])))

    overfit_metric_lombscargle.__doc__ = orig.__doc__
    return overfit_metric_lombscargle


_METRIC_MODULES = ("lightkurve.correctors.metrics", "lightkurve.correctors.cbvcorrector", "lightkurve.correctors.corrector")


def _set_overfit_metric(fn):
    """imported by name into cbvcorrector.py:33 and corrector.py:9: every namespace that holds it gets the replacement."""
    import importlib
    for modname in _METRIC_MODULES:
        try:
            mod = importlib.import_module(modname)
        except ImportError:
            continue
        if hasattr(mod, "overfit_metric_lombscargle"):
            mod.overfit_metric_lombscargle = fn


_SPLINE_MODULES = ("lightkurve.correctors.designmatrix", "lightkurve.correctors.pldcorrector",
                   "lightkurve.correctors.sffcorrector", "lightkurve.correctors")


def _set_spline_builders(dense, sparse):
    """The builders are imported by name into several modules (pldcorrector.py:26, sffcorrector.py:18): every namespace that
    holds them gets the replacement."""
    import importlib
    for modname in _SPLINE_MODULES:
        try:
            mod = importlib.import_module(modname)
        except ImportError:
            continue
        if hasattr(mod, "create_spline_matrix"):
            mod.create_spline_matrix = dense
        if hasattr(mod, "create_sparse_spline_matrix"):
            mod.create_sparse_spline_matrix = sparse


# ------------------------------------------------------------------------------------------------ install / uninstall
_LS_NAMES = ("fast", "fastchi2", "chi2", "slow", "cython", "scipy")     # astropy's whole registry (main.py:20-25)

def install(backend=None, lightkurve=True, full_loop=True):
    """Patch astropy (S1, S2) and, if it is importable and ``lightkurve`` is true, lightkurve (S3, S4) in place; returns
    the list of seams installed.  ``full_loop``: replace ``RegressionCorrector.correct`` (all clip iterations in one
    GPU call) in addition to ``_fit_coefficients``."""
    global _BACKEND
    _BACKEND = backend if backend is not None else _capi
    from astropy.timeseries.periodograms.bls import methods as bls_methods
    from astropy.timeseries.periodograms.lombscargle.implementations import main as ls_main
    for name in _LS_NAMES:
        _ORIG.setdefault(name, ls_main.METHODS.get(name + "_cpu", ls_main.METHODS[name]))
        ls_main.METHODS.setdefault(name + "_cpu", _ORIG[name])
    ls_main.METHODS["hip"] = lombscargle_hip
    # the exact single-term names.  'slow' is where lightkurve's own switch sends every request whose grid is not regular
    # in frequency (periodogram.py:933-946: any period= / minimum_period= call); 'cython' is what method='auto' picks for
    # short or irregular grids and 'scipy' what it picks without errors and floating mean (main.py:96-104).  All three
    # receive the raw frequency array, like 'hip'
    ls_main.METHODS["slow"] = lombscargle_hip
    ls_main.METHODS["cython"] = lombscargle_hip
    ls_main.METHODS["scipy"] = lombscargle_scipy_hip
    # multi-term fits (lightkurve nterms > 1 requires the name 'chi2' or 'fastchi2', periodogram.py:948-958):
    # 'chi2' receives the raw frequency array like 'hip' does, so it can be replaced one-to-one
    ls_main.METHODS["chi2"] = lombscargle_hip
    # the reference's DEFAULT method: lc.to_periodogram() reaches the GPU with no change on the caller's side
    ls_main.METHODS["fast"] = lombscargle_fast_hip
    ls_main.METHODS["fastchi2"] = lombscargle_fast_hip
    bls_methods._bls_fast_reference = getattr(bls_methods, "_bls_fast_reference", bls_methods.bls_fast)
    bls_methods.bls_fast = bls_fast_hip
    done = ["lombscargle:METHODS['hip']", "lombscargle:METHODS['slow']", "lombscargle:METHODS['cython']",
            "lombscargle:METHODS['scipy']", "lombscargle:METHODS['chi2']", "lombscargle:METHODS['fast']",
            "lombscargle:METHODS['fastchi2']", "bls:methods.bls_fast"]
    if not lightkurve:
        return done
    try:
        import lightkurve.lightcurve as lk_lc
        import lightkurve.correctors.regressioncorrector as lk_rc
        import lightkurve.correctors.pldcorrector as lk_pld
    except ImportError:
        return done
    _ORIG.setdefault("flatten", lk_lc.LightCurve.flatten)
    _ORIG.setdefault("_fit_coefficients", lk_rc.RegressionCorrector._fit_coefficients)
    _ORIG.setdefault("correct", lk_rc.RegressionCorrector.correct)
    _ORIG.setdefault("create_design_matrix", lk_pld.PLDCorrector.create_design_matrix)
    lk_lc.LightCurve.flatten = _make_flatten(lk_lc)
    done.append("lightkurve:LightCurve.flatten")
    lk_rc.RegressionCorrector._fit_coefficients = _fit_coefficients_hip
    done.append("lightkurve:RegressionCorrector._fit_coefficients")
    if full_loop:
        lk_rc.RegressionCorrector.correct = _make_correct(lk_rc)
        done.append("lightkurve:RegressionCorrector.correct")
    lk_pld.PLDCorrector.create_design_matrix = _make_create_design_matrix(lk_pld)
    done.append("lightkurve:PLDCorrector.create_design_matrix")
    import lightkurve.correctors.designmatrix as lk_dm
    _ORIG.setdefault("dm_pca", lk_dm.DesignMatrix.pca)
    _ORIG.setdefault("dm_standardize", lk_dm.DesignMatrix.standardize)
    lk_dm.DesignMatrix.pca = _make_dm_pca(lk_dm)
    lk_dm.DesignMatrix.standardize = _make_dm_standardize(lk_dm)
    done += ["lightkurve:DesignMatrix.pca", "lightkurve:DesignMatrix.standardize"]
    _ORIG.setdefault("create_spline_matrix", lk_dm.create_spline_matrix)
    _ORIG.setdefault("create_sparse_spline_matrix", lk_dm.create_sparse_spline_matrix)
    _set_spline_builders(*_make_spline_builders(lk_dm))
    done += ["lightkurve:create_spline_matrix", "lightkurve:create_sparse_spline_matrix"]
    # SURVEY §8(f) rows N2 / N3 and the CDPP metric (round 6)
    import lightkurve.periodogram as lk_pg
    import lightkurve.correctors.metrics as lk_met
    _ORIG.setdefault("pg_smooth", lk_pg.Periodogram.smooth)
    _ORIG.setdefault("estimate_cdpp", lk_lc.LightCurve.estimate_cdpp)
    _ORIG.setdefault("overfit_metric", lk_met.overfit_metric_lombscargle)
    lk_pg.Periodogram.smooth = _make_pg_smooth(lk_pg)
    lk_lc.LightCurve.estimate_cdpp = _make_estimate_cdpp(lk_lc)
    _set_overfit_metric(_make_overfit_metric(lk_met))
    done += ["lightkurve:Periodogram.smooth", "lightkurve:LightCurve.estimate_cdpp",
             "lightkurve:overfit_metric_lombscargle"]
    return done


def uninstall():
    """Undo ``install()`` (restores astropy's and lightkurve's own functions)."""
    global _BACKEND
    _BACKEND = _capi
    from astropy.timeseries.periodograms.bls import methods as bls_methods
    from astropy.timeseries.periodograms.lombscargle.implementations import main as ls_main
    for name in _LS_NAMES:
        if name in _ORIG:
            ls_main.METHODS[name] = _ORIG[name]
    ls_main.METHODS.pop("hip", None)
    if hasattr(bls_methods, "_bls_fast_reference"):
        bls_methods.bls_fast = bls_methods._bls_fast_reference
    try:
        import lightkurve.lightcurve as lk_lc
        import lightkurve.correctors.regressioncorrector as lk_rc
        import lightkurve.correctors.pldcorrector as lk_pld
    except ImportError:
        return
    if "flatten" in _ORIG:
        lk_lc.LightCurve.flatten = _ORIG["flatten"]
    if "_fit_coefficients" in _ORIG:
        lk_rc.RegressionCorrector._fit_coefficients = _ORIG["_fit_coefficients"]
    if "correct" in _ORIG:
        lk_rc.RegressionCorrector.correct = _ORIG["correct"]
    if "create_design_matrix" in _ORIG:
        lk_pld.PLDCorrector.create_design_matrix = _ORIG["create_design_matrix"]
    import lightkurve.correctors.designmatrix as lk_dm
    if "dm_pca" in _ORIG:
        lk_dm.DesignMatrix.pca = _ORIG["dm_pca"]
    if "dm_standardize" in _ORIG:
        lk_dm.DesignMatrix.standardize = _ORIG["dm_standardize"]
    if "create_spline_matrix" in _ORIG:
        _set_spline_builders(_ORIG["create_spline_matrix"], _ORIG["create_sparse_spline_matrix"])
    if "pg_smooth" in _ORIG:
        import lightkurve.periodogram as lk_pg
        lk_pg.Periodogram.smooth = _ORIG["pg_smooth"]
    if "estimate_cdpp" in _ORIG:
        lk_lc.LightCurve.estimate_cdpp = _ORIG["estimate_cdpp"]
    if "overfit_metric" in _ORIG:
        _set_overfit_metric(_ORIG["overfit_metric"])
