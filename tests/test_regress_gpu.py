"""GPU parity: RegressionCorrector numerics (MFMA Gram + LU + sigma-clip loop) vs reference golden vectors and
the numpy oracle.  Tolerances (stated): outlier masks identical; corrected flux / model within 1e-9 * std(flux)
absolute; coefficients within 1e-7 relative of their scale (the solve is LU like LAPACK gesv but a different
elimination order, so coefficients of ill-conditioned columns agree less tightly than the model they produce)."""
import numpy as np
import pytest

from lightkurve_amd import _capi
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


def test_golden_k8_with_priors_mask_outliers(golden):
    g = golden("regress_k8")
    n = len(g["flux"])
    r = _capi.regress_batch(g["X"], g["flux"], [0, n], err=g["flux_err"], cadence_mask=g["cadence_mask"],
                            prior_mu=g["prior_mu"], prior_sigma=g["prior_sigma"])
    assert np.array_equal(r["outlier_mask"], g["outlier_mask"])
    assert np.allclose(r["coefficients"][0], g["coefficients"], rtol=1e-8, atol=1e-12)
    assert np.allclose(g["flux"] - r["model"], g["corrected"], rtol=0, atol=1e-11)
    assert np.allclose(r["model"], g["model"], rtol=0, atol=1e-11)


def test_golden_kat_and_no_errors(golden):
    g = golden("regress_kat")
    X = np.array([[1., 1.], [1., 2.]])
    r = _capi.regress_batch(X, [5., 10.], [0, 2], err=[1., 1.])
    assert np.allclose(r["coefficients"][0], g["noprior"], atol=1e-9)        # [0, 5]
    r = _capi.regress_batch(X, [5., 10.], [0, 2], err=[1., 1.], prior_mu=[99., 99.], prior_sigma=[1e-6, 1e-6])
    assert np.allclose(r["coefficients"][0], g["tight"], atol=1e-7)          # [99, 99]
    g = golden("regress_noerr")
    n = len(g["flux"])
    r = _capi.regress_batch(g["X"], g["flux"], [0, n])
    assert np.array_equal(r["outlier_mask"], g["outlier_mask"])
    assert np.allclose(g["flux"] - r["model"], g["corrected"], rtol=0, atol=1e-11)
    with pytest.raises(ValueError, match="both"):
        _capi.regress_batch(X, [5., 10.], [0, 2], prior_mu=[0., 0.])


def make_problem(rng, n, k, noutl, smooth=True):
    t = np.linspace(0, 30, n)
    if smooth:   # sinusoid regressors (mildly collinear, like real systematics bases)
        cols = [np.sin(2 * np.pi * t * rng.uniform(0.05, 3.0) + rng.uniform(0, 6)) for _ in range(k - 1)]
    else:        # well-conditioned: white regressors (hundreds of sinusoids in 30 d would be numerically rank deficient)
        cols = [rng.normal(0, 1, n) for _ in range(k - 1)]
    X = np.column_stack(cols + [np.ones(n)])
    w = rng.normal(0, 1e-3, k)
    w[-1] = 1.0
    err = rng.uniform(0.5, 2.0, n) * 2e-4
    y = X @ w + rng.normal(0, 1, n) * err
    y[rng.integers(0, n, noutl)] += rng.choice([-1, 1], noutl) * 0.01
    cm = np.ones(n, bool)
    cm[n // 3:n // 3 + n // 50] = False
    return X, y, err, cm


def test_ragged_batch_k135_vs_oracle():
    """K = 135 (C5's design-matrix width: 3 blocks of 64 incl. a partial one) on ragged N, priors on half the columns."""
    rng = np.random.default_rng(42)
    K = 135
    ns = [3500, 900, 2048, 3499, 400]
    Xs, ys, es, cms = zip(*[make_problem(rng, n, K, 12) for n in ns])
    off = np.r_[0, np.cumsum(ns)]
    mu = np.zeros((len(ns), K))
    sg = np.full((len(ns), K), np.inf)
    sg[:, ::2] = 0.05
    r = _capi.regress_batch(np.vstack(Xs), np.concatenate(ys), off, err=np.concatenate(es),
                            cadence_mask=np.concatenate(cms), prior_mu=mu, prior_sigma=sg)
    for b, n in enumerate(ns):
        ref = O.regression_correct(Xs[b], ys[b], es[b], cms[b], mu[b], sg[b])
        s = slice(off[b], off[b + 1])
        assert np.array_equal(r["outlier_mask"][s], ref["outlier_mask"]), b
        assert np.max(np.abs(r["model"][s] - ref["model"])) < 1e-9 * np.std(ys[b]), b
        assert np.allclose(r["coefficients"][b], ref["coefficients"], rtol=1e-6, atol=1e-9), b


def test_clip_loop_delta_gram_and_its_overflow_vs_oracle():
    """Passes 2.. of the clip loop take the previous normal matrix minus the newly clipped cadences (an ascending list of
    at most 256 per target); a target whose clip removes more than that at once is recomputed in full.  Both routes, and
    a target that converges at once, against the oracle: identical masks, models within 1e-9 std(flux)."""
    K = 40
    ns = [9000, 3000, 2500]
    rng = np.random.default_rng(5)
    X0, y0, e0, c0 = make_problem(rng, ns[0], K, 0)
    y0[rng.choice(ns[0], 300, replace=False)] += 1.0      # 3.3 % at one amplitude: 5.4 sigma, all clipped by the first clip
    rng = np.random.default_rng(6)
    X1, y1, e1, c1 = make_problem(rng, ns[1], K, 0)
    for amp, cnt in ((0.5, 40), (0.02, 40), (0.004, 40)):  # the small ones only stand out once the large ones are gone
        y1[rng.choice(ns[1], cnt, replace=False)] += amp * rng.choice([-1, 1], cnt)
    X2, y2, e2, c2 = make_problem(rng, ns[2], K, 0)
    Xs, ys, es, cms = (X0, X1, X2), (y0, y1, y2), (e0, e1, e2), (c0, c1, c2)
    off = np.r_[0, np.cumsum(ns)]
    r = _capi.regress_batch(np.vstack(Xs), np.concatenate(ys), off, err=np.concatenate(es), cadence_mask=np.concatenate(cms))
    n_out = []
    for b, n in enumerate(ns):
        ref = O.regression_correct(Xs[b], ys[b], es[b], cms[b])
        s = slice(off[b], off[b + 1])
        assert np.array_equal(r["outlier_mask"][s], ref["outlier_mask"]), b
        assert np.max(np.abs(r["model"][s] - ref["model"])) < 1e-9 * np.std(ys[b]), b
        n_out.append(int(ref["outlier_mask"].sum()))
    assert n_out[0] >= 300 and n_out[1] >= 100  # the cases are what the docstring says they are


@pytest.mark.parametrize("K", [3, 16, 17, 33, 50, 70, 90, 110, 128, 141, 143, 145])
def test_narrow_gram_kernel_every_tile_count(K):
    """[X | y] of K + 1 <= 144 columns takes the one-workgroup-per-matrix Gram kernel (T = 1 .. 9 tile columns, its waves'
    tile runs differ with T); 145 is the first width on the 64 x 64-block kernel again.  Ragged N (partial last stage),
    weights, a cadence mask: coefficients and model against the oracle."""
    rng = np.random.default_rng(100 + K)
    ns = [517, 1000, 33]
    if K >= 33:
        ns[2] = K + 40
    Xs, ys, es, cms = zip(*[make_problem(rng, n, K, 3, smooth=False) for n in ns])
    off = np.r_[0, np.cumsum(ns)]
    r = _capi.regress_batch(np.vstack(Xs), np.concatenate(ys), off, err=np.concatenate(es), cadence_mask=np.concatenate(cms))
    for b, n in enumerate(ns):
        ref = O.regression_correct(Xs[b], ys[b], es[b], cms[b])
        s = slice(off[b], off[b + 1])
        assert np.array_equal(r["outlier_mask"][s], ref["outlier_mask"]), (K, b)
        assert np.max(np.abs(r["model"][s] - ref["model"])) < 1e-9 * np.std(ys[b]), (K, b)
        assert np.allclose(r["coefficients"][b], ref["coefficients"], rtol=1e-6, atol=1e-9), (K, b)


def test_full_width_k465_properties():
    """K = 465 (the N = 20000 design-matrix width): residual orthogonality X^T W r ~ 0 on the unclipped cadences
    (a size-independent property of the normal equations), and agreement with the oracle's model."""
    rng = np.random.default_rng(7)
    n, K = 6000, 465
    X, y, err, cm = make_problem(rng, n, K, 20, smooth=False)
    r = _capi.regress_batch(X, y, [0, n], err=err, cadence_mask=cm)
    m = cm & ~r["outlier_mask"]
    model_raw = X @ r["coefficients"][0]
    grad = X[m].T @ ((y[m] - model_raw[m]) / err[m] ** 2)
    scale = np.abs(X[m].T) @ (np.abs(y[m]) / err[m] ** 2)
    assert np.max(np.abs(grad) / scale) < 1e-9
    ref = O.regression_correct(X, y, err, cm)
    assert np.array_equal(r["outlier_mask"], ref["outlier_mask"])
    assert np.max(np.abs(r["model"] - ref["model"])) < 1e-8 * np.std(y)


def test_regressioncorrector_api_like_reference(golden):
    """reference tests/correctors/test_regressioncorrector.py:13-118 (priors KAT, sinusoid removal, validation)."""
    from lightkurve_amd import LightCurve
    from lightkurve_amd.correctors import DesignMatrix, DesignMatrixCollection, RegressionCorrector
    lc = LightCurve(flux=[5, 10], flux_err=[1, 1], time=[1, 2])
    dm = DesignMatrix({"a": [1., 1.], "b": [1., 2.]})
    rc = RegressionCorrector(lc)
    rc.correct(dm)
    assert np.allclose(rc.coefficients, [0, 5], atol=1e-7)
    dm = DesignMatrix({"a": [1., 1.], "b": [1., 2.]}, prior_mu=[99., 99.], prior_sigma=[1e-6, 1e-6])
    rc.correct(dm)
    assert np.allclose(rc.coefficients, [99, 99], atol=1e-6)
    # sinusoid removal, with and without flux_err
    size = 100
    time = np.linspace(1, 100, size)
    true = np.ones(size)
    noise = np.sin(time / 5)
    for err in (np.ones(size), None):
        lc = LightCurve(time=time, flux=true + noise, flux_err=err)
        dmc = DesignMatrixCollection([DesignMatrix({"noise": noise}, name="noise"),
                                      DesignMatrix({"offset": np.ones(size)}, name="offset")])
        rc = RegressionCorrector(lc)
        clc = rc.correct(dmc)
        assert np.allclose(clc.flux / np.median(clc.flux), true, atol=1e-7)      # reference: corrected_lc.normalize()
        assert set(rc.diagnostic_lightcurves) == {"noise", "offset"}
    with pytest.raises(ValueError, match="NaNs in time or flux"):
        RegressionCorrector(LightCurve(time=time, flux=np.r_[np.nan, true[1:]], flux_err=np.ones(size)))
    with pytest.raises(ValueError, match="smaller than or equal to zero"):
        RegressionCorrector(LightCurve(time=time, flux=true, flux_err=np.zeros(size)))
    g = golden("regress_k8")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    rc = RegressionCorrector(lc)
    clc = rc.correct(DesignMatrix(g["X"], name="X", prior_mu=g["prior_mu"], prior_sigma=g["prior_sigma"]),
                     cadence_mask=g["cadence_mask"])
    assert np.array_equal(rc.outlier_mask, g["outlier_mask"]) and np.allclose(clc.flux, g["corrected"], atol=1e-11)


def test_cbv_gaussian_prior_golden(golden):
    """Row A14: CBVCorrector.correct_gaussian_prior = [CBVs, Constant] with one ridge width on every column, fitted by
    the regression kernels; golden from the reference RegressionCorrector on the same collection (1e-9 relative)."""
    from lightkurve_amd.correctors.cbvcorrector import CBVCorrector
    from lightkurve_amd.lightcurve import LightCurve
    g = golden("cbv_ridge")
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    for tag in ("weak", "ridge", "none"):
        cor = CBVCorrector(lc, g["cbvs"])
        out = cor.correct_gaussian_prior(cbv_indices=np.arange(1, 9), alpha=float(g["alpha_" + tag]),
                                         cadence_mask=g["cadence_mask"])
        assert np.array_equal(cor.outlier_mask, g["outlier_" + tag])
        assert np.max(np.abs(out.flux - g["corrected_" + tag])) <= 1e-9 * np.max(np.abs(g["corrected_" + tag]))
        assert np.allclose(cor.coefficients, g["coefficients_" + tag], rtol=1e-7, atol=1e-9 * np.abs(g["coefficients_" + tag]).max())


def test_propagate_errors_covariance_vs_reference_golden(golden):
    """propagate_errors=True: the coefficient covariance (inverse normal matrix of the last fit, Gauss-Jordan on the GPU)
    against the reference's np.linalg.inv (1e-9 of each entry's scale) and, through the mirror class with the same numpy
    seed, the sampled model uncertainty (the draws go through numpy's SVD of our covariance: 1e-6 relative)."""
    from lightkurve_amd import LightCurve
    from lightkurve_amd.correctors import DesignMatrix, RegressionCorrector
    g = golden("regress_cov")
    n = len(g["flux"])
    r = _capi.regress_batch(g["X"], g["flux"], [0, n], err=g["flux_err"], cadence_mask=g["cadence_mask"],
                            prior_mu=g["prior_mu"], prior_sigma=g["prior_sigma"], return_cov=True)
    cov, ref = r["coefficients_cov"][0], g["coefficients_err"]
    scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
    assert np.max(np.abs(cov - ref) / scale) < 1e-9
    assert np.allclose(cov, cov.T, rtol=0, atol=1e-12 * np.max(np.abs(cov)))
    lc = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    dm = DesignMatrix(g["X"], name="X", prior_mu=g["prior_mu"], prior_sigma=g["prior_sigma"])
    rc = RegressionCorrector(lc)
    np.random.seed(int(g["seed"]))
    clc = rc.correct(dm, cadence_mask=g["cadence_mask"], propagate_errors=True)
    assert np.allclose(rc.model_lc.flux_err, g["model_err"], rtol=1e-6, atol=0)
    assert np.allclose(clc.flux_err, g["corrected_err"], rtol=1e-8, atol=0)


def test_covariance_batch_k135_vs_oracle():
    rng = np.random.default_rng(7)
    Xs, ys, es, cms = [], [], [], []
    for n in (900, 1400):
        X, y, err, cm = make_problem(rng, n, 135, 10, smooth=False)
        Xs.append(X), ys.append(y), es.append(err), cms.append(cm)
    off = np.concatenate([[0], np.cumsum([len(y) for y in ys])])
    ps = np.full((2, 135), 10.0)
    r = _capi.regress_batch(np.vstack(Xs), np.concatenate(ys), off, err=np.concatenate(es),
                            cadence_mask=np.concatenate(cms), prior_mu=np.zeros((2, 135)), prior_sigma=ps, return_cov=True)
    for b in range(2):
        o = O.regression_correct(Xs[b], ys[b], es[b], cadence_mask=cms[b], prior_mu=np.zeros(135), prior_sigma=ps[b])
        assert np.array_equal(r["outlier_mask"][off[b]:off[b + 1]], o["outlier_mask"])
        ref = o["coefficients_cov"]
        scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
        assert np.max(np.abs(r["coefficients_cov"][b] - ref) / scale) < 1e-8


def test_nan_regressor_column_gives_nan_coefficients_not_a_fault():
    """an all-NaN column of the Gram matrix left the fused pivot search of solve_lds_kernel without a candidate
    (row index INT_MAX).  numpy.linalg.solve returns NaN coefficients for such a system (reference
    regressioncorrector.py:166-168); so must the kernel — and the other target of the batch must be untouched."""
    rng = np.random.default_rng(3)
    K, ns = 12, [500, 400]
    Xs, ys, es, cms = zip(*[make_problem(rng, n, K, 4) for n in ns])
    Xs = [x.copy() for x in Xs]
    Xs[0][:, 5] = np.nan
    off = np.r_[0, np.cumsum(ns)]
    r = _capi.regress_batch(np.vstack(Xs), np.concatenate(ys), off, err=np.concatenate(es), cadence_mask=np.concatenate(cms))
    assert np.all(np.isnan(r["coefficients"][0]))
    ref = O.regression_correct(Xs[1], ys[1], es[1], cms[1], None, None)
    assert np.allclose(r["coefficients"][1], ref["coefficients"], rtol=1e-6, atol=1e-9)
    assert np.array_equal(r["outlier_mask"][off[1]:], ref["outlier_mask"])
