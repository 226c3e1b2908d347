"""GPU: the standalone design-matrix operations — lk_pca_batch, lk_spline_basis_batch, lk_standardize_batch
and their mirrors DesignMatrix.pca / .standardize / .split, create_spline_matrix — against outputs of the reference itself
(tests/golden/designmatrix_ops.npz: lightkurve's DesignMatrix methods and patsy's bs(), written by oracle/gen_golden.py).
Tolerances (stated): PCA bases are compared as SUBSPACES (the reference's fbpca is randomised; a basis is defined up to
sign / rotation): ||P_ref - P_got||_2 <= 1e-6; splines and standardize element-wise 1e-12; split exactly."""
import os

import numpy as np
import pytest

from lightkurve_amd import _capi
from lightkurve_amd.correctors import DesignMatrix
from lightkurve_amd.correctors.designmatrix import create_spline_matrix

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "designmatrix_ops.npz"))


def _subspace_gap(U, V):
    Qu, Qv = np.linalg.qr(U)[0], np.linalg.qr(V)[0]
    return np.linalg.norm(Qu @ Qu.T - Qv @ Qv.T, 2)


def test_pca_matches_the_reference_subspace_and_is_orthonormal():
    A = G["A"]
    for k, key in ((6, "pca6"), (3, "pca3")):
        U = DesignMatrix(A, name="a").pca(k).values
        assert U.shape == (A.shape[0], k)
        assert np.max(np.abs(U.T @ U - np.eye(k))) < 1e-10           # left singular vectors
        assert _subspace_gap(U, G[key]) < 1e-6, k
        # each column is the reference's up to sign where the spectrum is non-degenerate
        for j in range(k):
            assert min(np.max(np.abs(U[:, j] - G[key][:, j])), np.max(np.abs(U[:, j] + G[key][:, j]))) < 1e-6, (k, j)
    # batch of differently scaled copies in one call; nterms larger than the column count is clipped like the reference
    Ub = _capi.pca_batch(np.stack([A, 3.0 * A[::-1]]), 4)
    assert _subspace_gap(Ub[0], G["pca6"][:, :4]) < 1e-6 and _subspace_gap(Ub[1][::-1], G["pca6"][:, :4]) < 1e-6
    assert DesignMatrix(A[:, :2]).pca(5).shape == (A.shape[0], 2)
    with pytest.raises(ValueError):
        _capi.pca_batch(A, 49)


def test_spline_matrix_matches_patsy():
    x = G["x"]
    for kw, key in ((dict(n_knots=20, degree=3), "spline_n20_d3"),
                    (dict(n_knots=12, degree=5, include_intercept=False), "spline_n12_d5_noint"),
                    (dict(knots=list(G["knots_given"]), degree=3), "spline_knots_d3")):
        dm = create_spline_matrix(x, **kw)
        assert dm.shape == G[key].shape, key
        assert np.max(np.abs(dm.values - G[key])) < 1e-12, key
        assert dm.columns[0] == "knot1" and dm.name == "spline"
    with pytest.raises(ValueError):
        create_spline_matrix(x, n_knots=2, degree=3)


def test_standardize_and_split_match_the_reference():
    got = DesignMatrix(G["S"], name="s").standardize().values
    assert np.max(np.abs(got - G["standardized"])) < 1e-12
    assert np.array_equal(got[:, 3], G["S"][:, 3]) and np.all(got[:, 5] == 0)          # constant / all-zero columns
    dm3 = DesignMatrix(G["A"][:, :3], name="three", prior_mu=[1.0, 2.0, 3.0], prior_sigma=[0.1, 0.2, 0.3])
    sp = dm3.split([200, 450])
    assert np.array_equal(sp.values, G["split"])
    assert np.array_equal(sp.prior_mu, G["split_mu"]) and np.array_equal(sp.prior_sigma, G["split_sigma"])
    assert dm3.split([]) is dm3 and dm3.split([0]) is dm3


def test_pca_wide_block_with_many_components_falls_to_the_subspace_iteration():
    """P = 134..138 with k in the 40s needs more than 160 KB of LDS in the direct tridiagonal solver; such
    shapes must take the subspace iteration instead of failing (reference designmatrix.py:252-282 has no such limit)."""
    rng = np.random.default_rng(8)
    N, P, k = 600, 138, 48
    A = rng.normal(size=(N, 60)) @ rng.normal(size=(60, P)) * np.geomspace(1.0, 1e-3, P) + 1e-6 * rng.normal(size=(N, P))
    U = _capi.pca_batch(A, k)
    Ac = A - A.mean(axis=0)
    ref = np.linalg.svd(Ac, full_matrices=False)[0][:, :k]
    assert U.shape == (N, k) and np.max(np.abs(U.T @ U - np.eye(k))) < 1e-9
    assert _subspace_gap(U, ref) < 1e-6


def test_sparse_spline_matrix_matches_reference_golden(golden):
    """create_sparse_spline_matrix (reference designmatrix.py:896-949, the spline block of PLDCorrector(sparse=True)): the
    reference's Python Cox-de Boor recursion vs the de Boor kernel on the reference's knots — 1e-12 (measured 4e-16)."""
    from lightkurve_amd.correctors import create_sparse_spline_matrix
    g = golden("pld_k2sin_order3_sparse")
    w = int(g["block_widths"][-1])
    sp = create_sparse_spline_matrix(g["time"], n_knots=10, degree=5).append_constant()
    assert sp.shape == (len(g["time"]), w)
    assert np.max(np.abs(sp.X - g["X"][:, -w:])) < 1e-12


def test_direct_solver_with_repeated_singular_values_both_gram_schmidt_paths():
    """The direct tridiagonal solver (64 < P <= 138) orthogonalises eigenvectors of clustered eigenvalues by modified
    Gram-Schmidt: with a wave per vector in registers for k <= 16, by one wave on the output array beyond.  Repeated singular
    values (exact multiplets inside the wanted set, the cut between distinct values) exercise both: orthonormal columns, the
    reference's subspace (np.linalg.svd of the centred matrix; designmatrix.py:252-282 keeps the leading left vectors)."""
    rng = np.random.default_rng(21)
    N, P = 400, 100
    sv = np.concatenate([[5, 5, 5, 4, 4, 3, 3, 3, 3, 2, 1.5, 1.5, 1.2, 1.1, 1.0, 0.9, 0.8, 0.8, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3],
                         np.geomspace(0.1, 1e-4, P - 24)])
    Qn = np.linalg.qr(rng.normal(size=(N, P)))[0]
    Qp = np.linalg.qr(rng.normal(size=(P, P)))[0]
    A = (Qn * sv) @ Qp.T
    Ac = A - A.mean(axis=0)
    Uref, sref, _ = np.linalg.svd(Ac, full_matrices=False)
    for k in (10, 16, 23):                                             # cuts between distinct singular values of the centred matrix
        assert sref[k - 1] - sref[k] > 0.05 * sref[k - 1], k
        U = _capi.pca_batch(A, k)
        assert U.shape == (N, k)
        assert np.max(np.abs(U.T @ U - np.eye(k))) < 1e-9, k
        assert _subspace_gap(U, Uref[:, :k]) < 1e-6, k
