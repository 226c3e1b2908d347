"""Deterministic stand-in for fbpca.pca (not installed anywhere here; the real one is randomized).

lightkurve uses only U (designmatrix.py:279-281).  Exact thin SVD of the column-centred matrix.
"""
import numpy as np


def pca(A, k=6, raw=False, n_iter=2, l=None):
    A = np.asarray(A, dtype=float)
    if not raw:
        A = A - A.mean(axis=0)
    U, s, Va = np.linalg.svd(A, full_matrices=False)
    return U[:, :k], s[:k], Va[:k]
