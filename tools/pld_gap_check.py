"""Subspace gaps of the PCA blocks of the k2sin order-3 golden for the library LK_LIB_PATH selects (development aid:
how close tests/test_pld_gpu.py::test_golden_third_order_path runs to its thresholds)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_pld_gpu import subspace_gap, PixelCube, PLDCorrector  # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pld_k2sin_order3.npz"))
pld = PLDCorrector(PixelCube(g["time"], g["flux"], g["flux_err"]))
clc = pld.correct(pld_order=3, pca_components=16, pld_aperture_mask="all", normalize_background_pixels=True)
X = pld.design_matrix_collection.X
w = g["block_widths"]
print("corrected relerr %.3e | gaps order1 %.2e order2 %.2e order3 %.2e background %.2e" % (
    np.max(np.abs(clc.flux - g["corrected"])) / np.median(g["corrected"]),
    subspace_gap(X[:, :16], g["X"][:, :16]), subspace_gap(X[:, 16:32], g["X"][:, 16:32]),
    subspace_gap(X[:, 32:48], g["X"][:, 32:48]), subspace_gap(X[:, 48:48 + w[1]], g["X"][:, 48:48 + w[1]])))
