"""GPU parity: PLDCorrector (design matrix by MFMA Gram + subspace eigen-solver + B-splines, then the regression
kernels) vs golden vectors produced by the reference itself (fbpca replaced by the exact SVD).
Parity is stated on the CORRECTED FLUX (<= 1e-6 relative; identical outlier masks) and on block SUBSPACES, never
on X columns or coefficients — PCA bases are defined up to a rotation inside each block."""
import numpy as np
import pytest

from lightkurve_amd.correctors import PixelCube, PLDCorrector, pld_correct_batch
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu


def subspace_gap(U, V):
    """sin of the largest principal angle between span(U) and span(V)."""
    Qu, _ = np.linalg.qr(U)
    Qv, _ = np.linalg.qr(V)
    s = np.linalg.svd(Qu.T @ Qv, compute_uv=False)
    return np.sqrt(max(0.0, 1 - s.min() ** 2))


def test_golden_third_order_path(golden):
    g = golden("pld_k2sin_order3")
    cube = PixelCube(g["time"], g["flux"], g["flux_err"])
    assert np.array_equal(cube.create_threshold_mask(3), g["threshold_mask"])
    pld = PLDCorrector(cube)                                   # default aperture: threshold mask
    assert np.array_equal(pld.aperture_mask, g["aperture_mask"])
    assert np.allclose(pld.lc.flux, g["lc_flux"], rtol=1e-6)
    clc = pld.correct(pld_order=3, pca_components=16, pld_aperture_mask="all", normalize_background_pixels=True)
    X = pld.design_matrix_collection.X
    assert X.shape == g["X"].shape
    assert np.array_equal(pld.outlier_mask, g["outlier_mask"])
    assert np.max(np.abs(clc.flux - g["corrected"])) / np.median(g["corrected"]) < 1e-6
    # spline block: same basis element-wise; PCA blocks: same subspaces
    assert np.allclose(X[:, -11:], g["X"][:, -11:], rtol=0, atol=1e-12)
    assert np.allclose(pld.design_matrix_collection.prior_sigma, g["prior_sigma"], rtol=1e-6)
    w = g["block_widths"]
    assert subspace_gap(X[:, :16], g["X"][:, :16]) < 1e-6          # order 1 (the reference's re-PCA only rotates it)
    assert subspace_gap(X[:, 16:32], g["X"][:, 16:32]) < 1e-5      # order 2 (136 -> 16)
    assert subspace_gap(X[:, 32:48], g["X"][:, 32:48]) < 1e-4      # order 3 (816 -> 16)
    # background: float32 row-sum normalisation + noise-dominated trailing components (2.5e-5 even CPU vs CPU)
    assert subspace_gap(X[:, 48:48 + w[1]], g["X"][:, 48:48 + w[1]]) < 1e-3


def test_golden_default_path_and_factory_cutout(golden):
    g = golden("pld_k2sin_default")                            # no MISSION: order 1, 3 PCA terms, 'empty' PLD pixels
    pld = PLDCorrector(PixelCube(g["time"], g["flux"], g["flux_err"]))
    clc = pld.correct()
    assert pld.design_matrix_collection.X.shape == g["X"].shape == (500, 14)
    assert np.array_equal(pld.outlier_mask, g["outlier_mask"])
    assert np.max(np.abs(clc.flux - g["corrected"])) / np.median(g["corrected"]) < 1e-6
    g = golden("pld_factory11_order2")                         # K2-like 11x11 factory cutout, order 2, 8 comps, degree 3
    pld = PLDCorrector(PixelCube(g["time"], g["flux"], g["flux_err"], mission="K2"), aperture_mask="all")
    clc = pld.correct(pld_order=2, pca_components=8, pld_aperture_mask="all", background_aperture_mask="all",
                      spline_degree=3)
    assert pld.design_matrix_collection.X.shape == g["X"].shape
    assert np.array_equal(pld.outlier_mask, g["outlier_mask"])
    assert np.max(np.abs(clc.flux - g["corrected"])) / np.median(g["corrected"]) < 1e-6


def test_golden_bench_shape(golden):
    """BASELINE configs[4] at its real shape — 11x11 pixels x 3500 cadences, pld_order 3, 16 components, all pixels —
    against the REFERENCE's PLDCorrector output (tests/golden/pld_c5.npz, made by oracle/gen_golden.py from lightkurve
    itself; the synthetic cutouts are regenerated here and checked by SHA-256)."""
    import hashlib
    from lightkurve_amd import synth
    g = golden("pld_c5")
    n = int(g["n_cutouts"])
    cubes = []
    for i in range(n):
        t, flux, err, _ = synth.pld_cutout(4, i, n=3500, npix=11)
        assert hashlib.sha256(t.tobytes() + flux.tobytes() + err.tobytes()).hexdigest() == str(g["sha_%d" % i])
        cubes.append(PixelCube(g["time_%d" % i], flux, err, mission="K2"))
    corrected, outl = pld_correct_batch(cubes, pld_order=3, pca_components=16)
    for i in range(n):
        assert np.array_equal(outl[i], g["outlier_mask_%d" % i]), i
        assert np.max(np.abs(corrected[i] - g["corrected_%d" % i])) / np.median(g["corrected_%d" % i]) < 1e-6, i


def test_batch_of_cutouts_vs_oracle():
    """config[4] shape at reduced cadence count: 4 cutouts 11x11, order 3, 16 components, all pixels."""
    from lightkurve_amd import synth
    cubes, refs = [], []
    for i in range(4):
        t, flux, err, truth = synth.pld_cutout(4, i, n=1000, npix=11)
        cubes.append(PixelCube(t, flux, err, mission="K2"))
    corrected, outl = pld_correct_batch(cubes, pld_order=3, pca_components=16)
    allm = np.ones((11, 11), bool)
    for i, c in enumerate(cubes):
        r = O.pld_correct(c.time, c.flux, c.flux_err, allm, allm, allm, pld_order=3, pca_components=16, spline_degree=5)
        assert np.array_equal(outl[i], r["outlier_mask"]), i
        assert np.max(np.abs(corrected[i] - r["corrected"])) / np.median(r["corrected"]) < 1e-6, i


@pytest.mark.parametrize("order,comps,npix", [(2, 24, 11), (3, 20, 9), (2, 40, 11)])
def test_wide_bases_vs_oracle(order, comps, npix):
    """More than 16 PCA components: the 40- to 56-wide subspace (4 column tiles per wave in the eigen-solver, 2 to 3 in
    the projection), product blocks of 300 / 1540 / 820 columns through the moment-form Gram (the 12-register stage
    variant of pld_moment_gram_kernel: k1 > 16)."""
    from lightkurve_amd import synth
    cubes = []
    for i in range(2):
        t, flux, err, truth = synth.pld_cutout(4, 10 + i, n=700, npix=npix)
        cubes.append(PixelCube(t, flux, err, mission="K2"))
    corrected, outl = pld_correct_batch(cubes, pld_order=order, pca_components=comps)
    allm = np.ones((npix, npix), bool)
    for i, c in enumerate(cubes):
        r = O.pld_correct(c.time, c.flux, c.flux_err, allm, allm, allm, pld_order=order, pca_components=comps,
                          spline_degree=5)
        assert np.array_equal(outl[i], r["outlier_mask"]), i
        assert np.max(np.abs(corrected[i] - r["corrected"])) / np.median(r["corrected"]) < 1e-6, i


def test_fourth_order_vs_oracle():
    """pld_order = 4 with 16 components: 3876 product columns (the widest block the path admits) through the order-4
    instantiations of the moment-form Gram (8th moments), its 15-million-entry index table and the product projection."""
    from lightkurve_amd import synth
    t, flux, err, truth = synth.pld_cutout(4, 30, n=400, npix=5)
    cube = PixelCube(t, flux, err, mission="K2")
    corrected, outl = pld_correct_batch([cube], pld_order=4, pca_components=16)
    allm = np.ones((5, 5), bool)
    r = O.pld_correct(cube.time, cube.flux, cube.flux_err, allm, allm, allm, pld_order=4, pca_components=16, spline_degree=5)
    assert np.array_equal(outl[0], r["outlier_mask"])
    assert np.max(np.abs(corrected[0] - r["corrected"])) / np.median(r["corrected"]) < 1e-6


@pytest.mark.parametrize("npix", [14, 15])
def test_large_cutouts_vs_oracle(npix):
    """196 / 225 PLD pixels: the pixel and background blocks go through the 128 x 128-tile Gram kernel (gram128_kernel,
    even and odd column counts) and the subspace iteration on a 196- / 225-column Gram matrix; the product block of the
    second order through the moment form."""
    from lightkurve_amd import synth
    cubes = []
    for i in range(2):
        t, flux, err, truth = synth.pld_cutout(4, 20 + i, n=600, npix=npix)
        cubes.append(PixelCube(t, flux, err, mission="K2"))
    corrected, outl = pld_correct_batch(cubes, pld_order=2, pca_components=16)
    allm = np.ones((npix, npix), bool)
    for i, c in enumerate(cubes):
        r = O.pld_correct(c.time, c.flux, c.flux_err, allm, allm, allm, pld_order=2, pca_components=16, spline_degree=5)
        assert np.array_equal(outl[i], r["outlier_mask"]), i
        assert np.max(np.abs(corrected[i] - r["corrected"])) / np.median(r["corrected"]) < 1e-6, i


def test_golden_sparse_design_matrix_branch(golden):
    """PLDCorrector.correct(sparse=True) (reference pldcorrector.py:194-199, regressioncorrector.py:170-176): the sparse
    collection carries a different spline basis (create_sparse_spline_matrix); densified and fitted on the GPU."""
    from lightkurve_amd.correctors import SparseDesignMatrixCollection
    g = golden("pld_k2sin_order3_sparse")
    pld = PLDCorrector(PixelCube(g["time"], g["flux"], g["flux_err"]))
    clc = pld.correct(pld_order=3, pca_components=16, pld_aperture_mask="all", normalize_background_pixels=True, sparse=True)
    dmc = pld.design_matrix_collection
    assert isinstance(dmc, SparseDesignMatrixCollection)
    assert dmc.X.shape == g["X"].shape
    w = int(g["block_widths"][-1])
    assert np.max(np.abs(dmc.X[:, -w:] - g["X"][:, -w:])) < 1e-12      # the sparse spline block: the de Boor kernel on the
    #                                                                    reference's knots vs its Python recursion (4e-16)
    assert np.allclose(dmc.prior_sigma, g["prior_sigma"], rtol=1e-6)
    assert np.array_equal(pld.outlier_mask, g["outlier_mask"])
    assert np.max(np.abs(clc.flux - g["corrected"])) / np.median(g["corrected"]) < 1e-6


def test_pixel_periodograms_batch_vs_reference(golden):
    """PixelCube.pixel_periodograms == the loop of TargetPixelFile.plot_pixels(periodogram=True) (per pixel: one-pixel
    aperture, remove_outliers, to_periodogram) run by lightkurve itself: same surviving cadences, same default grids,
    power within 1e-9 of the peak; with a shared grid all pixels are one launch."""
    g = golden("pixel_pg")
    cube = PixelCube(g["time"], g["flux"], g["flux_err"])
    pgs = cube.pixel_periodograms()
    assert len(pgs) == 49
    for j in g["pixels"]:
        pg = pgs[int(j)]
        assert pg is not None and len(pg.frequency) == len(g["freq_%d" % j])
        assert np.allclose(pg.frequency, g["freq_%d" % j], rtol=1e-13, atol=0)
        ref = g["power_%d" % j]
        ok = np.isfinite(ref)
        assert np.array_equal(ok, np.isfinite(pg.power))
        assert np.max(np.abs(pg.power[ok] - ref[ok])) / np.max(ref[ok]) < 1e-9
    grid = np.linspace(0.1, 20, 1500)
    pgs2 = cube.pixel_periodograms(frequency=grid)
    for j in g["pixels"]:
        ref = g["power_grid_%d" % j]
        assert np.max(np.abs(pgs2[int(j)].power - ref)) / np.max(ref) < 1e-9


@pytest.mark.parametrize("fname,tag,kw", [
    ("kepler_tpf.fits", "ktpf_default", {}),
    ("kepler_tpf.fits", "ktpf_none", dict(quality_bitmask="none")),
    ("tess_tpf.fits", "ttpf_default", {}),
    ("tess_tpf.fits", "ttpf_none", dict(quality_bitmask="none")),
    ("tess_tpf.fits", "ttpf_hard", dict(quality_bitmask="hard")),
])
def test_target_pixel_file_to_cube_vs_reference_classes(golden, fname, tag, kw):
    """FITS target-pixel file -> PixelCube on the device (lk_fits_unpack_cube): time, quality, flux / flux_err / flux_bkg
    cubes and the pipeline mask bit-identical to KeplerTargetPixelFile / TessTargetPixelFile on the same file."""
    import os
    g = golden("fits_ingest")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fits", fname)
    cube = PixelCube.from_fits(path, **kw)
    assert np.array_equal(cube.time, g[tag + "_time"], equal_nan=True)
    assert np.array_equal(cube.quality, g[tag + "_quality"])
    assert cube.flux.dtype == np.float32 and np.array_equal(cube.flux, g[tag + "_flux"], equal_nan=True)
    assert np.array_equal(cube.flux_err, g[tag + "_flux_err"], equal_nan=True)
    assert np.array_equal(cube.flux_bkg, g[tag + "_flux_bkg"], equal_nan=True)
    assert np.array_equal(cube.pipeline_mask, g[tag + "_pipeline_mask"])
    assert cube.meta["MISSION"] in ("K2", "TESS") and cube.meta["TARGETID"] == 4242
