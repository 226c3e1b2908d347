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
    # order 3 (816 -> 16): the iteration stops at a 1e-7 residual and the 16th / 17th eigenvalues are close — numerically equivalent
    # builds of the first-order solver move this gap between 8e-6 and 1.4e-4 (tools/pld_gap_check.py prints it; shipped: 2.4e-5)
    assert subspace_gap(X[:, 32:48], g["X"][:, 32:48]) < 1e-4
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


def test_one_call_correct_equals_design_then_regress():
    """lk_pld_correct_batch (design matrices kept in device memory) against the two host-pointer calls it fuses,
    lk_pld_design_batch + lk_regress_batch: same kernels on the same inputs, so coefficients, model and outlier mask are the
    same bits; the spline block's share of the model against numpy on the downloaded design matrix."""
    from lightkurve_amd import _capi, synth
    from lightkurve_amd.correctors.pldcorrector import _percentile_knots
    B, n, npix = 3, 900, 7
    pix = np.empty((B, n, npix * npix), np.float32)
    t = np.empty((B, n))
    y, err = np.empty((B, n)), np.empty((B, n))
    for b in range(B):
        tb, flux, ferr, _ = synth.pld_cutout(4, 40 + b, n=n, npix=npix)
        pix[b], t[b] = flux.reshape(n, -1), tb
        y[b], err[b] = flux.reshape(n, -1).sum(axis=1), np.sqrt((ferr.reshape(n, -1).astype(np.float64) ** 2).sum(axis=1))
    lcf = y.astype(np.float32)
    knots = np.stack([_percentile_knots(t[b], n // 50, 5) for b in range(B)])
    cm = np.ones((B, n), bool)
    cm[1, 100:140] = False
    for cmask in (None, cm):
        X, ps = _capi.pld_design_batch(pix, pix, lcf, t, knots, 2, 8, 5, True)
        K = X.shape[2]
        two = _capi.regress_batch(X.reshape(B * n, K), y.ravel(), np.arange(B + 1) * n, err=err.ravel(),
                                  cadence_mask=None if cmask is None else cmask.ravel(), prior_mu=np.zeros((B, K)),
                                  prior_sigma=ps, sigma=5, niters=5)
        one = _capi.pld_correct_batch(pix, pix, lcf, t, knots, y, err, 2, 8, 5, True, cadence_mask=cmask, sigma=5, niters=5)
        assert np.array_equal(one["coefficients"], two["coefficients"])
        assert np.array_equal(one["model"].ravel(), two["model"])
        assert np.array_equal(one["outlier_mask"].ravel(), two["outlier_mask"])
        nsp = n // 50 + 1
        sp = np.einsum("bnk,bk->bn", X[:, :, K - nsp:], two["coefficients"][:, K - nsp:])
        assert np.max(np.abs(one["spline"] - sp)) <= 1e-12 * np.max(np.abs(sp))
    # separate (copied) pixel arrays for the two blocks take the two-upload route: same result
    sep = _capi.pld_correct_batch(pix, pix.copy(), lcf, t, knots, y, err, 2, 8, 5, True, cadence_mask=cm, sigma=5, niters=5)
    assert np.array_equal(sep["model"], one["model"]) and np.array_equal(sep["coefficients"], one["coefficients"])
    no_sp = _capi.pld_correct_batch(pix, pix, lcf, t, knots, y, err, 2, 8, 5, True, cadence_mask=cm, want_spline=False)
    assert no_sp["spline"] is None and np.array_equal(no_sp["model"], one["model"])


def test_batch_front_end_matches_the_per_object_corrector():
    """pld_correct_batch's own host work (aperture sums, NaN-cadence removal, pixel gathers on the thread pool) against
    PLDCorrector(c).correct(...) per cutout — with a partial aperture, distinct PLD / background masks, cadences whose flux
    is all NaN or all zero (dropped by the corrector, pldcorrector.py:109-120) and restore_trend on and off."""
    from lightkurve_amd import synth
    rng = np.random.default_rng(8)
    ap = np.zeros((9, 9), bool)
    ap[2:7, 2:7] = True
    pm = np.zeros((9, 9), bool)
    pm[1:8, 1:8] = True
    bm = ~ap
    cubes = []
    for i in range(3):
        t, flux, err, _ = synth.pld_cutout(4, 50 + i, n=800, npix=9)
        flux, err = flux.copy(), err.copy()
        bad = rng.choice(800, 7, replace=False)       # the same COUNT of dropped cadences per cutout, at different places
        flux[bad[:4]] = np.nan
        flux[bad[4:]] = 0.0
        cubes.append(PixelCube(t, flux, err, mission="K2"))
    for restore in (True, False):
        corrected, outl = pld_correct_batch(cubes, aperture_mask=ap, pld_aperture_mask=pm, background_aperture_mask=bm,
                                            pld_order=2, pca_components=8, restore_trend=restore)
        assert corrected.shape == (3, 793)
        for i, c in enumerate(cubes):
            pld = PLDCorrector(c, aperture_mask=ap)
            clc = pld.correct(pld_order=2, pca_components=8, pld_aperture_mask=pm, background_aperture_mask=bm,
                              normalize_background_pixels=True, restore_trend=restore)
            assert len(clc.flux) == 793
            assert np.array_equal(outl[i], pld.outlier_mask), i
            assert np.max(np.abs(corrected[i] - clc.flux)) <= 1e-9 * np.median(clc.flux), i
    with pytest.raises(ValueError):
        pld_correct_batch(cubes + [PixelCube(cubes[0].time[:-1], cubes[0].flux[:-1], cubes[0].flux_err[:-1])], pld_order=2,
                          pca_components=8)


def test_batch_resolves_data_dependent_masks_per_cutout():
    """``aperture_mask=None`` (the reference's default: ``create_threshold_mask(3)`` of EACH target-pixel file,
    pldcorrector.py:99-107), 'threshold' and 'background' are functions of a cutout's own pixels — the batch must give every
    cutout the SAP aperture (and NaN-cadence set) its own ``PLDCorrector(tpf)`` would, not the first cutout's."""
    from lightkurve_amd import synth
    cubes = []
    for i in range(3):
        t, flux, err, _ = synth.pld_cutout(4, 70 + i, n=700, npix=9)
        sh = i - 1                                     # the star sits on a different pixel in every cutout
        cubes.append(PixelCube(t, np.roll(flux, (sh, -sh), axis=(1, 2)), np.roll(err, (sh, -sh), axis=(1, 2)), mission="K2"))
    masks = [c.create_threshold_mask(3) for c in cubes]
    assert not (np.array_equal(masks[0], masks[1]) and np.array_equal(masks[0], masks[2]))
    corrected, outl = pld_correct_batch(cubes, aperture_mask=None, pld_order=2, pca_components=8)
    for i, c in enumerate(cubes):
        pld = PLDCorrector(c)                          # aperture_mask=None -> this cutout's own threshold mask
        assert np.array_equal(pld.aperture_mask, masks[i])
        clc = pld.correct(pld_order=2, pca_components=8, pld_aperture_mask="all", background_aperture_mask="all",
                          normalize_background_pixels=True)
        assert np.array_equal(outl[i], pld.outlier_mask), i
        assert np.max(np.abs(corrected[i] - clc.flux)) <= 1e-9 * np.median(clc.flux), i
    # the order of the batch does not matter any more (before: every cutout got cubes[0]'s aperture)
    c2, _ = pld_correct_batch(cubes[::-1], aperture_mask=None, pld_order=2, pca_components=8)
    assert np.array_equal(c2[::-1], corrected)
    # data-dependent PLD / background masks: per cutout too; one design-matrix width per call
    nb = [int((~c.create_threshold_mask(threshold=0, reference_pixel=None)).sum()) for c in cubes]
    if len(set(nb)) > 1:
        with pytest.raises(ValueError, match="different numbers of pixels"):
            pld_correct_batch(cubes, aperture_mask=None, background_aperture_mask="background", pld_order=2, pca_components=8)
    else:
        cb, ob = pld_correct_batch(cubes, aperture_mask=None, background_aperture_mask="background", pld_order=2, pca_components=8)
        for i, c in enumerate(cubes):
            pld = PLDCorrector(c)
            clc = pld.correct(pld_order=2, pca_components=8, pld_aperture_mask="all", background_aperture_mask="background",
                              normalize_background_pixels=True)
            assert np.array_equal(ob[i], pld.outlier_mask) and np.max(np.abs(cb[i] - clc.flux)) <= 1e-9 * np.median(clc.flux)


def _pld_first_block(pix32, flux32, k, N):
    """X[:, :k] of a first-order PLD design matrix (PCA(PCA(pixels / flux)), pldcorrector.py:233-262) for one cutout."""
    from lightkurve_amd import _capi
    t = np.linspace(0.0, 30.0, N)
    nkn, deg = 8, 3
    knots = np.concatenate([[t.min()], np.percentile(t, np.linspace(0, 100, nkn - deg - 1 + 2)[1:-1]), [t.max()]])
    X, _ = _capi.pld_design_batch(pix32[None], pix32[None, :, :4], flux32[None], t[None], knots[None], 1, k, deg, True)
    return X[0][:, :k]


def test_pld_pca_block_backward_error_on_a_near_degenerate_spectrum():
    """the PLD blocks' subspace iteration stops at a relative residual of 1e-7 (pld.hip PLD_EIG_TOL; the goldens
    do not move down to 1e-6) — bound the SUBSPACE error itself where it is hardest: singular values 8, 9 and 10 agree to
    1e-4, so a principal-angle test against an SVD is ill-posed there, but the backward error is not: with G = A A^T of the
    centred float32 ratios and Q the returned basis, ||G Q - Q (Q^T G Q)||_F <= 1e-6 ||G||_2 sqrt(k) (stated), Q orthonormal to
    1e-10.  A well-separated spectrum is then compared with the exact SVD: ||P_svd - P_got||_2 <= 1e-6.  P = 180 columns >
    138: the blocks take pld_topk_eig_kernel, not the direct tridiagonal solver."""
    rng = np.random.default_rng(17)
    N, P, k = 900, 180, 8
    U0 = np.linalg.qr(rng.standard_normal((N, P)))[0]
    V0 = np.linalg.qr(rng.standard_normal((P, P)))[0]
    flux = (1000.0 * (1.0 + 0.02 * np.sin(np.linspace(0, 20, N)))).astype(np.float32)
    from lightkurve_amd import _capi
    h = _capi.Handle.get(0)
    for name, s in (("near-degenerate", np.r_[5, 4.5, 4, 3.5, 3, 2.5, 2.0, 1.5, 1.4999, 1.4998, 0.3 * rng.random(P - 10)]),
                    ("separated", np.r_[5, 4.5, 4, 3.5, 3, 2.5, 2.0, 1.5, 0.3 * rng.random(P - 8)])):
        A = (U0 * s) @ V0.T
        pix32 = ((1.0 + 0.01 * A) * flux[:, None].astype(np.float64)).astype(np.float32)
        ratio = (pix32 / flux[:, None]).astype(np.float64)            # float32 division, as the reference's numpy does
        Ac = ratio - ratio.mean(axis=0)
        G = Ac @ Ac.T
        Us, ss, _ = np.linalg.svd(Ac, full_matrices=False)
        for tol in (0.0, 1e-10):
            h.pld_set_eig_tolerance(tol)
            try:
                Q = _pld_first_block(pix32, flux, k, N)
            finally:
                h.pld_set_eig_tolerance(0.0)
            assert np.max(np.abs(Q.T @ Q - np.eye(k))) < 1e-10, name
            R = G @ Q - Q @ (Q.T @ G @ Q)
            bound = (1e-6 if tol == 0.0 else 1e-8) * ss[0] ** 2 * np.sqrt(k)
            assert np.linalg.norm(R) <= bound, (name, tol, np.linalg.norm(R) / (ss[0] ** 2 * np.sqrt(k)))
            assert np.trace(Q.T @ G @ Q) >= np.sum(ss[:k] ** 2) * (1 - 1e-5), name
            if name == "separated":
                assert np.linalg.norm(Us[:, :k] @ Us[:, :k].T - Q @ Q.T, 2) <= 1e-6


@pytest.mark.parametrize("order,comps,npix", [(3, 16, 11), (2, 24, 11), (2, 16, 15)])
def test_phase_split_eigen_iteration_matches_the_one_kernel_form(order, comps, npix):
    """lk_pld_set_eig_mode(1): the subspace iteration of the wide PCA blocks as one launch per phase (pld_eigs_* kernels) against
    the default one-kernel form: corrected flux equal to 1e-9 between the two (the PCA bases may differ by a rotation inside a
    block, the regression does not see it — SURVEY App. B.8), identical outlier masks, both within the stated 1e-6 of the oracle.
    Shapes: the 816-column third-order block (float32 matrix cores, 16 + 16 basis columns), a 300-column second-order block with
    24 + 16 basis columns (the four-tile basis), 15 x 15 cutouts whose 225-pixel blocks have no float32 copy of C (float64
    products only, mirrored Gram blocks)."""
    from lightkurve_amd import synth, _capi
    cubes = []
    for i in range(3):
        t, flux, err, truth = synth.pld_cutout(4, 40 + i, n=700, npix=npix)
        cubes.append(PixelCube(t, flux, err, mission="K2"))
    h = _capi.Handle.get(0)
    out = {}
    for mode in (0, 1):
        h.pld_set_eig_mode(mode)
        try:
            out[mode] = pld_correct_batch(cubes, pld_order=order, pca_components=comps)
        finally:
            h.pld_set_eig_mode(0)
    allm = np.ones((npix, npix), bool)
    for i, c in enumerate(cubes):
        assert np.array_equal(out[0][1][i], out[1][1][i]), i
        med = np.median(out[0][0][i])
        assert np.max(np.abs(out[0][0][i] - out[1][0][i])) / med < 1e-9, i
        r = O.pld_correct(c.time, c.flux, c.flux_err, allm, allm, allm, pld_order=order, pca_components=comps, spline_degree=5)
        assert np.array_equal(out[1][1][i], r["outlier_mask"]), i
        assert np.max(np.abs(out[1][0][i] - r["corrected"])) / np.median(r["corrected"]) < 1e-6, i
