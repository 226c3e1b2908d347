"""PLDCorrector over the GPU design-matrix + regression path
(reference: src/lightkurve/correctors/pldcorrector.py:109-120, 125-287, 304-427).

The reference takes a ``TargetPixelFile``; that container (FITS I/O, WCS, plotting) is out of scope, so the mirror
takes a ``PixelCube``: time[N], flux[N, ny, nx], flux_err[N, ny, nx] and an optional mission string — the only
things ``PLDCorrector`` reads from the TPF — with the two TPF helpers it calls
(``create_threshold_mask`` targetpixelfile.py:680-742, aperture photometry :868-923)."""
import logging

import numpy as np

from .. import _capi
from ..lightcurve import LightCurve
from .designmatrix import (DesignMatrix, DesignMatrixCollection, SparseDesignMatrixCollection,
                           create_sparse_spline_matrix)
from .regressioncorrector import RegressionCorrector

log = logging.getLogger(__name__)

__all__ = ["PixelCube", "PLDCorrector", "pld_correct_batch"]


class PixelCube(object):
    """Minimal stand-in for the slice of TargetPixelFile that PLDCorrector touches."""

    def __init__(self, time, flux, flux_err, mission=None, targetid=None):
        self.time = np.asarray(time, dtype=np.float64)
        self.flux = np.asarray(flux)
        self.flux_err = np.asarray(flux_err)
        if self.flux.ndim != 3 or self.flux.shape != self.flux_err.shape or len(self.time) != len(self.flux):
            raise ValueError("flux and flux_err must be (cadence, row, column) cubes matching time")
        self.meta = {"MISSION": mission, "TARGETID": targetid}

    @classmethod
    def from_fits(cls, path, quality_bitmask="default", device=0):
        """A Kepler / K2 / TESS target-pixel file -> the arrays PLDCorrector works on, with the reference's cadence
        selection (``KeplerTargetPixelFile`` / ``TessTargetPixelFile``: targetpixelfile.py:2120-2122, 2794-2801).  Headers
        are parsed on the host (fitsio.py), the table's bytes are turned into time / quality / float32 cubes on the GPU
        (``lk_fits_unpack_cube``).  Also sets ``flux_bkg`` (when the file has it), ``quality`` and ``pipeline_mask``
        (aperture extension & 2, targetpixelfile.py:308-322)."""
        from .. import _capi, fitsio
        tab = fitsio.read_fits_table(path, ext=1)
        pc = fitsio.pixel_columns(tab, quality_bitmask=quality_bitmask)
        t, q, cubes = _capi.fits_unpack_cube(tab.raw, pc["off_time"], pc["code_time"], pc["off_quality"], pc["code_quality"],
                                             pc["bitmask"], pc["keep_nan_time"], pc["col_offsets"], pc["npix"], device=device)
        shape = (len(t),) + tuple(pc["shape"])
        by_name = {n: cubes[i].reshape(shape) for i, n in enumerate(pc["columns"])}
        flux = by_name["flux"]
        err = by_name.get("flux_err", np.full(shape, np.nan, dtype=np.float32))
        mission = tab.primary.get("MISSION", tab.primary.get("TELESCOP"))
        out = cls(t, flux, err, mission=mission, targetid=tab.primary.get("KEPLERID", tab.primary.get("TICID")))
        out.quality = q
        out.flux_bkg = by_name.get("flux_bkg")
        aper = fitsio.read_fits_image(path, 2)
        out.pipeline_mask = (aper & 2) > 0 if aper is not None and aper.shape == shape[1:] else np.ones(shape[1:], dtype=bool)
        out.meta.update({"FILENAME": str(path), "QUALITY_BITMASK": quality_bitmask, "LABEL": tab.primary.get("OBJECT")})
        return out

    @property
    def shape(self):
        return self.flux.shape

    def __getitem__(self, key):
        c = PixelCube(self.time[key], self.flux[key], self.flux_err[key])
        c.meta = dict(self.meta)
        return c

    def create_threshold_mask(self, threshold=3, reference_pixel="center"):
        """Pixels whose median flux exceeds median + threshold * 1.4826 * MAD; with a reference pixel, only the
        4-connected region closest to it (targetpixelfile.py:680-742)."""
        if reference_pixel == "center":
            reference_pixel = (self.shape[2] / 2, self.shape[1] / 2)
        with np.errstate(all="ignore"):
            median_image = np.nanmedian(np.asarray(self.flux, dtype=np.float64), axis=0)
        vals = median_image[np.isfinite(median_image)].flatten()
        mad = np.median(np.abs(vals - np.median(vals)))
        mad_cut = (1.4826 * mad * threshold) + np.nanmedian(median_image)
        mask = np.nan_to_num(median_image) >= mad_cut
        if reference_pixel is None or not mask.any():
            return mask
        labels = np.zeros(mask.shape, dtype=int)
        current = 0
        for r in range(mask.shape[0]):
            for c in range(mask.shape[1]):
                if mask[r, c] and labels[r, c] == 0:
                    current += 1
                    labels[r, c] = current
                    stack = [(r, c)]
                    while stack:
                        a, b = stack.pop()
                        for da, db in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                            aa, bb = a + da, b + db
                            if (0 <= aa < mask.shape[0] and 0 <= bb < mask.shape[1] and mask[aa, bb]
                                    and labels[aa, bb] == 0):
                                labels[aa, bb] = current
                                stack.append((aa, bb))
        args = np.argwhere(labels > 0)
        dist = [np.hypot(a[0] - reference_pixel[1], a[1] - reference_pixel[0]) for a in args]
        closest = args[int(np.argmin(dist))]
        return labels == labels[closest[0], closest[1]]

    def _parse_aperture_mask(self, aperture_mask):
        """'all' / None, 'threshold', 'background' (= not threshold), 'empty', or a boolean (row, column) array
        (targetpixelfile.py:603-678; 'pipeline' needs the FITS aperture extension and is not available here)."""
        if aperture_mask is None or (isinstance(aperture_mask, str) and aperture_mask == "all"):
            return np.ones(self.shape[1:], dtype=bool)
        if isinstance(aperture_mask, str):
            if aperture_mask == "threshold":
                return self.create_threshold_mask()
            if aperture_mask == "background":
                return ~self.create_threshold_mask(threshold=0, reference_pixel=None)
            if aperture_mask == "empty":
                return np.zeros(self.shape[1:], dtype=bool)
            raise ValueError("aperture_mask '{}' is not supported here".format(aperture_mask))
        m = np.asarray(aperture_mask, dtype=bool)
        if m.shape != self.shape[1:]:
            raise ValueError("`aperture_mask` has shape {}, but the flux data has shape {}".format(m.shape,
                                                                                                 self.shape[1:]))
        return m

    def to_lightcurve(self, aperture_mask=None):
        """Simple aperture photometry, flux_method='sum' (targetpixelfile.py:868-923); numpy keeps float32 cubes
        float32 here exactly like the reference does."""
        flux, flux_err = self._aperture_sums(self._parse_aperture_mask(aperture_mask))
        lc = LightCurve(time=self.time, flux=flux, flux_err=flux_err, meta=dict(self.meta))
        lc._flux_f32 = np.asarray(flux, dtype=np.float32)   # the value the reference carries (float32 for FITS cubes)
        return lc


    def _aperture_sums(self, ap):
        """(flux, flux_err) of the aperture `ap` (boolean image) per cadence, in the cube's dtype.  The gather `cube[:, ap]`
        is kept even for a full aperture: numpy lays its result out pixel-major, so the float32 sums below add the pixels
        of a cadence one after the other — reshaping the cube instead would sum pairwise and round differently (1e-7)."""
        n = len(self.time)
        with np.errstate(all="ignore"):
            pix = self.flux[:, ap]
            perr = self.flux_err[:, ap]
            flux = np.asarray(np.sum(pix, axis=1))
            flux_err = np.sum(perr ** 2, axis=1)
            if pix.shape[1] and np.all(np.isfinite(flux)) and np.all(np.isfinite(flux_err)):
                # every aperture pixel is finite (a NaN / inf would have reached its row's sum): nansum == sum, no cadence
                # is empty, and only a row that sums to zero can be all zeros
                for r in np.flatnonzero(flux == 0):
                    if np.all(self.flux[r] == 0):
                        flux[r] = np.nan
                return flux, flux_err ** 0.5
            flux = np.asarray(np.nansum(pix, axis=1))
            flux[~np.any(np.isfinite(pix), axis=1)] = np.nan
            flux[np.all(self.flux.reshape(n, -1) == 0, axis=1)] = np.nan
            flux_err = np.nansum(perr ** 2, axis=1) ** 0.5
        return flux, flux_err

    def pixel_periodograms(self, frequency=None, corrector="remove_outliers", sigma=5.0, normalization="amplitude",
                           ls_method="fast", device=0, **kwargs):
        """One Lomb-Scargle periodogram per pixel — the loop of ``TargetPixelFile.plot_pixels(periodogram=True)``
        (reference src/lightkurve/targetpixelfile.py:1958-1974: per pixel a one-pixel aperture light curve,
        ``corrector_func`` (default ``remove_outliers``), ``to_periodogram(**kwargs)``) — as batched GPU calls: ONE sigma
        clip over all pixels and ONE periodogram launch per group of pixels that share a frequency grid (all of them when
        ``frequency`` is given; otherwise each pixel's default grid depends on which of its cadences survived the clip).
        Returns a list (row-major over the pixels) of ``LombScarglePeriodogram`` objects, ``None`` where a pixel has no
        usable cadence — the reference's ``pixel_list``."""
        from .. import _capi as capi
        from ..batch import lombscargle_batch
        from ..periodogram import LombScarglePeriodogram, _ls_plan
        n, nr, nc = self.shape
        flux = np.asarray(self.flux, dtype=np.float64).reshape(n, nr * nc)
        ferr = np.asarray(self.flux_err, dtype=np.float64).reshape(n, nr * nc)
        npx = nr * nc
        y = np.ascontiguousarray(flux.T).ravel()                       # pixel-major ragged batch, n cadences each
        off = np.arange(npx + 1, dtype=np.int64) * n
        if corrector == "remove_outliers":
            bad = capi.sigma_clip_batch(y, off, sigma=sigma, device=device).reshape(npx, n)
        elif corrector is None:
            bad = np.zeros((npx, n), bool)
        else:
            raise ValueError("corrector must be 'remove_outliers' or None on the batched path")
        lcs, idx = [], []
        for j in range(npx):
            keep = ~bad[j]
            if not np.any(keep & np.isfinite(flux[:, j])):
                continue
            lcs.append(LightCurve(time=self.time[keep], flux=flux[keep, j], flux_err=ferr[keep, j], meta=dict(self.meta)))
            idx.append(j)
        out = [None] * npx
        # group the pixels by frequency grid (one launch per group)
        groups = {}
        plans = {}
        for lc, j in zip(lcs, idx):
            try:
                plan = _ls_plan(lc, frequency=frequency, normalization=normalization, ls_method=ls_method, **kwargs)
            except IndexError:          # the reference appends None for a light curve too short for a grid
                continue
            plans[j] = plan
            groups.setdefault(plan["frequency"].tobytes(), []).append(j)
        by_pix = dict(zip(idx, lcs))
        for key, members in groups.items():
            p0 = plans[members[0]]
            power = lombscargle_batch([by_pix[j] for j in members], p0["frequency"], normalization=normalization,
                                      freq_unit=p0["freq_unit"], ls_method=ls_method, device=device, gather=False,
                                      oversample_factor=kwargs.get("oversample_factor"), nterms=kwargs.get("nterms", 1))
            for row, j in zip(power, members):
                pl = plans[j]
                out[j] = LombScarglePeriodogram(frequency=pl["frequency"], power=row, nyquist=pl["nyquist"],
                                                default_view=pl["default_view"], ls_method=pl["ls_method"],
                                                nterms=pl["nterms"], meta=pl["lc"].meta,
                                                frequency_unit=pl["freq_unit"], power_unit=pl["power_unit"])
        return out


def _percentile_knots(time, n_knots, degree):
    """[min, interior knots, max] of patsy's bs(x, df=n_knots, degree, include_intercept=True)."""
    order = degree + 1
    n_inner = n_knots - order
    if n_inner < 0:
        raise ValueError("df={} is too small for degree={}; must be >= {}".format(n_knots, degree, order))
    inner = np.percentile(time, np.linspace(0, 100, n_inner + 2)[1:-1]) if n_inner > 0 else np.zeros(0)
    return np.concatenate([[np.min(time)], inner, [np.max(time)]])


def _finite_columns(cube2d):
    return cube2d[:, np.all(np.isfinite(cube2d), axis=0)]


class PLDCorrector(RegressionCorrector):
    def __init__(self, tpf, aperture_mask=None):
        if aperture_mask is None:
            aperture_mask = tpf.create_threshold_mask(3)
        self.aperture_mask = tpf._parse_aperture_mask(aperture_mask)
        lc = tpf.to_lightcurve(aperture_mask=self.aperture_mask)
        nan_mask = np.isnan(lc.flux) | np.isnan(lc.flux_err)
        f32 = lc._flux_f32[~nan_mask]
        lc = lc[~nan_mask]
        lc._flux_f32 = f32
        self.tpf = tpf[~nan_mask]
        super().__init__(lc=lc)

    def __repr__(self):
        return "PLDCorrector (ID: {})".format(self.lc.label)

    def _resolve(self, pld_order, pca_components, pld_aperture_mask, normalize_background_pixels):
        """mission defaults of .correct() (pldcorrector.py:382-403)."""
        k2 = self.tpf.meta.get("MISSION") == "K2"
        if pld_order is None:
            pld_order = 3 if k2 else 1
        if pca_components is None:
            pca_components = 16 if k2 else 3
        if pld_aperture_mask is None:
            pld_aperture_mask = "threshold" if k2 else "empty"
        if normalize_background_pixels is None:
            normalize_background_pixels = bool(k2)
        return pld_order, pca_components, pld_aperture_mask, normalize_background_pixels

    def create_design_matrix(self, pld_order=3, pca_components=16, pld_aperture_mask=None,
                             background_aperture_mask="background", spline_n_knots=None, spline_degree=3,
                             normalize_background_pixels=None, sparse=False, device=0):
        """DesignMatrixCollection [pixel_series | background | spline] built on the GPU (one cutout = a batch of 1)."""
        if pca_components is None or pca_components < 1:
            raise NotImplementedError("pca_components must be >= 1 on the HIP path")
        # None -> all pixels, exactly as the reference's create_design_matrix treats it (pldcorrector.py:203-207;
        # correct() resolves the mission-dependent default before it calls this)
        self.pld_aperture_mask = self.tpf._parse_aperture_mask(pld_aperture_mask)
        self.background_aperture_mask = self.tpf._parse_aperture_mask(background_aperture_mask)
        n = len(self.lc)
        if spline_n_knots is None:
            spline_n_knots = int(n / 50)
        if normalize_background_pixels is None:
            normalize_background_pixels = False
        cube = self.tpf.flux
        pld_pix = _finite_columns(cube[:, self.pld_aperture_mask].reshape(n, -1))
        bkg_pix = _finite_columns(cube[:, self.background_aperture_mask].reshape(n, -1))
        knots = _percentile_knots(self.lc.time, spline_n_knots, spline_degree)
        X, ps = _capi.pld_design_batch(pld_pix[None] if pld_pix.shape[1] else None, bkg_pix[None],
                                       self.lc._flux_f32[None], self.lc.time[None], knots[None], pld_order,
                                       pca_components, spline_degree, normalize_background_pixels, device=device)
        X, ps = X[0], ps[0]
        nsp = spline_n_knots + 1
        kb = min(pca_components, bkg_pix.shape[1])
        npld = X.shape[1] - nsp - kb
        mats = []
        if npld > 0:
            mats.append(DesignMatrix(X[:, :npld], name="pixel_series", prior_sigma=ps[:npld]))
        mats.append(DesignMatrix(X[:, npld:npld + kb], name="background", prior_sigma=ps[npld:npld + kb]))
        if sparse:
            # reference :194-199, 226-230: the sparse collection carries a DIFFERENT spline basis
            # (create_sparse_spline_matrix); pixel and background blocks are the same PCA'd matrices
            sp = create_sparse_spline_matrix(self.lc.time, n_knots=spline_n_knots, degree=spline_degree).append_constant()
            sp.prior_sigma = np.ones(sp.shape[1]) * ps[-1]
            mats.append(sp)
            return SparseDesignMatrixCollection(mats)
        mats.append(DesignMatrix(X[:, npld + kb:], name="spline", prior_sigma=ps[npld + kb:],
                                 columns=["knot{}".format(i + 1) for i in range(spline_n_knots)] + ["offset"]))
        return DesignMatrixCollection(mats)

    def correct(self, pld_order=None, pca_components=None, pld_aperture_mask=None,
                background_aperture_mask="background", spline_n_knots=None, spline_degree=5,
                normalize_background_pixels=None, restore_trend=True, sparse=False, cadence_mask=None, sigma=5,
                niters=5, propagate_errors=False, device=0):
        """Same contract and defaults as the reference (pldcorrector.py:304-427)."""
        self.restore_trend = restore_trend
        pld_order, pca_components, pld_aperture_mask, normalize_background_pixels = self._resolve(
            pld_order, pca_components, pld_aperture_mask, normalize_background_pixels)
        dm = self.create_design_matrix(pld_aperture_mask=pld_aperture_mask,
                                       background_aperture_mask=background_aperture_mask, pld_order=pld_order,
                                       pca_components=pca_components, spline_n_knots=spline_n_knots,
                                       spline_degree=spline_degree,
                                       normalize_background_pixels=normalize_background_pixels, sparse=sparse,
                                       device=device)
        clc = super().correct(dm, cadence_mask=cadence_mask, sigma=sigma, niters=niters,
                              propagate_errors=propagate_errors, device=device)
        if restore_trend:
            sp = self.diagnostic_lightcurves["spline"].flux
            clc.flux = clc.flux + (sp - np.median(sp))
        return clc


def _all_finite(a):
    """np.all(np.isfinite(a)) without the boolean temporary in the common case: a finite float64 total implies it."""
    with np.errstate(all="ignore"):
        return bool(np.isfinite(a.sum(dtype=np.float64))) or bool(np.all(np.isfinite(a)))


def _shared_mask(first, spec, sap=False):
    """A mask every cutout of the batch shares — given as an array, or by a name that depends on the cutout's SHAPE only
    ('all', 'empty'; None = all pixels for the PLD / background masks) — as a boolean image; None for the specs the reference
    evaluates on each target-pixel file's own DATA ('threshold', 'background'; None for the photometric aperture =
    ``create_threshold_mask(3)``, pldcorrector.py:99-107): those are resolved per cutout in ``_batch_cutout``."""
    if spec is None:
        return None if sap else first._parse_aperture_mask(None)
    if isinstance(spec, str) and spec in ("threshold", "background"):
        return None
    return first._parse_aperture_mask(spec)


def _batch_cutout(c, ap, pm, bm, specs):
    """One cutout's share of pld_correct_batch: what ``PLDCorrector(c, aperture_mask)`` keeps (the SAP light curve without its
    NaN cadences, pldcorrector.py:109-120) and the pixel series of the two apertures, without the object construction.
    ``ap`` / ``pm`` / ``bm``: shared boolean images, or None = evaluate ``specs`` (aperture, PLD, background) on THIS cutout as
    the per-object corrector does — the photometric aperture on the whole cutout (``PLDCorrector.__init__``), the other
    two on the cutout without its NaN cadences (``create_design_matrix`` sees ``self.tpf = tpf[~nan_mask]``).
    Returns (time, flux float64, flux_err float64, flux float32, pld pixels (n, P), background pixels (n, Pb))."""
    if ap is None:
        ap = c._parse_aperture_mask(c.create_threshold_mask(3) if specs[0] is None else specs[0])
    flux, ferr = c._aperture_sums(ap)
    keep = ~(np.isnan(flux) | np.isnan(ferr))
    everything = bool(keep.all())
    n_all = len(c.time)
    if pm is None or bm is None:
        ck = c if everything else c[keep]
        pm = ck._parse_aperture_mask(specs[1]) if pm is None else pm
        bm = ck._parse_aperture_mask(specs[2]) if bm is None else bm

    def pixels(mask):
        px = c.flux.reshape(n_all, -1) if bool(np.all(mask)) else c.flux[:, mask]
        return px if everything else px[keep]

    pld = pixels(pm)
    bkg = pld if (pm.shape == bm.shape and np.array_equal(pm, bm)) else pixels(bm)
    if not everything:
        flux, ferr = flux[keep], ferr[keep]
    return (c.time if everything else c.time[keep], np.asarray(flux, dtype=np.float64), np.asarray(ferr, dtype=np.float64),
            np.asarray(flux, dtype=np.float32), pld, bkg)


def pld_correct_batch(cubes, aperture_mask="all", pld_aperture_mask="all", background_aperture_mask="all",
                      pld_order=3, pca_components=16, spline_n_knots=None, spline_degree=5,
                      normalize_background_pixels=True, restore_trend=True, sigma=5, niters=5, device=0):
    """PLDCorrector(...).correct(...) for a list of same-shaped cutouts in ONE GPU call (``lk_pld_correct_batch``: design
    matrices, regression and the spline block's share of the model; the design matrices never leave the device).  Masks
    given as arrays (or 'all' / 'empty') are shared by the batch; the data-dependent ones — ``aperture_mask=None`` (the
    reference's default, ``create_threshold_mask(3)``), 'threshold', 'background' — are evaluated on EVERY cutout, as a loop
    over ``PLDCorrector(tpf, aperture_mask).correct(...)`` would (pldcorrector.py:99-107, 203-207); such PLD / background
    masks must then select the same NUMBER of pixels in every cutout (one design-matrix width per call).  The per-cutout host
    work (aperture sums, NaN-cadence removal, pixel gathers, percentile knots) runs on the packing thread pool and lands in
    page-locked staging buffers.  Returns (corrected_flux[B, N], outlier_mask[B, N])."""
    from .. import packed
    cubes = list(cubes)
    if not cubes:
        raise ValueError("pld_correct_batch needs at least one cutout")
    if pca_components is None or pca_components < 1:
        raise NotImplementedError("pca_components must be >= 1 on the HIP path")
    first = cubes[0]
    specs = (aperture_mask, pld_aperture_mask, background_aperture_mask)
    ap = _shared_mask(first, aperture_mask, sap=True)
    pm = _shared_mask(first, pld_aperture_mask)
    bm = _shared_mask(first, background_aperture_mask)
    for c in cubes:
        if c.shape[1:] != first.shape[1:]:
            raise ValueError("pld_correct_batch needs cutouts of one shape (got %s and %s)" % (c.shape, first.shape))
    parts = packed._pmap(_batch_cutout, [(c, ap, pm, bm, specs) for c in cubes])
    n = len(parts[0][0])
    if any(len(p[0]) != n for p in parts):
        raise ValueError("pld_correct_batch needs cutouts with the same number of valid cadences")
    B, P, Pb = len(parts), parts[0][4].shape[1], parts[0][5].shape[1]
    if any(p[4].shape[1] != P or p[5].shape[1] != Pb for p in parts):
        raise ValueError("pld_correct_batch: the per-cutout '%s' / '%s' masks select different numbers of pixels (%s PLD, %s "
                         "background); pass masks of one size or correct these cutouts one by one"
                         % (pld_aperture_mask, background_aperture_mask, sorted({p[4].shape[1] for p in parts}),
                            sorted({p[5].shape[1] for p in parts})))
    shared = all(p[5] is p[4] for p in parts)
    if spline_n_knots is None:
        spline_n_knots = int(n / 50)

    def staged(key, shape, dtype):
        count = int(np.prod(shape))
        try:
            return _capi.pinned_pool("pld_" + key, count, dtype).reshape(shape)
        except (OSError, RuntimeError, MemoryError):
            return np.empty(shape, dtype=dtype)

    pld = staged("pix", (B, n, P), np.float32)
    bkg = pld if shared else staged("bkg", (B, n, Pb), np.float32)
    lcf = staged("lcf", (B, n), np.float32)
    t, y, err = (staged(k, (B, n), np.float64) for k in ("t", "y", "err"))
    knots = np.empty((B, max(spline_n_knots - spline_degree - 1, 0) + 2), dtype=np.float64)

    def fill(b):
        tb, fb, eb, f32, px, bx = parts[b]
        t[b], y[b], err[b], lcf[b] = tb, fb, eb, f32
        pld[b] = px
        if not shared:
            bkg[b] = bx
        knots[b] = _percentile_knots(tb, spline_n_knots, spline_degree)
        return _all_finite(px) and (shared or _all_finite(bx))

    if not all(packed._pmap(fill, [(b,) for b in range(B)])):
        raise ValueError("pld_correct_batch needs finite pixels inside the masks")
    res = _capi.pld_correct_batch(pld if P else None, bkg, lcf, t, knots, y, err, pld_order, pca_components, spline_degree,
                                  normalize_background_pixels, sigma=sigma, niters=niters, want_spline=restore_trend,
                                  device=device)
    corrected = y - res["model"]
    if restore_trend:
        sp = res["spline"]
        corrected += sp - np.median(sp, axis=1)[:, None]
    return corrected, res["outlier_mask"]
