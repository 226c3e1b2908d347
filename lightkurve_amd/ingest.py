"""Ragged batch container for light curves + the device-side ingest steps.

A pipeline over a ``LightCurveCollection`` (reference src/lightkurve/collections.py:145) calls ``remove_nans`` /
``normalize`` / ``bin`` / ``create_transit_mask`` / ``fold`` per light curve in Python, each building astropy Time / Quantity
/ Table objects; with the kernels at milliseconds per thousand targets that object construction is the wall.
``LightCurveBatch`` keeps B light curves as three concatenated float64 arrays plus prefix offsets — the layout every
``lk_*_batch`` entry point takes — and runs those steps for the whole batch in one launch each.

    batch = LightCurveBatch.from_lightcurves(lcs)          # anything with .time / .flux / .flux_err
    batch = batch.remove_nans().normalize()                # lk_ingest_batch
    mask  = batch.create_transit_mask(period, transit_time, duration)     # lk_transit_mask_batch
    power = batch.to_periodogram_power(frequency)          # lombscargle_batch over the same arrays
"""
import numpy as np

from . import _capi
from .lightcurve import LightCurve

__all__ = ["LightCurveBatch"]


from .packed import values as _values  # noqa: E402,F401  (kept under its old name for callers)


class LightCurveBatch(object):
    def __init__(self, time, flux, flux_err, n_off, meta=None):
        # contiguous float64 inputs are kept as they are (page-locked arrays from ``_capi.pinned_empty`` stay page-locked)
        self.time = np.ascontiguousarray(time, dtype=np.float64)
        self.flux = np.ascontiguousarray(flux, dtype=np.float64)
        self.flux_err = np.ascontiguousarray(flux_err, dtype=np.float64)
        self.n_off = np.ascontiguousarray(n_off, dtype=np.int64)
        if not (self.time.shape == self.flux.shape == self.flux_err.shape) or self.time.ndim != 1:
            raise ValueError("time, flux, flux_err must be 1-D arrays of one length")
        if self.n_off[0] != 0 or self.n_off[-1] != self.time.size or np.any(np.diff(self.n_off) < 0):
            raise ValueError("n_off must be non-decreasing prefix offsets over the arrays")
        self.meta = list(meta) if meta is not None else [{} for _ in range(len(self))]
        self.quality = None   # per-cadence quality flags (set by from_fits; carried through remove_nans / normalize)

    @classmethod
    def from_lightcurves(cls, lcs, pinned=False):
        """From an iterable of light curves (this package's or lightkurve's own: Time / Quantity columns are read through
        ``.value``): one concatenation per column.  ``pinned``: keep the three arrays in page-locked memory (lk_host_alloc)
        so that every later ``to_periodogram_*`` / ``bls`` / ``flatten_trend`` call DMAs them without staging."""
        from . import packed
        lcs = list(lcs)
        meta = [dict(getattr(lc, "meta", {}) or {}) for lc in lcs]
        (t, f, e), off = packed.pack_columns(lcs, ("time", "flux", "flux_err"), pinned="own" if pinned else False)
        return cls(t, f, e, off, meta)

    @classmethod
    def from_fits(cls, paths, flux_column=None, quality_bitmask="default", ext=1, device=0):
        """Light-curve FITS files -> one batch, with the reference readers' defaults (``lk.read(path)``: PDCSAP flux and
        the mission's default quality bitmask for Kepler / K2 / TESS products, FLUX / QUALITY otherwise; rows with NaN time
        dropped; reference src/lightkurve/io/generic.py:21-207, io/kepler.py, io/tess.py).  The host parses the headers,
        the GPU turns the tables' bytes into the arrays (``lk_fits_unpack_batch``).  ``batch.quality`` holds the kept
        cadences' flags, ``batch.meta[b]`` OBJECT / MISSION / the bitmask used, like the reference's ``lc.meta``."""
        from . import fitsio
        raws, descs, masks, meta = [], [], [], []
        for path in paths:
            tab = fitsio.read_fits_table(path, ext=ext)
            desc, bitmask, mission = fitsio.lightcurve_columns(tab, flux_column=flux_column, quality_bitmask=quality_bitmask)
            raws.append(tab.raw), descs.append(desc), masks.append(bitmask)
            meta.append({"FILENAME": str(path), "LABEL": tab.primary.get("OBJECT"),
                         "MISSION": tab.primary.get("MISSION", tab.primary.get("TELESCOP")), "RA": tab.primary.get("RA_OBJ"),
                         "DEC": tab.primary.get("DEC_OBJ"), "QUALITY_BITMASK": quality_bitmask,
                         "BJDREFI": tab.header.get("BJDREFI"), "READER_MISSION": mission})
        t, f, e, q, off = _capi.fits_unpack_batch(raws, descs, masks, device=device)
        out = cls(t, f, e, off, meta)
        out.quality = q
        return out

    def __len__(self):
        return len(self.n_off) - 1

    def __getitem__(self, b):
        s = slice(int(self.n_off[b]), int(self.n_off[b + 1]))
        return LightCurve(time=self.time[s], flux=self.flux[s], flux_err=self.flux_err[s], meta=self.meta[b])

    def to_lightcurves(self):
        return [self[b] for b in range(len(self))]

    # ---------------------------------------------------------------- device-side steps
    def _ingest(self, normalize, device):
        t, f, e, off, med = _capi.ingest_batch(self.time, self.flux, self.n_off, flux_err=self.flux_err,
                                               normalize=normalize, device=device)
        out = LightCurveBatch(t, f, e, off, [dict(m) for m in self.meta])
        out.median_flux = med
        if self.quality is not None:
            # lk_ingest_batch keeps exactly the cadences with finite flux, in order: the flags follow by the same test
            keep = ~np.isnan(self.flux)
            if int(keep.sum()) != len(t):
                raise RuntimeError("quality flags out of step with the ingest kernel (kept %d of %d cadences, expected %d)"
                                   % (len(t), len(self.flux), int(keep.sum())))
            out.quality = np.ascontiguousarray(np.asarray(self.quality)[keep])
        return out

    def remove_nans(self, device=0):
        """Every light curve without the cadences whose flux is NaN (reference lightcurve.py:1300-1327)."""
        return self._ingest(False, device)

    def normalize(self, device=0):
        """flux and flux_err divided by nanmedian(flux) per light curve (reference :1216-1292); NaN-flux cadences are dropped
        first — what pipelines do anyway (``lc.remove_nans().normalize()``, e.g. correctors/metrics.py:60) — because the
        packed layout has no place for them."""
        out = self._ingest(True, device)
        for m in out.meta:
            m["NORMALIZED"] = True
        return out

    def create_transit_mask(self, period, transit_time, duration, planet_off=None, device=0):
        """Boolean array over all cadences of the batch, True in transit (reference :2967-3037).  Scalars / 1-D arrays
        apply to every light curve; with ``planet_off`` each light curve gets its own planets."""
        return _capi.transit_mask_batch(self.time, self.n_off, period, duration, transit_time, planet_off=planet_off,
                                        device=device)

    def bin(self, time_bin_size=0.5, time_bin_start=None, device=0):
        """Equal-width time bins (reference :1558-1763 with ``time_bin_size`` in days): nanmean flux, rms flux_err."""
        t, f, e, boff = _capi.bin_batch(self.time, self.flux, self.n_off, flux_err=self.flux_err,
                                        time_bin_size=time_bin_size, time_bin_start=time_bin_start, device=device)
        return LightCurveBatch(t, f, e, boff, [dict(m) for m in self.meta])   # binned cadences have no quality flag

    def to_periodogram_power(self, frequency, normalization="amplitude", ls_method="fast", device=None, **kw):
        """Lomb-Scargle power of every light curve on one shared grid -> float64[B, M]: the packed arrays go straight to
        the C ABI (``batch.lombscargle_batch`` plans per batch and never rebuilds per-target objects).  ``out=``: a
        preallocated (e.g. ``_capi.pinned_empty``) float64[B, M]."""
        from .batch import lombscargle_batch
        return lombscargle_batch(self, frequency, normalization=normalization, ls_method=ls_method, device=device, **kw)

    def to_periodogram_peaks(self, frequency, normalization="amplitude", device=None, **kw):
        """(max power, argmax) per light curve of the default-method periodogram -> float64[B, 2]; the spectra stay in HBM
        (``batch.lombscargle_peaks_batch``; reference periodogram.py:127-140)."""
        from .batch import lombscargle_peaks_batch
        return lombscargle_peaks_batch(self, frequency, normalization=normalization, device=device, **kw)

    def bls(self, period, duration=None, objective="likelihood", oversample=10, device=None, **kw):
        """The seven BLS statistics of every light curve on one shared period grid -> float64[B, 7, nP]
        (``batch.bls_batch``)."""
        from .batch import bls_batch
        return bls_batch(self, period, duration, objective=objective, oversample=oversample, device=device, **kw)

    def flatten_trend(self, device=0, **kw):
        """The trend ``LightCurve.flatten`` divides by, for every light curve -> concatenated array in batch layout."""
        return _capi.savgol_trend_batch(self.time, self.flux, self.n_off, device=device, **kw)
