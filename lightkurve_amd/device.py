"""Device-resident batch of light curves ("FITS -> ragged device arrays, remove_nans / normalize / bin /
fold / create_transit_mask on device").

``LightCurveBatch`` (ingest.py) keeps the packed arrays on the HOST: every stage of a chained pipeline is H2D -> kernel ->
D2H.  ``DeviceLightCurveBatch`` uploads once and chains the ``lk_*_batch_dev`` entry points on ONE stream of ONE handle;
only what the caller asks back crosses PCIe.  The reference loop this replaces (per object, per stage, on the host):

    for lc in collection:                                     # src/lightkurve/collections.py:145
        lc = lc.remove_nans().normalize()                     # lightcurve.py:1300-1327, 1216-1292
        lc = lc.flatten(window_length=401)                    # :943-1078
        pg = lc.to_periodogram(frequency=f)                   # :2490-2535 -> periodogram.py:636-989
        lc.fold(period=pg.period_at_max_power)                # :1089-1214

    batch = DeviceLightCurveBatch.from_lightcurves(lcs)       # or .from_fits(paths) / .from_batch(host_batch)
    flat = batch.remove_nans().normalize().flatten(window_length=401)
    peaks = flat.to_periodogram_peaks(f)                      # float64[B, 2] on the host: 16 B per target cross PCIe
    folded = flat.fold(period=1 / f[peaks[:, 1].astype(int)]).to_host()

No torch: device memory, copies and the stream come from liblkhip.so itself (``lk_dev_alloc`` / ``lk_memcpy_*`` /
``lk_stream_*``).  Results are bit-identical to the staged host path — the same kernels run on the same numbers.
"""
import ctypes
import threading
import weakref

import numpy as np

from . import _capi
from . import packed

__all__ = ["DeviceBuffer", "DeviceLightCurveBatch", "DeviceFoldedBatch", "DeviceBLSResult", "release_device_pool"]

_vp = ctypes.c_void_p
_ip = ctypes.POINTER(ctypes.c_int64)
_dp = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)


# ------------------------------------------------------------------------------------------------ device memory
class _Pool(object):
    """Free list of device allocations per handle.  ``lk_dev_free`` (hipFree) synchronises the whole device, and a pipeline
    drops an intermediate batch at every stage: a dropped buffer goes back to this list instead and is handed out again
    to the next request it fits (work on one handle is stream-ordered — include/lkhip.h "Conventions" — so a kernel still
    reading the old contents finishes before the new owner's kernels start).  ``release_device_pool()`` really frees."""

    def __init__(self, handle):
        self.handle = handle
        self.free = []                  # (capacity, ptr)
        self.lock = threading.Lock()

    def take(self, nbytes):
        nbytes = max(int(nbytes), 1)
        with self.lock:
            fit = [i for i, (cap, _) in enumerate(self.free) if nbytes <= cap <= 2 * nbytes + (1 << 16)]
            if fit:
                return self.free.pop(min(fit, key=lambda i: self.free[i][0]))
        cap = (nbytes + 255) & ~255
        ptr = _vp()
        _capi._check(_capi._lib.lk_dev_alloc(self.handle._h, ctypes.byref(ptr), cap))
        return cap, ptr.value

    def give(self, cap, ptr):
        with self.lock:
            self.free.append((cap, ptr))

    def release(self):
        with self.lock:
            items, self.free = self.free, []
        for _cap, ptr in items:
            _capi._lib.lk_dev_free(self.handle._h, _vp(ptr))


_POOLS = {}


def _pool(handle):
    p = _POOLS.get(handle.device)
    if p is None or p.handle is not handle:
        p = _POOLS[handle.device] = _Pool(handle)
    return p


def release_device_pool():
    """hipFree every idle buffer of the device free lists (buffers still owned by live objects are not touched)."""
    for p in list(_POOLS.values()):
        p.release()


class DeviceBuffer(object):
    """``nbytes`` of HBM owned by this object (returned to the free list when it is garbage collected)."""
    __slots__ = ("ptr", "nbytes", "capacity", "handle", "__weakref__")

    def __init__(self, handle, nbytes):
        pool = _pool(handle)
        self.capacity, self.ptr = pool.take(nbytes)
        self.nbytes = int(nbytes)
        self.handle = handle
        weakref.finalize(self, pool.give, self.capacity, self.ptr)

    def upload(self, host, stream=0):
        host = np.ascontiguousarray(host)
        if host.nbytes > self.capacity:
            raise ValueError("host array of %d bytes into a device buffer of %d" % (host.nbytes, self.capacity))
        _capi._check(_capi._lib.lk_memcpy_h2d(self.handle._h, _vp(self.ptr), _vp(host.ctypes.data), host.nbytes,
                                             _vp(stream or None)))
        return host          # (the caller keeps it alive until the stream has consumed it)

    def download(self, dtype, count, out=None, stream=0, offset_bytes=0):
        """``count`` elements of ``dtype`` starting ``offset_bytes`` into the buffer -> host array (synchronises ``stream``)."""
        dtype = np.dtype(dtype)
        count = int(count)
        if out is None:
            out = np.empty(count, dtype=dtype)
        if out.dtype != dtype or out.size != count or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous %s array of %d elements" % (dtype, count))
        if offset_bytes + count * dtype.itemsize > self.capacity:
            raise ValueError("read past the end of the device buffer")
        h = self.handle
        _capi._check(_capi._lib.lk_memcpy_d2h(h._h, _vp(out.ctypes.data), _vp(self.ptr + offset_bytes), out.nbytes,
                                             _vp(stream or None)))
        _capi._check(_capi._lib.lk_stream_synchronize(h._h, _vp(stream or None)))
        return out


def _upload(handle, host, stream, dtype=np.float64):
    host = np.ascontiguousarray(host, dtype=dtype)
    buf = DeviceBuffer(handle, host.nbytes)
    keep = buf.upload(host, stream)
    return buf, keep


def _off_ptr(n_off):
    return n_off.ctypes.data_as(_ip)


# ------------------------------------------------------------------------------------------------ the batch
class DeviceLightCurveBatch(object):
    """B light curves as packed float64 arrays IN HBM (``d_time`` / ``d_flux`` / ``d_flux_err``: ``DeviceBuffer``) plus the
    prefix offsets ``n_off`` on the host (every launcher sizes its grid from them).  All work goes to ``stream`` (an opaque
    stream handle: ``0`` = the null stream, ``lk_stream_create``'s, or e.g. ``torch.cuda.current_stream().cuda_stream``) of
    the process's handle for ``device``."""

    def __init__(self, d_time, d_flux, d_flux_err, n_off, meta=None, device=0, stream=0, nan_free=False, is_sorted=None):
        self.handle = _capi.Handle.get(device)
        self.device = int(device)
        self.stream = int(stream or 0)
        self.d_time, self.d_flux, self.d_flux_err = d_time, d_flux, d_flux_err
        self.n_off = np.ascontiguousarray(n_off, dtype=np.int64)
        if self.n_off.ndim != 1 or self.n_off.size < 1 or self.n_off[0] != 0 or np.any(np.diff(self.n_off) < 0):
            raise ValueError("n_off must be non-decreasing prefix offsets starting at 0")
        need = int(self.n_off[-1]) * 8
        for b in (d_time, d_flux, d_flux_err):
            if b is not None and b.capacity < need:
                raise ValueError("device buffer smaller than the batch it is said to hold")
        self.meta = list(meta) if meta is not None else [{} for _ in range(len(self))]
        self.nan_free = bool(nan_free)          # no NaN flux: the periodogram front ends need not compact first
        self.is_sorted = is_sorted              # times non-decreasing per light curve (None: not checked yet)
        self.d_quality = None                   # int32 flags (from_fits), carried through remove_nans / normalize
        self.median_flux = None                 # DeviceBuffer float64[B] after remove_nans / normalize
        self._keep = []                         # host staging the stream may still be reading

    # ---------------------------------------------------------------- construction
    @classmethod
    def from_arrays(cls, time, flux, flux_err, n_off, meta=None, device=0, stream=0):
        """H2D once: packed host arrays (page-locked ones from ``_capi.pinned_empty`` / the staging pool go by DMA)."""
        time = np.ascontiguousarray(time, dtype=np.float64)
        n_off = _capi._offsets(n_off, time.size)
        h = _capi.Handle.get(device)
        srt = packed.check_sorted(time, n_off)
        bufs, keep = [], []
        for a in (time, flux, flux_err):
            if a is None:
                bufs.append(None)
                continue
            if np.shape(a) != time.shape:
                raise ValueError("time, flux, flux_err must be 1-D arrays of one length")
            b, k = _upload(h, a, stream)
            bufs.append(b), keep.append(k)
        out = cls(bufs[0], bufs[1], bufs[2], n_off, meta, device, stream, is_sorted=srt)
        out._keep = keep
        return out

    @classmethod
    def from_lightcurves(cls, lcs, device=0, stream=0):
        """From an iterable of light curves (this package's or lightkurve's own): one concatenation per column into the
        page-locked staging pool, then ONE upload."""
        lcs = list(lcs)
        meta = [dict(getattr(lc, "meta", {}) or {}) for lc in lcs]
        (t, f, e), off = packed.pack_columns(lcs, ("time", "flux", "flux_err"), pinned="auto", pool_prefix="devbatch")
        out = cls.from_arrays(t, f, e, off, meta, device, stream)
        out.synchronize()        # the staging pool is reused by the next packing call
        out._keep = []
        return out

    @classmethod
    def from_batch(cls, batch, device=0, stream=0):
        """From a host ``LightCurveBatch``."""
        out = cls.from_arrays(batch.time, batch.flux, batch.flux_err, batch.n_off, [dict(m) for m in batch.meta], device, stream)
        if getattr(batch, "quality", None) is not None:
            out.d_quality, k = _upload(out.handle, batch.quality, stream, np.int32)
            out._keep.append(k)
        return out

    @classmethod
    def from_fits(cls, paths, flux_column=None, quality_bitmask="default", ext=1, device=0, stream=0):
        """Light-curve FITS files -> a device-resident batch (``LightCurveBatch.from_fits`` without the way back: reference
        ``lk.read(path)`` per file, src/lightkurve/io/generic.py:21-207, io/kepler.py, io/tess.py).  The host parses the
        headers; the tables' bytes go to HBM as they are and ``lk_fits_unpack_batch_dev`` turns them into the arrays."""
        from . import fitsio
        h = _capi.Handle.get(device)
        raws, descs, masks, meta = [], [], [], []
        for path in paths:
            tab = fitsio.read_fits_table(path, ext=ext)
            desc, bitmask, mission = fitsio.lightcurve_columns(tab, flux_column=flux_column, quality_bitmask=quality_bitmask)
            raws.append(tab.raw), descs.append(desc), masks.append(bitmask)
            meta.append({"FILENAME": str(path), "LABEL": tab.primary.get("OBJECT"),
                         "MISSION": tab.primary.get("MISSION", tab.primary.get("TELESCOP")), "RA": tab.primary.get("RA_OBJ"),
                         "DEC": tab.primary.get("DEC_OBJ"), "QUALITY_BITMASK": quality_bitmask,
                         "BJDREFI": tab.header.get("BJDREFI"), "READER_MISSION": mission})
        B = len(raws)
        desc = np.ascontiguousarray(descs, dtype=np.int32).reshape(B, 10)
        mask = np.ascontiguousarray(masks, dtype=np.int64).reshape(B)
        raw_off = np.zeros(B + 1, dtype=np.int64)
        for b, r in enumerate(raws):
            nbytes = int(desc[b, 0]) * int(desc[b, 1])
            if np.asarray(r).size != nbytes:
                raise ValueError("file %d: %d bytes of table data, descriptor says %d rows x %d bytes"
                                 % (b, np.asarray(r).size, desc[b, 1], desc[b, 0]))
            raw_off[b + 1] = raw_off[b] + ((nbytes + 3 + 15) // 16) * 16
        try:
            raw = _capi.pinned_pool("devbatch:fits", int(raw_off[-1]), np.uint8)
        except (OSError, RuntimeError, MemoryError):
            raw = np.empty(int(raw_off[-1]), dtype=np.uint8)
        for b, r in enumerate(raws):
            r = np.asarray(r, dtype=np.uint8).reshape(-1)
            raw[raw_off[b]:raw_off[b] + r.size] = r
            raw[raw_off[b] + r.size:raw_off[b + 1]] = 0
        rows = int(desc[:, 1].sum())
        d_raw = DeviceBuffer(h, raw.nbytes)
        d_raw.upload(raw, stream)
        d_t, d_f, d_e = (DeviceBuffer(h, rows * 8) for _ in range(3))
        d_q = DeviceBuffer(h, rows * 4)
        new_off = np.zeros(B + 1, dtype=np.int64)
        _capi._check(_capi._lib.lk_fits_unpack_batch_dev(h._h, B, _vp(d_raw.ptr), _off_ptr(raw_off), desc.ctypes.data_as(_i32p),
                                                         _off_ptr(mask), _vp(d_t.ptr), _vp(d_f.ptr), _vp(d_e.ptr), _vp(d_q.ptr),
                                                         _off_ptr(new_off), _vp(stream or None)))     # (synchronises)
        out = cls(d_t, d_f, d_e, new_off, meta, device, stream)
        out.d_quality = d_q
        return out

    # ---------------------------------------------------------------- plumbing
    def __len__(self):
        return len(self.n_off) - 1

    @property
    def n_cadences(self):
        return int(self.n_off[-1])

    def synchronize(self):
        _capi._check(_capi._lib.lk_stream_synchronize(self.handle._h, _vp(self.stream or None)))
        self._keep = []

    def _new(self, d_time, d_flux, d_err, n_off, **kw):
        out = DeviceLightCurveBatch(d_time, d_flux, d_err, n_off, [dict(m) for m in self.meta], self.device, self.stream, **kw)
        return out

    def _host(self, buf, dtype=np.float64):
        if buf is None:
            return None
        return buf.download(dtype, self.n_cadences, stream=self.stream)

    def time_host(self):
        return self._host(self.d_time)

    def flux_host(self):
        return self._host(self.d_flux)

    def flux_err_host(self):
        return self._host(self.d_flux_err)

    def quality_host(self):
        return self._host(self.d_quality, np.int32)

    def to_host(self):
        """D2H of the three columns -> ``LightCurveBatch`` (what a caller asks back at the END of a chain)."""
        from .ingest import LightCurveBatch
        n = self.n_cadences
        e = self.flux_err_host() if self.d_flux_err is not None else np.full(n, np.nan)
        out = LightCurveBatch(self.time_host(), self.flux_host(), e, self.n_off.copy(), [dict(m) for m in self.meta])
        if self.d_quality is not None:
            out.quality = self.quality_host()
        return out

    def _sorted(self):
        if self.is_sorted is None:
            B = len(self)
            desc = np.zeros(max(B, 1), dtype=np.int64)
            _capi._check(_capi._lib.lk_segment_probe_batch_dev(self.handle._h, B, _off_ptr(self.n_off), _vp(self.d_time.ptr), None,
                                                               _off_ptr(desc), None, _vp(self.stream or None)))
            self.is_sorted = not bool(desc[:B].any())
        return self.is_sorted

    # ---------------------------------------------------------------- remove_nans / normalize
    def _ingest(self, normalize):
        h, B, n = self.handle, len(self), self.n_cadences
        d_t, d_f = DeviceBuffer(h, n * 8), DeviceBuffer(h, n * 8)
        d_e = DeviceBuffer(h, n * 8) if self.d_flux_err is not None else None
        d_med = DeviceBuffer(h, max(B, 1) * 8)
        new_off = np.zeros(B + 1, dtype=np.int64)
        _capi._check(_capi._lib.lk_ingest_batch_dev(
            h._h, B, _off_ptr(self.n_off), _vp(self.d_time.ptr), _vp(self.d_flux.ptr),
            _vp(self.d_flux_err.ptr if self.d_flux_err is not None else None), int(bool(normalize)), _vp(d_t.ptr), _vp(d_f.ptr),
            _vp(d_e.ptr if d_e is not None else None), _off_ptr(new_off), _vp(d_med.ptr), _vp(self.stream or None)))
        out = self._new(d_t, d_f, d_e, new_off, nan_free=True, is_sorted=self.is_sorted)
        out.median_flux = d_med
        if self.d_quality is not None:
            d_q = DeviceBuffer(h, n * 4)
            cin, cout = (_vp * 1)(self.d_quality.ptr), (_vp * 1)(d_q.ptr)
            _capi._check(_capi._lib.lk_compact_columns_batch_dev(h._h, B, _off_ptr(self.n_off), _off_ptr(new_off), _vp(self.d_flux.ptr),
                                                                 1, 4, cin, cout, _vp(self.stream or None)))
            out.d_quality = d_q
        return out

    def remove_nans(self):
        """Every light curve without the cadences whose flux is NaN (reference lightcurve.py:1300-1327); repacked in HBM."""
        return self._ingest(False)

    def normalize(self):
        """flux and flux_err divided by nanmedian(flux) per light curve (reference :1216-1292); NaN-flux cadences are
        dropped first, as ``LightCurveBatch.normalize`` does."""
        out = self._ingest(True)
        for m in out.meta:
            m["NORMALIZED"] = True
        return out

    # ---------------------------------------------------------------- flatten
    def flatten_trend(self, window_length=101, polyorder=2, break_tolerance=5, niters=3, sigma=3, mask=None):
        """The trend ``LightCurve.flatten`` divides by (reference :996-1063) -> ``DeviceBuffer`` float64[sum N].
        ``mask``: bool / uint8 array over all cadences of the batch (host, uploaded) or a ``DeviceBuffer`` of bytes,
        1 = excluded from the fit."""
        if polyorder >= window_length:
            polyorder = window_length - 1
        if window_length % 2 != 1:
            raise ValueError("window_length must be odd (scipy.signal.savgol_filter with mode='interp')")
        if not self._sorted():
            raise ValueError("flatten needs the light curve sorted by time")
        h, n = self.handle, self.n_cadences
        d_m = None
        if mask is not None:
            if isinstance(mask, DeviceBuffer):
                d_m = mask
            else:
                mk = np.ascontiguousarray(mask, dtype=np.uint8)
                if mk.shape != (n,):
                    raise ValueError("mask must have one entry per cadence (got shape %s, need (%d,))" % (mk.shape, n))
                d_m, k = _upload(h, mk, self.stream, np.uint8)
                self._keep.append(k)
        bt = float("nan") if break_tolerance is None else float(break_tolerance)
        d_tr = DeviceBuffer(h, n * 8)
        _capi._check(_capi._lib.lk_savgol_trend_batch_dev(h._h, len(self), _off_ptr(self.n_off), _vp(self.d_time.ptr),
                                                          _vp(self.d_flux.ptr), _vp(d_m.ptr if d_m is not None else None),
                                                          int(window_length), int(polyorder), bt, int(niters), float(sigma),
                                                          _vp(d_tr.ptr), None, _vp(self.stream or None)))
        return d_tr

    def flatten(self, window_length=101, polyorder=2, return_trend=False, break_tolerance=5, niters=3, sigma=3, mask=None):
        """``lc.flatten(...)`` for the whole batch, result resident: flux / trend and flux_err / trend (reference
        :1064-1070).  ``return_trend``: also a batch whose flux is the trend (the reference's ``trend_lc``)."""
        d_tr = self.flatten_trend(window_length, polyorder, break_tolerance, niters, sigma, mask)
        h, n = self.handle, self.n_cadences
        d_f = DeviceBuffer(h, n * 8)
        d_e = DeviceBuffer(h, n * 8) if self.d_flux_err is not None else None
        _capi._check(_capi._lib.lk_flatten_apply_batch_dev(h._h, n, _vp(self.d_flux.ptr),
                                                           _vp(self.d_flux_err.ptr if self.d_flux_err is not None else None),
                                                           _vp(d_tr.ptr), _vp(d_f.ptr), _vp(d_e.ptr if d_e is not None else None),
                                                           _vp(self.stream or None)))
        out = self._new(self.d_time, d_f, d_e, self.n_off, nan_free=False, is_sorted=self.is_sorted)
        out.d_quality = self.d_quality
        for m in out.meta:
            m["NORMALIZED"] = True
        if return_trend:
            tr = self._new(self.d_time, d_tr, self.d_flux_err, self.n_off, nan_free=False, is_sorted=self.is_sorted)
            return out, tr
        return out

    # ---------------------------------------------------------------- Lomb-Scargle
    def _ls_ready(self):
        """The batch the periodogram kernels see: NaN-flux cadences dropped (LombScarglePeriodogram.from_lightcurve,
        periodogram.py:869-872), at least two cadences each."""
        src = self if self.nan_free else self.remove_nans()
        counts = np.diff(src.n_off)
        if len(counts) and counts.min() < 2:
            raise ValueError("The light curve needs at least two cadences to build a periodogram.")
        return src

    def _ls_scale(self, src, plan):
        """Device array of lightkurve's per-target psd factor (periodogram.py:865-868, 969-975), or None for 'amplitude'."""
        if plan.normalization != "psd":
            return None
        B = len(src)
        idx = np.concatenate([src.n_off[:-1], src.n_off[1:] - 1]).astype(np.int64)
        ends = np.empty(2 * B, dtype=np.float64)
        _capi._check(_capi._lib.lk_gather_f64_dev(src.handle._h, 2 * B, _off_ptr(idx), _vp(src.d_time.ptr), ends.ctypes.data_as(_dp),
                                                  _vp(src.stream or None)))
        counts = np.diff(src.n_off)
        fs = (1.0 / (ends[B:] - ends[:B])) / plan.oversample_factor * plan.unit
        scale = 2.0 / (counts * plan.oversample_factor * fs)
        d_s, k = _upload(src.handle, scale, src.stream)
        src._keep.append(k)
        return d_s

    def to_periodogram_power(self, frequency, normalization="amplitude", freq_unit=None, oversample_factor=None,
                             ls_method="fast", nterms=1, out=None, to_host=True, want_peaks=False):
        """Lomb-Scargle power of every light curve on one shared grid (``batch.lombscargle_batch`` for a resident batch:
        same plan, same kernels).  ``to_host``: float64[B, M] on the host (``out=`` a preallocated / page-locked array), else
        the ``DeviceBuffer`` holding it.  ``want_peaks``: also return float64[B, 2] (max power, argmax)."""
        plan = packed.ls_grid_plan(frequency, normalization, freq_unit, oversample_factor, ls_method, nterms)
        src = self._ls_ready()
        h, B, M, st = src.handle, len(src), len(plan.f_day), _vp(src.stream or None)
        lib = _capi._lib
        d_pow = DeviceBuffer(h, max(B * M, 1) * 8)
        d_max = DeviceBuffer(h, max(B, 1) * 8)
        d_arg = DeviceBuffer(h, max(B, 1) * 8)
        if B and M:
            d_s = self._ls_scale(src, plan)
            sp = _vp(d_s.ptr if d_s is not None else None)
            f_day, norm = plan.f_day, _capi.NORM[plan.norm]
            if plan.nterms == 1 and plan.ls_method in ("fast", "fastchi2"):
                _capi._check(lib.lk_ls_fast_peaks_lc_batch_dev(h._h, B, _off_ptr(src.n_off), _vp(src.d_time.ptr), _vp(src.d_flux.ptr),
                                                               None, float(f_day[0]), float(f_day[1] - f_day[0]), M, 1, 1, norm, sp, 5,
                                                               _vp(d_pow.ptr), _vp(d_max.ptr), _vp(d_arg.ptr), st))
            else:
                d_trel = DeviceBuffer(h, src.n_cadences * 8)
                _capi._check(lib.lk_rebase_times_batch_dev(h._h, B, _off_ptr(src.n_off), _vp(src.d_time.ptr), _vp(d_trel.ptr), st))
                if plan.nterms > 1 and plan.ls_method == "fastchi2":
                    _capi._check(lib.lk_ls_fastchi2_batch_dev(h._h, B, _off_ptr(src.n_off), _vp(d_trel.ptr), _vp(src.d_flux.ptr), None,
                                                              float(f_day[0]), float(f_day[1] - f_day[0]), M, plan.nterms, 1, 1, norm,
                                                              sp, 5, _vp(d_pow.ptr), st))
                elif plan.exact is not None:
                    _capi._check(lib.lk_ls_chi2_batch_dev(h._h, B, _off_ptr(src.n_off), _vp(d_trel.ptr), _vp(src.d_flux.ptr), None, None,
                                                          float(plan.exact[0]), float(plan.exact[1]), M, plan.nterms, 1, 1, norm, sp,
                                                          _vp(d_pow.ptr), st))
                else:
                    d_fr, k = _upload(h, f_day, src.stream)
                    src._keep.append(k)
                    _capi._check(lib.lk_ls_chi2_batch_dev(h._h, B, _off_ptr(src.n_off), _vp(d_trel.ptr), _vp(src.d_flux.ptr), None,
                                                          _vp(d_fr.ptr), 0.0, 0.0, M, plan.nterms, 1, 1, norm, sp, _vp(d_pow.ptr), st))
                if want_peaks:
                    _capi._check(lib.lk_argmax_batch_dev(h._h, B, M, _vp(d_pow.ptr), _vp(d_max.ptr), _vp(d_arg.ptr), st))
        peaks = None
        if want_peaks:
            mx = d_max.download(np.float64, B, stream=src.stream)
            am = d_arg.download(np.int64, B, stream=src.stream)
            peaks = np.column_stack([mx, am.astype(np.float64)]) if B else np.zeros((0, 2))
        if to_host:
            if out is not None and (out.shape != (B, M) or out.dtype != np.float64 or not out.flags.c_contiguous):
                raise ValueError("out must be a C-contiguous float64 array of shape (B, M)")
            host = out if out is not None else _capi.result_empty((B, M))
            d_pow.download(np.float64, B * M, out=host.reshape(-1), stream=src.stream)
            res = host
        else:
            res = d_pow
        src._keep = []
        return (res, peaks) if want_peaks else res

    def to_periodogram_peaks(self, frequency, normalization="amplitude", freq_unit=None, oversample_factor=None):
        """(max power, argmax) per light curve of the default-method periodogram -> float64[B, 2] on the host; the spectra
        stay in HBM (``Periodogram.max_power`` / ``frequency_at_max_power``, reference periodogram.py:127-140)."""
        plan = packed.ls_grid_plan(frequency, normalization, freq_unit, oversample_factor, "fast", 1)
        if plan.ls_method != "fast":
            raise ValueError("to_periodogram_peaks needs a regular frequency grid (the reference switches to 'slow')")
        _pow, peaks = self.to_periodogram_power(frequency, normalization, freq_unit, oversample_factor, "fast", 1,
                                                to_host=False, want_peaks=True)
        return peaks

    # ---------------------------------------------------------------- BLS
    def bls(self, period, duration=None, objective="likelihood", oversample=10):
        """The seven BLS statistics of every light curve on one shared period grid, resident (``batch.bls_batch``:
        reference periodogram.py:1093-1169 over astropy ``bls_fast``) -> ``DeviceBLSResult``."""
        from .batch import _bls_options
        period, duration, objective, oversample = _bls_options(period, duration, objective, oversample)
        src = self if self.nan_free else self.remove_nans()
        counts = np.diff(src.n_off)
        if len(counts) and counts.min() < 1:
            raise ValueError("a light curve of the batch has no finite flux")
        h, B, n, nP, st = src.handle, len(src), src.n_cadences, len(period), _vp(src.stream or None)
        d_t, d_y, d_w = (DeviceBuffer(h, max(n, 1) * 8) for _ in range(3))
        d_ref = DeviceBuffer(h, max(B, 1) * 8)
        d_out = DeviceBuffer(h, max(7 * B * nP, 1) * 8)
        if B:
            lib = _capi._lib
            _capi._check(lib.lk_bls_prepare_batch_dev(h._h, B, _off_ptr(src.n_off), _vp(src.d_time.ptr), _vp(src.d_flux.ptr),
                                                      _vp(src.d_flux_err.ptr if src.d_flux_err is not None else None), _vp(d_t.ptr),
                                                      _vp(d_y.ptr), _vp(d_w.ptr), _vp(d_ref.ptr), st))
            d_per, k = _upload(h, period, src.stream)
            _capi._check(lib.lk_bls_batch_dev(h._h, B, _off_ptr(src.n_off), _vp(d_t.ptr), _vp(d_y.ptr), _vp(d_w.ptr),
                                              period.ctypes.data_as(_dp), _vp(d_per.ptr), nP, duration.ctypes.data_as(_dp),
                                              len(duration), int(oversample), int(objective == "likelihood"), _vp(d_out.ptr), st))
            src._keep.append(k)
        return DeviceBLSResult(src, d_out, d_ref, period, duration)

    # ---------------------------------------------------------------- fold / transit mask / bin
    def fold(self, period, epoch_time=None, epoch_phase=0.0, wrap_phase=None, normalize_phase=False):
        """``lc.fold(period, epoch_time, ...)`` for every light curve (reference :1089-1214): phases + stable sort + the flux
        columns gathered into phase order, resident -> ``DeviceFoldedBatch``.  ``period`` / ``epoch_time`` / ``wrap_phase``:
        scalars or one value per light curve; ``epoch_time`` defaults to each light curve's first time."""
        h, B, n = self.handle, len(self), self.n_cadences
        period = np.ascontiguousarray(np.broadcast_to(np.asarray(period, dtype=np.float64), (B,)))
        if not np.all(np.isfinite(period)) or np.any(period == 0):
            raise ValueError("period must be finite and non-zero")
        if epoch_time is None:
            epoch_time = np.empty(B, dtype=np.float64)
            _capi._check(_capi._lib.lk_gather_f64_dev(h._h, B, _off_ptr(np.ascontiguousarray(self.n_off[:-1])), _vp(self.d_time.ptr),
                                                      epoch_time.ctypes.data_as(_dp), _vp(self.stream or None)))
        epoch_time = np.ascontiguousarray(np.broadcast_to(np.asarray(epoch_time, dtype=np.float64), (B,)))
        if wrap_phase is None:
            wrap_phase = np.full(B, 0.5) if normalize_phase else period / 2.0
        wrap_phase = np.ascontiguousarray(np.broadcast_to(np.asarray(wrap_phase, dtype=np.float64), (B,)))
        cols = [self.d_flux] + ([self.d_flux_err] if self.d_flux_err is not None else [])
        outs = [DeviceBuffer(h, max(n, 1) * 8) for _ in cols]
        d_ph, d_ord = DeviceBuffer(h, max(n, 1) * 8), DeviceBuffer(h, max(n, 1) * 8)
        cin = (_vp * len(cols))(*[c.ptr for c in cols])
        cout = (_vp * len(cols))(*[c.ptr for c in outs])
        _capi._check(_capi._lib.lk_fold_batch_dev(h._h, B, _off_ptr(self.n_off), _vp(self.d_time.ptr), period.ctypes.data_as(_dp),
                                                  epoch_time.ctypes.data_as(_dp), float(epoch_phase), wrap_phase.ctypes.data_as(_dp),
                                                  int(bool(normalize_phase)), len(cols), cin, cout, _vp(d_ph.ptr), _vp(d_ord.ptr),
                                                  _vp(self.stream or None)))
        return DeviceFoldedBatch(self, d_ph, outs[0], outs[1] if len(outs) > 1 else None, d_ord, period, epoch_time)

    def create_transit_mask(self, period, transit_time, duration, planet_off=None, to_host=True):
        """In-transit flags over all cadences of the batch (reference :2967-3037) -> bool array on the host, or the
        ``DeviceBuffer`` of bytes (usable as ``flatten(mask=...)``) with ``to_host=False``."""
        h, B, n = self.handle, len(self), self.n_cadences
        period, duration, transit_time = (np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)
                                          for a in (period, duration, transit_time))
        if not (period.shape == duration.shape == transit_time.shape):
            raise ValueError("period, duration, and transit_time must have the same number of values.")
        if planet_off is None:
            k = period.size
            period, duration, transit_time = (np.tile(a, B) for a in (period, duration, transit_time))
            planet_off = np.arange(B + 1, dtype=np.int32) * k
        planet_off = np.ascontiguousarray(planet_off, dtype=np.int32)
        if planet_off.shape != (B + 1,) or planet_off[0] != 0 or planet_off[-1] != period.size:
            raise ValueError("planet_off must be B + 1 prefix offsets over the planet arrays")
        d_m = DeviceBuffer(h, max(n, 1))
        _capi._check(_capi._lib.lk_transit_mask_batch_dev(h._h, B, _off_ptr(self.n_off), _vp(self.d_time.ptr),
                                                          planet_off.ctypes.data_as(_i32p), period.ctypes.data_as(_dp),
                                                          duration.ctypes.data_as(_dp), transit_time.ctypes.data_as(_dp),
                                                          _vp(d_m.ptr), _vp(self.stream or None)))
        if not to_host:
            return d_m
        return d_m.download(np.uint8, n, stream=self.stream).astype(bool)

    def bin(self, time_bin_size=0.5, time_bin_start=None):
        """Equal-width time bins (reference :1558-1763 with ``time_bin_size`` in days), resident: nanmean flux, rms flux_err."""
        h, B, n, st = self.handle, len(self), self.n_cadences, _vp(self.stream or None)
        size_sec = float(time_bin_size) * 86400.0
        if not size_sec > 0:
            raise ValueError("time_bin_size must be positive")
        if not self._sorted():
            raise ValueError("bin needs the light curve sorted by time")
        idx = np.concatenate([self.n_off[:-1], np.maximum(self.n_off[1:] - 1, 0)]).astype(np.int64)
        ends = np.zeros(2 * B, dtype=np.float64)
        if n:
            _capi._check(_capi._lib.lk_gather_f64_dev(h._h, 2 * B, _off_ptr(np.minimum(idx, n - 1)), _vp(self.d_time.ptr),
                                                      ends.ctypes.data_as(_dp), st))
        counts = np.diff(self.n_off)
        fin = np.zeros(max(B, 1), dtype=np.int64)
        if self.d_flux_err is not None and B:
            _capi._check(_capi._lib.lk_segment_probe_batch_dev(h._h, B, _off_ptr(self.n_off), None, _vp(self.d_flux_err.ptr), None,
                                                               _off_ptr(fin), st))
        first, last = ends[:B], ends[B:]
        start = first.copy() if time_bin_start is None else np.broadcast_to(np.asarray(time_bin_start, float), (B,)).copy()
        start[counts == 0] = 0.0
        nb = np.where(counts > 0, np.maximum(0, np.ceil((last - start) * 86400.0 / size_sec)), 0).astype(np.int64)
        has_err = (fin[:B] > 0).astype(np.uint8)
        bin_off = np.zeros(B + 1, dtype=np.int64)
        bin_off[1:] = np.cumsum(nb)
        edges = np.cumsum(np.hstack([0.0, np.repeat(size_sec, int(nb.max()) if B else 0)]))
        nbt = int(bin_off[-1])
        d_t, d_f, d_e = (DeviceBuffer(h, max(nbt, 1) * 8) for _ in range(3))
        u8p = ctypes.POINTER(ctypes.c_uint8)
        _capi._check(_capi._lib.lk_bin_batch_dev(h._h, B, _off_ptr(self.n_off), _vp(self.d_time.ptr), _vp(self.d_flux.ptr),
                                                 _vp(self.d_flux_err.ptr if self.d_flux_err is not None else None), _off_ptr(bin_off),
                                                 start.ctypes.data_as(_dp), edges.ctypes.data_as(_dp), int(edges.size), size_sec,
                                                 has_err.ctypes.data_as(u8p), _vp(d_t.ptr), _vp(d_f.ptr), _vp(d_e.ptr), st))
        return self._new(d_t, d_f, d_e, bin_off, nan_free=False, is_sorted=True)


class DeviceFoldedBatch(object):
    """Folded light curves in HBM: phase (sorted), flux / flux_err in phase order, ``order`` (index of the cadence, relative to
    its light curve, at each sorted slot)."""

    def __init__(self, parent, d_phase, d_flux, d_flux_err, d_order, period, epoch_time):
        self.handle, self.stream, self.n_off = parent.handle, parent.stream, parent.n_off
        self.d_phase, self.d_flux, self.d_flux_err, self.d_order = d_phase, d_flux, d_flux_err, d_order
        self.period, self.epoch_time = period, epoch_time
        self.meta = parent.meta
        self._parent = parent            # (keeps the unfolded columns alive while the fold kernels may still read them)

    def __len__(self):
        return len(self.n_off) - 1

    def to_host(self):
        """-> dict(phase, flux, flux_err, order, n_off) of host arrays."""
        n = int(self.n_off[-1])
        return dict(phase=self.d_phase.download(np.float64, n, stream=self.stream),
                    flux=self.d_flux.download(np.float64, n, stream=self.stream),
                    flux_err=None if self.d_flux_err is None else self.d_flux_err.download(np.float64, n, stream=self.stream),
                    order=self.d_order.download(np.int64, n, stream=self.stream), n_off=self.n_off.copy())


class DeviceBLSResult(object):
    """out7[7, B, nP] in HBM (``_capi.BLS_FIELDS`` order; transit_time relative to ``t_ref``, the reference's
    ``min(t)`` per light curve) + what a pipeline keeps of it."""

    def __init__(self, batch, d_out7, d_t_ref, period, duration):
        self.handle, self.stream, self.B = batch.handle, batch.stream, len(batch)
        self.d_out7, self.d_t_ref, self.period, self.duration = d_out7, d_t_ref, period, duration
        self._batch = batch

    def to_host(self):
        """float64[B, 7, nP] as ``batch.bls_batch`` returns it (transit_time absolute, like the reference)."""
        B, nP = self.B, len(self.period)
        raw = self.d_out7.download(np.float64, 7 * B * nP, stream=self.stream).reshape(7, B, nP)
        t_ref = self.d_t_ref.download(np.float64, B, stream=self.stream)
        out = np.ascontiguousarray(np.transpose(raw, (1, 0, 2)))
        out[:, 4, :] += t_ref[:, None]
        return out

    def peaks(self):
        """Per light curve, at the maximum of the power: dict(max_power, argmax, period, transit_time (absolute), duration,
        depth) — ``BoxLeastSquaresPeriodogram.*_at_max_power`` (reference periodogram.py:1229-1260); 48 B per target cross PCIe."""
        h, B, nP, st = self.handle, self.B, len(self.period), _vp(self.stream or None)
        d_max, d_arg = DeviceBuffer(h, max(B, 1) * 8), DeviceBuffer(h, max(B, 1) * 8)
        if B == 0:
            z = np.zeros(0)
            return dict(max_power=z, argmax=z.astype(np.int64), period=z, transit_time=z, duration=z, depth=z)
        _capi._check(_capi._lib.lk_argmax_batch_dev(h._h, B, nP, _vp(self.d_out7.ptr), _vp(d_max.ptr), _vp(d_arg.ptr), st))
        mx = d_max.download(np.float64, B, stream=self.stream)
        am = d_arg.download(np.int64, B, stream=self.stream)
        a = np.clip(am, 0, nP - 1)
        rows = {"depth": 1, "duration": 3, "transit_time": 4}
        idx = np.concatenate([r * B * nP + np.arange(B) * nP + a for r in rows.values()]).astype(np.int64)
        vals = np.empty(idx.size, dtype=np.float64)
        _capi._check(_capi._lib.lk_gather_f64_dev(h._h, idx.size, _off_ptr(idx), _vp(self.d_out7.ptr), vals.ctypes.data_as(_dp), st))
        t_ref = self.d_t_ref.download(np.float64, B, stream=self.stream)
        got = {k: vals[i * B:(i + 1) * B] for i, k in enumerate(rows)}
        return dict(max_power=mx, argmax=am, period=self.period[a], transit_time=got["transit_time"] + t_ref,
                    duration=got["duration"], depth=got["depth"])
