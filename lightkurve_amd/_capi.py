"""ctypes binding of liblkhip.so (the C ABI in include/lkhip.h).

No CPU fallback exists anywhere in this package: if the library is missing or there is no GPU the
first compute call raises.  numpy arrays in -> numpy arrays out (host-pointer entry points); the
``*_dev`` variants take raw device pointers (ints) for callers that already hold data in HBM
(bench.py / batch.py pass ``torch.Tensor.data_ptr()``).
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LK_LIB_PATH") or os.path.join(_HERE, "liblkhip.so")   # LK_LIB_PATH: A/B builds of the library

LK_OK, LK_EINVAL, LK_ENOMEM, LK_EHIP = 0, 1, 2, 3
NORM = {"standard": 0, "psd": 1, "lk_amplitude": 2, "lk_psd": 3}
BLS_FIELDS = ("power", "depth", "depth_err", "duration", "transit_time", "depth_snr", "log_likelihood")

_c_dp = ctypes.POINTER(ctypes.c_double)
_c_ip = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p
_c_u8p = ctypes.POINTER(ctypes.c_uint8)
_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_fp = ctypes.POINTER(ctypes.c_float)

# (name, restype, argtypes) for EVERY symbol include/lkhip.h declares — tests/test_capi_symbols.py checks the list
SIGNATURES = [
    ("lk_version", ctypes.c_int, []),
    ("lk_last_error", ctypes.c_char_p, []),
    ("lk_device_count", ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    ("lk_init", ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    ("lk_destroy", None, [_vp]),
    ("lk_set_host_chunk_mb", ctypes.c_int, [_vp, ctypes.c_int]),
    ("lk_bls_max_period", ctypes.c_int, [_c_dp, ctypes.c_int, ctypes.c_int, _c_dp]),
    ("lk_bls_set_ordered_histogram", ctypes.c_int, [_vp, ctypes.c_int]),
    ("lk_pld_set_eig_tolerance", ctypes.c_int, [_vp, ctypes.c_double]),
    ("lk_pld_set_eig_mode", ctypes.c_int, [_vp, ctypes.c_int]),
    ("lk_workspace_bytes", ctypes.c_int64, [_vp]),
    ("lk_synchronize", ctypes.c_int, [_vp]),
    ("lk_ls_power_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_dp, _c_dp, _c_dp, _c_dp, ctypes.c_double, ctypes.c_double, ctypes.c_int64,
      ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp, _c_dp]),
    ("lk_ls_power_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_int64,
      ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    ("lk_ls_chi2_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_dp, _c_dp, _c_dp, _c_dp, ctypes.c_double, ctypes.c_double, ctypes.c_int64,
      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp, _c_dp]),
    ("lk_ls_chi2_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_int64,
      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    ("lk_ls_fastchi2_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_dp, _c_dp, _c_dp, ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
      ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp, ctypes.c_int, _c_dp]),
    ("lk_ls_fastchi2_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
      ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp]),
    ("lk_ls_fast_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_dp, _c_dp, _c_dp, ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
      ctypes.c_int, ctypes.c_int, _c_dp, ctypes.c_int, _c_dp]),
    ("lk_ls_fast_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
      ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp]),
    ("lk_ls_fast_peaks_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_dp, _c_dp, _c_dp, ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
      ctypes.c_int, ctypes.c_int, _c_dp, ctypes.c_int, _c_dp, _c_dp, _c_ip]),
    ("lk_ls_fast_peaks_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
      ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp, _vp]),
    ("lk_ls_fast_peaks_lc_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_dp, _c_dp, _c_dp, ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
      ctypes.c_int, ctypes.c_int, _c_dp, ctypes.c_int, _c_dp, _c_dp, _c_ip]),
    ("lk_dev_alloc", ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_size_t]),
    ("lk_dev_free", ctypes.c_int, [_vp, _vp]),
    ("lk_stream_create", ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    ("lk_stream_destroy", ctypes.c_int, [_vp, _vp]),
    ("lk_stream_synchronize", ctypes.c_int, [_vp, _vp]),
    ("lk_memcpy_h2d", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, _vp]),
    ("lk_memcpy_d2h", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, _vp]),
    ("lk_memcpy_d2d", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, _vp]),
    ("lk_flatten_apply_batch_dev", ctypes.c_int, [_vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("lk_ls_fast_peaks_lc_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_int64, ctypes.c_int,
      ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, _vp, _vp]),
    ("lk_rebase_times_batch_dev", ctypes.c_int, [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp]),
    ("lk_segment_probe_batch_dev", ctypes.c_int, [_vp, ctypes.c_int, _c_ip, _vp, _vp, _c_ip, _c_ip, _vp]),
    ("lk_bls_prepare_batch_dev", ctypes.c_int, [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("lk_compact_columns_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_ip, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp]),
    ("lk_gather_f64_dev", ctypes.c_int, [_vp, ctypes.c_int, _c_ip, _vp, _c_dp, _vp]),
    ("lk_shader_clock_mhz", ctypes.c_int, [_vp, ctypes.c_double, _c_dp]),
    ("lk_host_alloc", ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_size_t]),
    ("lk_host_free", ctypes.c_int, [_vp]),
    ("lk_fold_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_dp, _c_dp, _c_dp, ctypes.c_double, _c_dp, ctypes.c_int, ctypes.c_int,
      ctypes.POINTER(_c_dp), ctypes.POINTER(_c_dp), _c_dp, _c_ip]),
    ("lk_fold_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _vp, _c_dp, _c_dp, ctypes.c_double, _c_dp, ctypes.c_int, ctypes.c_int,
      ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _vp, _vp]),
    ("lk_pg_logmedian_batch", ctypes.c_int,
     [_vp, ctypes.c_int, ctypes.c_int64, _c_dp, ctypes.c_int, _c_i32p, _c_i32p, _c_i32p, _c_i32p, ctypes.c_double,
      _c_dp]),
    ("lk_pg_logmedian_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, ctypes.c_int64, _vp, ctypes.c_int, _c_i32p, _c_i32p, _c_i32p, _c_i32p, ctypes.c_double, _vp,
      _vp]),
    ("lk_pg_boxsmooth_batch", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64, _c_dp, _c_dp, ctypes.c_int, _c_dp]),
    ("lk_pg_boxsmooth_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, ctypes.c_int64, _vp, _c_dp, ctypes.c_int, _vp, _vp]),
    ("lk_pg_acf2d_batch", ctypes.c_int,
     [_vp, ctypes.c_int, ctypes.c_int64, _c_dp, ctypes.c_int, _c_i32p, ctypes.c_int, _c_dp, _c_dp]),
    ("lk_pg_acf2d_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, ctypes.c_int64, _vp, ctypes.c_int, _c_i32p, ctypes.c_int, _vp, _vp, _vp]),
    ("lk_sigma_clip_batch", ctypes.c_int, [_vp, ctypes.c_int, _c_ip, _c_dp, ctypes.c_double, ctypes.c_int, _c_u8p]),
    ("lk_sigma_clip_batch_dev", ctypes.c_int, [_vp, ctypes.c_int, _c_ip, _vp, ctypes.c_double, ctypes.c_int, _vp, _vp]),
    ("lk_ingest_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_dp, _c_dp, _c_dp, ctypes.c_int, _c_dp, _c_dp, _c_dp, _c_ip, _c_dp]),
    ("lk_ingest_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _c_ip, _vp, _vp]),
    ("lk_fits_unpack_batch", ctypes.c_int, [_vp, ctypes.c_int, _c_u8p, _c_ip, _c_i32p, _c_ip, _c_dp, _c_dp, _c_dp, _c_i32p, _c_ip]),
    ("lk_fits_unpack_batch_dev", ctypes.c_int, [_vp, ctypes.c_int, _vp, _c_ip, _c_i32p, _c_ip, _vp, _vp, _vp, _vp, _c_ip, _vp]),
    ("lk_fits_unpack_cube", ctypes.c_int, [_vp, _c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_i32p, ctypes.c_int, _c_dp,
                                            _c_i32p, ctypes.POINTER(ctypes.c_float), _c_ip]),
    ("lk_fits_unpack_cube_dev", ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _c_i32p, ctypes.c_int, _vp,
                                                _vp, _vp, _c_ip, _vp]),
    ("lk_transit_mask_batch", ctypes.c_int, [_vp, ctypes.c_int, _c_ip, _c_dp, _c_i32p, _c_dp, _c_dp, _c_dp, _c_u8p]),
    ("lk_transit_mask_batch_dev", ctypes.c_int, [_vp, ctypes.c_int, _c_ip, _vp, _c_i32p, _c_dp, _c_dp, _c_dp, _vp, _vp]),
    ("lk_bin_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_dp, _c_dp, _c_dp, _c_ip, _c_dp, _c_dp, ctypes.c_int64, ctypes.c_double, _c_u8p, _c_dp,
      _c_dp, _c_dp]),
    ("lk_bin_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp, _c_ip, _c_dp, _c_dp, ctypes.c_int64, ctypes.c_double, _c_u8p, _vp, _vp, _vp,
      _vp]),
    ("lk_argmax_batch", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64, _c_dp, _c_dp, _c_ip]),
    ("lk_argmax_batch_dev", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    ("lk_bls_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_dp, _c_dp, _c_dp, _c_dp, ctypes.c_int64, _c_dp, ctypes.c_int, ctypes.c_int,
      ctypes.c_int, _c_dp]),
    ("lk_bls_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp, _c_dp, _vp, ctypes.c_int64, _c_dp, ctypes.c_int, ctypes.c_int,
      ctypes.c_int, _vp, _vp]),
    ("lk_regress_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, ctypes.c_int, _c_dp, _c_dp, _c_dp, _c_u8p, _c_dp, _c_dp, ctypes.c_double,
      ctypes.c_int, _c_dp, _c_dp, _c_u8p]),
    ("lk_regress_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_int, _vp,
      _vp, _vp, _vp]),
    ("lk_regress_cov_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, ctypes.c_int, _c_dp, _c_dp, _c_dp, _c_u8p, _c_dp, _c_dp, ctypes.c_double,
      ctypes.c_int, _c_dp, _c_dp, _c_u8p, _c_dp]),
    ("lk_regress_cov_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_int, _vp,
      _vp, _vp, _vp, _vp]),
    ("lk_savgol_trend_batch", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _c_dp, _c_dp, _c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
      ctypes.c_double, _c_dp, _c_u8p]),
    ("lk_savgol_trend_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, _c_ip, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
      ctypes.c_double, _vp, _vp, _vp]),
    ("lk_savgol_design", ctypes.c_int, [ctypes.c_int, ctypes.c_int, _c_dp, _c_dp]),
    ("lk_pld_design_width", ctypes.c_int, [ctypes.c_int] * 5),
    ("lk_pld_design_batch", ctypes.c_int,
     [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_fp, _c_fp, _c_fp, _c_dp, _c_dp, ctypes.c_int,
      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp, _c_dp]),
    ("lk_pld_design_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_int,
      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    ("lk_pld_correct_batch", ctypes.c_int,
     [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_fp, _c_fp, _c_fp, _c_dp, _c_dp, ctypes.c_int,
      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp, _c_dp, _c_u8p,
      ctypes.c_double, ctypes.c_int, _c_dp, _c_dp, _c_u8p, _c_dp]),
    ("lk_pca_batch", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp, _c_dp]),
    ("lk_pca_batch_dev", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    ("lk_spline_basis_batch", ctypes.c_int,
     [_vp, ctypes.c_int, ctypes.c_int, _c_dp, _c_dp, ctypes.c_int, ctypes.c_int, _c_dp]),
    ("lk_spline_basis_batch_dev", ctypes.c_int,
     [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    ("lk_standardize_batch", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_dp, _c_dp]),
    ("lk_standardize_batch_dev", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
]

_lib = None


class LkHipError(RuntimeError):
    """HIP runtime failure inside liblkhip.so (LK_EHIP)."""


def load_library(path=None):
    """dlopen liblkhip.so and type every entry point.  Raises OSError if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise OSError(
            "%s not found: build it first (python -c 'import __graft_entry__ as g; g.build()' or "
            "make -C lightkurve_amd/csrc).  lightkurve_amd has no CPU fallback." % path)
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(path)
    for name, restype, argtypes in SIGNATURES:
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's); if
    liblkhip.so pulls in /opt/rocm's copy first and torch is imported LATER, torch ends up on a second runtime and sees
    no GPU (round 1's import-order trap).  So when torch is installed but not imported yet, its bundled runtime is
    loaded first with RTLD_GLOBAL: liblkhip.so's DT_NEEDED libamdhip64.so.7 then resolves to that already-loaded
    object, and a later `import torch` reuses it — the situation bench.py (torch first) has always been in.
    LK_NO_TORCH_HIP=1 disables this (e.g. to force /opt/rocm's runtime)."""
    import sys
    if "torch" in sys.modules or os.environ.get("LK_NO_TORCH_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        libdir = os.path.join(os.path.dirname(spec.origin), "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            cand = os.path.join(libdir, name)
            if os.path.exists(cand):
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:      # never fatal: fall back to the plain load
        pass


def _check(rc):
    if rc == LK_OK:
        return
    msg = (_lib.lk_last_error() or b"").decode("utf-8", "replace")
    if rc == LK_EINVAL:
        raise ValueError(msg)
    if rc == LK_ENOMEM:
        raise MemoryError(msg)
    raise LkHipError(msg)


class Handle:
    """One lk_handle = one GPU.  Created lazily per device id and cached."""

    _cache = {}

    def __init__(self, device=0):
        lib = load_library()
        self._h = _vp()
        _check(lib.lk_init(int(device), ctypes.byref(self._h)))
        self.device = int(device)

    @classmethod
    def get(cls, device=0):
        device = int(device)
        if device not in cls._cache:
            cls._cache[device] = cls(device)
        return cls._cache[device]

    def close(self):
        if self._h:
            _lib.lk_destroy(self._h)
            self._h = _vp()

    def synchronize(self):
        _check(_lib.lk_synchronize(self._h))

    def workspace_bytes(self):
        return int(_lib.lk_workspace_bytes(self._h))

    def bls_set_ordered_histogram(self, on):
        """Force (True) the atomic-free BLS histogram — the form the library falls back to by itself on a device whose LDS
        ds_add_f64 is not lane-ordered; False returns to the automatic choice."""
        _check(_lib.lk_bls_set_ordered_histogram(self._h, int(bool(on))))

    def pld_set_eig_tolerance(self, tol):
        """Stop of the subspace iteration behind the PLD design matrices' PCA blocks (relative residual; 0 = the default 1e-7)."""
        _check(_lib.lk_pld_set_eig_tolerance(self._h, float(tol)))

    def pld_set_eig_mode(self, mode):
        """0 (default): the one-kernel subspace iteration behind the wide PCA blocks; 1: its phase-split form (one launch per phase
        over all matrices) — same results to rounding, kept for per-phase profiling (include/lkhip.h)."""
        _check(_lib.lk_pld_set_eig_mode(self._h, int(mode)))

    def set_host_chunk_mb(self, mb):
        """MiB of spectra per chunk of the pinned host pipeline behind ls_fast_batch / ls_fast_peaks_batch (default 64, or
        LK_HOST_CHUNK_MB when the handle was created)."""
        _check(_lib.lk_set_host_chunk_mb(self._h, int(mb)))


def device_count():
    lib = load_library()
    n = ctypes.c_int(0)
    rc = lib.lk_device_count(ctypes.byref(n))
    return n.value if rc == LK_OK else 0


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, typ=_c_dp):
    return a.ctypes.data_as(typ) if a is not None else None


def _offsets(n_off, total):
    n_off = np.ascontiguousarray(n_off, dtype=np.int64)
    if n_off.ndim != 1 or n_off.size < 1 or n_off[0] != 0 or n_off[-1] != total or np.any(np.diff(n_off) < 0):
        raise ValueError("n_off must be non-decreasing prefix offsets starting at 0 and ending at len(t)")
    return n_off


# --------------------------------------------------------------------------------------------- Lomb-Scargle
MAX_NTERMS = 8  # LK_MAX_NTERMS in include/lkhip.h (5..8: exact sums for both 'chi2' and 'fastchi2')


def ls_power_batch(t, y, n_off, dy=None, frequency=None, f0=0.0, df=0.0, M=None, fit_mean=True,
                   center_data=True, normalization="psd", scale=None, device=0, nterms=1):
    """Exact GLS power for B ragged targets -> float64[B, M].

    ``t`` relative times [d], ``y`` flux, ``dy`` errors or None (uniform weights), concatenated over
    targets with prefix offsets ``n_off``.  Either ``frequency`` (any 1-D array, 1/d) or the regular grid
    ``f0 + df*arange(M)`` (the fast kernel).  ``nterms`` > 1: astropy's multi-term ``chi2`` periodogram."""
    if not 1 <= int(nterms) <= MAX_NTERMS:
        raise ValueError("nterms must be between 1 and %d on the HIP path (got %r)" % (MAX_NTERMS, nterms))
    h = Handle.get(device)
    t, y = _f64(t), _f64(y)
    n_off = _offsets(n_off, t.size)
    if y.shape != t.shape:
        raise ValueError("t and y must have the same length")
    dy = None if dy is None else _f64(np.broadcast_to(dy, t.shape))
    B = n_off.size - 1
    if frequency is not None:
        frequency = _f64(frequency).ravel()
        M = frequency.size
    M = int(M)
    scale = None if scale is None else _f64(np.broadcast_to(scale, (B,)))
    power = result_empty((B, M))
    _check(_lib.lk_ls_chi2_batch(h._h, B, _ptr(n_off, _c_ip), _ptr(t), _ptr(y), _ptr(dy), _ptr(frequency),
                                 float(f0), float(df), M, int(nterms), int(bool(fit_mean)), int(bool(center_data)),
                                 NORM[normalization], _ptr(scale), _ptr(power)))
    return power


def ls_power_batch_dev(handle, B, n_off_host, t_ptr, y_ptr, dy_ptr, freq_ptr, f0, df, M, fit_mean, center_data,
                       normalization, scale_ptr, power_ptr, stream=0, nterms=1):
    """Device-pointer variant (ints from tensor.data_ptr()); enqueues on ``stream`` and returns."""
    n_off_host = np.ascontiguousarray(n_off_host, dtype=np.int64)
    _check(_lib.lk_ls_chi2_batch_dev(handle._h, int(B), _ptr(n_off_host, _c_ip), _vp(t_ptr), _vp(y_ptr),
                                     _vp(dy_ptr or None), _vp(freq_ptr or None), float(f0), float(df), int(M),
                                     int(nterms), int(bool(fit_mean)), int(bool(center_data)), NORM[normalization],
                                     _vp(scale_ptr or None), _vp(power_ptr), _vp(stream or None)))


def ls_fast_batch(t, y, n_off, dy=None, f0=0.0, df=0.0, M=0, fit_mean=True, center_data=True, normalization="psd",
                  scale=None, oversampling=5, device=0, nterms=1):
    """The reference's default ``ls_method="fast"`` (extirpolation + FFT) for B ragged targets on the regular grid
    ``f0 + df*arange(M)`` -> float64[B, M].  ``nterms`` > 1: its multi-term sibling ``"fastchi2"``."""
    if not 1 <= int(nterms) <= MAX_NTERMS:
        raise ValueError("nterms must be between 1 and %d on the HIP path (got %r)" % (MAX_NTERMS, nterms))
    h = Handle.get(device)
    t, y = _f64(t), _f64(y)
    n_off = _offsets(n_off, t.size)
    if y.shape != t.shape:
        raise ValueError("t and y must have the same length")
    dy = None if dy is None else _f64(np.broadcast_to(dy, t.shape))
    B, M = n_off.size - 1, int(M)
    scale = None if scale is None else _f64(np.broadcast_to(scale, (B,)))
    power = result_empty((B, M))
    _check(_lib.lk_ls_fastchi2_batch(h._h, B, _ptr(n_off, _c_ip), _ptr(t), _ptr(y), _ptr(dy), float(f0), float(df), M,
                                     int(nterms), int(bool(fit_mean)), int(bool(center_data)), NORM[normalization],
                                     _ptr(scale), int(oversampling), _ptr(power)))
    return power


def ls_fast_batch_dev(handle, B, n_off_host, t_ptr, y_ptr, dy_ptr, f0, df, M, fit_mean, center_data, normalization,
                      scale_ptr, oversampling, power_ptr, stream=0):
    n_off_host = np.ascontiguousarray(n_off_host, dtype=np.int64)
    _check(_lib.lk_ls_fast_batch_dev(handle._h, int(B), _ptr(n_off_host, _c_ip), _vp(t_ptr), _vp(y_ptr),
                                     _vp(dy_ptr or None), float(f0), float(df), int(M), int(bool(fit_mean)),
                                     int(bool(center_data)), NORM[normalization], _vp(scale_ptr or None),
                                     int(oversampling), _vp(power_ptr), _vp(stream or None)))


def ls_fast_peaks_batch_dev(handle, B, n_off_host, t_ptr, y_ptr, dy_ptr, f0, df, M, fit_mean, center_data,
                            normalization, scale_ptr, oversampling, power_ptr, max_ptr, arg_ptr, stream=0):
    """lk_ls_fast_peaks_batch_dev: ls_method='fast' + per-target (nanmax, nanargmax), all device pointers."""
    n_off_host = np.ascontiguousarray(n_off_host, dtype=np.int64)
    _check(_lib.lk_ls_fast_peaks_batch_dev(handle._h, int(B), _ptr(n_off_host, _c_ip), _vp(t_ptr), _vp(y_ptr),
                                           _vp(dy_ptr or None), float(f0), float(df), int(M), int(bool(fit_mean)),
                                           int(bool(center_data)), NORM[normalization], _vp(scale_ptr or None),
                                           int(oversampling), _vp(power_ptr), _vp(max_ptr or None), _vp(arg_ptr or None),
                                           _vp(stream or None)))


def pinned_empty(shape, dtype=np.float64):
    """numpy array over page-locked host memory (lk_host_alloc): the host-pointer entry points DMA such buffers
    directly instead of staging them.  Freed when the array (and every view of it) is garbage collected."""
    import weakref
    load_library()
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    ptr = _vp()
    _check(_lib.lk_host_alloc(ctypes.byref(ptr), max(nbytes, 1)))
    buf = (ctypes.c_char * max(nbytes, 1)).from_address(ptr.value)
    weakref.finalize(buf, _lib.lk_host_free, _vp(ptr.value))
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)


_POOL = {}
_POOL_LOCK = threading.Lock()


def pinned_pool(key, count, dtype=np.float64):
    """float64[count] view of a page-locked staging buffer that is kept (and only ever grown) per ``key`` AND per calling
    thread: the packing step of the batch front end writes the concatenated light curves here so that the host-pointer
    entry points DMA them without the runtime's pageable staging.  The contents are valid until the next request for the
    same key from the same thread (ctypes releases the GIL inside lk_* calls: two threads packing into one buffer would
    overwrite each other's batch while its DMA is in flight — ADVICE r5)."""
    dtype = np.dtype(dtype)
    count = int(count)
    key = (threading.get_ident(), key)
    with _POOL_LOCK:
        buf = _POOL.get(key)
        if buf is None or buf.nbytes < count * dtype.itemsize:
            _POOL.pop(key, None)
            cap = max(count * dtype.itemsize + (count * dtype.itemsize >> 3), 1 << 16)
            buf = _POOL[key] = pinned_empty(cap, np.uint8)
    return buf[: count * dtype.itemsize].view(dtype)


def release_pinned_pool():
    """Free the page-locked staging buffers of every thread and the idle blocks of the result pool."""
    with _POOL_LOCK:
        _POOL.clear()
    if _RESULT_POOL is not None:
        _RESULT_POOL.release()


# Recycled RESULT memory.  A fresh 160-MB / 800-MB numpy result costs more in first-touch page faults (10 / 50 ms) than its
# transfer from the device (3 / 15 ms), so large results are placed in page-locked blocks that are kept for the next call.
# Ownership is numpy's own: every array handed out is built over a fresh LEASE object (its ``base`` chain ends there), and a
# block returns to the free list only when that lease is garbage collected — i.e. exactly when numpy would have freed the
# memory of an ordinary array: after the array, all its views and every memoryview / buffer export of it are gone.  (Rounds
# 4-5 looked at ``sys.getrefcount`` of a shared root array instead, which a buffer export does not raise — VERDICT r5 #8.)
# Idle blocks are capped (LK_RESULT_POOL_MB, default 2048: beyond it a returned block is freed at once);
# ``release_pinned_pool()`` frees them all; LK_RESULT_POOL=0 turns the recycling off (plain ``np.empty``).
_RESULT_POOL_MIN_BYTES = 1 << 22


class _ResultPool(object):
    def __init__(self, alloc, free, idle_limit_bytes):
        self._alloc, self._free, self.idle_limit = alloc, free, int(idle_limit_bytes)
        self.idle = []                   # (capacity, address) of blocks nobody holds
        self.lock = threading.Lock()
        self.leased = 0                  # blocks currently out (diagnostics / tests)

    def idle_bytes(self):
        with self.lock:
            return sum(c for c, _ in self.idle)

    def take(self, need):
        """(capacity, address) of a block of at least ``need`` bytes: the smallest idle one that is not wastefully large,
        else a new allocation."""
        with self.lock:
            fit = [i for i, (c, _) in enumerate(self.idle) if need <= c <= 2 * need + (1 << 20)]
            if fit:
                self.leased += 1
                return self.idle.pop(min(fit, key=lambda i: self.idle[i][0]))
        cap = need + (need >> 4)
        addr = self._alloc(cap)
        with self.lock:
            self.leased += 1
        return cap, addr

    def give(self, cap, addr):
        with self.lock:
            self.leased -= 1
            keep = sum(c for c, _ in self.idle) + cap <= self.idle_limit
            if keep:
                self.idle.append((cap, addr))
        if not keep:
            self._free(addr)

    def release(self):
        with self.lock:
            items, self.idle = self.idle, []
        for _c, addr in items:
            self._free(addr)

    def empty(self, shape, dtype):
        import weakref
        need = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        cap, addr = self.take(need)
        lease = (ctypes.c_char * cap).from_address(addr)              # does not own the memory; the finalizer returns it
        weakref.finalize(lease, self.give, cap, addr)
        return np.frombuffer(lease, dtype=dtype, count=need // dtype.itemsize).reshape(shape)


_RESULT_POOL = None
_RESULT_POOL_INIT = threading.Lock()


def _pinned_alloc(nbytes):
    ptr = _vp()
    _check(_lib.lk_host_alloc(ctypes.byref(ptr), max(int(nbytes), 1)))
    return ptr.value


def _pinned_free(addr):
    _lib.lk_host_free(_vp(addr))


def _result_pool():
    global _RESULT_POOL
    with _RESULT_POOL_INIT:
        if _RESULT_POOL is None:
            load_library()
            try:
                mb = int(os.environ.get("LK_RESULT_POOL_MB", "2048"))
            except ValueError:
                mb = 2048
            _RESULT_POOL = _ResultPool(_pinned_alloc, _pinned_free, max(mb, 0) << 20)
    return _RESULT_POOL


def result_empty(shape, dtype=np.float64):
    """Uninitialised array for a large result of a batch call (``np.empty`` semantics).  Results of 4 MB and more live in
    page-locked memory that is RECYCLED: once the array and everything that views or exports it has been garbage collected,
    the next large result may be placed at the same address.  An address obtained through ``arr.ctypes.data`` is valid for
    as long as ``arr`` (or a view of it) is alive — as for any numpy array.  LK_RESULT_POOL=0: plain ``np.empty``."""
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
    dtype = np.dtype(dtype)
    need = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if need < _RESULT_POOL_MIN_BYTES or os.environ.get("LK_RESULT_POOL", "1") == "0":
        return np.empty(shape, dtype=dtype)
    try:
        return _result_pool().empty(shape, dtype)
    except (OSError, RuntimeError, MemoryError):          # no library / no GPU runtime / pinned memory exhausted
        return np.empty(shape, dtype=dtype)


def ls_fast_peaks_batch(t, y, n_off, dy=None, f0=0.0, df=0.0, M=0, fit_mean=True, center_data=True,
                        normalization="psd", scale=None, oversampling=5, device=0, out=None, want_power=True,
                        want_peaks=True, absolute_time=False):
    """ls_method='fast' for B ragged targets through the pipelined host-pointer entry point.
    Returns (power[B, M] or None, max_power[B] or None, argmax[B] or None).  ``out``: preallocated float64[B, M]
    (e.g. from ``pinned_empty``) to receive the spectra; ``want_power=False`` keeps the spectra on the device and
    returns only the per-target peaks (Periodogram.max_power / nanargmax).  ``absolute_time``: ``t`` holds the light
    curves' own (absolute) times and each is rebased to ``t - t[first]`` on the device (lk_ls_fast_peaks_lc_batch)."""
    h = Handle.get(device)
    t, y = _f64(t), _f64(y)
    n_off = _offsets(n_off, t.size)
    if y.shape != t.shape:
        raise ValueError("t and y must have the same length")
    dy = None if dy is None else _f64(np.broadcast_to(dy, t.shape))
    B, M = n_off.size - 1, int(M)
    scale = None if scale is None else _f64(np.broadcast_to(scale, (B,)))
    power = None
    if want_power:
        power = out if out is not None else result_empty((B, M))
        if power.shape != (B, M) or power.dtype != np.float64 or not power.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array of shape (B, M)")
    if not want_power and not want_peaks:
        raise ValueError("nothing requested")
    mx = np.empty(B, dtype=np.float64) if want_peaks else None
    am = np.empty(B, dtype=np.int64) if want_peaks else None
    entry = _lib.lk_ls_fast_peaks_lc_batch if absolute_time else _lib.lk_ls_fast_peaks_batch
    _check(entry(h._h, B, _ptr(n_off, _c_ip), _ptr(t), _ptr(y), _ptr(dy), float(f0), float(df), M,
                 int(bool(fit_mean)), int(bool(center_data)), NORM[normalization], _ptr(scale),
                 int(oversampling), _ptr(power), _ptr(mx), _ptr(am, _c_ip)))
    return power, mx, am


# --------------------------------------------------------------------------------------------- Periodogram.smooth
def pg_logmedian_batch(power, win_lo, win_hi, klo, khi, corr=(8.0 / 9.0) ** 3, device=0):
    """Moving nanmedian in log-frequency windows for B periodograms on one grid (power float64[B, M]); the window
    tables come from ``periodogram._logmedian_windows`` -> float64[B, M]."""
    h = Handle.get(device)
    power = np.ascontiguousarray(np.atleast_2d(power), dtype=np.float64)
    B, M = power.shape
    tabs = [np.ascontiguousarray(a, dtype=np.int32) for a in (win_lo, win_hi, klo, khi)]
    if tabs[0].shape != tabs[1].shape or tabs[2].shape != (M,) or tabs[3].shape != (M,):
        raise ValueError("window tables do not match the grid")
    out = np.empty_like(power)
    _check(_lib.lk_pg_logmedian_batch(h._h, B, M, _ptr(power), int(tabs[0].size), _ptr(tabs[0], _c_i32p),
                                      _ptr(tabs[1], _c_i32p), _ptr(tabs[2], _c_i32p), _ptr(tabs[3], _c_i32p),
                                      float(corr), _ptr(out)))
    return out


def pg_boxsmooth_batch(power, kernel, device=0):
    """astropy.convolution.convolve(power[b], kernel) (boundary='fill', normalised, NaN-interpolating) for every row."""
    h = Handle.get(device)
    power = np.ascontiguousarray(np.atleast_2d(power), dtype=np.float64)
    B, M = power.shape
    taps = np.ascontiguousarray(np.asarray(kernel, dtype=np.float64)[::-1])
    out = np.empty_like(power)
    _check(_lib.lk_pg_boxsmooth_batch(h._h, B, M, _ptr(power), _ptr(taps), int(taps.size), _ptr(out)))
    return out


def pg_acf2d_batch(power, win_start, W, device=0):
    """Windowed autocorrelations of B periodograms on one grid (power float64[B, M]): for every window start the ACF of
    the W mean-subtracted samples and the mean collapsed correlation.  Returns (acf2d[B, n_win, W], metric[B, n_win])."""
    h = Handle.get(device)
    power = np.ascontiguousarray(np.atleast_2d(power), dtype=np.float64)
    B, M = power.shape
    ws = np.ascontiguousarray(win_start, dtype=np.int32).ravel()
    W = int(W)
    if ws.size and (ws.min() < 0 or int(ws.max()) + W > M):
        raise ValueError("a window reaches outside the spectrum")
    acf = np.empty((B, ws.size, W), dtype=np.float64)
    met = np.empty((B, ws.size), dtype=np.float64)
    _check(_lib.lk_pg_acf2d_batch(h._h, B, M, _ptr(power), int(ws.size), _ptr(ws, _c_i32p), W, _ptr(acf), _ptr(met)))
    return acf, met


def argmax_batch(x, device=0):
    """Row-wise (nanmax, nanargmax) of float64[B, M]; first maximum wins; all-NaN row -> (nan, -1)."""
    h = Handle.get(device)
    x = _f64(x)
    if x.ndim != 2:
        raise ValueError("x must be 2-D")
    B, M = x.shape
    mx = np.empty(B, dtype=np.float64)
    am = np.empty(B, dtype=np.int64)
    _check(_lib.lk_argmax_batch(h._h, B, M, _ptr(x), _ptr(mx), _ptr(am, _c_ip)))
    return mx, am


def argmax_batch_dev(handle, B, M, x_ptr, max_ptr, arg_ptr, stream=0):
    _check(_lib.lk_argmax_batch_dev(handle._h, int(B), int(M), _vp(x_ptr), _vp(max_ptr), _vp(arg_ptr),
                                    _vp(stream or None)))


# --------------------------------------------------------------------------------------------- fold
def fold_batch(t, n_off, period, epoch_time, epoch_phase=0.0, wrap_phase=None, normalize_phase=False, columns=(),
               device=0):
    """LightCurve.fold for B ragged targets: returns (phase, order, [columns gathered into phase order]).
    ``period`` / ``epoch_time`` / ``wrap_phase``: scalars or one value per target; ``wrap_phase`` defaults to
    period/2 (0.5 with ``normalize_phase``), like astropy.  ``order`` indexes cadences relative to their target."""
    h = Handle.get(device)
    t = _f64(t)
    n_off = _offsets(n_off, t.size)
    B = n_off.size - 1
    period = _f64(np.broadcast_to(np.asarray(period, dtype=np.float64), (B,)))
    epoch_time = _f64(np.broadcast_to(np.asarray(epoch_time, dtype=np.float64), (B,)))
    if wrap_phase is None:
        wrap_phase = np.full(B, 0.5) if normalize_phase else period / 2.0
    wrap_phase = _f64(np.broadcast_to(np.asarray(wrap_phase, dtype=np.float64), (B,)))
    if not np.all(np.isfinite(period)) or np.any(period == 0):
        raise ValueError("period must be finite and non-zero")
    cols = [_f64(c) for c in columns]
    for c in cols:
        if c.shape != t.shape:
            raise ValueError("every column must have the length of t")
    outs = [np.empty_like(t) for _ in cols]
    arr_t = _c_dp * max(len(cols), 1)
    cin = arr_t(*[_ptr(c) for c in cols]) if cols else arr_t()
    cout = arr_t(*[_ptr(o) for o in outs]) if cols else arr_t()
    phase = np.empty_like(t)
    order = np.empty(t.size, dtype=np.int64)
    _check(_lib.lk_fold_batch(h._h, B, _ptr(n_off, _c_ip), _ptr(t), _ptr(period), _ptr(epoch_time), float(epoch_phase),
                              _ptr(wrap_phase), int(bool(normalize_phase)), len(cols), cin, cout, _ptr(phase),
                              _ptr(order, _c_ip)))
    return phase, order, outs


# --------------------------------------------------------------------------------------------- BLS
def bls_batch(t, y, ivar, n_off, period, duration, oversample=10, use_likelihood=True, device=0):
    """astropy ``bls_fast`` for B ragged targets (inputs already t-min(t), y-median(y), ivar).
    Returns a dict of float64[B, nP] arrays keyed by BLS_FIELDS (transit_time is the phase in (0, period))."""
    h = Handle.get(device)
    t, y, ivar = _f64(t), _f64(y), _f64(ivar)
    n_off = _offsets(n_off, t.size)
    if y.shape != t.shape or ivar.shape != t.shape:
        raise ValueError("t, y, ivar must have the same length")
    period, duration = _f64(period).ravel(), _f64(duration).ravel()
    B, nP = n_off.size - 1, period.size
    out = result_empty((7, B, nP))
    _check(_lib.lk_bls_batch(h._h, B, _ptr(n_off, _c_ip), _ptr(t), _ptr(y), _ptr(ivar), _ptr(period), nP,
                             _ptr(duration), duration.size, int(oversample), int(bool(use_likelihood)), _ptr(out)))
    return {k: out[i] for i, k in enumerate(BLS_FIELDS)}


def bls_max_period(duration, oversample=10):
    """The longest period the LDS kernels of ``bls_batch`` take for these durations; longer ones run the (slower, bit-identical)
    global-memory kernel.  Host-only."""
    load_library()
    duration = _f64(np.atleast_1d(duration)).ravel()
    out = ctypes.c_double(0.0)
    _check(_lib.lk_bls_max_period(_ptr(duration), duration.size, int(oversample), ctypes.byref(out)))
    return float(out.value)


def bls_batch_dev(handle, B, n_off_host, t_ptr, y_ptr, ivar_ptr, period_host, period_ptr, duration_host, oversample,
                  use_likelihood, out7_ptr, stream=0):
    n_off_host = np.ascontiguousarray(n_off_host, dtype=np.int64)
    period_host, duration_host = _f64(period_host).ravel(), _f64(duration_host).ravel()
    _check(_lib.lk_bls_batch_dev(handle._h, int(B), _ptr(n_off_host, _c_ip), _vp(t_ptr), _vp(y_ptr), _vp(ivar_ptr),
                                 _ptr(period_host), _vp(period_ptr), period_host.size, _ptr(duration_host),
                                 duration_host.size, int(oversample), int(bool(use_likelihood)), _vp(out7_ptr),
                                 _vp(stream or None)))


# --------------------------------------------------------------------------------------------- regression
def regress_batch(X, y, n_off, err=None, cadence_mask=None, prior_mu=None, prior_sigma=None, sigma=5.0, niters=5,
                  device=0, return_cov=False):
    """RegressionCorrector.correct numerics for B ragged targets sharing K columns.
    X: (sum N, K); returns dict(coefficients[B,K], model[sum N] (median-subtracted), outlier_mask[sum N] bool);
    ``return_cov``: also coefficients_cov[B,K,K] = inverse normal matrix of the last fit (propagate_errors=True)."""
    h = Handle.get(device)
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim != 2:
        raise ValueError("X must be 2-D (cadences x regressors)")
    ntot, K = X.shape
    y = _f64(y)
    n_off = _offsets(n_off, ntot)
    if y.shape != (ntot,):
        raise ValueError("y must have one value per row of X")
    B = n_off.size - 1
    err = None if err is None else _f64(np.broadcast_to(err, (ntot,)))
    cm = None if cadence_mask is None else np.ascontiguousarray(cadence_mask, dtype=np.uint8)
    if cm is not None and cm.shape != (ntot,):
        raise ValueError("cadence_mask must have one entry per row of X (got shape %s, need (%d,))" % (cm.shape, ntot))
    if (prior_mu is None) != (prior_sigma is None):
        raise ValueError("Please specify both `prior_mu` and `prior_sigma`")
    if prior_mu is not None:
        prior_mu = _f64(np.broadcast_to(np.asarray(prior_mu, dtype=np.float64), (B, K)))
        prior_sigma = _f64(np.broadcast_to(np.asarray(prior_sigma, dtype=np.float64), (B, K)))
    w = np.empty((B, K), dtype=np.float64)
    model = np.empty(ntot, dtype=np.float64)
    outl = np.empty(ntot, dtype=np.uint8)
    cov = np.empty((B, K, K), dtype=np.float64) if return_cov else None
    _check(_lib.lk_regress_cov_batch(h._h, B, _ptr(n_off, _c_ip), K, _ptr(X), _ptr(y), _ptr(err), _ptr(cm, _c_u8p),
                                     _ptr(prior_mu), _ptr(prior_sigma), float(sigma), int(niters), _ptr(w), _ptr(model),
                                     _ptr(outl, _c_u8p), _ptr(cov)))
    res = dict(coefficients=w, model=model, outlier_mask=outl.astype(bool))
    if return_cov:
        res["coefficients_cov"] = cov
    return res


# --------------------------------------------------------------------------------------------- flatten
def savgol_design(window, polyorder):
    """(taps[window], edge[2, window//2, window]) exactly as the kernel uses them (host-only, no GPU needed)."""
    load_library()
    half = window // 2
    c = np.empty(window, dtype=np.float64)
    e = np.empty((2, half, window), dtype=np.float64)
    _check(_lib.lk_savgol_design(int(window), int(polyorder), _ptr(c), _ptr(e)))
    return c, e


def savgol_trend_batch(t, flux, n_off, mask=None, window_length=101, polyorder=2, break_tolerance=5, niters=3,
                       sigma=3, return_fit_mask=False, device=0):
    """The trend LightCurve.flatten divides by, for B ragged targets.  ``mask``: True = exclude from the fit."""
    h = Handle.get(device)
    t, flux = _f64(t), _f64(flux)
    n_off = _offsets(n_off, t.size)
    if flux.shape != t.shape:
        raise ValueError("t and flux must have the same length")
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    if m is not None and m.shape != t.shape:
        raise ValueError("mask must have one entry per cadence (got shape %s, need %s)" % (m.shape, t.shape))
    bt = float("nan") if break_tolerance is None else float(break_tolerance)
    trend = result_empty(t.size)
    fm = np.empty(t.size, dtype=np.uint8) if return_fit_mask else None
    _check(_lib.lk_savgol_trend_batch(h._h, n_off.size - 1, _ptr(n_off, _c_ip), _ptr(t), _ptr(flux), _ptr(m, _c_u8p),
                                      int(window_length), int(polyorder), bt, int(niters), float(sigma), _ptr(trend),
                                      _ptr(fm, _c_u8p)))
    return (trend, fm.astype(bool)) if return_fit_mask else trend


# --------------------------------------------------------------------------------------------- PLD design matrix
def pld_design_width(P, Pb, pld_order, pca_components, n_knots):
    load_library()
    return int(_lib.lk_pld_design_width(int(P), int(Pb), int(pld_order), int(pca_components), int(n_knots)))


def pld_design_batch(pld_pix, bkg_pix, lc_flux, time, knots, pld_order, pca_components, spline_degree,
                     normalize_bkg=True, device=0):
    """PLD design matrices for B same-shaped cutouts.  pld_pix (B, N, P) or None, bkg_pix (B, N, Pb), lc_flux (B, N)
    float32; time (B, N); knots (B, n_inner + 2) = [min t, interior knots, max t].
    Returns (X[B, N, K], prior_sigma[B, K])."""
    h = Handle.get(device)
    bkg_pix = np.ascontiguousarray(bkg_pix, dtype=np.float32)
    B, N, Pb = bkg_pix.shape
    P = 0
    if pld_pix is not None and np.size(pld_pix):
        pld_pix = np.ascontiguousarray(pld_pix, dtype=np.float32)
        if pld_pix.ndim != 3 or pld_pix.shape[:2] != (B, N):
            raise ValueError("pld_pix must be (B, N, P) with the B, N of bkg_pix (got %s)" % (pld_pix.shape,))
        P = pld_pix.shape[2]
    else:
        pld_pix = None
    lc_flux = np.ascontiguousarray(lc_flux, dtype=np.float32)
    time, knots = _f64(time), _f64(knots)
    if lc_flux.shape != (B, N) or time.shape != (B, N) or knots.shape[0] != B:
        raise ValueError("inconsistent PLD batch shapes")
    n_inner = knots.shape[1] - 2
    n_knots = n_inner + int(spline_degree) + 1
    K = pld_design_width(P, Pb, pld_order, pca_components, n_knots)
    X = np.empty((B, N, K), dtype=np.float64)
    ps = np.empty((B, K), dtype=np.float64)
    _check(_lib.lk_pld_design_batch(h._h, B, N, P, Pb, _ptr(pld_pix, _c_fp), _ptr(bkg_pix, _c_fp), _ptr(lc_flux, _c_fp),
                                    _ptr(time), _ptr(knots), n_inner, int(pld_order), int(pca_components), n_knots,
                                    int(spline_degree), int(bool(normalize_bkg)), K, _ptr(X), _ptr(ps)))
    return X, ps


def pld_correct_batch(pld_pix, bkg_pix, lc_flux, time, knots, y, err, pld_order, pca_components, spline_degree,
                      normalize_bkg=True, cadence_mask=None, sigma=5.0, niters=5, want_spline=True, device=0):
    """``PLDCorrector.correct`` numerics for B same-shaped cutouts in ONE call (design matrices, regression + clip loop, the
    spline block's share of the model); the design matrices stay in device memory.  Inputs as ``pld_design_batch`` plus
    y / err (B, N) float64 = the SAP light curve; pass the SAME array as ``pld_pix`` and ``bkg_pix`` when both apertures
    are equal (uploaded once).  Returns dict(coefficients[B, K], model[B, N] (median removed), outlier_mask[B, N] bool,
    spline[B, N] or None)."""
    h = Handle.get(device)
    same = pld_pix is bkg_pix
    bkg_pix = np.ascontiguousarray(bkg_pix, dtype=np.float32)
    B, N, Pb = bkg_pix.shape
    P = 0
    if pld_pix is not None and np.size(pld_pix):
        pld_pix = bkg_pix if same else np.ascontiguousarray(pld_pix, dtype=np.float32)
        if pld_pix.ndim != 3 or pld_pix.shape[:2] != (B, N):
            raise ValueError("pld_pix must be (B, N, P) with the B, N of bkg_pix (got %s)" % (pld_pix.shape,))
        P = pld_pix.shape[2]
    else:
        pld_pix = None
    lc_flux = np.ascontiguousarray(lc_flux, dtype=np.float32)
    time, knots, y = _f64(time), _f64(knots), _f64(y)
    err = None if err is None else _f64(err)
    if lc_flux.shape != (B, N) or time.shape != (B, N) or knots.shape[0] != B or y.shape != (B, N) or (
            err is not None and err.shape != (B, N)):
        raise ValueError("inconsistent PLD batch shapes")
    cm = None if cadence_mask is None else np.ascontiguousarray(cadence_mask, dtype=np.uint8)
    if cm is not None and cm.shape != (B, N):
        raise ValueError("cadence_mask must be (B, N)")
    n_inner = knots.shape[1] - 2
    n_knots = n_inner + int(spline_degree) + 1
    K = pld_design_width(P, Pb, pld_order, pca_components, n_knots)
    w = np.empty((B, K), dtype=np.float64)
    model = np.empty((B, N), dtype=np.float64)
    outl = np.empty((B, N), dtype=np.uint8)
    sp = np.empty((B, N), dtype=np.float64) if want_spline else None
    _check(_lib.lk_pld_correct_batch(h._h, B, N, P, Pb, _ptr(pld_pix, _c_fp), _ptr(bkg_pix, _c_fp), _ptr(lc_flux, _c_fp),
                                     _ptr(time), _ptr(knots), n_inner, int(pld_order), int(pca_components), n_knots,
                                     int(spline_degree), int(bool(normalize_bkg)), K, _ptr(y), _ptr(err), _ptr(cm, _c_u8p),
                                     float(sigma), int(niters), _ptr(w), _ptr(model), _ptr(outl, _c_u8p), _ptr(sp)))
    return dict(coefficients=w, model=model, outlier_mask=outl.astype(bool), spline=sp)


# --------------------------------------------------------------------------------------------- design-matrix operations
def pca_batch(A, nterms, device=0):
    """``DesignMatrix.pca`` (reference correctors/designmatrix.py:252-282) for B same-shaped matrices: A (B, N, P) or (N, P)
    -> the first ``nterms`` left singular vectors of each column-centred matrix, (B, N, nterms) / (N, nterms)."""
    h = Handle.get(device)
    A = np.ascontiguousarray(A, dtype=np.float64)
    single = A.ndim == 2
    if single:
        A = A[None]
    if A.ndim != 3:
        raise ValueError("A must be (B, N, P) or (N, P)")
    B, N, P = A.shape
    k = int(nterms)
    U = np.empty((B, N, k), dtype=np.float64)
    _check(_lib.lk_pca_batch(h._h, B, N, P, k, _ptr(A), _ptr(U)))
    return U[0] if single else U


def spline_basis_batch(x, knots, degree=3, device=0):
    """Clamped B-spline basis (patsy ``bs(x, knots=..., degree, include_intercept=True)``) of x (B, N) or (N,) on
    knots (B, n_inner + 2) / (n_inner + 2,) = [lower bound, interior knots, upper bound] -> (B, N, n_inner + degree + 1)."""
    h = Handle.get(device)
    x, knots = _f64(x), _f64(knots)
    single = x.ndim == 1
    if single:
        x, knots = x[None], knots[None]
    B, N = x.shape
    if knots.ndim != 2 or knots.shape[0] != B or knots.shape[1] < 2:
        raise ValueError("knots must be (B, n_inner + 2)")
    n_inner = knots.shape[1] - 2
    out = np.empty((B, N, n_inner + int(degree) + 1), dtype=np.float64)
    _check(_lib.lk_spline_basis_batch(h._h, B, N, _ptr(x), _ptr(knots), n_inner, int(degree), _ptr(out)))
    return out[0] if single else out


def standardize_batch(A, device=0):
    """``DesignMatrix.standardize`` (reference correctors/designmatrix.py:215-250): A (B, N, P) or (N, P) -> same shape."""
    h = Handle.get(device)
    A = np.ascontiguousarray(A, dtype=np.float64)
    single = A.ndim == 2
    if single:
        A = A[None]
    B, N, P = A.shape
    out = np.empty_like(A)
    _check(_lib.lk_standardize_batch(h._h, B, N, P, _ptr(A), _ptr(out)))
    return out[0] if single else out


# --------------------------------------------------------------------------------------------- batch ingest (N4)
def sigma_clip_batch(y, n_off, sigma=5.0, maxiters=5, device=0):
    """astropy.stats.sigma_clip(y_b, sigma, maxiters).mask for B ragged arrays -> bool[sum N] (True = clipped / non-finite)."""
    h = Handle.get(device)
    y = _f64(y)
    n_off = _offsets(n_off, y.size)
    out = np.zeros(y.size, dtype=np.uint8)
    _check(_lib.lk_sigma_clip_batch(h._h, n_off.size - 1, _ptr(n_off, _c_ip), _ptr(y), float(sigma), int(maxiters),
                                    _ptr(out, _c_u8p)))
    return out.astype(bool)


def ingest_batch(t, flux, n_off, flux_err=None, normalize=True, device=0):
    """remove_nans (+ normalize) for B ragged light curves.  Returns (t, flux, flux_err or None, new_off, median[B])."""
    h = Handle.get(device)
    t, flux = _f64(t), _f64(flux)
    n_off = _offsets(n_off, t.size)
    if flux.shape != t.shape:
        raise ValueError("t and flux must have the same length")
    err = None if flux_err is None else _f64(np.broadcast_to(flux_err, t.shape))
    B = n_off.size - 1
    to, fo = np.empty_like(t), np.empty_like(t)
    eo = np.empty_like(t) if err is not None else None
    new_off = np.zeros(B + 1, dtype=np.int64)
    med = np.empty(B, dtype=np.float64)
    _check(_lib.lk_ingest_batch(h._h, B, _ptr(n_off, _c_ip), _ptr(t), _ptr(flux), _ptr(err), int(bool(normalize)), _ptr(to),
                                _ptr(fo), _ptr(eo), _ptr(new_off, _c_ip), _ptr(med)))
    k = int(new_off[-1])
    return to[:k], fo[:k], (eo[:k] if eo is not None else None), new_off, med


def fits_unpack_batch(raws, descs, bitmasks, device=0):
    """FITS binary tables -> packed (time, flux, flux_err, quality, n_off) for B files (see lightkurve_amd/fitsio.py).
    ``raws``: list of uint8 arrays (rows x record bytes, as in the file); ``descs``: B x 10 int32 (fitsio.lightcurve_columns);
    ``bitmasks``: B ints.  Rows with NaN time or (quality & bitmask) != 0 are dropped on the device."""
    h = Handle.get(device)
    B = len(raws)
    desc = np.ascontiguousarray(descs, dtype=np.int32).reshape(B, 10)
    mask = np.ascontiguousarray(bitmasks, dtype=np.int64).reshape(B)
    raw_off = np.zeros(B + 1, dtype=np.int64)
    for b, r in enumerate(raws):
        nbytes = int(desc[b, 0]) * int(desc[b, 1])
        if np.asarray(r).size != nbytes:
            raise ValueError("file %d: %d bytes of table data, descriptor says %d rows x %d bytes" % (b, np.asarray(r).size, desc[b, 1], desc[b, 0]))
        raw_off[b + 1] = raw_off[b] + ((nbytes + 3 + 15) // 16) * 16   # 3 spare bytes, then 16-byte alignment
    raw = np.zeros(int(raw_off[-1]), dtype=np.uint8)
    for b, r in enumerate(raws):
        raw[raw_off[b]:raw_off[b] + np.asarray(r).size] = np.asarray(r, dtype=np.uint8).reshape(-1)
    rows = int(desc[:, 1].sum())
    t, f, e = (np.empty(rows, dtype=np.float64) for _ in range(3))
    q = np.empty(rows, dtype=np.int32)
    new_off = np.zeros(B + 1, dtype=np.int64)
    _check(_lib.lk_fits_unpack_batch(h._h, B, _ptr(raw, _c_u8p), _ptr(raw_off, _c_ip), _ptr(desc, _c_i32p), _ptr(mask, _c_ip),
                                     _ptr(t), _ptr(f), _ptr(e), _ptr(q, _c_i32p), _ptr(new_off, _c_ip)))
    k = int(new_off[-1])
    return t[:k], f[:k], e[:k], q[:k], new_off


def fits_unpack_cube(raw, off_time, code_time, off_quality, code_quality, bitmask, keep_nan_time, col_offsets, npix, device=0):
    """One target-pixel file's table -> (time[k], quality[k], cubes[ncols, k, npix] float32) for the kept cadences."""
    h = Handle.get(device)
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    n_rows, row_bytes = raw.shape
    cols = np.ascontiguousarray(col_offsets, dtype=np.int32)
    ncols = cols.size
    t = np.empty(n_rows, dtype=np.float64)
    q = np.empty(n_rows, dtype=np.int32)
    cubes = np.empty((ncols, n_rows, npix), dtype=np.float32)
    kept = np.zeros(1, dtype=np.int64)
    _check(_lib.lk_fits_unpack_cube(h._h, _ptr(raw.reshape(-1), _c_u8p), int(row_bytes), int(n_rows), int(off_time), int(code_time),
                                    int(off_quality), int(code_quality), ctypes.c_int64(int(bitmask)), int(bool(keep_nan_time)),
                                    int(ncols), _ptr(cols, _c_i32p), int(npix), _ptr(t), _ptr(q, _c_i32p),
                                    cubes.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _ptr(kept, _c_ip)))
    k = int(kept[0])
    return t[:k], q[:k], cubes[:, :k, :]


def transit_mask_batch(t, n_off, period, duration, transit_time, planet_off=None, device=0):
    """create_transit_mask for B ragged light curves.  ``period`` / ``duration`` / ``transit_time``: flat arrays over all
    planets, ``planet_off[B + 1]`` says which belong to which target (default: every target gets all of them)."""
    h = Handle.get(device)
    t = _f64(t)
    n_off = _offsets(n_off, t.size)
    B = n_off.size - 1
    period, duration, transit_time = (_f64(np.atleast_1d(a)) for a in (period, duration, transit_time))
    if not (period.shape == duration.shape == transit_time.shape):
        raise ValueError("period, duration, and transit_time must have the same number of values.")
    if planet_off is None:
        k = period.size
        period, duration, transit_time = (np.tile(a, B) for a in (period, duration, transit_time))
        planet_off = np.arange(B + 1, dtype=np.int32) * k
    planet_off = np.ascontiguousarray(planet_off, dtype=np.int32)
    if planet_off.shape != (B + 1,) or planet_off[0] != 0 or planet_off[-1] != period.size:
        raise ValueError("planet_off must be B + 1 prefix offsets over the planet arrays")
    mask = np.zeros(t.size, dtype=np.uint8)
    _check(_lib.lk_transit_mask_batch(h._h, B, _ptr(n_off, _c_ip), _ptr(t), _ptr(planet_off, _c_i32p), _ptr(period),
                                      _ptr(duration), _ptr(transit_time), _ptr(mask, _c_u8p)))
    return mask.astype(bool)


def bin_batch(t, flux, n_off, flux_err=None, time_bin_size=0.5, time_bin_start=None, device=0):
    """LightCurve.bin(time_bin_size=...) for B ragged, time-sorted light curves.
    Returns (t_binned, flux_binned, flux_err_binned, bin_off[B + 1])."""
    h = Handle.get(device)
    t, flux = _f64(t), _f64(flux)
    n_off = _offsets(n_off, t.size)
    B = n_off.size - 1
    err = None if flux_err is None else _f64(np.broadcast_to(flux_err, t.shape))
    size_sec = float(time_bin_size) * 86400.0
    if not size_sec > 0:
        raise ValueError("time_bin_size must be positive")
    start = np.empty(B, dtype=np.float64)
    nb = np.zeros(B, dtype=np.int64)
    has_err = np.zeros(B, dtype=np.uint8)
    for b in range(B):
        lo, hi = int(n_off[b]), int(n_off[b + 1])
        if hi == lo:
            start[b] = 0.0
            continue
        tb = t[lo:hi]
        if np.any(np.diff(tb) < 0):
            raise ValueError("bin needs the light curve sorted by time")
        s0 = tb[0] if time_bin_start is None else np.broadcast_to(np.asarray(time_bin_start, float), (B,))[b]
        start[b] = s0
        nb[b] = max(0, int(np.ceil((tb[-1] - s0) * 86400.0 / size_sec)))     # downsample.py:75-76
        has_err[b] = 1 if (err is not None and np.any(np.isfinite(err[lo:hi]))) else 0
    bin_off = np.zeros(B + 1, dtype=np.int64)
    bin_off[1:] = np.cumsum(nb)
    edges = np.cumsum(np.hstack([0.0, np.repeat(size_sec, int(nb.max()) if B else 0)]))   # downsample.py:82
    nbt = int(bin_off[-1])
    to, fo, eo = (np.empty(nbt, dtype=np.float64) for _ in range(3))
    _check(_lib.lk_bin_batch(h._h, B, _ptr(n_off, _c_ip), _ptr(t), _ptr(flux), _ptr(err), _ptr(bin_off, _c_ip), _ptr(start),
                             _ptr(edges), int(edges.size), size_sec, _ptr(has_err, _c_u8p), _ptr(to), _ptr(fo), _ptr(eo)))
    return to, fo, eo, bin_off
