"""CPU: the C-ABI library loads and exports every symbol include/lkhip.h declares (no compute calls)."""
import ctypes
import os
import re

from lightkurve_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lkhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lk_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_and_exports_header():
    assert os.path.exists(_capi.LIB_PATH), "liblkhip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(_capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), "header declares %s but the library does not export it" % n
    assert sorted(s[0] for s in _capi.SIGNATURES) == names   # the ctypes table covers the whole header
    assert _capi.load_library().lk_version() >= 100


def test_no_gpu_fails_loudly_not_silently():
    """Without a GPU every compute entry point must raise (no CPU fallback exists)."""
    import numpy as np
    import pytest
    if _capi.device_count() > 0:
        pytest.skip("GPU present")
    t = np.linspace(0, 1, 10)
    with pytest.raises((RuntimeError, ValueError)):
        _capi.ls_power_batch(t, t, [0, 10], f0=0.1, df=0.1, M=4)


def test_kernel_resource_table_names_every_kernel():
    """profiles/kernel_resources.md (tools/kernel_resources.py: registers / spills / occupancy straight from the compiler, what
    DESIGN.md quotes) must list every __global__ kernel of the sources — a kernel added or renamed without regenerating the
    table fails here (the numbers themselves are checked by `tools/kernel_resources.py --check`, which compiles)."""
    import re
    csrc = os.path.join(ROOT, "lightkurve_amd", "csrc")
    table = open(os.path.join(ROOT, "profiles", "kernel_resources.md")).read()
    missing = []
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(".hip"):
            continue
        src = open(os.path.join(csrc, f)).read()
        for m in re.finditer(r"__global__[^;{]*?\bvoid\s+(\w+)\s*\(", src):
            if ("`%s" % m.group(1)) not in table:
                missing.append((f, m.group(1)))
    assert not missing, missing
