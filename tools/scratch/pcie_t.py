# PCIe-inclusive rate of the host-pointer C ABI (lk_ls_fast_batch) on configs[1]
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from lightkurve_amd import _capi, synth
B, N, M = 1000, 20000, 100000
t, y, dy, off = synth.ls_batch(1, B, N)
for b in range(B):
    t[off[b]:off[b + 1]] -= t[off[b]]
df = 360.0 / M
for rep in range(3):
    t0 = time.perf_counter()
    p = _capi.ls_fast_batch(t, y, off, f0=df, df=df, M=M, normalization="lk_amplitude")
    dt = time.perf_counter() - t0
    print("ls_fast host API: %.1f ms -> %.3g freq*targets/s (PCIe + pageable numpy buffers included)" % (dt * 1e3, B * M / dt))
