#!/usr/bin/env python
"""Phase profile of flatten_kernel on the GPU box: cumulative kernel time up to each phase (LK_FLAT_STOP), B x N light
curves resident in HBM.  Prints one line per stop point and the per-phase differences."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (before liblkhip.so)
import ctypes  # noqa: E402
from lightkurve_amd import _capi, synth  # noqa: E402

B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 20000
t, y, dy, off = synth.ls_batch(6, B, N)
dev = torch.device("cuda", 0)
d_t, d_y = torch.from_numpy(t).to(dev), torch.from_numpy(y).to(dev)
d_tr = torch.empty_like(d_y)
h = _capi.Handle.get(0)
lib = _capi.load_library()
vp = ctypes.c_void_p
stream = torch.cuda.current_stream().cuda_stream


def run():
    _capi._check(lib.lk_savgol_trend_batch_dev(h._h, B, off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), vp(d_t.data_ptr()),
                                               vp(d_y.data_ptr()), None, 401, 2, 5.0, 3, 3.0, vp(d_tr.data_ptr()), None,
                                               vp(stream)))


for base, dbg, what in [(b, d, w) for b in (100, 200) for d, w in ((0, "sample gather"), (1, "sample sort"), (2, "collect pass"), (3, "counts"), (4, "candidate sort"), (5, "rank k"), (6, "rank k+1"), (7, "sync"), (9, "whole median"))]:
    os.environ["LK_FLAT_STOP"] = str(base + dbg)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("%s up to %-16s %7.3f ms" % ("median0" if base == 100 else "dtmedian", what, e0.elapsed_time(e1) / 5))

names = ["stats", "median0", "mask0", "compact", "gather", "dtcount", "dtmedian", "segs", "fir", "clip", "interp"]
stops = [(0, k) for k in range(10)] + [(1, 3), (1, 9), (2, 3), (2, 8), (2, 9), (2, 10), (5, 0)]
prev = 0.0
for it, k in stops:
    os.environ["LK_FLAT_STOP"] = str(16 * it + k)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    label = "full" if it == 5 else "it%d:%s" % (it, names[k])
    print("stop %-14s cumulative %7.3f ms   delta %7.3f ms" % (label, ms, ms - prev))
    prev = ms
