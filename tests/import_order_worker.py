"""Runs in a fresh interpreter on the GPU box: lightkurve_amd computes FIRST, torch is imported AFTERWARDS and must still
see the GPU (and share device memory with the library)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lightkurve_amd import _capi, synth  # noqa: E402

assert "torch" not in sys.modules
t, y, e, _ = synth.ls_target(1, 0, 800)
p = _capi.ls_fast_batch(t - t[0], y, [0, len(t)], f0=0.01, df=0.01, M=2000)[0]
import torch  # noqa: E402

assert torch.cuda.is_available(), "torch lost its GPU to the library load order"
d = torch.from_numpy(t - t[0]).cuda()
dy = torch.from_numpy(y).cuda()
out = torch.empty((1, 2000), dtype=torch.float64, device="cuda")
h = _capi.Handle.get(0)
_capi.ls_fast_batch_dev(h, 1, np.array([0, len(t)]), d.data_ptr(), dy.data_ptr(), 0, 0.01, 0.01, 2000, True, True, "psd", 0, 5,
                        out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert np.allclose(out.cpu().numpy()[0], p, rtol=1e-12, atol=0, equal_nan=True)
print("IMPORT_ORDER_OK")
