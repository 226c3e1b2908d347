import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np
from lightkurve_amd import _capi, synth
for n in (4000, 20000):
    t, y, e, _ = synth.bls_target(11, 0, n, cadence_days=30.0/1440.0)
    t = t - t.min(); y = y - np.median(y); iv = 1.0/e**2
    period = np.exp(np.linspace(np.log(0.5), np.log(12.0), 5000))
    dur = np.array([0.05, 0.08, 0.1, 0.15, 0.2, 0.25])
    _capi.bls_batch(t, y, iv, [0, n], period, dur)
    t0 = time.perf_counter()
    for _ in range(10): _capi.bls_batch(t, y, iv, [0, n], period, dur)
    print("N", n, "BLS B=1 5000 periods x 6 durations: ms per C call", (time.perf_counter()-t0)/10*1e3)
