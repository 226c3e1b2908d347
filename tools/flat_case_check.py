"""Debug aid: per-case largest relative difference between the HIP flatten trend and the numpy oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lightkurve_amd import LightCurve, synth
from lightkurve_amd.flatten import flatten_trend_batch
from oracle import np_oracle as O

rng = np.random.default_rng(21)
lcs, masks = [], []
for i, n in enumerate([3000, 150, 999, 20000, 60]):
    t, y, e, _ = synth.ls_target(6, i, n)
    y = y * (1 + 0.02 * np.sin(2 * np.pi * t / 2.1) + 1e-3 * t)
    y[rng.integers(0, n, max(1, n // 300))] += 0.03
    if n > 200:
        y[rng.integers(0, n, 3)] = np.nan
    lcs.append(LightCurve(time=t, flux=y))
    mk = np.zeros(n, bool)
    mk[n // 2:n // 2 + n // 40] = i % 2 == 0
    masks.append(mk)
for w, p, bt, ni, sg in [(101, 2, 5, 3, 3), (31, 3, 2, 4, 2.5), (401, 2, 5, 3, 3), (257, 3, 3, 5, 2.5), (101, 4, 5, 3, 3),
                         (61, 5, 5, 3, 3), (75, 6, 5, 2, 3)]:
    trends = flatten_trend_batch(lcs, window_length=w, polyorder=p, break_tolerance=bt, niters=ni, sigma=sg, masks=masks)
    for lc, mk, tr in zip(lcs, masks, trends):
        ref, _ = O.flatten_trend(lc.time, lc.flux, w, p, bt, ni, sg, mask=mk)
        ok = np.isfinite(tr) & np.isfinite(ref)
        d = np.abs(tr[ok] - ref[ok]) / np.abs(ref[ok])
        j = int(np.argmax(d)) if d.size else -1
        print("w %3d p %d n %5d: max rel diff %.3e at %d of %d (nan pattern equal %s)" % (
            w, p, len(lc.time), d.max() if d.size else 0.0, np.flatnonzero(ok)[j] if d.size else -1, len(tr),
            bool((np.isfinite(tr) == np.isfinite(ref)).all())))
