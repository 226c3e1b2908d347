"""GPU parity: over-fitting goodness metric (reference correctors/metrics.py:24-138) — the three
default-method Lomb-Scargle periodograms per sample run on the GPU FFT path (the noise ones as one batch).

Tolerance (stated): with the same numpy seed the metric equals the reference's value to 1e-8 absolute (the periodograms
agree to 1e-9 relative; the metric is a ratio of sums of them pushed through a sigmoid).
"""
import numpy as np
import pytest

from lightkurve_amd.correctors.metrics import overfit_metric_lombscargle
from lightkurve_amd.lightcurve import LightCurve

pytestmark = pytest.mark.gpu


def test_golden_metric_values(golden):
    g = golden("overfit_metric")
    orig = LightCurve(time=g["time"], flux=g["flux"], flux_err=g["flux_err"])
    for name in ("over", "mild", "clean", "nan"):
        np.random.seed(1234)
        m = overfit_metric_lombscargle(orig, LightCurve(time=g["time"], flux=g["flux_" + name], flux_err=g["flux_err"]),
                                       n_samples=int(g["nsamples_" + name]))
        assert abs(m - float(g["metric_" + name])) < 1e-8, name


def test_reference_sanity_cases():
    """tests/correctors/test_metrics.py:14-35 of the reference, verbatim expectations."""
    time = np.arange(1, 100, 0.1)
    lc_flat = LightCurve(time=time, flux=np.ones_like(time), flux_err=0.0)
    lc_sine = LightCurve(time=time, flux=np.sin(time) + 1, flux_err=0.0)
    assert overfit_metric_lombscargle(lc_flat, lc_flat) == 1.0
    assert overfit_metric_lombscargle(lc_sine, lc_sine) == 1.0
    assert overfit_metric_lombscargle(lc_sine, lc_flat) == 1.0
    assert overfit_metric_lombscargle(lc_flat, lc_sine) == 0.0
    lc_flat.flux_err += 0.5
    lc_sine.flux_err += 0.5
    assert overfit_metric_lombscargle(lc_flat, lc_sine) > 0.5
