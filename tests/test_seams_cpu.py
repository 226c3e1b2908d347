"""CPU (this container only: needs the conda interpreter with astropy AND the reference checkout): the four seams
installed into the REAL lightkurve.  `lightkurve_amd.seams.install(backend=...)` is given a stand-in backend answered by
the oracle, so what is verified here is the WIRING — that an unmodified ``lc.to_periodogram()`` / ``method="bls"`` /
``lc.flatten()`` / ``RegressionCorrector.correct`` / ``PLDCorrector.correct`` really go through the seam entry points, get
the right arrays, and hand back lightkurve's own LightCurve / Periodogram objects with the same units, metadata and values —
and that the reference's own tests pass with the seams active.  The numerics of the real (HIP) backend behind the same
functions are checked on the GPU box (tests/test_seams_gpu.py, where /root/reference does not exist)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA = "/opt/conda/bin/python3.9"
REF = "/root/reference"


def _env():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "oracle", "shims"), os.path.join(REF, "src"), ROOT]))
    std = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if os.path.exists(std):
        env["LD_PRELOAD"] = std
    return env


def _need():
    if not os.path.exists(CONDA) or not os.path.isdir(os.path.join(REF, "src", "lightkurve")):
        pytest.skip("needs the conda interpreter and the reference checkout (this container only)")
    p = subprocess.run([CONDA, "-W", "ignore", "-c", "import lightkurve, astropy"], env=_env(), capture_output=True)
    if p.returncode != 0:
        pytest.skip("reference lightkurve not importable: " + p.stderr.decode()[-300:])


def test_all_seams_return_lightkurve_objects_with_reference_values():
    _need()
    p = subprocess.run([CONDA, "-W", "ignore", os.path.join(ROOT, "tests", "seams_lk_worker.py"), "compare", "oracle"],
                       env=_env(), capture_output=True, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("SEAMS_LK_RESULT ")][-1]
    res = json.loads(line[len("SEAMS_LK_RESULT "):])
    assert len(res["installed"]) == 19
    # every seam was really taken
    assert set(res["calls"]) >= {"ls_fast_batch", "ls_power_batch", "bls_batch", "savgol_trend_batch", "regress_batch",
                                 "pld_design_batch", "pca_batch", "standardize_batch", "spline_basis_batch"}
    assert res["types"]["ls_default"] == "LombScarglePeriodogram" and res["types"]["bls"] == "BoxLeastSquaresPeriodogram"
    assert res["errors"]["bls"] == 0.0


def test_reference_test_suites_pass_with_seams_active():
    _need()
    files = [os.path.join(REF, "tests", "test_periodogram.py"),
             os.path.join(REF, "tests", "correctors", "test_regressioncorrector.py"),
             os.path.join(REF, "tests", "correctors", "test_designmatrix.py"),
             os.path.join(REF, "tests", "correctors", "test_metrics.py") + "::test_overfit_metric_lombscargle"]
    files += [os.path.join(REF, "tests", "test_lightcurve.py") + "::" + t for t in
              ("test_flatten_with_nans", "test_flatten_robustness", "test_flatten_returns_normalized",
               "test_iterative_flatten", "test_cdpp")]
    p = subprocess.run([CONDA, "-W", "ignore", os.path.join(ROOT, "tests", "seams_lk_worker.py"), "reftests", "oracle"] + files,
                       env=_env(), capture_output=True, timeout=2400, cwd=os.path.join(REF, "tests"))
    tail = p.stdout.decode()[-2500:] + p.stderr.decode()[-1500:]
    assert p.returncode == 0, tail
    assert "SEAMS_LK_REFTESTS rc=0" in p.stdout.decode(), tail
