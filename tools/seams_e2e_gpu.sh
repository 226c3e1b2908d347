#!/bin/bash
# End-to-end proof on the GPU box: an UNMODIFIED lightkurve (staged by tools/stage_reference.sh, unpacked to /tmp — outside
# the repo) with lightkurve_amd.seams installed returns its own LightCurve / Periodogram objects from liblkhip.so:
#   1. tests/seams_lk_worker.py compare hip   — all nineteen seams through lightkurve's public API against the reference path
#   2. tests/seams_lk_worker.py reftests hip  — the reference's own periodogram / corrector / flatten tests, seams active
# The logs are what profiles/r06_seams_e2e_gpu.log and r06_seams_latency.txt hold.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/seams_e2e; mkdir -p $O
if [ ! -f .stage/lkref.tar.gz ]; then echo "no .stage/lkref.tar.gz: run tools/stage_reference.sh first"; exit 2; fi
rm -rf /tmp/lkref && mkdir -p /tmp/lkref && tar -C /tmp/lkref -xzf .stage/lkref.tar.gz
export LK_REFERENCE_ROOT=/tmp/lkref
R="$PWD"
export PYTHONPATH="$R/oracle/shims:/tmp/lkref/src:$R"
export LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6
CONDA=/opt/conda/bin/python3.9
{
echo "# $(date -u) seams end-to-end on the GPU box ($(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 -o 'gfx9[0-9a-z]*'))"
echo "## compare hip"
LK_SEAMS_LOG=DEBUG $CONDA -W ignore tests/seams_lk_worker.py compare hip 2>&1 | grep -v "No period specified\|No duration specified\|No transit time specified" | tail -40
echo "## reference test files with the seams active (hip backend)"
( cd /tmp/lkref/tests && $CONDA -W ignore "$R/tests/seams_lk_worker.py" reftests hip \
    test_periodogram.py correctors/test_regressioncorrector.py correctors/test_designmatrix.py \
    "correctors/test_metrics.py::test_overfit_metric_lombscargle" \
    "test_lightcurve.py::test_flatten_with_nans" "test_lightcurve.py::test_flatten_robustness" \
    "test_lightcurve.py::test_flatten_returns_normalized" "test_lightcurve.py::test_iterative_flatten" \
    "test_lightcurve.py::test_cdpp" 2>&1 | tail -15 )
} | tee $O/seams_e2e_gpu.log
# 3. B = 1 latency table (tools/seams_latency.py) and the PLD block with lightkurve itself as the CPU baseline
{
echo "# $(date -u) B = 1 latency through the seams on the GPU box"
$CONDA -W ignore tools/seams_latency.py 2>&1 | grep -v "Warning\|warn" | tail -40
} | tee $O/seams_latency.txt
unset LD_PRELOAD PYTHONPATH
python bench.py --workload pld --steps 5 --warmup 2 > $O/bench_pld_reference_baseline.json 2> $O/bench_pld.err
grep -o '"cpu_baseline": {[^}]*' $O/bench_pld_reference_baseline.json | head -c 900
