#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2h; mkdir -p $O
timeout 1200 python -m pytest tests/test_ingest_gpu.py tests/test_seismology_gpu.py tests/test_pld_gpu.py tests/test_api_gpu.py tests/test_metrics_gpu.py tests/test_seams_gpu.py -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $O/pytest.log | tail -15 | cut -c1-300
echo done
