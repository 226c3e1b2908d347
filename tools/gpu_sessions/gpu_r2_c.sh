#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2c; mkdir -p $O
for v in BOTH PLAIN NOLAP ""; do
  lib=lightkurve_amd/liblkhip${v:+_$v}.so
  echo "== variant ${v:-CURRENT} ($lib)"
  LK_LIB_PATH=$PWD/$lib timeout 600 python -m pytest tests/test_flatten_gpu.py -q -x --timeout=300 > $O/pytest_${v:-CUR}.log 2>&1; echo "rc=$?"; tail -2 $O/pytest_${v:-CUR}.log | cut -c1-200
done
echo "== regress + api tests (current lib)"; timeout 900 python -m pytest tests/test_regress_gpu.py tests/test_api_gpu.py tests/test_pld_gpu.py -q --timeout=600 > $O/pytest_rest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_rest.log | cut -c1-300
echo done
