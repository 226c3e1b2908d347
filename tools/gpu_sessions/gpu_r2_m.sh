#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2m; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log | cut -c1-600
echo done
