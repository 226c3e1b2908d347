#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2f; mkdir -p $O
timeout 600 python -m pytest tests/test_flatten_gpu.py tests/test_api_gpu.py -q -x --timeout=300 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
timeout 600 python tools/flat_phase_profile.py 1024 2>&1 | tee $O/flat_phases.txt | tail -18
for NT in 512 1024; do
LK_FLAT_NT=$NT timeout 300 python bench.py --workload flatten --targets 1024 --no-cpu-baseline --steps 10 --warmup 2 > $O/flat_$NT.json 2> $O/flat_$NT.err
python -c "import json;d=json.load(open('$O/flat_$NT.json'));print('flatten NT=$NT ms/step',d['ms_per_step'])"
done
echo done
