#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2n; mkdir -p $O
timeout 900 python -m pytest tests/test_bls_gpu.py tests/test_api_gpu.py -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
timeout 600 python bench.py --workload bls --no-cpu-baseline --targets 250 --steps 1 --warmup 1 > $O/bls.json 2> $O/bls.err; python -c "import json;d=json.load(open('$O/bls.json'));print('bls TS (250 targets) ms/step',d['ms_per_step'])"
LK_BLS_GENERIC=1 timeout 600 python bench.py --workload bls --no-cpu-baseline --targets 250 --steps 1 --warmup 1 > $O/bls_g.json 2> $O/bls_g.err; python -c "import json;d=json.load(open('$O/bls_g.json'));print('bls generic (250 targets) ms/step',d['ms_per_step'])"
LK_BLS_PROF=1 timeout 600 python bench.py --workload bls --no-cpu-baseline --targets 128 --steps 1 --warmup 0 > $O/bls_prof.json 2> $O/bls_prof.err; grep "bls prof" $O/bls_prof.err | cut -c1-260 > $O/bls_prof.txt; tail -4 $O/bls_prof.txt
echo done
