#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2r; mkdir -p $O
timeout 1200 python -m pytest tests/test_pld_gpu.py tests/test_seams_gpu.py -q --timeout=600 2>&1 | tail -2
LK_PLD_EIG_NT=1024 timeout 600 python -m pytest tests/test_pld_gpu.py -q --timeout=600 2>&1 | tail -2
timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 5 --warmup 2 > $O/pld.json 2> $O/pld.err; python -c "import json;d=json.load(open('$O/pld.json'));print('pld ms/step',d['ms_per_step'])"
LK_PLD_ITERS=1 timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $O/pld_prof.err; grep "pld eig" $O/pld_prof.err | cut -c1-330 | sort | uniq
echo done
