#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2p; mkdir -p $O
timeout 1200 python -m pytest tests/test_regress_gpu.py tests/test_pld_gpu.py tests/test_metrics_gpu.py tests/test_api_gpu.py tests/test_seams_gpu.py -q --timeout=600 2>&1 | tail -2
for E in 1 0; do
LK_REGRESS_EARLY=$E timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 3 --warmup 1 > $O/pld_$E.json 2> $O/pld_$E.err; python -c "import json;d=json.load(open('$O/pld_$E.json'));print('pld early=$E ms/step',d['ms_per_step'])"
LK_REGRESS_EARLY=$E timeout 300 python bench.py --workload regress --no-cpu-baseline --steps 3 --warmup 1 > $O/reg_$E.json 2> $O/reg_$E.err; python -c "import json;d=json.load(open('$O/reg_$E.json'));print('regress early=$E ms/step',d['ms_per_step'])"
done
echo done
