#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2u; mkdir -p $O
timeout 1200 python -m pytest tests/test_pld_gpu.py tests/test_seams_gpu.py tests/test_metrics_gpu.py -q --timeout=600 2>&1 | tail -2
timeout 600 python bench.py --workload pld --steps 5 --warmup 2 > $O/pld.json 2> $O/pld.err; python -c "import json;d=json.load(open('$O/pld.json'));print('pld ms/step',d['ms_per_step'],d['value'],d.get('accuracy'))"
LK_PLD_POWER=4 timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 5 --warmup 2 > $O/pld4.json 2> $O/pld4.err; python -c "import json;d=json.load(open('$O/pld4.json'));print('pld pow=4 ms/step',d['ms_per_step'])"
LK_PLD_POWER=2 timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 5 --warmup 2 > $O/pld2.json 2> $O/pld2.err; python -c "import json;d=json.load(open('$O/pld2.json'));print('pld pow=2 ms/step',d['ms_per_step'])"
echo done
