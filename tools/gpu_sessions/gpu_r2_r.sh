#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2r; mkdir -p $O
for NT in 512 256; do
LK_PLD_EIG_NT=$NT timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 5 --warmup 2 > $O/pld_$NT.json 2> $O/pld_$NT.err; python -c "import json;d=json.load(open('$O/pld_$NT.json'));print('pld eig nt=$NT ms/step',d['ms_per_step'])"
done
LK_PLD_EIG_NT=256 timeout 600 python -m pytest tests/test_pld_gpu.py -q --timeout=600 2>&1 | tail -2
echo done
