#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2i; mkdir -p $O
timeout 900 python -m pytest tests/test_pld_gpu.py tests/test_distributed_gpu.py -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
for C in 0 1; do for PW in 3 4; do
LK_PLD_CHEB=$C LK_PLD_POWER=$PW LK_PLD_ITERS=1 timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 1 --warmup 0 --cutouts 200 > /dev/null 2> $O/pld_iters_$C_$PW.err; grep "pld eig" $O/pld_iters_$C_$PW.err | sort | uniq -c | head -6
LK_PLD_CHEB=$C LK_PLD_POWER=$PW timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 3 --warmup 1 > $O/pld_$C_$PW.json 2> $O/pld_$C_$PW.err; python -c "import json;d=json.load(open('$O/pld_$C_$PW.json'));print('pld cheb=$C pow=$PW ms/step',d['ms_per_step'])"
done; done
echo done
