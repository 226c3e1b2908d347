#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2u; mkdir -p $O
for PP in 4 6 8 12; do
LK_PLD_POWER_PROD=$PP LK_PLD_ITERS=1 timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $O/prof_$PP.err; grep "pld eig" $O/prof_$PP.err | grep "P=136" | cut -c1-330
LK_PLD_POWER_PROD=$PP timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 5 --warmup 2 > $O/pld_$PP.json 2> $O/pld_$PP.err; python -c "import json;d=json.load(open('$O/pld_$PP.json'));print('pld pow_prod=$PP ms/step',d['ms_per_step'])"
done
echo done
