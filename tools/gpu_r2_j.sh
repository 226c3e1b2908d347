#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2j; mkdir -p $O
timeout 900 python -m pytest tests/test_pld_gpu.py -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-400
LK_PLD_ITERS=1 timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $O/pld_prof.err; grep "pld eig" $O/pld_prof.err | cut -c1-400
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/pld_trace -o pld -- python $R/bench.py --workload pld --no-cpu-baseline --steps 3 --warmup 1 > $R/$O/pld_trace.json 2> $R/$O/pld_trace.err
cd $R
timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 3 --warmup 1 > $O/pld.json 2> $O/pld.err; python -c "import json;d=json.load(open('$O/pld.json'));print('pld ms/step',d['ms_per_step'])"
echo done
