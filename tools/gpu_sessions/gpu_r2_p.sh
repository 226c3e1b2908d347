#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2p; mkdir -p $O
timeout 1200 python -m pytest tests/test_regress_gpu.py tests/test_pld_gpu.py -q --timeout=600 2>&1 | tail -2
timeout 300 python bench.py --workload pld --no-cpu-baseline --steps 5 --warmup 2 > $O/pld.json 2> $O/pld.err; python -c "import json;d=json.load(open('$O/pld.json'));print('pld ms/step',d['ms_per_step'])"
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/tr -o p -- python $R/bench.py --workload pld --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> $R/$O/tr.err
cd $R; python tools/rocprof_summary.py $O/tr/p_results.db "bench.py --workload pld --steps 3 --warmup 1 (round 2 final)" > $O/pld_trace_summary.txt; head -16 $O/pld_trace_summary.txt; rm -rf $O/tr
echo done
