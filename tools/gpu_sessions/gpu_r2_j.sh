#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2j; mkdir -p $O
timeout 1200 python -m pytest tests/test_pld_gpu.py tests/test_regress_gpu.py tests/test_api_gpu.py tests/test_metrics_gpu.py tests/test_seams_gpu.py -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-400
timeout 600 python bench.py --workload pld --steps 5 --warmup 2 > $O/bench_pld.json 2> $O/bench_pld.err; python -c "import json;d=json.load(open('$O/bench_pld.json'));print('pld ms/step',d['ms_per_step'], d['value'], d['unit'], d.get('roofline'), d.get('cpu_baseline'))"
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/pld_trace -o pld -- python $R/bench.py --workload pld --no-cpu-baseline --steps 3 --warmup 1 > $R/$O/pld_trace.json 2> $R/$O/pld_trace.err
cd $R; python tools/rocprof_summary.py $O/pld_trace/pld_results.db "bench.py --workload pld --steps 3 --warmup 1 (round 2: MFMA eigen-solver pieces, gram128, LDS-free projection)" > $O/pld_trace_summary.txt; head -22 $O/pld_trace_summary.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/$O/pld_pmc -o p -- python $R/bench.py --workload pld --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $R/$O/pld_pmc.err
python tools/rocprof_summary.py $O/pld_pmc/p_results.db "bench.py --workload pld --steps 1 (round 2 final kernels), SQ counters" > $O/pld_pmc_sq.txt; rm -rf $O/pld_pmc
echo done
