#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2k; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
run_pmc() { # name, counters...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/$O/pmc_$n -o p -- python $R/bench.py --workload pld --no-cpu-baseline --steps 1 --warmup 0 > $R/$O/pmc_$n.json 2> $R/$O/pmc_$n.err
  python $R/tools/rocprof_summary.py $R/$O/pmc_$n/p_results.db "pld pmc $n" 2>&1 | grep -E "gram128|topk_eig|project" | grep -v "^#" > $R/$O/pmc_$n.txt; cat $R/$O/pmc_$n.txt
  rm -rf $R/$O/pmc_$n
}
run_pmc tcc TCC_HIT_sum TCC_MISS_sum
run_pmc sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run_pmc fetch FETCH_SIZE
echo done
