#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2q; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
run_pmc() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/$O/pmc_$n -o p -- python $R/bench.py --ls-method fast --no-bls --no-host --no-cpu-baseline --targets 170 --steps 2 --warmup 1 > /dev/null 2> $R/$O/pmc_$n.err
  python $R/tools/rocprof_summary.py $R/$O/pmc_$n/p_results.db "ls fast pmc $n (170 targets)" 2>&1 | grep -E "fft_rows_power" > $R/$O/pmc_$n.txt; cat $R/$O/pmc_$n.txt; rm -rf $R/$O/pmc_$n
}
export LK_FFT3=1 LK_FFT3_RT=4
run_pmc fft3 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS
run_pmc fft3b SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE
echo done
