#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2t; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $R/$O/pmc_$C -o p -- python $R/bench.py --workload pld --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $R/$O/pmc_$C.err
  python $R/tools/rocprof_summary.py $R/$O/pmc_$C/p_results.db "bench.py --workload pld --no-cpu-baseline --steps 2 --warmup 1, --pmc $C (round 2 final kernels)" > $R/$O/pld_pmc_$C.txt; rm -rf $R/$O/pmc_$C
  grep "$C" $R/$O/pld_pmc_$C.txt | awk '{s+=$(NF-2)} END {print "'$C' total KiB over 3 steps:", s}'
done
echo done
