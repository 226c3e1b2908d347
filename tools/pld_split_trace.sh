R=$PWD; export TMPDIR=/tmp
LK_PLD_SPLIT=0 LK_PLD_ITERS=1 LK_LIB_PATH=$R/build/ab/pld_dbg.so python bench.py --workload pld --no-cpu-baseline --steps 1 --warmup 1 2>&1 >/dev/null | grep "pld eig\]" | head -2
for sp in 0 1 2; do
cd /tmp; LK_PLD_SPLIT=$sp LK_LIB_PATH=$R/build/ab/pld_dbg.so rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6pld_s$sp/trace -o pld -- python $R/bench.py --workload pld --no-cpu-baseline --steps 4 --warmup 1 > /dev/null 2>&1; cd $R
db=$(ls gpurun_out/r6pld_s$sp/trace/*/*results.db gpurun_out/r6pld_s$sp/trace/*results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$db" "split $sp" --skip-frac 0.3 > gpurun_out/r6pld_s$sp/summary.txt
grep -A12 "steady state" gpurun_out/r6pld_s$sp/summary.txt | head -16
done
