#!/bin/bash
# Round-6 evidence in one GPU call: the GPU test suite, the default bench line, kernel traces and PMC passes of the LS, PLD and
# flatten benches, the PLD development-build clocks, the seams end-to-end run.  Everything lands under gpurun_out/r06/ (copy what is
# to be judged into profiles/).
out=gpurun_out/r06; mkdir -p $out
python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err
tools/trace_lsfast.sh $out/ls
tools/pmc_pass.sh $out ls_fetch FETCH_SIZE --no-bls --no-pld --no-flatten --no-host --no-cpu-baseline --ls-method fast --no-api --no-pipeline --c2-targets 0 --targets 180 --steps 3 --warmup 1 --repeats 1
tools/pmc_pass.sh $out ls_write WRITE_SIZE --no-bls --no-pld --no-flatten --no-host --no-cpu-baseline --ls-method fast --no-api --no-pipeline --c2-targets 0 --targets 180 --steps 3 --warmup 1 --repeats 1
tools/trace_pld.sh $out/pld
tools/pmc_pass.sh $out pld_fetch FETCH_SIZE --workload pld --no-cpu-baseline --no-api --steps 3 --warmup 1
tools/pmc_pass.sh $out pld_write WRITE_SIZE --workload pld --no-cpu-baseline --no-api --steps 3 --warmup 1
tools/trace_flatten.sh $out/flat
tools/pmc_pass.sh $out flat_fetch FETCH_SIZE --workload flatten --no-cpu-baseline --no-api --steps 3 --warmup 1
tools/pmc_pass.sh $out flat_write WRITE_SIZE --workload flatten --no-cpu-baseline --no-api --steps 3 --warmup 1
if [ -f build/ab/pld_dbg.so ]; then
  LK_PLD_ITERS=1 LK_LIB_PATH=$PWD/build/ab/pld_dbg.so python bench.py --workload pld --no-cpu-baseline --steps 1 --warmup 1 2>&1 >/dev/null | grep "pld tridiag\|pld eig" | head -8 > $out/pld_phase_clocks.txt
fi
find $out -name "*results.db" -size +30M -delete
if [ -f .stage/lkref.tar.gz ]; then bash tools/seams_e2e_gpu.sh > $out/seams.log 2>&1; fi
ls $out
