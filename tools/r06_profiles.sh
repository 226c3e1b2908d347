#!/bin/bash
# Round-6 evidence in one GPU call: the GPU test suite, the default bench line, kernel traces and PMC passes of the PLD and flatten
# benches.  Everything lands under gpurun_out/r06/ (copy what is to be judged into profiles/).
out=gpurun_out/r06; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.json | head -c 300; echo
tools/trace_pld.sh $out/pld
tools/pmc_pass.sh $out pld_fetch FETCH_SIZE --workload pld --no-cpu-baseline --steps 3 --warmup 1
tools/pmc_pass.sh $out pld_write WRITE_SIZE --workload pld --no-cpu-baseline --steps 3 --warmup 1
tools/trace_flatten.sh $out/flat
tools/pmc_pass.sh $out flat_fetch FETCH_SIZE --workload flatten --no-cpu-baseline --steps 3 --warmup 1
tools/pmc_pass.sh $out flat_write WRITE_SIZE --workload flatten --no-cpu-baseline --steps 3 --warmup 1
find $out -name "*results.db" -size +20M -delete
ls -la $out
