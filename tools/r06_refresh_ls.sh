#!/bin/bash
# Round-6 refresh after an LS-fast change: GPU test suite, default bench line, LS kernel trace, smoke.  Output under gpurun_out/r06b/.
out=gpurun_out/r06b; mkdir -p $out
python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err
tools/trace_lsfast.sh $out/ls
find $out -name "*results.db" -size +30M -delete
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
ls $out
