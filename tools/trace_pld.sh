#!/bin/bash
# kernel trace of the PLD bench: tools/trace_pld.sh <outdir> [VAR=value ...]  (e.g. LK_LIB_PATH=$PWD/build/ab/x.so)
out=$1; shift; mkdir -p "$out"; R=$PWD; export TMPDIR=/tmp; cd /tmp
env "$@" rocprofv3 --kernel-trace --stats -d "$R/$out/trace" -o pld -- python "$R/bench.py" --workload pld --no-cpu-baseline --no-api --steps 5 --warmup 2 > "$R/$out/bench.json" 2> "$R/$out/bench.err"
cd "$R"
db=$(ls $out/trace/*/*results.db $out/trace/*results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$db" "bench.py --workload pld under rocprofv3" --skip-frac 0.3 > "$out/summary.txt"
