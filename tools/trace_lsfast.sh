#!/bin/bash
# kernel trace of the LS-fast bench (steady-state table + timeline of the last dispatches): tools/trace_lsfast.sh <outdir> [env...]
set -u
out=$1; shift
mkdir -p "$out"
R=$PWD
export TMPDIR=/tmp
cd /tmp
env "$@" rocprofv3 --kernel-trace --stats -d "$R/$out/trace" -o ls -- python "$R/bench.py" --no-bls --no-pld --no-flatten --no-host --no-cpu-baseline --ls-method fast --steps 10 --warmup 3 > "$R/$out/bench.json" 2> "$R/$out/bench.err"
cd "$R"
db=$(ls $out/trace/*/*results.db $out/trace/*results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$db" "bench.py --ls-method fast under rocprofv3 ($*)" --skip-frac 0.3 --timeline 80 > "$out/summary.txt"
