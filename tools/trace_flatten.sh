#!/bin/bash
# kernel trace of the flatten bench (per-phase times of the pipeline + the timeline of the last step): tools/trace_flatten.sh <outdir> [bench args]
out=$1; shift; mkdir -p "$out"; R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d "$R/$out/trace" -o fl -- python "$R/bench.py" --workload flatten --no-cpu-baseline --no-api --steps 10 --warmup 2 "$@" > "$R/$out/bench.json" 2> "$R/$out/bench.err"
cd "$R"
db=$(ls $out/trace/*/*results.db $out/trace/*results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$db" "bench.py --workload flatten --steps 10 --warmup 2 under rocprofv3" --skip-frac 0.3 --timeline 16 > "$out/summary.txt"
rm -rf "$out/trace"
