#!/bin/bash
# One rocprofv3 PMC pass of a bench.py command (counters in their own run, kernel-trace/stats only — never with the
# hip/hsa/memory trace domains): tools/pmc_pass.sh <outdir> <tag> "<COUNTER [COUNTER...]>" <bench.py args...>
out=$1; tag=$2; ctr=$3; shift 3
mkdir -p "$out"; R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --pmc $ctr -d "$R/$out/pmc_$tag" -o p -- python "$R/bench.py" "$@" > "$R/$out/pmc_$tag.json" 2> "$R/$out/pmc_$tag.err"
cd "$R"
db=$(ls $out/pmc_$tag/*/*results.db $out/pmc_$tag/*results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$db" "bench.py $* with --pmc $ctr" > "$out/pmc_$tag.txt"
rm -rf "$out/pmc_$tag"
