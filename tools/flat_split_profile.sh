#!/bin/bash
# Phase costs inside the select kernels of the phase-split flatten pipeline: a development build of the library
# (make DEBUG=1, or -DLK_FLAT_PROFILE on flatten.hip) returns from flat_init_kernel / flat_dtseg_kernel at stop point k of the
# sampled select (LK_FLAT_STOP=100+k / 200+k; 99 = before the select); kernel-time differences between successive stop
# points are the phase costs.  tools/flat_split_profile.sh <outdir> <lib.so> [bench args...]
out=$1; lib=$2; shift 2
mkdir -p "$out"; R=$PWD; export TMPDIR=/tmp
for stop in ${STOPS:--1 100 101 102 103 104 107 200 201 202 203 204 207 208}; do
  cd /tmp
  LK_FLAT_STOP=$stop LK_LIB_PATH=$R/$lib rocprofv3 --kernel-trace --stats -d "$R/$out/t$stop" -o fl -- python "$R/bench.py" --workload flatten --no-cpu-baseline --steps 5 --warmup 2 "$@" > /dev/null 2> "$R/$out/err$stop.txt"
  cd "$R"
  db=$(ls $out/t$stop/*/*results.db $out/t$stop/*results.db 2>/dev/null | head -1)
  python tools/rocprof_summary.py "$db" "stop $stop" --skip-frac 0.3 | awk -v s=$stop '/steady state/{f=1} f && /flat_(init|dtseg)_kernel/{printf "stop %4d  %-28s avg %8.1f us\n", s, $1, $(NF-1)}'
  rm -rf "$out/t$stop"
done
