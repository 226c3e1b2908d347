#!/bin/bash
# A/B of liblkhip.so builds for flatten on ONE box: tools/ab_flatten.sh <outdir> <reps> lib1.so lib2.so ...
out=$1; reps=$2; shift 2
mkdir -p $out
for r in $(seq $reps); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    LK_LIB_PATH=$PWD/$lib python bench.py --workload flatten --no-cpu-baseline --steps 10 --warmup 2 > $out/$tag.a$r.json 2> $out/$tag.a$r.err
    LK_LIB_PATH=$PWD/$lib python bench.py --workload flatten --cadences 4500 --flatten-window 101 --no-cpu-baseline --steps 10 --warmup 2 > $out/$tag.b$r.json 2> $out/$tag.b$r.err
    LK_LIB_PATH=$PWD/$lib python bench.py --workload flatten --cadences 3500 --flatten-window 101 --no-cpu-baseline --steps 10 --warmup 2 > $out/$tag.c$r.json 2> $out/$tag.c$r.err
    echo "$tag rep $r 20000: $(grep -o 'ms_per_step[^,]*' $out/$tag.a$r.json | head -1)  4500: $(grep -o 'ms_per_step[^,]*' $out/$tag.b$r.json | head -1)  3500: $(grep -o 'ms_per_step[^,]*' $out/$tag.c$r.json | head -1)"
  done
done
