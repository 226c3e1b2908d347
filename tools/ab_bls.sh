#!/bin/bash
# A/B of liblkhip.so builds for the BLS bench (128 targets) on ONE box: tools/ab_bls.sh <outdir> <reps> lib1.so lib2.so ...
out=$1; reps=$2; shift 2
mkdir -p $out
for r in $(seq $reps); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    LK_LIB_PATH=$PWD/$lib timeout 300 python bench.py --workload bls --targets ${BLS_TARGETS:-128} --steps 2 --warmup 1 --no-cpu-baseline > $out/$tag.$r.json 2> $out/$tag.$r.err
    echo "$tag rep $r $(grep -o 'ms_per_step[^,]*' $out/$tag.$r.json | head -1)"
  done
done
