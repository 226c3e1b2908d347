#!/bin/bash
# A/B of liblkhip.so builds on ONE box (boxes differ by +-2 %): tools/ab_lsfast.sh <outdir> <reps> lib1.so lib2.so ...
out=$1; reps=$2; shift 2
mkdir -p $out
for r in $(seq $reps); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    LK_LIB_PATH=$PWD/$lib python bench.py --no-bls --no-pld --no-flatten --no-host --no-cpu-baseline --ls-method fast --no-api --steps 20 --warmup 3 > $out/$tag.$r.json 2> $out/$tag.$r.err
    echo "$tag rep $r $(grep -o 'ms_per_step[^,]*' $out/$tag.$r.json | head -1)"
  done
done
