#!/bin/bash
# Shape experiments of the BLS launcher on the GPU box (DEBUG build of bls.hip in build/ab/lib_blsdbg.so):
#   tools/bls_shape_sweep.sh <outdir> "VAR=val VAR=val" "..." ...   one bench run (128 targets, LK_BLS_PROF=1) per setting
out=$1; shift
mkdir -p $out
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg LK_LIB_PATH=$PWD/build/ab/lib_blsdbg.so LK_BLS_PROF=${PROF:-1} timeout 200 python bench.py --workload bls --targets 128 --steps 2 --warmup 1 --no-cpu-baseline > $out/run$i.json 2> $out/run$i.err
  echo "== $cfg : $(grep -o '"ms_per_step": [0-9.]*' $out/run$i.json)"
done
