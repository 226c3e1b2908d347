#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests/test_lsfast_gpu.py tests/test_lsfast_variants_gpu.py -q --timeout=600 2>&1 | tail -2
run() { L=$1; shift
  env "$@" timeout 600 python bench.py --ls-method fast --no-bls --no-host --steps 5 --warmup 2 > $O/ls_$L.json 2> $O/ls_$L.err; python -c "
import json;d=json.load(open('$O/ls_$L.json'));print('$L ms/step',d['ms_per_step'],'frac',d['roofline']['frac'],(d.get('accuracy') or {}).get('ls_fast',{}).get('max_power_relerr_max'),(d.get('accuracy') or {}).get('ls_fast',{}).get('argmax_equal'))"
}
run fused LK_LSF_FUSED_SPREAD=1
run unfused LK_LSF_FUSED_SPREAD=0
R=$PWD
cd /tmp && export TMPDIR=/tmp
LK_LSF_FUSED_SPREAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/tr -o ls -- python $R/bench.py --ls-method fast --no-bls --no-host --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $R/$O/tr.err
cd $R; python tools/rocprof_summary.py $O/tr/ls_results.db "fused spreader" | head -10; rm -rf $O/tr
echo done
