#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2s; mkdir -p $O
timeout 900 python -m pytest tests/test_lsfast_gpu.py tests/test_lsfast_variants_gpu.py tests/test_api_gpu.py -q --timeout=600 2>&1 | tail -2
run() { L=$1; shift
  env "$@" timeout 600 python bench.py --ls-method fast --no-bls --no-host --steps 5 --warmup 2 > $O/ls_$L.json 2> $O/ls_$L.err; python -c "
import json;d=json.load(open('$O/ls_$L.json'));print('$L ms/step',d['ms_per_step'],'frac',d['roofline']['frac'],(d.get('accuracy') or {}).get('ls_fast',{}).get('max_power_relerr_max'),(d.get('accuracy') or {}).get('ls_fast',{}).get('argmax_equal'))"
}
run rows_stream LK_DUMMY=1
run one_stream LK_LSF_ROWS_STREAM=0
run rows_stream_nospreadstream LK_LSF_STREAMS=0
run rows_stream_chunk1024 LK_FAST_CHUNK_MB=1024
echo done
