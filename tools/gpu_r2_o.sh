#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2o; mkdir -p $O
LK_FFT3=1 timeout 900 python -m pytest tests/test_lsfast_gpu.py tests/test_lsfast_variants_gpu.py -q --timeout=600 2>&1 | tail -2
run() { L=$1; shift
  env "$@" timeout 300 python bench.py --ls-method fast --no-bls --no-host --no-cpu-baseline --steps 5 --warmup 2 > $O/ls_$L.json 2> $O/ls_$L.err; python -c "import json;d=json.load(open('$O/ls_$L.json'));print('$L ms/step',d['ms_per_step'])"
}
run default LK_DUMMY=1
run fft3_rt4 LK_FFT3=1 LK_FFT3_RT=4
run default2 LK_DUMMY=1
run fft3_rt4b LK_FFT3=1 LK_FFT3_RT=4
run fft3_rt8 LK_FFT3=1 LK_FFT3_RT=8
echo done
