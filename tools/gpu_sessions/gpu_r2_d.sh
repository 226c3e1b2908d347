#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest lsfast + flatten"; timeout 900 python -m pytest tests/test_flatten_gpu.py tests/test_lsfast_gpu.py tests/test_lsfast_variants_gpu.py tests/test_lschi2_gpu.py -q -x --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
for NT in 512 1024; do
  LK_FLAT_NT=$NT timeout 300 python bench.py --workload flatten --targets 1024 --no-cpu-baseline --steps 10 --warmup 2 > $O/flat_$NT.json 2> $O/flat_$NT.err
  python -c "import json;d=json.load(open('$O/flat_$NT.json'));print('flatten NT=$NT ms/step',d['ms_per_step'])"
  LK_LIB_PATH=$PWD/lightkurve_amd/liblkhip_prof.so LK_FLAT_PROF=1 LK_FLAT_NT=$NT timeout 300 python bench.py --workload flatten --targets 1024 --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2> $O/flat_prof_$NT.err; grep "flatten prof" $O/flat_prof_$NT.err | tail -1
done
echo "== LS fast variants"
run() { name=$1; shift; env "$@" timeout 300 python bench.py --ls-method fast --no-bls --no-host --no-cpu-baseline --steps 10 --warmup 3 > $O/ls_$name.json 2> $O/ls_$name.err; python - "$name" <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/r2d/ls_%s.json'%sys.argv[1])); print('%-28s ms/step %.3f kernel_ms %.3f frac %.3f'%(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac']))
except Exception as e: print(sys.argv[1],'failed',e)
PY
}
run tables_default   LK_LSF_STREAMS=0
run notables         LK_LSF_STREAMS=0 LK_LSF_TABLES=0
run tables_streams   LK_LSF_STREAMS=1
run tables_rt16      LK_LSF_STREAMS=0 LK_FFT_RT=16
run tables_rt4       LK_LSF_STREAMS=0 LK_FFT_RT=4
echo done
