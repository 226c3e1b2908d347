#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2e; mkdir -p $O
timeout 600 python tools/flat_phase_profile.py 1024 2>&1 | tee $O/flat_phases.txt | tail -20
run() { name=$1; shift; env "$@" timeout 300 python bench.py --ls-method fast --no-bls --no-host --no-cpu-baseline --steps 10 --warmup 3 > $O/ls_$name.json 2> $O/ls_$name.err; python - "$name" <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/r2e/ls_%s.json'%sys.argv[1])); print('%-28s ms/step %.3f kernel_ms %.3f frac %.3f'%(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac']))
except Exception as e: print(sys.argv[1],'failed',e)
PY
}
run rt2  LK_FFT_RT=2
run rt4  LK_FFT_RT=4
run rt8  LK_FFT_RT=8
echo done
