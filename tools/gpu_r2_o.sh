#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests/test_lsfast_gpu.py tests/test_lsfast_variants_gpu.py tests/test_seams_gpu.py -q --timeout=600 2>&1 | tail -2
timeout 900 python bench.py --ls-method fast --no-bls --no-host --no-cpu-baseline --steps 5 --warmup 2 > $O/ls.json 2> $O/ls.err; python -c "
import json;d=json.load(open('$O/ls.json'));print('ms/step',d['ms_per_step'],'frac',d['roofline']['frac'])"
echo done
