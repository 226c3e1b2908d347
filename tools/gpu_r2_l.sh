#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONPATH=$PWD
O=gpurun_out/r2l; mkdir -p $O
for cfg in "W6 512 4896" "W6 512 4224" "W8 512 4224" "W8 512 3840" "W6 384 4896" "W8 384 4224"; do set -- $cfg
LK_LIB_PATH=$PWD/lightkurve_amd/liblkhip_$1.so LK_FLAT_NT=$2 LK_FLAT_FIR=$3 timeout 300 python bench.py --workload flatten --no-cpu-baseline --steps 5 --warmup 2 > $O/flat_$1_$2_$3.json 2> $O/flat_$1_$2_$3.err; python -c "import json;d=json.load(open('$O/flat_$1_$2_$3.json'));print('flatten $1 nt=$2 fir=$3 ms/step',d['ms_per_step'])"
done
echo done
