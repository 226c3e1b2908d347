#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONPATH=$PWD
O=gpurun_out/r2l; mkdir -p $O
timeout 900 python -m pytest tests/test_flatten_gpu.py tests/test_seams_gpu.py -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-400
LK_FLAT_NEAR=0 python tools/flat_quad_check.py run $O/a.npz 401 2>/dev/null; python tools/flat_quad_check.py run $O/b.npz 401 2>/dev/null; python tools/flat_quad_check.py compare $O/b.npz $O/a.npz
rm -f $O/*.npz
timeout 300 python bench.py --workload flatten --no-cpu-baseline --steps 5 --warmup 2 > $O/flat.json 2> $O/flat.err; python -c "import json;d=json.load(open('$O/flat.json'));print('flatten ms/step',d['ms_per_step'])"
LK_FLAT_NEAR=0 timeout 300 python bench.py --workload flatten --no-cpu-baseline --steps 5 --warmup 2 > $O/flat0.json 2> $O/flat0.err; python -c "import json;d=json.load(open('$O/flat0.json'));print('flatten (near off) ms/step',d['ms_per_step'])"
python tools/flat_phase_profile.py 1000 2>/dev/null | grep stop > $O/phase.txt; cat $O/phase.txt
echo done
