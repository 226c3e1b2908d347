#!/bin/bash
# round-2 GPU session B: flatten phase profile, regress cov tests, LS-fast kernel trace + HBM PMC
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest flatten/regress"; timeout 900 python -m pytest tests/test_flatten_gpu.py tests/test_regress_gpu.py tests/test_api_gpu.py -q -x --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for NT in 512 1024; do
  echo "== flatten NT=$NT"
  LK_FLAT_NT=$NT timeout 300 python bench.py --workload flatten --targets 1024 --no-cpu-baseline --steps 10 --warmup 2 > $O/flat_$NT.json 2> $O/flat_$NT.err
  python -c "import json;d=json.load(open('$O/flat_$NT.json'));print('flatten NT=$NT ms/step',d['ms_per_step'])"
  LK_FLAT_PROF=1 LK_FLAT_NT=$NT timeout 300 python bench.py --workload flatten --targets 1024 --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2> $O/flat_prof_$NT.err; grep "flatten prof" $O/flat_prof_$NT.err | tail -1
done
echo "== LS fast trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/ls_trace -o ls -- python $OLDPWD/bench.py --ls-method fast --no-bls --no-host --no-cpu-baseline --steps 5 --warmup 2 > $OLDPWD/$O/ls_trace.json 2> $OLDPWD/$O/ls_trace.err)
DB=$(find $O/ls_trace -name "*results.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "bench.py --ls-method fast --no-bls --no-host --steps 5 --warmup 2 (round 2: pruned column FFT)" > $O/ls_trace_summary.txt && rm -rf $O/ls_trace && head -14 $O/ls_trace_summary.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OLDPWD/$O/ls_pmc_$C -o p -- python $OLDPWD/bench.py --ls-method fast --no-bls --no-host --no-cpu-baseline --targets 170 --steps 2 --warmup 1 > /dev/null 2> $OLDPWD/$O/ls_pmc_$C.err)
  DB=$(find $O/ls_pmc_$C -name "*results.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "LS fast, 170 targets, --steps 2 --warmup 1, --pmc $C" > $O/ls_pmc_$C.txt && rm -rf $O/ls_pmc_$C && grep -E "fft_|lsf_" $O/ls_pmc_$C.txt | grep $C
done
echo done
