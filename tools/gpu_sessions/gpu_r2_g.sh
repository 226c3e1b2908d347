#!/bin/bash
# round-2 GPU session G: full parity suite, default bench, flatten + LS profiles (trace + HBM PMC)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-400
echo "== default bench"; timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2g/bench_default.json'))
print('fast ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'acc', d.get('accuracy'), 'h2h', {k:v.get('ms_per_step') for k,v in d.get('host_to_host',{}).items() if isinstance(v,dict)}, 'bls', d['bls']['ms_per_step'], d['bls'].get('accuracy'))
PY
echo "== flatten bench + profiles"
timeout 300 python bench.py --workload flatten --targets 1024 --steps 10 --warmup 2 > $O/bench_flatten.json 2> $O/bench_flatten.err; python -c "import json;d=json.load(open('$O/bench_flatten.json'));print('flatten ms/step',d['ms_per_step'],'cpu',d.get('speedup_vs_cpu_baseline'))"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OLDPWD/$O/flat_pmc_$C -o p -- python $OLDPWD/bench.py --workload flatten --targets 1024 --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $OLDPWD/$O/flat_pmc_$C.err)
  DB=$(find $O/flat_pmc_$C -name "*results.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "flatten 1024 x 20000, window 401, --steps 2 --warmup 1, --pmc $C" > $O/flat_pmc_$C.txt && rm -rf $O/flat_pmc_$C && grep -E "flatten_kernel" $O/flat_pmc_$C.txt
done
echo "== LS fast trace (default path)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/ls_trace -o ls -- python $OLDPWD/bench.py --ls-method fast --no-bls --no-host --no-cpu-baseline --steps 5 --warmup 2 > $OLDPWD/$O/ls_trace.json 2> $OLDPWD/$O/ls_trace.err)
DB=$(find $O/ls_trace -name "*results.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "bench.py --ls-method fast --no-bls --no-host --steps 5 --warmup 2 (round 2 final: pruned column FFT + spreader tables)" > $O/ls_trace_summary.txt && rm -rf $O/ls_trace && head -14 $O/ls_trace_summary.txt
echo done
