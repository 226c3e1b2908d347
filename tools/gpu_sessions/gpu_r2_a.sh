#!/bin/bash
# round-2 GPU session A: parity suite, default bench, LS-fast variant sweep, flatten bench, PLD MFMA counters
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest" ; timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
echo "== flatten bench"; timeout 300 python bench.py --workload flatten --targets 1024 --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_flatten.json 2> $O/bench_flatten.err; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r2a/bench_flatten.json')); print('flatten ms/step', d['ms_per_step'], 'value', d['value'])
except Exception as e: print('flatten bench failed', e)
PY
echo "== LS fast variants"
run() { name=$1; shift; env "$@" timeout 300 python bench.py --ls-method fast --no-bls --no-host --no-cpu-baseline --steps 10 --warmup 3 > $O/ls_$name.json 2> $O/ls_$name.err; python - "$name" <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/r2a/ls_%s.json'%sys.argv[1])); print('%-28s ms/step %.3f kernel_ms %.3f frac %.3f'%(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac']))
except Exception as e: print(sys.argv[1],'failed',e)
PY
}
run old            LK_LSF_PRUNED=0 LK_LSF_STREAMS=0
run old_streams    LK_LSF_PRUNED=0 LK_LSF_STREAMS=1
run pruned         LK_LSF_PRUNED=1 LK_LSF_PERM=0 LK_LSF_STREAMS=0
run pruned_perm    LK_LSF_PRUNED=1 LK_LSF_PERM=1 LK_LSF_STREAMS=0
run pruned_streams LK_LSF_PRUNED=1 LK_LSF_PERM=0 LK_LSF_STREAMS=1
run prunedperm_str LK_LSF_PRUNED=1 LK_LSF_PERM=1 LK_LSF_STREAMS=1
run def_chunk256   LK_FAST_CHUNK_MB=256
run def_chunk512   LK_FAST_CHUNK_MB=512
run def_chunk1024  LK_FAST_CHUNK_MB=1024
run def_chunk4096  LK_FAST_CHUNK_MB=4096
echo "== default bench (with astropy accuracy)"; timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; head -c 600 $O/bench_default.json; echo; tail -3 $O/bench_default.err
echo "== PLD MFMA counters"
rocprofv3 --list-avail 2>/dev/null | grep -i -E "MFMA|VALU_BUSY|SQ_BUSY_CY" | head -40 > $O/avail_mfma.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 -d $OLDPWD/$O/pld_pmc -o pld -- python $OLDPWD/bench.py --workload pld --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/pld_pmc.log 2>&1); echo "pld pmc rc=$?"
DB=$(find $O/pld_pmc -name "*results.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py "$DB" "PLD bench (--steps 2 --warmup 1): MFMA counters" > $O/pld_pmc_summary.txt 2>&1 && rm -rf $O/pld_pmc && tail -25 $O/pld_pmc_summary.txt
echo done
