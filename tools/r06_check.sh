python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do python -m pytest tests/test_batch_api_gpu.py tests/test_lsfast_variants_gpu.py tests/test_lschi2_gpu.py tests/test_determinism_gpu.py -m gpu -x -q 2>&1 | tail -1; done
LK_PLD_ITERS=1 LK_LIB_PATH=$PWD/build/ab/pld_dbg.so python bench.py --workload pld --no-cpu-baseline --steps 1 --warmup 1 2>&1 >/dev/null | grep "pld tridiag\|pld eig" | head -4
for r in 1 2 3; do python bench.py --workload pld --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pld', d['ms_per_step'], d['accuracy']['vs_reference']['corrected_flux_relerr_max'], d['accuracy']['vs_reference']['outlier_masks_equal'])"; done
