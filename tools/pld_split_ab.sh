#!/bin/bash
# A/B of the three eigen-iteration modes (LK_PLD_SPLIT = 0 one kernel, 1 phase-split, 2 phase-split in two halves) with a development build of
# pld.hip (tools/build_variant.sh pld_dbg lightkurve_amd/csrc/pld.hip "-DLK_PLD_DEBUG") on ONE box, parity tests first
python -m pytest tests/test_pld_gpu.py tests/test_designmatrix_gpu.py tests/test_regress_gpu.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do for sp in 0 1 2; do
  LK_PLD_SPLIT=$sp LK_LIB_PATH=$PWD/build/ab/pld_dbg.so python bench.py --workload pld --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $sp rep $r', d['ms_per_step'], d['accuracy']['vs_reference']['corrected_flux_relerr_max'], d['accuracy']['vs_reference']['outlier_masks_equal'])"
done; done
for sp in 0 1; do LK_PLD_SPLIT=$sp LK_PLD_ITERS=1 LK_LIB_PATH=$PWD/build/ab/pld_dbg.so python bench.py --workload pld --no-cpu-baseline --steps 1 --warmup 1 2>&1 >/dev/null | grep "pld eig" | head -2; done
