out=gpurun_out/r06; mkdir -p $out
tools/pmc_pass.sh $out pld_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" --workload pld --no-cpu-baseline --no-api --steps 2 --warmup 1
tools/pmc_pass.sh $out pld_sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" --workload pld --no-cpu-baseline --no-api --steps 2 --warmup 1
grep "moment_gram\|counter" $out/pmc_pld_sq1.txt | tail -20; grep "moment_gram" $out/pmc_pld_sq2.txt | tail -24
