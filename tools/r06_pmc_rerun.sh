out=gpurun_out/r06; mkdir -p $out
tools/trace_pld.sh $out/pld
tools/pmc_pass.sh $out pld_fetch FETCH_SIZE --workload pld --no-cpu-baseline --no-api --steps 3 --warmup 1
tools/pmc_pass.sh $out pld_write WRITE_SIZE --workload pld --no-cpu-baseline --no-api --steps 3 --warmup 1
tools/pmc_pass.sh $out flat_fetch FETCH_SIZE --workload flatten --no-cpu-baseline --no-api --steps 3 --warmup 1
tools/pmc_pass.sh $out flat_write WRITE_SIZE --workload flatten --no-cpu-baseline --no-api --steps 3 --warmup 1
find $out -name "*results.db" -size +30M -delete
