R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/r5q
for stop in ${STOPS:--1 300 301 302 303}; do
  cd /tmp
  LK_FLAT_STOP=$stop LK_LIB_PATH=$R/build/ab/prof.so rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r5q/t$stop" -o fl -- python "$R/bench.py" --workload flatten --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2> /dev/null
  cd "$R"
  db=$(ls gpurun_out/r5q/t$stop/*/*results.db gpurun_out/r5q/t$stop/*results.db 2>/dev/null | head -1)
  echo "stop $stop: $(python tools/rocprof_summary.py "$db" x --timeline 16 | grep flat_trend | awk '{printf "%s ", $(NF-1)}')"
  rm -rf gpurun_out/r5q/t$stop
done
