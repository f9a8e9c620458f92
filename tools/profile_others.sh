#!/bin/bash
# gpurun -- bash tools/profile_others.sh : rocprofv3 kernel traces of the 3-D bench configurations (c5b 1 M, q1h, p2t) -> gpurun_out/final/prof_<cfg>/
mkdir -p gpurun_out/final
( while true; do sleep 60; date >> gpurun_out/final/heartbeat.log; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
for cfg in c5b q1h p2t; do
  bash tools/profile_bench.sh --config $cfg --steps 6 > gpurun_out/final/profile_$cfg.log 2>&1
  mkdir -p gpurun_out/final/prof_$cfg && cp gpurun_out/prof/*.csv gpurun_out/prof/bench_line.json gpurun_out/final/prof_$cfg/ && echo $cfg ok && head -6 gpurun_out/prof/kernel_stats.csv | cut -c1-150
done
