#!/bin/bash
# gpurun -- bash tools/profile_8m.sh : by-grid kernel traces of the 8 M-DOF configurations for the scaling model
mkdir -p gpurun_out/final
( while true; do sleep 60; date >> gpurun_out/final/heartbeat.log; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
for cfg in c5 c5b8; do
  if [ $cfg = c5 ]; then bash tools/profile_bench.sh --config c5 --steps 6 > gpurun_out/final/profile_$cfg.log 2>&1; else bash tools/profile_bench.sh --config c5b --res3 1e-4 --steps 6 > gpurun_out/final/profile_$cfg.log 2>&1; fi
  mkdir -p gpurun_out/final/prof_$cfg && cp gpurun_out/prof/*.csv gpurun_out/prof/bench_line.json gpurun_out/final/prof_$cfg/ && echo $cfg ok
done
