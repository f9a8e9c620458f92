#!/bin/bash
mkdir -p gpurun_out/final
( while true; do sleep 60; date >> gpurun_out/final/heartbeat.log; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/final/gputests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/final/gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/final/smoke.log
