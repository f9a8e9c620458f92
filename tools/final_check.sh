#!/bin/bash
# gpurun -- bash tools/final_check.sh : headline kernel trace + PMC passes + the bench line at the driver's settings + the whole GPU suite + smoke, at the current sources
mkdir -p gpurun_out/final
( while true; do sleep 60; date >> gpurun_out/final/heartbeat.log; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
bash tools/profile_bench.sh > gpurun_out/final/profile.log 2>&1 || { echo profile failed; tail -5 gpurun_out/final/profile.log; exit 1; }
mkdir -p gpurun_out/final/prof_c3 && cp gpurun_out/prof/*.csv gpurun_out/prof/bench_line.json gpurun_out/final/prof_c3/
bash tools/profile_pmc.sh > gpurun_out/final/pmc.log 2>&1 || { echo pmc failed; tail -5 gpurun_out/final/pmc.log; exit 1; }
cp profiles/pmc_traffic.json gpurun_out/final/pmc_traffic.json
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/c3.json 2> gpurun_out/final/c3.err || { echo bench failed; tail -5 gpurun_out/final/c3.err; exit 1; }
python - <<'PY'
import json
l=json.loads(open('gpurun_out/final/c3.json').read().strip().splitlines()[-1])
print({k:l.get(k) for k in ('value','ms_per_step','krylov_its_per_step')}); print(l['roofline'])
PY
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > gpurun_out/final/gputests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/final/gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"
