#!/bin/bash
mkdir -p gpurun_out/final
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/c3.json 2> gpurun_out/final/c3.err || { echo bench failed; tail -5 gpurun_out/final/c3.err; exit 1; }
python - <<'PY'
import json
l=json.loads(open('gpurun_out/final/c3.json').read().strip().splitlines()[-1])
print({k:l.get(k) for k in ('value','ms_per_step','krylov_its_per_step','speedup_vs_cpu_baseline','end_to_end_measured','zero_initial_guess_check','solves_stopped_at_attainable_accuracy')})
print(l['roofline']); print(l['parity']); print(l['cpu_baseline'])
PY
bash tools/_final_b.sh c2 c4 c5b q1 q1h p2t p2
