#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_gen3.py -x -q -m gpu > gpurun_out/r4_q_tests.log 2>&1
echo "rc=$?"; tail -8 gpurun_out/r4_q_tests.log
for cfg in q1h p2t; do
timeout -k 10 600 python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r4_q_$cfg.json 2> gpurun_out/r4_q_$cfg.err
python - <<PY
import json
l=json.loads(open('gpurun_out/r4_q_$cfg.json').read().strip().splitlines()[-1])
print('$cfg', {k:l.get(k) for k in ('value','ms_per_step','ms_assemble_per_step','ms_solve_per_step','krylov_its_per_step')})
PY
done
