#!/bin/bash
mkdir -p gpurun_out/final
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final/c3_again.json 2> gpurun_out/final/c3_again.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/final/c3_again.json').read().strip().splitlines()[-1])
print('c3 again', {k:l.get(k) for k in ('value','ms_per_step','krylov_its_per_step','ms_assemble_per_step','ms_solve_per_step')})
PY
bash tools/_final_b.sh q1 c5 p2s
ASM_BENCH_3D=2e-4 timeout -k 10 300 python tools/asm_bench.py > gpurun_out/final/asm_bench_3d.log 2>&1; tail -12 gpurun_out/final/asm_bench_3d.log
