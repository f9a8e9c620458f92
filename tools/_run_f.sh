timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4_f_tests.log 2>&1; tail -5 gpurun_out/r4_f_tests.log
echo "== gen asm (Q1 bench mesh)"; timeout -k 10 300 python tools/gen_asm_time.py 2>&1 | tail -3
echo "== tet asm (1 M DOF)"; ASM_BENCH_3D=2e-4 timeout -k 10 300 python tools/asm_bench.py 2>&1 | tail -4
for cfg in q1 p2; do timeout -k 10 400 python bench.py --config $cfg --steps 10 --warmup 3 > gpurun_out/r4_f_$cfg.json 2> gpurun_out/r4_f_$cfg.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r4_f_$cfg.json"))
    print("$cfg", round(d["value"],2), d["krylov_its_per_step"], d["newton_its_per_step"], "asm ms/step", round(d["ms_assemble_per_step"],3), d.get("end_to_end_measured"), d["end_to_end_steps_per_s"], [(k["kernel"], round(k["avg_us"],1)) for k in d["kernels"][:3]])
except Exception as e: print("$cfg failed", e)
PY
done
timeout -k 10 400 python bench.py --config c5b --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_f_c5b.json 2> gpurun_out/r4_f_c5b.err; python -c "
import json; d=json.load(open('gpurun_out/r4_f_c5b.json')); print('c5b', round(d['value'],2), d['krylov_its_per_step'], [(k['kernel'], round(k['avg_us'],1)) for k in d['kernels'][:4]])"
