timeout -k 10 600 python -m pytest tests/test_gpu_gen3.py -x -q > gpurun_out/r4_i_gen3.log 2>&1; tail -4 gpurun_out/r4_i_gen3.log
for cfg in p2 p2t; do timeout -k 10 700 python bench.py --config $cfg --steps 10 --warmup 3 > gpurun_out/r4_i_$cfg.json 2> gpurun_out/r4_i_$cfg.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r4_i_$cfg.json"))
    print("$cfg", round(d["value"],2), "its", d["krylov_its_per_step"], "newton", d["newton_its_per_step"], "asm ms/step", round(d["ms_assemble_per_step"],2), "cpu", d.get("cpu_baseline",{}).get("value"), "parity", {k: v for k,v in d.get("parity",{}).items() if k.endswith("_rel")}, "e2e", d.get("end_to_end_measured",{}).get("steps"), d.get("end_to_end_measured",{}).get("stopped"), [(k["kernel"], round(k["avg_us"],1)) for k in d["kernels"][:3]], d["config"]["workload"][:100])
except Exception as e: print("$cfg failed", e); print(open("gpurun_out/r4_i_$cfg.err").read()[-600:])
PY
done
