timeout -k 10 300 python tools/p2_dfg_run.py 60 100 2>&1 | tail -3 | cut -c1-1200
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4_h_tests.log 2>&1; tail -5 gpurun_out/r4_h_tests.log
for cfg in q1 q1h p2t; do timeout -k 10 500 python bench.py --config $cfg --steps 10 --warmup 3 > gpurun_out/r4_h_$cfg.json 2> gpurun_out/r4_h_$cfg.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r4_h_$cfg.json"))
    print("$cfg", round(d["value"],2), "its", d["krylov_its_per_step"], "newton", d["newton_its_per_step"], "asm ms/step", round(d["ms_assemble_per_step"],2), "cpu", d.get("cpu_baseline",{}).get("value"), "parity", {k: v for k,v in d.get("parity",{}).items() if k.endswith("_rel")}, [(k["kernel"], round(k["avg_us"],1)) for k in d["kernels"][:3]], d["config"]["workload"][:100])
except Exception as e: print("$cfg failed", e); import subprocess; print(open("gpurun_out/r4_h_$cfg.err").read()[-800:])
PY
done
