bash tools/prof_fp32.sh > gpurun_out/r4_b_prof_fp32.log 2>&1; tail -25 gpurun_out/r4_b_prof_fp32.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_b_tests.log 2>&1; tail -5 gpurun_out/r4_b_tests.log
for lag in 0 9; do CFDH_KSP_LAG=$lag timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_b_c3_lag$lag.json 2> gpurun_out/r4_b_c3_lag$lag.err; python - <<PY
import json
d=json.load(open("gpurun_out/r4_b_c3_lag$lag.json"))
print("lag $lag", d["value"], d["krylov_its_per_step"], d["per_krylov_iteration"], d["zero_initial_guess_check"]["steps_per_s"], d.get("end_to_end_measured",{}).get("steps_per_s"))
PY
done
