for eta in 1e-2 1e-6; do for lag in 0 9; do
CFDH_GS_ETA2=$eta CFDH_KSP_LAG=$lag timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 --host-loop-steps 0 > gpurun_out/r4_c_c3_$eta_$lag.json 2> gpurun_out/r4_c_c3.err; python - <<PY
import json
d=json.load(open("gpurun_out/r4_c_c3_$eta_$lag.json"))
print("c3 eta $eta lag $lag", round(d["value"],1), d["krylov_its_per_step"], {k: round(v,2) for k,v in d["per_krylov_iteration"].items()}, "zero guess:", round(d["zero_initial_guess_check"]["steps_per_s"],1), d["zero_initial_guess_check"]["krylov_its_per_step"], "e2e", round(d.get("end_to_end_measured",{}).get("steps_per_s",0),1))
PY
done; done
for cfg in "c2 288 30" "c4 115 16" "c5b 2e-4 12"; do for eta in 1e-2 1e-6; do
echo "== $cfg eta $eta"; CFDH_GS_ETA2=$eta timeout -k 10 400 python tools/amg_dev_check.py $cfg 2>&1 | tail -2 | cut -c1-600
done; done
for ml in 2 3; do for cfg in "c3 200 20" "c2 288 20" "c4 115 10" "c5b 2e-4 10"; do
echo "== A_MAXLEV $ml $cfg"; CFDH_A_MAXLEV=$ml timeout -k 10 400 python tools/amg_dev_check.py $cfg 2>&1 | tail -2 | cut -c1-600
done; done
echo "== baseline c3 200 20"; timeout -k 10 400 python tools/amg_dev_check.py c3 200 20 2>&1 | tail -2 | cut -c1-600
