export CFDH_SHARE_GPU=1 CFDH_RCCL_LIB=$GRAFT_REPO_ROOT/tests/fake_rccl/libfake_rccl.so
make -C tests/fake_rccl -s
for cfg in "c3 --steps 10 --warmup 3" "c4 --steps 6 --warmup 3" "c5b --steps 6 --warmup 3"; do set -- $cfg; for g in 1 0; do
CFDH_RAS_GHOST_RHS=$g timeout -k 10 600 python bench.py --gpus 4 --config $cfg --no-cpu-baseline --prof-steps 0 --host-loop-steps 0 > gpurun_out/r4_l_n4_$1_ras$g.json 2> gpurun_out/r4_l_n4_$1_ras$g.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r4_l_n4_$1_ras$g.json"))
    print("$1 ras_ghost_rhs $g: its/step", d["krylov_its_per_step"], {k: round(v,2) for k,v in d["per_krylov_iteration"].items()}, round(d["value"],2))
except Exception as e: print("$1 $g failed", e)
PY
done; done
