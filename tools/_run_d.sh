export CFDH_SHARE_GPU=1 CFDH_RCCL_LIB=$GRAFT_REPO_ROOT/tests/fake_rccl/libfake_rccl.so
make -C tests/fake_rccl -s
for g in 1 0; do
CFDH_DL0_GHOST_RHS=$g timeout -k 10 500 python bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 --host-loop-steps 0 > gpurun_out/r4_d_n4_ghost$g.json 2> gpurun_out/r4_d_n4_ghost$g.err
python - <<PY
import json
d=json.load(open("gpurun_out/r4_d_n4_ghost$g.json"))
print("ghost_rhs $g:", d["config"]["parallelism"][:80], "its/step", d["krylov_its_per_step"], {k: round(v,2) for k,v in d["per_krylov_iteration"].items()}, round(d["value"],1))
PY
done
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --prof-steps 0 --host-loop-steps 0 > gpurun_out/r4_d_n1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r4_d_n1.json')); print('n1 its/step', d['krylov_its_per_step'], d['value'])"
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q > gpurun_out/r4_d_tests.log 2>&1; tail -5 gpurun_out/r4_d_tests.log
