#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "built_on_the_device" > gpurun_out/r4_o_tests.log 2>&1
echo "rc=$?"; tail -15 gpurun_out/r4_o_tests.log
for n in 1 4; do
CFDH_SHARE_GPU=1 CFDH_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so timeout -k 10 600 python bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_o_n$n.json 2> gpurun_out/r4_o_n$n.err
python - <<PY
import json
l=json.loads(open('gpurun_out/r4_o_n$n.json').read().strip().splitlines()[-1])
print($n, {k:l.get(k) for k in ('value','ms_per_step','hierarchy_build_s','first_step_s','setup_s','krylov_its_per_step')})
PY
done
CFDH_PC_HOST_ASSEMBLY=1 CFDH_SHARE_GPU=1 CFDH_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so timeout -k 10 600 python bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_o_n4h.json 2> gpurun_out/r4_o_n4h.err
python - <<PY
import json
l=json.loads(open('gpurun_out/r4_o_n4h.json').read().strip().splitlines()[-1])
print('4 host', {k:l.get(k) for k in ('value','ms_per_step','hierarchy_build_s','first_step_s','setup_s','krylov_its_per_step')})
PY
