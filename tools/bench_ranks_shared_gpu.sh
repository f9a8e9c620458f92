#!/bin/bash
# gpurun -- bash tools/bench_ranks_shared_gpu.sh : ranks sharing the one GPU through the RCCL stand-in (communication counts, iteration ratios), matching 1-rank windows
mkdir -p gpurun_out/final
( while true; do sleep 60; date >> gpurun_out/final/heartbeat.log; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
export CFDH_SHARE_GPU=1
FAKE=$PWD/tests/fake_rccl/libfake_rccl.so
[ -f $FAKE ] || make -C tests/fake_rccl -s
run() { name=$1; shift; timeout -k 10 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/final/$name.json 2> gpurun_out/final/$name.err; rc=$?
  python - <<PY
import json
try:
    l=json.loads(open('gpurun_out/final/$name.json').read().strip().splitlines()[-1])
    print('$name', $rc, {k:l.get(k) for k in ('n_gpus','value','ms_per_step','krylov_its_per_step','per_krylov_iteration','pressure_level1_rows','hierarchy_build_s')})
except Exception as e:
    print('$name', $rc, 'no line', e)
PY
}
for cfg in c3 c4 c5b; do
  st=10; [ $cfg != c3 ] && st=6
  run n1_$cfg --config $cfg --gpus 1 --steps $st --warmup 3
  for n in 2 4; do CFDH_RCCL_LIB=$FAKE run n${n}_$cfg --config $cfg --gpus $n --steps $st --warmup 3; done
done
