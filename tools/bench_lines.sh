#!/bin/bash
# bench lines of the named configurations at the driver's --steps 20 --warmup 5 -> gpurun_out/final/<name>.json (tools/collect_lines.py copies them
# into profiles/r04_bench_line_*.json):  gpurun -- 'bash tools/bench_lines.sh c2 c4 c5 c5b c5b8 q1 q1h p2 p2s p2t c5bdf2'
mkdir -p gpurun_out/final
( while true; do sleep 60; date >> gpurun_out/final/heartbeat.log; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
run() { name=$1; shift; timeout -k 10 1100 python bench.py --steps 20 --warmup 5 "$@" > gpurun_out/final/$name.json 2> gpurun_out/final/$name.err; rc=$?
  python - <<PY
import json
try:
    l=json.loads(open('gpurun_out/final/$name.json').read().strip().splitlines()[-1])
    print('$name', $rc, {k:l.get(k) for k in ('value','ms_per_step','krylov_its_per_step','ms_assemble_per_step')}, (l.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('$name', $rc, 'no line', e)
PY
}
for cfg in "$@"; do
case $cfg in
 c2) run c2 --config c2 ;;
 c4) run c4 --config c4 ;;
 c5) run c5 --config c5 ;;
 c5bdf2) run c5_bdf2_vmax0.5 --config c5 --solver stabilized_schur_bdf2 --v-max 0.5 --no-cpu-baseline ;;
 c5b) run c5b_1m --config c5b ;;
 c5bmean) run c5b_1m_remove_p_mean1 --config c5b --remove-p-mean 1 ;;
 c5b8) run c5b_8m --config c5b --res3 1e-4 --no-cpu-baseline --parity-steps 0 ;;
 q1) run q1 --config q1 ;;
 q1h) run q1h --config q1h ;;
 p2) run p2 --config p2 ;;
 p2s) run p2s --config p2s --no-cpu-baseline ;;
 p2t) run p2t --config p2t ;;
esac
done
