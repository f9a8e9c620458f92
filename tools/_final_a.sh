#!/bin/bash
# final batch A: headline profile (kernel trace + by-grid), PMC traffic, then the default bench line with the fresh traffic file
mkdir -p gpurun_out/final
bash tools/profile_bench.sh > gpurun_out/final/profile.log 2>&1 || { echo profile failed; tail -5 gpurun_out/final/profile.log; exit 1; }
mkdir -p gpurun_out/final/prof_c3 && cp gpurun_out/prof/*.csv gpurun_out/prof/bench_line.json gpurun_out/final/prof_c3/
bash tools/profile_pmc.sh > gpurun_out/final/pmc.log 2>&1 || { echo pmc failed; tail -5 gpurun_out/final/pmc.log; exit 1; }
cp profiles/pmc_traffic.json gpurun_out/final/pmc_traffic.json
timeout -k 10 900 python bench.py > gpurun_out/final/c3.json 2> gpurun_out/final/c3.err || { echo bench failed; tail -5 gpurun_out/final/c3.err; exit 1; }
python - <<'PY'
import json
l=json.loads(open('gpurun_out/final/c3.json').read().strip().splitlines()[-1])
print({k:l.get(k) for k in ('value','ms_per_step','krylov_its_per_step','roofline','cpu_baseline','vs_baseline')})
PY
