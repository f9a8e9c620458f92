# by-grid kernel statistics for the scaling model: c3 (1 M), c5 (8.2 M, 2-D), c5b 8 M (3-D)
prof() { tag=$1; shift
  bash tools/profile_bench.sh "$@" > gpurun_out/r4_e_prof_$tag.log 2>&1
  mkdir -p gpurun_out/r4_e_$tag; cp gpurun_out/prof/kernel_stats.csv gpurun_out/prof/kernel_stats_by_grid.csv gpurun_out/prof/bench_line.json gpurun_out/r4_e_$tag/ 2>/dev/null
  tail -3 gpurun_out/r4_e_prof_$tag.log | cut -c1-200; }
prof c3 --prof-steps 0 --host-loop-steps 0
prof c5 --config c5 --warmup 35 --prof-steps 0 --host-loop-steps 0
prof c5b8m --config c5b --res3 1e-4 --steps 10 --prof-steps 0 --host-loop-steps 0
# 4 ranks on the one GPU through the RCCL stand-in: with / without the exchange of the ghost layer of the pressure right-hand side
( export CFDH_SHARE_GPU=1 CFDH_RCCL_LIB=$GRAFT_REPO_ROOT/tests/fake_rccl/libfake_rccl.so
for cfg in "c4 --steps 6 --warmup 3" "c5b --steps 6 --warmup 3" "c2 --steps 10 --warmup 3"; do set -- $cfg; for g in 1 0; do
CFDH_DL0_GHOST_RHS=$g timeout -k 10 600 python bench.py --gpus 4 --config $cfg --no-cpu-baseline --prof-steps 0 --host-loop-steps 0 > gpurun_out/r4_e_n4_$1_ghost$g.json 2> gpurun_out/r4_e_n4_$1_ghost$g.err
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r4_e_n4_$1_ghost$g.json"))
    print("$1 ghost_rhs $g: its/step", d["krylov_its_per_step"], {k: round(v,2) for k,v in d["per_krylov_iteration"].items()}, round(d["value"],2))
except Exception as e: print("$1 ghost $g failed", e)
PY
done; done )
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_e_tests.log 2>&1; tail -5 gpurun_out/r4_e_tests.log
