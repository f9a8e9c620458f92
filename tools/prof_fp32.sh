#!/bin/bash
# rocprofv3 kernel trace of a driver that does NOT import torch (libcfdh.so initialises HIP itself and therefore runs on the
# image's /opt/rocm HIP runtime, not on the one torch bundles): the command of VERDICT round 3 item 2 / ADVICE round 3.
#   gpurun -- 'bash tools/prof_fp32.sh [config size steps]'      default: c4 115 12, CFDH_KRYLOV_FP32 = 0 and 1
# The program stands directly after `--` (no env / bash -c hop between the profiler and python3).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
cfg=${1:-c4}; size=${2:-115}; steps=${3:-12}
for mode in 0 1; do
  out=gpurun_out/prof_fp32_$mode
  rm -rf $out && mkdir -p $out
  export CFDH_KRYLOV_FP32=$mode CFDH_DUMP_MAPS=$out/maps.txt
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out -o run -- python3 tools/amg_dev_check.py $cfg $size $steps > $out/log.txt 2>&1
  echo "mode $mode: exit code $?"
  tail -3 $out/log.txt | cut -c1-400
  db=$(find $out -name "*_results.db" | head -1)
  if [ -n "$db" ]; then python3 tools/rocpd_stats.py "$db" $out/kernel_stats.csv; rm -f "$db"; head -8 $out/kernel_stats.csv | cut -c1-150; fi
done
