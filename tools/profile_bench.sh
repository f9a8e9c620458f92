#!/bin/bash
# rocprofv3 kernel trace of the default bench on the GPU box: gpurun -- 'bash tools/profile_bench.sh [bench args]'
# Output: gpurun_out/prof/kernel_stats.csv (+ the bench line); copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o bench -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $out/bench_line.json 2> $out/bench.err
db=$(find $out -name "*_results.db" | head -1)
if [ -n "$db" ]; then python3 tools/rocpd_stats.py "$db" $out/kernel_stats.csv --by-grid; rm -f "$db"; fi
ls -la $out | head; head -12 $out/kernel_stats.csv | cut -c1-150
