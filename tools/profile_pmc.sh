#!/bin/bash
# HBM traffic counters of the default bench (separate --pmc passes, kernel trace only): gpurun -- 'bash tools/profile_pmc.sh'
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc
rm -rf $out && mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/$ctr -o pmc -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $out/$ctr.json 2> $out/$ctr.err || exit 1
done
python3 tools/pmc_summary.py $out/FETCH_SIZE $out/WRITE_SIZE
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete
ls -R $out | head -20
