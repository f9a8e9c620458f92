#!/bin/bash
# One-off diagnosis of the SIGSEGV under rocprofv3: the same small driver under the profiler with
# (a) nothing changed, (b) hipGraph replay off, (c) torch imported first (= torch's bundled HIP runtime), (d) the runtime's
# graph packet capture off.  Every run dumps /proc/self/maps so that the frames of a backtrace can be resolved.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  tag=$1; shift
  out=gpurun_out/segv_$tag
  rm -rf $out && mkdir -p $out
  export CFDH_DUMP_MAPS=$out/maps.txt
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out -o run -- python3 tools/amg_dev_check.py c3 100 3 > $out/log.txt 2>&1
  echo "== $tag: exit code $?"
  grep -c . $out/maps.txt 2>/dev/null
  tail -2 $out/log.txt | cut -c1-300
  find $out -name "*_results.db" -delete
}
run asis
CFDH_NO_GRAPH=1 run nograph
CFDH_IMPORT_TORCH_FIRST=1 run torchfirst
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run nopktcapture
