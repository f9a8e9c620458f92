#!/bin/bash
# Build an experimental copy of libcfdh.so with extra compiler flags: tools/build_variant.sh NAME "-DCFDH_MAX_INC=128 ..."
# Output: cfd_hemodynamic_amd/variants/libcfdh_NAME.so (git-ignored; tools/asm_bench.py loads it by name).
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/cfd_hemodynamic_amd/csrc
tmp=/tmp/cfdh_variant_$name
mkdir -p $tmp $root/cfd_hemodynamic_amd/variants
FLAGS="-O3 -std=c++17 -fPIC -fopenmp -I$root/include -I$src -Wno-unused-result -Wno-unused-function --offload-arch=gfx950 $extra"
pids=()
for f in cfdh_kernels.hip cfdh3_kernels.hip cfdh_amg_dev.hip cfdh_gen.hip cfdh_setup.cpp cfdh3_setup.cpp cfdh_solver.cpp cfdh_api.cpp cfdh_comm.cpp; do
  /opt/rocm/bin/hipcc $FLAGS -x hip -c $src/$f -o $tmp/${f%.*}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC -fopenmp --offload-arch=gfx950 -o $root/cfd_hemodynamic_amd/variants/libcfdh_$name.so $tmp/*.o -ldl
echo built $root/cfd_hemodynamic_amd/variants/libcfdh_$name.so
