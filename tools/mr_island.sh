#!/bin/bash
# two ranks on one GPU (RCCL stand-in), lid cavity with rank 1 owning an interior island, at the tolerances given:
#   gpurun -- 'bash tools/mr_island.sh 1e-11 1e-9 [verbose]'   -> gpurun_out/island_r{0,1}.log
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
make -C tests/fake_rccl -s
PORT=$((20000 + RANDOM % 20000))
for r in 0 1; do
  RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT OMP_NUM_THREADS=2 CFDH_HOST_THREADS=2 CFDH_TEST_BACKEND=${BACKEND:-rccl} \
  CFDH_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so CFDH_TEST_CASE=lid CFDH_TEST_PARTITION=${PARTITION:-interior_island} CFDH_TEST_SNES_RTOL=$1 CFDH_TEST_KSP_RTOL=$2 \
  CFDH_TEST_VERBOSE=${3:-1} timeout -k 10 300 python tests/_gpu_rank_worker.py /tmp/island.npz > gpurun_out/island_r$r.log 2>&1 &
done
wait
grep -c "fgmres" gpurun_out/island_r0.log; grep -v "fgmres " gpurun_out/island_r0.log | tail -40
