#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "generic_elements" > gpurun_out/r4_n_tests.log 2>&1
echo "rc=$?"; tail -30 gpurun_out/r4_n_tests.log
