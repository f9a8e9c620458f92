#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --config q1 --steps 3 --warmup 1 --verbose 2 --no-cpu-baseline > gpurun_out/r4_r_q1.json 2> gpurun_out/r4_r_q1.err
echo rc=$?; grep -v "amdgpu.ids" gpurun_out/r4_r_q1.err | head -60
