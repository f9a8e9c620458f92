timeout -k 10 300 python -m pytest tests/test_gpu_gen3.py -x -q -k p_grade_2 > gpurun_out/r4_j_gen3.log 2>&1; tail -3 gpurun_out/r4_j_gen3.log
for m in 70 80 100; do timeout -k 10 200 python tools/p2_dfg_run.py $m 12 2>&1 | tail -3 | cut -c1-500; done
VERBOSE=1 timeout -k 10 200 python tools/p2_dfg_run.py 100 2 2>&1 | grep -v "^\[cfdh\]     fgmres" | head -60 | cut -c1-260
