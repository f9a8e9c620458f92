timeout -k 10 900 python -m pytest tests/test_gpu_gen3.py -x -q > gpurun_out/r4_g_gen3.log 2>&1; tail -15 gpurun_out/r4_g_gen3.log
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q > gpurun_out/r4_g_mr.log 2>&1; tail -5 gpurun_out/r4_g_mr.log
# config 5 nearer the reference's operating point with the BDF2 plugin: 1.1 M-DOF mesh of the same domain, 200 steps
for v in 0.5 1.5; do for ramp in 0.03 0; do
timeout -k 10 400 python tools/c5_range.py $v 2e-5 200 0 $ramp stabilized_schur_bdf2 2>&1 | tail -1 | cut -c1-900
done; done
# P2 stenosis variants (which workload reaches its own T): v_max, severity
for a in "100 4 40 1e-5 5" "100 4 40 1e-5 20 0.0" "100 4 40 1e-5 10 0.3"; do timeout -k 10 500 python tools/p2_long_run.py $a 2>&1 | tail -2 | cut -c1-700; done
