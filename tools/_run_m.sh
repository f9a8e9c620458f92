timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -k "generic_elements" > gpurun_out/r4_m_mr.log 2>&1; tail -25 gpurun_out/r4_m_mr.log | cut -c1-300
timeout -k 10 600 python -m pytest tests/test_gpu_gen.py tests/test_gpu_gen3.py -x -q > gpurun_out/r4_m_gen.log 2>&1; tail -3 gpurun_out/r4_m_gen.log
