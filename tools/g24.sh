mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_unstr.py tests/test_dist.py -m gpu -q -k "randomised or rehearsal or launches" 2>&1 | tail -15 > gpurun_out/g24_tests.log
cat gpurun_out/g24_tests.log
