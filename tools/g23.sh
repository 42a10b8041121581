mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/g23_tests.log
cat gpurun_out/g23_tests.log
