mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_unstr.py tests/test_gpu_kernels.py -q -k "ilu0" 2>&1 | tail -5 > gpurun_out/g13_tests.log
cat gpurun_out/g13_tests.log
for a in "fem:40,40,41 asis" "fem:80,80,81 asis" "unstr:80,80,80 rcm" "hpcg:128 asis"; do timeout -k 10 200 python tools/ilu_bench.py $a >> gpurun_out/g13_ilu.log 2>&1; done
cat gpurun_out/g13_ilu.log
