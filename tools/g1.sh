set -x
mkdir -p gpurun_out
export BIS_TRSV_CHAIN_STATS=1
timeout -k 10 500 python -m pytest tests/test_gpu_unstr.py -x -q 2>&1 | tail -25 > gpurun_out/g1_tests.log
cat gpurun_out/g1_tests.log
timeout -k 10 200 python tools/sweep_bench.py fem:40,40,41 asis "tiled=0,chain=0" "tiled=0,chain=1" tiled=-1 > gpurun_out/g1_bench_small.log 2>&1
cat gpurun_out/g1_bench_small.log
timeout -k 10 300 python tools/sweep_bench.py fem:80,80,81 asis "tiled=0,chain=0" "tiled=0,chain=1" tiled=-1 > gpurun_out/g1_bench_fem.log 2>&1
cat gpurun_out/g1_bench_fem.log
timeout -k 10 300 python tools/sweep_bench.py unstr:80,80,80 rcm chain=0 chain=1 > gpurun_out/g1_bench_unstr_rcm.log 2>&1
cat gpurun_out/g1_bench_unstr_rcm.log
