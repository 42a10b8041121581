mkdir -p gpurun_out
export BIS_TRSV_CHAIN_STATS=1
timeout -k 10 300 python -m pytest tests/test_gpu_unstr.py -x -q -k "not raw_anderson" 2>&1 | tail -8 > gpurun_out/g2_tests.log
cat gpurun_out/g2_tests.log
timeout -k 10 200 python tools/sweep_bench.py fem:40,40,41 asis "tiled=0,chain=0" "tiled=0,chain=1" "tiled=0,chain=1,chain_pairs=128" "tiled=0,chain=1,chain_idle=1000000" "tiled=0,chain=1,chain_pause=32" 2>&1 | grep -v "plan (" > gpurun_out/g2_bench_small.log
cat gpurun_out/g2_bench_small.log
timeout -k 10 300 python tools/sweep_bench.py fem:80,80,81 asis "tiled=0,chain=0" "tiled=0,chain=1" "tiled=0,chain=1,chain_pairs=256" "tiled=0,chain=1,chain_pause=2"  2>&1 | grep -v "plan (" > gpurun_out/g2_bench_fem.log
cat gpurun_out/g2_bench_fem.log
timeout -k 10 300 python tools/sweep_bench.py unstr:80,80,80 rcm chain=0 chain=1 "chain=1,chain_pairs=256" "chain=1,chain_pairs=128" "chain=1,chain_pause=2" "chain=1,chain_pause=32,chain_idle=1" 2>&1 | grep -v "plan (" > gpurun_out/g2_bench_unstr_rcm.log
cat gpurun_out/g2_bench_unstr_rcm.log
