mkdir -p gpurun_out
timeout -k 10 200 python tools/chain_probe.py 16384 1 13 35 64 100 > gpurun_out/g17_probe.log 2>&1
cat gpurun_out/g17_probe.log
timeout -k 10 300 python -m pytest tests/test_gpu_unstr.py -x -q -k "not raw_anderson" 2>&1 | tail -5 > gpurun_out/g17_tests.log
cat gpurun_out/g17_tests.log
timeout -k 10 200 python tools/sweep_bench.py fem:40,40,41 asis "tiled=0,chain=0" "tiled=0,chain=1" "tiled=0,chain=1,chain_pairs=128" 2>&1 | grep -v "plan (" > gpurun_out/g17_bench_small.log
cat gpurun_out/g17_bench_small.log
timeout -k 10 300 python tools/sweep_bench.py fem:80,80,81 asis "tiled=0,chain=0" "tiled=0,chain=1" "tiled=0,chain=1,chain_pairs=256" tiled=-1 2>&1 | grep -v "plan (" > gpurun_out/g17_bench_fem.log
cat gpurun_out/g17_bench_fem.log
timeout -k 10 300 python tools/sweep_bench.py unstr:80,80,80 rcm chain=0 chain=1 "chain=1,chain_pairs=256" "chain=1,chain_pairs=128" 2>&1 | grep -v "plan (" > gpurun_out/g17_bench_unstr_rcm.log
cat gpurun_out/g17_bench_unstr_rcm.log
