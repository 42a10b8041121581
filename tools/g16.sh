mkdir -p gpurun_out
python tools/dist_overhead.py 256 32 200 2>&1 | grep "rows" > gpurun_out/g16_dist.log
python tools/dist_overhead.py 256 128 200 2>&1 | grep "rows" >> gpurun_out/g16_dist.log
cat gpurun_out/g16_dist.log
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -q -k "cg" 2>&1 | tail -3
timeout -k 10 200 python tools/sweep_bench.py fem:80,80,81 asis "tiled=0,chain=1" "tiled=0,chain=1,chain_idle=1000000" "tiled=0,chain=1,chain_idle=1000000,chain_pairs=192" "tiled=0,chain=1,chain_pause=2" 2>&1 | grep -v "plan (" > gpurun_out/g16_fem.log
cat gpurun_out/g16_fem.log
timeout -k 10 200 python tools/sweep_bench.py unstr:80,80,80 rcm chain=1 "chain=1,chain_idle=1000000" "chain=1,chain_idle=1000000,chain_pairs=96" "chain=1,chain_idle=4,chain_pause=2" 2>&1 | grep -v "plan (" > gpurun_out/g16_unstr.log
cat gpurun_out/g16_unstr.log
