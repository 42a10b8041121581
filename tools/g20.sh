mkdir -p gpurun_out
timeout -k 10 200 python tools/chain_probe.py 16384 1 13 35 64 > gpurun_out/g22_probe.log 2>&1; cat gpurun_out/g22_probe.log
timeout -k 10 300 python tools/sweep_bench.py fem:80,80,81 asis "tiled=0,chain=1" "tiled=0,chain=1,chain_pairs=231" 2>&1 | grep -v "plan (" > gpurun_out/g22_fem.log
cat gpurun_out/g22_fem.log
timeout -k 10 300 python tools/sweep_bench.py unstr:80,80,80 rcm chain=1 "chain=1,chain_pairs=90" 2>&1 | grep -v "plan (" > gpurun_out/g22_unstr.log
cat gpurun_out/g22_unstr.log
timeout -k 10 300 python tools/sweep_bench.py fem:40,40,41 asis "tiled=0,chain=1" tiled=-1 2>&1 | grep -v "plan (" > gpurun_out/g22_small.log
cat gpurun_out/g22_small.log
