mkdir -p gpurun_out
timeout -k 10 300 python tools/sweep_bench.py fem:80,80,81 asis "tiled=0,chain=1" "tiled=0,chain=1,chain_prefix=0" "tiled=0,chain=1,chain_prefix=0,chain_pairs=231" "tiled=0,chain=1,chain_pairs=231" 2>&1 | grep -v "plan (" > gpurun_out/g19_fem.log
cat gpurun_out/g19_fem.log
timeout -k 10 300 python tools/sweep_bench.py unstr:80,80,80 rcm chain=1 "chain=1,chain_prefix=0" "chain=1,chain_prefix=0,chain_pairs=90" "chain=1,chain_pairs=90" 2>&1 | grep -v "plan (" > gpurun_out/g19_unstr.log
cat gpurun_out/g19_unstr.log
