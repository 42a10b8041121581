mkdir -p gpurun_out
timeout -k 10 300 python tools/chain_probe.py 16384 1 3 13 35 64 100 > gpurun_out/g3_probe.log 2>&1
cat gpurun_out/g3_probe.log
