mkdir -p gpurun_out
export PMC_GROUPS="1 2 10"
timeout -k 10 500 bash tools/spmv_pmc.sh r04_512_crs 0 hpcg 512 > gpurun_out/g15_pmc512_crs.log 2>&1
timeout -k 10 500 bash tools/spmv_pmc.sh r04_512_default -1 hpcg 512 > gpurun_out/g15_pmc512_def.log 2>&1
tail -3 gpurun_out/g15_pmc512_crs.log; cat gpurun_out/spmv_pmc_r04_512_crs/spmv_traffic.json | head -40
bash tools/dist_gap_trace.sh s32 256 32; bash tools/dist_gap_trace.sh s128 256 128
cat gpurun_out/dist_gap_s32.txt gpurun_out/dist_gap_s128.txt
timeout -k 10 300 python -m pytest tests/test_gpu_unstr.py -q -k "ilu0" 2>&1 | tail -3
