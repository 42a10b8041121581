mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_host_cli.py tests/test_gpu_unstr.py -m gpu -q -k "hpcg48 or mtx_file or raw or round4 or unstructured" 2>&1 | tail -5 > gpurun_out/g12_tests.log
cat gpurun_out/g12_tests.log
timeout -k 10 800 python bench.py --steps 20 --warmup 5 > gpurun_out/g12_bench.json 2> gpurun_out/g12_bench.err
tail -c 600 gpurun_out/g12_bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/g12_bench.json'))
r=j['roofline']
print('value',j['value'],'ms/step',j['ms_per_step'],'frac',r['frac'],'kernel',r['kernel'],'traffic',r['traffic'],'moved_frac',r['moved_frac'], r['priced_on'])
c=j.get('compressed_stream',{})
print('compressed',c.get('cg_iterations_per_s'),c.get('roofline',{}).get('frac'),c.get('roofline',{}).get('kernel'))
for k in ('cpu_baseline','cpu_baseline_socket','cpu_baseline_first_touch'):
    if k in j: print(k,j[k]['value'],j[k]['cores'],j[k].get('matrix_first_touch'))
print('parity',j.get('parity_max_dr_over_r0'),j.get('parity_samples'))
t=j.get('target_512',{})
print('t512',t.get('cg_iterations_per_s'),t.get('roofline',{}).get('frac'),t.get('roofline',{}).get('traffic'),'compressed',t.get('compressed_stream',{}).get('cg_iterations_per_s'))
print('config5',j.get('config5_spmv',{}).get('roofline',{}).get('frac'))
PY
