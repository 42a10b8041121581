mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_host_cli.py -m gpu -q -k "hpcg48 or mtx_file" 2>&1 | grep -v "^  \|^$" | head -120 > gpurun_out/g11_tests.log
cat gpurun_out/g11_tests.log | head -100
basic_iterative_solvers_amd/host/basic_iterative_solvers hpcg:48 -cg | grep "converged\|iterations" | head
