rm -f gpurun_out/r06h_cov.txt
GEMX_COVERAGE_FILE=$PWD/gpurun_out/r06h_cov.txt python -m pytest tests/ -q -m gpu -n 4 2>&1 | tail -4 > gpurun_out/r06h_gpu_tests.txt
python tools/bench_matrix.py --solver default > gpurun_out/r06h_matrix_default_solver.md 2>/dev/null
python bench.py > gpurun_out/r06h_bench.log 2>&1; tail -1 gpurun_out/r06h_bench.log > gpurun_out/r06h_bench_line.json; cp bench_extras.json gpurun_out/r06h_bench_extras.json
