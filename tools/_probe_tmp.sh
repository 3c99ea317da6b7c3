python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06g_gpu_tests_serial.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06g_smoke.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06g_bench.log 2>&1
tail -1 gpurun_out/r06g_bench.log > gpurun_out/r06g_bench_line.json
