#!/bin/bash
# Instantiation coverage of everything the repo runs on a GPU (run through gpurun from the repo root):
#   tools/collect_coverage.sh r06a  ->  gpurun_out/<tag>_cov_*.txt (+ the suite's / bench's own records)
# then, in the build container:  python tools/instantiation_coverage.py gpurun_out/<tag>_cov_*.txt > profiles/<round>_instantiation_coverage.md
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
rm -f $OUT/${TAG}_cov_*.txt
cd $R
GEMX_COVERAGE_FILE=$OUT/${TAG}_cov_tests.txt timeout 2400 python -m pytest tests -m gpu -q -n 4 > $OUT/${TAG}_gpu_tests_full.txt 2>&1
tail -15 $OUT/${TAG}_gpu_tests_full.txt > $OUT/${TAG}_gpu_tests.txt
GEMX_COVERAGE_FILE=$OUT/${TAG}_cov_bench.txt timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench.log 2>&1
tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench_line.json
cp bench_extras.json $OUT/${TAG}_bench_extras.json 2>/dev/null
GEMX_COVERAGE_FILE=$OUT/${TAG}_cov_smoke.txt timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.txt 2>&1
GEMX_COVERAGE_FILE=$OUT/${TAG}_cov_matrix.txt timeout 900 python tools/bench_matrix.py --solver default > $OUT/${TAG}_matrix_default_solver.md 2>$OUT/${TAG}_matrix_err.txt
GEMX_COVERAGE_FILE=$OUT/${TAG}_cov_scan.txt timeout 900 python tests/solver_scan.py > $OUT/${TAG}_solver_scan.md 2>/dev/null; echo "solver_scan.py exit status $?" >> $OUT/${TAG}_gpu_tests.txt
GEMX_COVERAGE_FILE=$OUT/${TAG}_cov_parity.txt timeout 900 python tests/parity_report.py > $OUT/${TAG}_parity.md 2>&1; echo "parity_report.py exit status $?" >> $OUT/${TAG}_gpu_tests.txt
ls -la $OUT | tail -20
