R=$PWD
for V in q1 r2; do
echo "##### $V"; GEMX_UNIT_DIR=$R/variants/$V python tools/bench_matrix.py --solver default --only "random initial" 2>/dev/null | grep -v "^| case\|^|---" | cut -c1-100
done
echo "##### product (registers)"; python tools/bench_matrix.py --solver default --only "random initial" 2>/dev/null | grep -v "^| case\|^|---" | cut -c1-100
GEMX_UNIT_DIR=$R/variants/r2 python -m pytest tests/test_gpu_parity.py -q -x -k "prepared_draws or two_half_size or random_init or random_uniform or induction_machine" 2>&1 | tail -3
