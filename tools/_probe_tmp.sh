python -m pytest tests -m gpu -q -n 4 2>&1 | tail -12 > gpurun_out/r06f_gpu_tests.txt
for H in "" --half-actions; do
echo "##### actions: ${H:-fp32}"
python tools/bench_matrix.py --solver default --envs 65536 131072 $H --only "PMSM cont" "EESM cont" "DFIM cont" "SCIM cont SC" "ExtExDc cont" "PermExDc cont" 2>/dev/null | grep -v "^| case\|^|---\|random\|reward" | cut -c1-105
done
