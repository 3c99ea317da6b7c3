for H in "" --half-actions; do
echo "##### actions: ${H:-fp32}, 4 rotating chunks"
python tools/bench_matrix.py --solver default --envs 65536 131072 --chunks 4 $H --only "PMSM cont" "EESM cont" "DFIM cont" "SCIM cont SC" "ExtExDc cont" 2>/dev/null | grep -v "^| case\|^|---\|random\|reward" | cut -c1-105
done
