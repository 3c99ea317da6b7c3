for V in t_ph1 t_ph2; do
export GEMX_UNIT_DIR=$PWD/variants/$V
echo "##### $V"
python tools/probe_prepared_draws.py 2>&1 | grep -v amdgpu.ids | grep -v "PMSM finite" | awk '{print $1,$2,$3,$4,$5, $(NF-1), $NF}'
for N in 16384 131072; do
echo "=== SCIM cont CC rinit $N"
PROBE_RINIT=SCIM python tools/pipe_timing_probe.py $N 500 Cont-CC-SCIM-v0 2>&1 | grep -E "HIP events|integrator per block|loader" | head -3
PROBE_RINIT=SCIM python tools/pipe_timing_probe.py $N 500 Cont-CC-SCIM-v0 2>&1 | grep -E "phases"
echo "=== PMSM cont SC rinit $N"
PROBE_RINIT=1 python tools/pipe_timing_probe.py $N 500 Cont-SC-PMSM-v0 2>&1 | grep -E "HIP events|integrator per block|loader" | head -3
PROBE_RINIT=1 python tools/pipe_timing_probe.py $N 500 Cont-SC-PMSM-v0 2>&1 | grep -E "phases"
done
done
