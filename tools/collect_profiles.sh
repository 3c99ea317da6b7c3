#!/bin/bash
# Collects the judged artefacts of one round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02a [quick]  ->  gpurun_out/<tag>_*
# bench.py: one bench step = one fused launch of 1000 control steps; default 20 timed launches after 5.
set -u
TAG=${1:-rXX}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench.log 2>&1   # exactly what the driver runs (HBM traffic measured in-run by its own --pmc child passes)
tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench.json          # the compact stdout line (round 6: < 6 KB, printed last)
cp $R/bench_extras.json $OUT/${TAG}_bench_extras.json 2>/dev/null  # ... and the full record beside it
# (the runs under rocprofv3 below pass --no-pmc: no profiler inside the profiler)
ksub() { [ "$1" = permexdc ] && echo dc_stream || echo advance; }  # the dominant kernel of a workload (config 2: dc_stream_kernel)
for WL in pmsm permexdc scim scim_constspeed; do
  rm -rf /tmp/ks_$WL
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$WL -- python $R/bench.py --no-extras --no-pmc --workload $WL > /tmp/ks_$WL.log 2>&1
  grep "^{\"metric\"" /tmp/ks_$WL.log | tail -1 > $OUT/${TAG}_bench_under_rocprof_$WL.json
  cp $(find /tmp/ks_$WL -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats_$WL.csv
  # (the stats file averages over the settle and warm-up launches too: the timed region = the last 20 dispatches of the trace)
  python $R/tools/trace_tail_stats.py /tmp/ks_$WL $(ksub $WL) 20 > $OUT/${TAG}_bench_kernel_timed_region_$WL.txt 2>&1
done
# a SHORT launch (20 control steps per launch): kernel-trace durations vs the HIP-event figure
rm -rf /tmp/ks_short
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_short -- python $R/bench.py --no-extras --no-pmc --repeats 1 --steps-per-launch 20 --steps 200 --warmup 20 > /tmp/ks_short.log 2>&1
grep "^{\"metric\"" /tmp/ks_short.log | tail -1 > $OUT/${TAG}_bench_under_rocprof_short20.json
cp $(find /tmp/ks_short -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats_short20.csv
[ "$QUICK" = "quick" ] && exit 0
for WL in pmsm permexdc scim scim_constspeed; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --no-extras --no-pmc --repeats 1 --workload $WL --steps 5 --warmup 2 --settle-ms 0 > /tmp/pmc_$c.log 2>&1
    echo "$WL $c $(python $R/tools/pmc_sum.py /tmp/pmc_$c $(ksub $WL))" >> $OUT/${TAG}_pmc_raw.txt
  done
done
for WL in pmsm scim permexdc; do
  rm -rf /tmp/pmc_sq
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_sq -- python $R/bench.py --no-extras --no-pmc --repeats 1 --workload $WL --steps 5 --warmup 2 --settle-ms 0 > /tmp/pmc_sq.log 2>&1
  python $R/tools/pmc_sum.py /tmp/pmc_sq $(ksub $WL) | sed "s/^/$WL SQ /" >> $OUT/${TAG}_pmc_raw.txt
done
python $R/tools/bench_matrix.py > $OUT/${TAG}_matrix.md 2>/dev/null                          # plain RK4 on every row (rounds 1-3's matrix)
python $R/tools/bench_matrix.py --solver default > $OUT/${TAG}_matrix_default_solver.md 2>/dev/null  # what make(env_id) hands out (kink correction on the SC rows)
# round 4: the large batches limiter off -> on (same box, interleaved), and LONG launches at one workgroup per CU (paced <12, 3> from 1500 steps on)
bash $R/tools/ab_rate_limiter.sh > $OUT/${TAG}_pace_ab.txt 2>&1
python $R/tools/ab_rate_limiter_table.py $OUT/${TAG}_pace_ab.txt > $OUT/${TAG}_pace_ab.md 2>&1
for K in 1000 3000 6000; do
  python $R/tools/bench_matrix.py --envs 16384 --steps $K --only "PMSM finite (headline)" "SynRM" "PMSM cont" 2>/dev/null | grep -v "^| case\|^|---" | sed "s/^| /| $K steps per launch: /" >> $OUT/${TAG}_matrix_long.md
done
# the records that go with them: the GPU suite, the parity report, two ranks on this one GPU (gloo control plane) with the chunk gather
cd $R
rm -f $OUT/${TAG}_cov_tests.txt
GEMX_COVERAGE_FILE=$OUT/${TAG}_cov_tests.txt python -m pytest tests -m gpu -q -n 4 2>&1 | tail -3 > $OUT/${TAG}_gpu_tests.txt   # (+ the instantiation-coverage record)
python tools/latency_bound_rows.py > $OUT/${TAG}_latency_bound_rows.md 2>/dev/null   # needs variants/lat_a, lat_b (python tools/latency_bound_rows.py --build)
python tools/bench_matrix.py --solver default --chunks 4 --only "PMSM cont" "EESM cont" "DFIM cont" "ExtExDc cont" > $OUT/${TAG}_matrix_actions_from_hbm.md 2>/dev/null
python tools/bench_matrix.py --solver default --chunks 4 --half-actions --only "PMSM cont" "EESM cont" "DFIM cont" "ExtExDc cont" > $OUT/${TAG}_matrix_half_actions_from_hbm.md 2>/dev/null
# (round 5: both scripts exit non-zero on any failed comparison, and the collection records it -- a FAIL cell used to be swallowed)
python tests/parity_report.py > $OUT/${TAG}_parity.md 2>&1; echo "parity_report.py exit status $?" >> $OUT/${TAG}_gpu_tests.txt
python tests/solver_scan.py > $OUT/${TAG}_solver_scan.md 2>/dev/null; echo "solver_scan.py exit status $?" >> $OUT/${TAG}_gpu_tests.txt
python bench.py --gpus 2 --oversubscribe --gather chunk --steps 5 --warmup 2 --no-extras --no-pmc 2>&1 | tail -1 > $OUT/${TAG}_bench_gpus2_oversubscribe.json
