#!/bin/bash
# Collects the judged artefacts of one round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r01f   ->  gpurun_out/<tag>_*
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/${TAG}_bench.log 2>&1
tail -1 $OUT/${TAG}_bench.log > $OUT/${TAG}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --no-extras > /tmp/ks.log 2>&1
grep "^{\"metric\"" /tmp/ks.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --no-extras --steps 3000 --warmup 1000 > /tmp/pmc_$c.log 2>&1
  python $R/tools/pmc_sum.py /tmp/pmc_$c advance >> $OUT/${TAG}_pmc_raw.txt
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_sq -- python $R/bench.py --no-extras --steps 3000 --warmup 1000 > /tmp/pmc_sq.log 2>&1
python $R/tools/pmc_sum.py /tmp/pmc_sq advance >> $OUT/${TAG}_pmc_raw.txt
python $R/tools/bench_matrix.py > $OUT/${TAG}_matrix.md 2>/dev/null
