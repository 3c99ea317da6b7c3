#!/bin/bash
# Same-box A/B of the large-batch rate limiter (advance_pipe_kernel) over every row of tools/bench_matrix.py at 32768 / 65536 / 131072 envs,
# 1000 steps per launch, interleaved repeats: OFF (GEMX_PACE_GBPS=0), OPEN loop (the built-in targets, GEMX_PACE_CAL=0: round 4's limiter) and
# CLOSED loop (the default since round 5: per-handle calibration starting from those targets).
#   tools/ab_rate_limiter.sh > profiles/<round>_pace_ab.txt        (through gpurun, from the repo root)
cd ${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for pace in 0 open closed; do
  [ $rep = 2 ] && [ $pace = open ] && continue
  echo "== rep $rep pace $pace"
  unset GEMX_PACE_GBPS GEMX_PACE_CAL
  [ $pace = 0 ] && export GEMX_PACE_GBPS=0
  [ $pace = open ] && export GEMX_PACE_CAL=0
  python tools/bench_matrix.py --envs 32768 65536 131072 --steps 1000 --solver default 2>/dev/null | grep -v "^| case\|^|---" | awk -F'|' '{print $2,"|",$3,"|",$7}'
done
done
