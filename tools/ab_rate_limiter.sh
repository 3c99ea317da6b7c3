#!/bin/bash
# Same-box A/B of the large-batch rate limiter (advance_pipe_kernel; built-in targets) against GEMX_PACE_GBPS=0 over every row of
# tools/bench_matrix.py at 32768 / 65536 / 131072 envs, 1000 steps per launch, two interleaved repeats:
#   tools/ab_rate_limiter.sh > profiles/<round>_pace_ab.txt        (through gpurun, from the repo root)
cd ${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
for pace in 0 default; do
  echo "== rep $rep pace $pace"
  if [ $pace = default ]; then unset GEMX_PACE_GBPS; else export GEMX_PACE_GBPS=$pace; fi
  python tools/bench_matrix.py --envs 32768 65536 131072 --steps 1000 --solver default 2>/dev/null | grep -v "^| case\|^|---" | awk -F'|' '{print $2,"|",$3,"|",$7}'
done
done
