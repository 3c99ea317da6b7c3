#!/bin/bash
# Same-box A/B of the production library against gym_electric_motor_amd/libgemx_timing.so (an experimental build of some units) on
# the bench workloads: bash tools/ab_library.sh [workloads...]
for i in 1 2; do
for lib in "" "$PWD/gym_electric_motor_amd/libgemx_timing.so"; do
  export GEMX_LIBRARY=$lib; [ -z "$lib" ] && unset GEMX_LIBRARY
  echo "== ${lib:-production}"
  for wl in ${@:-pmsm scim permexdc}; do
    python bench.py --no-extras --workload $wl --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$wl', j['config']['envs_per_gpu'], round(j['value']/1e9,1), round(j['roofline']['frac'],3), j['roofline']['kernel'].split(' grid')[0][-14:])"
  done
done; done
