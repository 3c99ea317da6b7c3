#!/bin/bash
# Diagnostic build: ONE instantiation unit (system S, converter C, fp32) compiled with -DGEMX_TIMING (per-wave-role clock64 deltas,
# see advance_pipe_kernel), linked with the production objects of every other unit -> gym_electric_motor_amd/libgemx_timing.so.
#   tools/build_timing_lib.sh 0 0        then on the GPU box:  GEMX_LIBRARY=$PWD/gym_electric_motor_amd/libgemx_timing.so python tools/pipe_timing_probe.py ...
set -e
S=${1:-1}; C=${2:-1}; EXTRA=${EXTRA:-}
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/gym_electric_motor_amd
O=/tmp/gemx_timing_${S}_${C}.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fPIC -Wno-unused-variable -I$R/include -I$P/csrc -DGEMX_TIMING $EXTRA \
  -DGEMX_INST_SYS=$S -DGEMX_INST_CONV=$C -DGEMX_INST_F64=0 -c $P/csrc/gemx_inst.hip -o $O
OBJS=$(ls $P/build/*.o | grep -v "gemx_inst_${S}_${C}_0.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libgemx_timing.so $OBJS $O
ls -la $P/libgemx_timing.so
