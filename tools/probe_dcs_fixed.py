#!/usr/bin/env python3
"""Where the FIXED part of a dc_stream_kernel launch goes (needs a -DGEMX_TIMING build: tools/dev_build.py --units 0_0_0,capi --defs GEMX_TIMING
--slim; GEMX_LIBRARY=<that build>): wall-clock stamps (100 MHz) of workgroup 0 relative to its waves' kernel entry, beside the launch
time from HIP events, for several launch lengths.
    GEMX_LIBRARY=variants/libgemx_timing.so python tools/probe_dcs_fixed.py [envs] > profiles/<round>_dcs_fixed.txt"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gym_electric_motor_amd as ga  # noqa: E402
from gym_electric_motor_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = _lib.load()
L.gemx_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for K in (64, 250, 1000, 2000):
    env = ga.make("Cont-CC-PermExDc-v0", n_envs=n, ode_solver=ga.EulerSolver())
    ps = env.physical_system
    env.reset()
    act = torch.rand((K, n, 1), device="cuda") * 2 - 1
    obs = torch.empty((K, n, ps._n_out), device="cuda")
    done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
    for _ in range(200):
        ps.rollout(act, obs_out=obs, done_out=done)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        ps.rollout(act, obs_out=obs, done_out=done)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    buf = (C.c_ulonglong * 96)()
    L.gemx_debug_read(ps._handle, buf, 96)
    t = lambda i: buf[i] / 100.0  # noqa: E731  (ticks of 10 ns -> us)
    print(f"K={K:5d}  {us:6.2f} us per launch (HIP events, back to back)   {ps.last_launch().split(' grid')[0]}")
    print(f"   pre wave 0:  loads issued at {t(75):.2f} us, block 0 converted at {t(76):.2f}, done at {t(77):.2f}")
    print(f"   integrator:  at its first barrier at {t(70):.2f} us, last barrier passed at {t(71):.2f}  (loop: {t(71) - t(70):.2f} us = {1e3 * (t(71) - t(70)) / K:.2f} ns per step)")
    print(f"   out wave 0:  first barrier released at {t(73):.2f} us, last store issued at {t(74):.2f}")
    print("   first blocks (barrier b released at [us] / the integrator's own cycles in block b): " + "  ".join(f"{t(80 + b):.2f}/{buf[88 + b]}" for b in range(min(8, (K + 63) // 64))))
    print(f"   => before the loop {t(73):.2f} us, after it {t(74) - t(71):.2f} us, outside the waves' lifetime {us - t(74):.2f} us (dispatch, end-of-kernel release)")
    env.close()
