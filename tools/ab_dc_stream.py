#!/usr/bin/env python3
"""A/B of dc_stream_kernel against the pipelined kernel on one box: GEMX_DC_STREAM=0 / 2 over batch sizes and solvers (one process per
setting: the switch is read when the handle is created).   python tools/ab_dc_stream.py [env_id]  -> markdown rows"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
import gym_electric_motor_amd as ga
env_id, n, solver, K = sys.argv[1], int(sys.argv[2]), sys.argv[3], 1000
sol = dict(euler=ga.EulerSolver, rk4=ga.RK4Solver, dp5=ga.DormandPrince5Solver)[solver]()
env = ga.make(env_id, n_envs=n, device="cuda:0", ode_solver=sol, tau=1e-4, load=ga.ConstantSpeedLoad(omega_fixed=60.0))
ps = env.physical_system
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 4, (K, n), device="cuda", generator=g, dtype=torch.uint8) if ps._discrete else torch.rand((K, n, ps._n_act), device="cuda", generator=g) * 2 - 1
obs = torch.empty((K, n, ps._n_out), device="cuda"); done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
t_end = time.perf_counter() + 0.06
while time.perf_counter() < t_end:
    for _ in range(4): ps.rollout(acts, obs_out=obs, done_out=done)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ps.rollout(acts, obs_out=obs, done_out=done)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
b = (1 if ps._discrete else 4 * ps._n_act) + 4 * ps._n_out + 1
print(f"{us:.1f} {n * K / us / 1e3:.1f} {n * K * b / us / 1e6 / 8000:.3f} {ps.last_launch().split('<')[0].replace('gemx::', '')}")
''' % REPO

env_id = sys.argv[1] if len(sys.argv) > 1 else "Cont-CC-PermExDc-v0"
print("| env | envs | solver | GEMX_DC_STREAM | us per 1000 steps | G env-steps/s | of 8 TB/s | kernel |")
print("|---|---|---|---|---|---|---|---|")
for solver in ("euler", "rk4"):
    for n in (1024, 4096, 8192, 12288, 16384, 32768):
        for sw in ("0", "2"):
            env = dict(os.environ, GEMX_DC_STREAM=sw)
            r = subprocess.run([sys.executable, "-c", CHILD, env_id, str(n), solver], env=env, capture_output=True, text=True)
            out = [l for l in r.stdout.splitlines() if l and l[0].isdigit()]
            if not out:
                print(f"| {env_id} | {n} | {solver} | {sw} | failed: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ''} |")
                continue
            us, rate, frac, kern = out[-1].split()
            print(f"| {env_id} | {n} | {solver} | {sw} | {us} | {rate} | {frac} | {kern} |", flush=True)
