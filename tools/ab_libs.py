#!/usr/bin/env python3
"""Same-box A/B of libgemx.so builds (tools/dev_build.py variants) on one env id over several batch sizes:
    python tools/ab_libs.py <env id> <solver> <envs,envs,...> [variants/libgemx_x.so ...]   -> markdown rows (product library first).
One child process per (library, size): 60 ms of settling launches, then the median of 3 x 10 launches of 1000 steps (HIP events)."""
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
if solver == "default":  # what make(env_id) hands out (RK4 + kink correction behind a PolynomialStaticLoad)
    env = ga.make(env_id, n_envs=n, device="cuda:0")
else:
    sol = dict(euler=ga.EulerSolver, rk4=ga.RK4Solver, dp5=ga.DormandPrince5Solver)[solver]()
    env = ga.make(env_id, n_envs=n, device="cuda:0", ode_solver=sol, tau=1e-4)
ps = env.physical_system
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.randint(0, 4, (K, n), device="cuda", generator=g, dtype=torch.uint8) if ps._discrete else torch.rand((K, n, ps._n_act), device="cuda", generator=g) * 2 - 1
obs = torch.empty((K, n, ps._n_out), device="cuda"); done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
res = []
t_end = time.perf_counter() + 0.06
while time.perf_counter() < t_end:
    for _ in range(4): ps.rollout(acts, obs_out=obs, done_out=done)
    torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ps.rollout(acts, obs_out=obs, done_out=done)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 10 * 1e3)
res.sort()
us = res[1]
b = (1 if ps._discrete else 4 * ps._n_act) + 4 * ps._n_out + 1
print(f"{us:.1f} {res[0]:.1f} {res[2]:.1f} {n * K * b / us / 8e6:.3f} {ps.last_launch().split(' grid')[0].replace('gemx::', '')} {float(obs.double().sum()):.9e}")
''' % REPO

env_id, solver, sizes = sys.argv[1], sys.argv[2], [int(x) for x in sys.argv[3].split(",")]
libs = [None] + sys.argv[4:]
print("| library | env | envs | solver | us per 1000 steps (median of 3 x 10 launches) | min | max | of 8 TB/s | kernel | checksum |")
print("|---|---|---|---|---|---|---|---|---|---|")
for n in sizes:
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["GEMX_LIBRARY"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", CHILD, env_id, str(n), solver], env=env, capture_output=True, text=True)
        out = [l for l in r.stdout.splitlines() if l and l[0].isdigit()]
        tag = os.path.basename(lib) if lib else "libgemx.so (product)"
        if not out:
            print(f"| {tag} | {env_id} | {n} | {solver} | failed: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ''} |")
            continue
        us, lo, hi, frac, kern, chk = out[-1].split()
        print(f"| {tag} | {env_id} | {n} | {solver} | {us} | {lo} | {hi} | {frac} | {kern} | {chk} |", flush=True)
