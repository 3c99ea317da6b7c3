#!/usr/bin/env python3
"""Do the action reads of a launch come out of the 256 MB Infinity Cache?  The headline kernel at 1M envs x 100 steps (105 MB of uint8 actions per
launch, 1.7 % of its bytes) with 1 / 2 / 4 / 8 action chunks in rotation: python tools/probe_action_chunks.py > profiles/<round>_action_chunks.txt"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import gym_electric_motor_amd as ga
n, K = 1 << 20, 100
env = ga.make("Finite-CC-PMSM-v0", n_envs=n, ode_solver=ga.RK4Solver(), tau=1e-4)
ps = env.physical_system
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
obs = torch.empty((K, n, ps._n_out), device="cuda"); done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
for nb in (1, 2, 4, 8, 1, 4):
    acts = torch.randint(0, 8, (nb * K, n), device="cuda", generator=g, dtype=torch.uint8)
    bound = [env.bind_rollout(acts[j * K:(j + 1) * K], obs, done) for j in range(nb)]
    for i in range(16): bound[i % nb]()
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(16): bound[i % nb]()
        torch.cuda.synchronize()
        res.append(n * K * 58 / ((time.perf_counter() - t0) / 16) / 8e12)
    print(f"{nb} action chunks of {n * K / 1e6:.0f} MB in rotation: {sorted(res)[1]:.3f} of the roofline ({min(res):.3f}..{max(res):.3f})", flush=True)
    del bound, acts
