#!/usr/bin/env python3
"""Is the launch rate a function of the steps per launch or of the BYTES the launches touch?  The same rollout (one env id, one batch size)
written to 1 / 2 / 4 / 8 output buffers in rotation, at several steps per launch:
    python tools/probe_footprint.py [env id] [envs]      -> markdown rows: K, buffers, output footprint [GB], of the 8 TB/s
Round 4: at 16384 envs the headline holds 0.82 up to ~1 GB of output footprint and falls to 0.72 / 0.68 at 1.9 / 5.6 GB -- whether the
bytes belong to one launch of 2000 steps or to two buffers of 1000 steps each (profiles/r04q_footprint.md)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gym_electric_motor_amd as ga  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "Finite-CC-PMSM-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
print("| env | envs | steps per launch | output buffers in rotation | output footprint [GB] | of 8 TB/s | kernel |")
print("|---|---|---|---|---|---|---|")
for K, nbufs in ((250, 1), (250, 4), (250, 8), (500, 1), (500, 2), (500, 4), (1000, 1), (1000, 2), (1000, 4), (2000, 1), (2000, 2), (4000, 1)):
    env = ga.make(env_id, n_envs=n)
    ps = env.physical_system
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = (torch.randint(0, int(ps.action_space.n), (K, n), device="cuda", generator=g, dtype=torch.uint8) if ps._discrete
            else torch.rand((K, n, ps._n_act), device="cuda", generator=g) * 2 - 1)
    obs = [torch.empty((K, n, ps._n_out), device="cuda") for _ in range(nbufs)]
    done = [torch.empty((K, n), dtype=torch.uint8, device="cuda") for _ in range(nbufs)]
    b = (1 if ps._discrete else 4 * ps._n_act) + 4 * ps._n_out + 1
    i = 0

    def launch():
        global i
        ps.rollout(acts, obs_out=obs[i % nbufs], done_out=done[i % nbufs])
        i += 1

    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        for _ in range(4):
            launch()
        torch.cuda.synchronize()
    res = []
    for _ in range(3):
        nl = max(8, int(8000 / K)) // nbufs * nbufs
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nl):
            launch()
        torch.cuda.synchronize()
        res.append(n * K * b / ((time.perf_counter() - t0) / nl) / 8e12)
    res.sort()
    print(f"| {env_id} | {n} | {K} | {nbufs} | {nbufs * K * n * (4 * ps._n_out + 1) / 1e9:.2f} | {res[1]:.3f} | `{ps.last_launch().split(' grid')[0].replace('gemx::', '')}` |", flush=True)
    env.close()
    del obs, done, acts
    torch.cuda.empty_cache()
