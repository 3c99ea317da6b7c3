#!/usr/bin/env python3
"""Open-loop use: K control steps of N envs in ONE launch (data generation, shooting methods, policy evaluation with a fixed
action tape), with the reward evaluated in the same launch.

    python examples/fused_rollout.py [--env-id Finite-CC-PMSM-v0] [--envs 131072] [--steps 500]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env-id", default="Finite-CC-PMSM-v0")
    ap.add_argument("--envs", type=int, default=131072)
    ap.add_argument("--steps", type=int, default=500)
    args = ap.parse_args()
    import numpy as np
    import torch

    import gym_electric_motor_amd as ga

    env = ga.make(args.env_id, n_envs=args.envs, ode_solver=ga.RK4Solver())
    ps = env.physical_system
    K, n = args.steps, args.envs
    if ps._discrete:
        nflat = int(np.prod(ps.action_space.nvec)) if hasattr(ps.action_space, "nvec") else int(ps.action_space.n)
        actions = torch.randint(0, nflat, (K, n), dtype=torch.uint8, device="cuda")
    else:
        actions = torch.rand((K, n, ps._n_act), device="cuda") * 2 - 1
    referenced = [s for s in ("i_sd", "i_sq", "i", "i_a") if s in ps.state_positions][:2]
    gen = ga.BatchedWienerProcessReferenceGenerator(reference_states=referenced, seed=0).set_modules(ps)
    ps.set_reward(referenced_states=gen.reference_names)  # equal weights over the referenced states (the reference's default)
    gen.reset()
    refs = gen.rollout(K)
    env.reset()
    obs, done, reward = env.rollout(actions, references=refs)  # warm-up launch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    obs, done, reward = env.rollout(actions, references=refs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.env_id}: {n} envs x {K} steps in {dt * 1e3:.2f} ms = {n * K / dt / 1e9:.1f} G env-steps/s; obs {tuple(obs.shape)}, "
          f"terminations {int(done.sum())}, mean reward {float(reward.mean()):.4f}; kernel: {ps.last_launch().split(' grid')[0]}")
    env.close()
    gen.close()


if __name__ == "__main__":
    main()
