#!/usr/bin/env python3
"""Closed-loop use of the batched stepper: N current-controlled PMSM drives, device-side Wiener references, fused reward, a
trivial proportional dq controller as the "policy" -- everything stays on the GPU, one launch per control step.

    python examples/closed_loop.py [--envs 16384] [--steps 2000]

The same loop with the reference package is `env.step(policy(obs))` on ONE env per Python call (reference: core.py:329-372).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=2000)
    args = ap.parse_args()
    import torch

    import gym_electric_motor_amd as ga

    n = args.envs
    # Cont-CC-PMSM-v0 with (u_d, u_q) actions: the reference's DqToAbcActionProcessor folded into the kernel
    env = ga.make("Cont-CC-PMSM-v0", n_envs=n, ode_solver=ga.RK4Solver(), physical_system_wrappers=(ga.DqToAbcActionProcessor.make("PMSM"),))
    ps = env.physical_system
    gen = ga.BatchedWienerProcessReferenceGenerator(reference_states=("i_sd", "i_sq"), seed=1).set_modules(ps)
    ps.set_reward(reward_weights=dict(i_sd=0.5, i_sq=0.5), referenced_states=gen.reference_names)
    isd, isq = ps.state_positions["i_sd"], ps.state_positions["i_sq"]
    obs, _ = env.reset()
    gen.reset()
    ref = gen.rollout(1)[0]  # the reference the agent sees before acting
    ret = torch.zeros(n, device="cuda")
    n_done = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        err = ref - obs[:, [isd, isq]]
        action = (8.0 * err).clamp(-1, 1)                    # "policy": proportional current controller in dq
        obs, reward, terminated, _, _ = env.step(action, references=ref)
        ret += reward
        n_done += int(terminated.sum()) if k % 200 == 199 else 0
        gen.apply_done(terminated)                           # terminated envs get a fresh reference process (auto-reset envs)
        ref = gen.rollout(1)[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n} envs x {args.steps} closed-loop steps in {dt:.2f} s = {n * args.steps / dt / 1e6:.1f} M env-steps/s; "
          f"mean return {float(ret.mean()):.2f}; kernel: {ps.last_launch().split(' grid')[0]}")
    assert torch.isfinite(ret).all()
    env.close()
    gen.close()


if __name__ == "__main__":
    main()
