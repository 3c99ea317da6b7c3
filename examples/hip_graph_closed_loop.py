#!/usr/bin/env python3
"""Policy in the loop, captured in a HIP graph: S control steps of (policy -> one gemx_step launch) are recorded once with
torch.cuda.CUDAGraph (hipGraph on ROCm) and replayed, which removes the per-step Python and launch overhead of the closed loop.

    python examples/hip_graph_closed_loop.py [--envs 16384] [--steps 4096] [--capture 64]

`simulate()` enqueues exactly one kernel on the current stream and never synchronises, so it can be captured as it is (no
DeadTimeProcessor here: its queue position is a host-side launch argument).  The state lives in the handle's device buffers, so
replays continue the simulation; observations go to the system's internal buffer, which the policy reads in place.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--capture", type=int, default=64, help="control steps per graph")
    args = ap.parse_args()
    import torch

    import gym_electric_motor_amd as ga

    n, S = args.envs, args.capture
    env = ga.make("Cont-CC-PMSM-v0", n_envs=n, ode_solver=ga.RK4Solver(), physical_system_wrappers=(ga.DqToAbcActionProcessor.make("PMSM"),))
    ps = env.physical_system
    isd, isq = ps.state_positions["i_sd"], ps.state_positions["i_sq"]
    cols = torch.tensor([isd, isq], device="cuda")            # (device-resident index: no host -> device copy inside the capture)
    target = torch.tensor([0.0, 0.3], device="cuda")         # constant dq current reference (normalised)
    gain = torch.tensor(8.0, device="cuda")
    action = torch.zeros((n, 2), device="cuda")               # static buffers: graph replays reuse these addresses
    cost = torch.zeros(n, device="cuda")

    def control_step(obs):
        err = target - obs.index_select(1, cols)
        torch.clamp(gain * err, -1, 1, out=action)            # "policy": proportional current controller in dq
        cost.add_((err * err).sum(dim=1))
        return ps.simulate(action)                            # ONE kernel launch, observations in ps's internal buffer

    obs_buf, _ = env.reset()                                  # the system's internal observation buffer [N, S_out]: every step rewrites it

    def run_eager(steps):
        obs = obs_buf
        for _ in range(steps):
            obs = control_step(obs)

    run_eager(8)                                              # warm-up (also builds the one-step map of the handle)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_eager(args.steps)
    torch.cuda.synchronize()
    t_eager = time.perf_counter() - t0
    cost_eager = float(cost.mean()) / (args.steps + 8)

    env.reset()
    cost.zero_()
    run_eager(8)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run_eager(3)                                          # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    env.reset()
    cost.zero_()
    run_eager(8)
    with torch.cuda.graph(graph):
        run_eager(S)
    torch.cuda.synchronize()
    # the capture itself executed nothing: replay ceil(steps / S) times from the state after the 8 warm-up steps
    reps = args.steps // S
    t0 = time.perf_counter()
    for _ in range(reps):
        graph.replay()
    torch.cuda.synchronize()
    t_graph = time.perf_counter() - t0
    print(f"{n} envs, {args.steps} closed-loop steps: eager {n * args.steps / t_eager / 1e6:.1f} M env-steps/s ({t_eager / args.steps * 1e6:.1f} us/step), "
          f"HIP graph of {S} steps {n * reps * S / t_graph / 1e6:.1f} M env-steps/s ({t_graph / (reps * S) * 1e6:.1f} us/step); "
          f"mean cost per step eager {cost_eager:.6f}, graph {float(cost.mean()) / (reps * S + 8):.6f}")
    assert torch.isfinite(cost).all()
    env.close()


if __name__ == "__main__":
    main()
