#!/usr/bin/env python3
"""Small driver for rocprofv3: a few launches of the fused advance kernel on one workload.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python tools/prof_rollout.py --workload pmsm --envs 16384 --chunk 500 --launches 6
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pmsm")
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--chunk", type=int, default=500)
    ap.add_argument("--launches", type=int, default=6)
    ap.add_argument("--spb", type=int, default=0, help="steps per I/O block (0 = heuristic)")
    ap.add_argument("--layout", default="aos")
    ap.add_argument("--no-constraints", action="store_true")
    ap.add_argument("--last-only", action="store_true", help="obs_every=0: no per-step observation traffic (compute only)")
    ap.add_argument("--solver", default=None)
    args = ap.parse_args()
    import torch

    import bench
    import gym_electric_motor_amd as ga

    w = dict(bench.WORKLOADS[args.workload], key=args.workload)
    if args.solver:
        w["solver"] = args.solver
    sol = {"euler": ga.EulerSolver(), "rk4": ga.RK4Solver(), "dp5": ga.DormandPrince5Solver()}[w["solver"]]
    kw = dict(n_envs=args.envs, ode_solver=sol, tau=w["tau"], obs_layout=args.layout)
    if args.no_constraints:
        kw["constraints"] = ()
    env = ga.make(w["env_id"], **kw)
    ps = env.physical_system
    if args.spb:
        ps._L.gemx_set_steps_per_block(ps._handle, args.spb)
    dev = torch.device("cuda", 0)
    acts = bench.make_actions(torch, ps, args.chunk, args.envs, dev, 1)
    shape = (args.chunk, args.envs, ps._n_out) if args.layout == "aos" else (args.chunk, ps._n_out, args.envs)
    obs = torch.empty(shape, dtype=torch.float32, device=dev)
    done = torch.empty((args.chunk, args.envs), dtype=torch.uint8, device=dev)
    if args.last_only:
        obs, done = obs[0], done[0]
    kw2 = dict(last_only=True) if args.last_only else {}
    env.rollout(acts, obs_out=obs, done_out=done, **kw2)
    torch.cuda.synchronize()
    ev = []
    t0 = time.perf_counter()
    for _ in range(args.launches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.rollout(acts, obs_out=obs, done_out=done, **kw2)
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = [a.elapsed_time(b) for a, b in ev]
    b = bench.bytes_per_env_step_fused(w)
    per = sum(ms) / len(ms)
    print(f"{args.workload} N={args.envs} chunk={args.chunk} spb={args.spb} layout={args.layout} solver={w['solver']} last_only={args.last_only}: {per:.4f} ms/launch, "
          f"{per / args.chunk * 1e3:.3f} us/step, {args.envs * args.chunk / per / 1e6:.2f} G env-steps/s, "
          f"{args.envs * args.chunk * b / per / 1e6:.1f} GB/s; wall {dt * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
