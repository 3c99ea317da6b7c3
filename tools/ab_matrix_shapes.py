#!/usr/bin/env python3
"""Interleaved same-box A/B of the pipelined kernel's shapes (GEMX_PIPE_SHAPE) over the matrix cases.

    python tools/ab_matrix_shapes.py --envs 32768 65536 131072 [--only "PMSM cont" ...] [--reps 3] > profiles/<round>_shapes_ab.md

Run-to-run spread on one box is +-3-5 %, as large as most differences between shapes, so every (case, envs) cell is measured `--reps`
times with the shapes INTERLEAVED (auto, <12,3>, <12,6>, <4,2>, <2,2>, auto, ...) and the median per shape is reported; `best` names
the fastest forced shape and its margin over what the launcher picks on its own.  Timing as tools/bench_matrix.py (regions of >= 4 ms,
wall clock)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SHAPES = [("auto", None), ("<12,3>", "0"), ("<12,6>", "3"), ("<4,2>", "1"), ("<2,2>", "2")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[131072])
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--solver", choices=["rk4", "default"], default="default")
    args = ap.parse_args()
    import numpy as np
    import torch

    import bench_matrix as bm
    import gym_electric_motor_amd as ga

    K = args.steps
    print("| case | envs | " + " | ".join(s for s, _ in SHAPES) + " | auto picks | best forced (vs auto) |")
    print("|---|---|" + "---|" * (len(SHAPES) + 2))
    for label, env_id, kw, with_reward in bm.CASES:
        if args.only and not any(o in label for o in args.only):
            continue
        for n in args.envs:
            res = {s: [] for s, _ in SHAPES}
            picked = ""
            for _ in range(args.reps):
                for sname, sval in SHAPES:
                    if sval is None:
                        os.environ.pop("GEMX_PIPE_SHAPE", None)
                    else:
                        os.environ["GEMX_PIPE_SHAPE"] = sval
                    kw2 = dict(kw)
                    if kw2.pop("rc", False):
                        kw2["supply"] = ga.RCVoltageSupply(u_nominal=420.0, supply_parameter=dict(R=0.5, C=2e-3))
                    ws = []
                    for wname in kw2.pop("wrappers", ()):
                        ws.append(ga.DeadTimeProcessor(int(wname[4:])) if wname.startswith("dead") else ga.DqToAbcActionProcessor.make("PMSM"))
                    if args.solver == "rk4":
                        kw2["ode_solver"] = ga.RK4Solver()
                    env = ga.make(env_id, n_envs=n, physical_system_wrappers=tuple(ws), **kw2)
                    ps = env.physical_system
                    g = torch.Generator(device="cuda").manual_seed(1)
                    if ps._discrete:
                        nflat = int(np.prod(ps.action_space.nvec)) if hasattr(ps.action_space, "nvec") else int(ps.action_space.n)
                        acts = torch.randint(0, nflat, (K, n), device="cuda", generator=g, dtype=torch.uint8)
                        b = 1
                    else:
                        acts = torch.rand((K, n, ps._n_act), device="cuda", generator=g) * 2 - 1
                        b = 4 * ps._n_act
                    obs = torch.empty((K, n, ps._n_out), device="cuda")
                    done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
                    b += 4 * ps._n_out + 1
                    refs = rew = None
                    if with_reward:
                        rc = ps.set_reward(**bm.REWARD[env_id])
                        refs = torch.rand((K, n, int(rc.n_ref)), device="cuda", generator=g) - 0.5
                        rew = torch.empty((K, n), device="cuda")
                        b += 4 * int(rc.n_ref) + 4

                    def launch():
                        if with_reward:
                            ps.rollout(acts, obs_out=obs, done_out=done, references=refs, reward_out=rew)
                        else:
                            ps.rollout(acts, obs_out=obs, done_out=done)

                    t_end = time.perf_counter() + 0.03
                    while time.perf_counter() < t_end:
                        for _ in range(4):
                            launch()
                        torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        launch()
                    torch.cuda.synchronize()
                    nl = max(5, int(4e-3 / max((time.perf_counter() - t0) / 3, 1e-6)) + 1)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(nl):
                        launch()
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / nl
                    res[sname].append(n * K / dt * b / 8e12)
                    if sval is None:
                        picked = ps.last_launch().split("f32,")[1].split(">")[0] if "f32," in ps.last_launch() else ps.last_launch()[:40]
                    env.close()
            med = {s: sorted(v)[len(v) // 2] for s, v in res.items()}
            forced = {s: m for s, m in med.items() if s != "auto"}
            best = max(forced, key=forced.get)
            print(f"| {label} | {n} | " + " | ".join(f"{med[s]:.3f}" for s, _ in SHAPES) + f" | {picked} | {best} ({forced[best] - med['auto']:+.3f}) |", flush=True)
    os.environ.pop("GEMX_PIPE_SHAPE", None)


if __name__ == "__main__":
    main()
