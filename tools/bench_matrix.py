#!/usr/bin/env python3
"""Throughput matrix over the widened scope: every motor family, the action stage and the fused reward.

    python tools/bench_matrix.py [--envs 16384 131072] [--steps 500] > profiles/<round>_matrix.md

For each case: fused rollouts of `--steps` control steps (observation rows + done bytes written every step, default
constraints + auto-reset, RK4, fp32, uniformly random actions resident in HBM) timed the way bench.py times its legs (round 4: the
rows used to be 5 launches behind a 60-ms settle and were not comparable with the bench legs quoted beside them): 60 ms of the same
launches untimed (clock governor), 2 warm-up launches, then THREE timed regions of >= 4 ms of launches each, bracketed by
synchronize on both sides and priced by the WALL clock; the MEDIAN region is the row, `min..max` of the three beside it.
Algorithmic bytes per env-step = action + 4 * S_out + 1 (+ 4 * n_ref + 4 with the fused reward).
`--solver default` takes the solver make(env_id) hands out (RK4, kink-corrected where the id's load has kinks) instead of plain RK4;
`--shape S` forces GEMX_PIPE_SHAPE (A/B of the pipelined kernel's shapes over every family); `--only SUBSTR` selects rows."""
import argparse
import time
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [
    # label, env_id, make kwargs, reward?
    ("PermExDc cont", "Cont-CC-PermExDc-v0", {}, False),
    ("SeriesDc cont SC", "Cont-SC-SeriesDc-v0", {}, False),
    ("ShuntDc finite", "Finite-CC-ShuntDc-v0", {}, False),
    ("ExtExDc cont (2x4QC)", "Cont-CC-ExtExDc-v0", {}, False),
    ("ExtExDc finite (2x4QC)", "Finite-CC-ExtExDc-v0", {}, False),
    ("PMSM finite (headline)", "Finite-CC-PMSM-v0", {}, False),
    ("PMSM cont", "Cont-CC-PMSM-v0", {}, False),
    ("PMSM cont SC (poly load)", "Cont-SC-PMSM-v0", {}, False),
    ("SynRM finite", "Finite-CC-SynRM-v0", {}, False),
    ("EESM cont (B6+4QC)", "Cont-CC-EESM-v0", {}, False),
    ("EESM finite (B6+4QC)", "Finite-CC-EESM-v0", {}, False),
    ("SCIM cont SC", "Cont-SC-SCIM-v0", {}, False),
    ("SCIM finite", "Finite-CC-SCIM-v0", {}, False),
    ("DFIM cont (2xB6)", "Cont-CC-DFIM-v0", {}, False),
    ("DFIM finite (2xB6)", "Finite-CC-DFIM-v0", {}, False),
    ("PMSM finite + dead time 1us", "Finite-CC-PMSM-v0", {"converter": dict(interlocking_time=1e-6)}, False),
    ("PMSM cont control_space=dq", "Cont-CC-PMSM-v0", {"control_space": "dq"}, False),
    ("PMSM cont DqToAbc + DeadTime(1)", "Cont-CC-PMSM-v0", {"wrappers": ("dead1", "dq")}, False),
    ("PMSM finite DeadTime(2)", "Finite-CC-PMSM-v0", {"wrappers": ("dead2",)}, False),
    ("PMSM finite + RC supply", "Finite-CC-PMSM-v0", {"rc": True}, False),
    ("PMSM finite + random initial states", "Finite-CC-PMSM-v0", {"rinit": "PMSM"}, False),  # (round 5: prepared draws, RINIT instantiation)
    ("PMSM cont SC + random initial states", "Cont-SC-PMSM-v0", {"rinit": "PMSM", "rinit_load": True}, False),
    ("SCIM cont + random initial states", "Cont-CC-SCIM-v0", {"rinit": "SCIM"}, False),
    ("PMSM finite + fused reward", "Finite-CC-PMSM-v0", {}, True),
    ("SCIM cont SC + fused reward", "Cont-SC-SCIM-v0", {}, True),
]
REWARD = {"Finite-CC-PMSM-v0": dict(reward_weights=dict(i_sd=0.5, i_sq=0.5), referenced_states=("i_sd", "i_sq")),
          "Cont-SC-SCIM-v0": dict(reward_weights=dict(omega=1.0), referenced_states=("omega",))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[16384, 131072])
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--solver", choices=["rk4", "default"], default="rk4")
    ap.add_argument("--shape", default=None, help="GEMX_PIPE_SHAPE for every row (0: <12,3>, 1: <4,2>, 2: <2,2>, 3: <12,6>)")
    ap.add_argument("--only", nargs="*", default=None, help="substrings: run the rows whose label contains one of them")
    ap.add_argument("--device-actions", action="store_true", help="actions generated on the device (rollout_synthetic): no action tensor, B/env-step without the action bytes")
    ap.add_argument("--chunks", type=int, default=1, help="rotating action tensors (C x K steps): with C x K x N x bytes beyond the 256 MB Infinity Cache the "
                                                          "actions of every launch come from the HBM, as behind a policy (one re-read chunk can sit in the cache)")
    ap.add_argument("--half-actions", action="store_true", help="continuous rows: the action tensor as float16 (gemx_rollout_half): 2 instead of 4 bytes per duty cycle")
    args = ap.parse_args()
    if args.shape is not None:
        os.environ["GEMX_PIPE_SHAPE"] = args.shape
    import numpy as np
    import torch

    import gym_electric_motor_amd as ga

    K = args.steps
    print(f"| case | envs | G env-steps/s | B/env-step | GB/s (algorithmic) | frac of 8 TB/s | min..max of 3 regions | kernel |")
    print("|---|---|---|---|---|---|---|---|")
    for label, env_id, kw, with_reward in CASES:
        if args.only and not any(o in label for o in args.only):
            continue
        for n in args.envs:
            kw2 = dict(kw)
            if kw2.pop("rc", False):
                kw2["supply"] = ga.RCVoltageSupply(u_nominal=420.0, supply_parameter=dict(R=0.5, C=2e-3))
            rinit = kw2.pop("rinit", None)
            if rinit:
                kw2["motor"] = {"PMSM": ga.PermanentMagnetSynchronousMotor, "SCIM": ga.SquirrelCageInductionMotor}[rinit](motor_initializer=dict(random_init="uniform"))
                kw2["seed"] = 3
                if kw2.pop("rinit_load", False):
                    kw2["load"] = ga.PolynomialStaticLoad(load_initializer=dict(random_init="uniform"))
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
                a_bytes = 1
            else:
                acts = torch.rand((K, n, ps._n_act), device="cuda", generator=g) * 2 - 1
                a_bytes = 4 * ps._n_act
                if args.half_actions and not with_reward:
                    acts = acts.to(torch.float16)
                    a_bytes = 2 * ps._n_act
            obs = torch.empty((K, n, ps._n_out), device="cuda")
            done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
            refs = rew = None
            b = a_bytes + 4 * ps._n_out + 1
            if with_reward:
                rc = ps.set_reward(**REWARD[env_id])
                refs = torch.rand((K, n, int(rc.n_ref)), device="cuda", generator=g) - 0.5
                rew = torch.empty((K, n), device="cuda")
                b += 4 * int(rc.n_ref) + 4

            if args.device_actions and not with_reward:
                b -= a_bytes

            chunks = [acts] + [acts.clone() for _ in range(max(1, args.chunks) - 1)]
            turn = [0]

            def launch():
                turn[0] += 1
                acts = chunks[turn[0] % len(chunks)]
                if args.device_actions and not with_reward:
                    ps.rollout_synthetic(K, seed=1, step0=0, obs_out=obs, done_out=done)
                elif with_reward:
                    ps.rollout(acts, obs_out=obs, done_out=done, references=refs, reward_out=rew)
                else:
                    ps.rollout(acts, obs_out=obs, done_out=done)

            t_end = time.perf_counter() + 0.06  # clock governor: ~20 ms of load until the sustained clock (bench.py docstring)
            while time.perf_counter() < t_end:
                for _ in range(4):
                    launch()
                torch.cuda.synchronize()
            for _ in range(50):  # (the closed-loop rate limiter calibrates during the first paced launches: wait for it, bounded)
                if "limiter calibrating" not in ps.last_launch():
                    break
                for _ in range(8):
                    launch()
                torch.cuda.synchronize()
            for _ in range(2):
                launch()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                launch()
            e1.record()
            torch.cuda.synchronize()
            nl = max(5, int(4.0 / max(e0.elapsed_time(e1) / 3, 1e-3)) + 1)  # launches per timed region: >= 4 ms
            walls = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(nl):
                    launch()
                torch.cuda.synchronize()
                walls.append((time.perf_counter() - t0) / nl)
            ws_ = sorted(walls)
            rate = n * K / ws_[1]
            gbs = rate * b / 1e9
            lohi = f"{n * K / ws_[2] * b / 8e12:.3f}..{n * K / ws_[0] * b / 8e12:.3f}"
            kern = ps.last_launch().split(" grid")[0].replace("gemx::", "")
            print(f"| {label} | {n} | {rate / 1e9:.1f} | {b} | {gbs:.0f} | {gbs / 8000:.3f} | {lohi} | `{kern}` |", flush=True)
            assert torch.isfinite(obs).all()
            env.close()


if __name__ == "__main__":
    main()
