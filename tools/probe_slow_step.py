#!/usr/bin/env python3
"""Custom constraint sets / solver sub-steps: the pipelined kernel's SLOW instantiation against the single-wave kernel (GEMX_PIPE=0),
same box, same actions.  Prints G env-steps/s of both and whether the results are bit-identical.

    python tools/probe_slow_step.py > gpurun_out/slow_step.txt"""
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

CASES = [
    ("Finite-CC-PMSM-v0", "constraints=('i_sq',)"),
    ("Finite-CC-PMSM-v0", "ode_solver=ga.RK4Solver(nsteps=2)"),
    ("Cont-CC-PMSM-v0", "ode_solver=ga.EulerSolver(nsteps=4)"),
    ("Finite-CC-SCIM-v0", "constraints=('i_sa', 'i_sb', 'i_sc')"),
    ("Finite-CC-SCIM-v0", "ode_solver=ga.RK4Solver(nsteps=2)"),
    ("Cont-CC-ExtExDc-v0", "constraints=('i_a',)"),
]


def child(env_id, kw, n, K):
    import hashlib

    import torch

    import gym_electric_motor_amd as ga

    env = eval(f"ga.make(env_id, n_envs=n, {kw})", dict(ga=ga, env_id=env_id, n=n))
    ps = env.physical_system
    g = torch.Generator(device="cuda").manual_seed(5)
    if ps._discrete:
        acts = torch.randint(0, int(ps.action_space.n), (K, n), device="cuda", generator=g, dtype=torch.uint8)
    else:
        acts = (torch.rand((K, n, ps._n_act), device="cuda", generator=g) * 2 - 1).to(ps._tdtype)
    obs, done = env.rollout(acts)
    h = hashlib.sha256(obs.cpu().numpy().tobytes() + done.cpu().numpy().tobytes()).hexdigest()[:16]
    for _ in range(3):
        env.rollout(acts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 5
    for _ in range(R):
        env.rollout(acts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    nbytes = n * K * (ps._n_out * 4 + 1 + (1 if ps._discrete else ps._n_act * 4))
    print(f"{n * K / dt / 1e9:.2f} {nbytes / dt / 8e12:.3f} {h} {ps.last_launch().split('<')[0].split('::')[-1]}")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    print("| env | option | envs | single-wave G env-steps/s (of roofline) | pipelined SLOW | identical |")
    print("|---|---|---:|---:|---:|---|")
    for env_id, kw in CASES:
        for n in (32768, 131072):
            res = []
            for pipe in ("0", "1"):
                e = dict(os.environ, GEMX_PIPE=pipe, GEMX_QUIET="1")
                out = subprocess.run([sys.executable, __file__, env_id, kw, str(n), "500"], env=e, capture_output=True, text=True)
                res.append(out.stdout.strip().split() if out.returncode == 0 else ["fail", "-", out.stderr.strip()[-300:].replace("\n", " / "), "-"])
                if out.returncode != 0:
                    print(res[-1][2], file=sys.stderr)
            print(f"| {env_id} | `{kw}` | {n} | {res[0][0]} ({res[0][1]}) {res[0][3]} | {res[1][0]} ({res[1][1]}) {res[1][3]} | {res[0][2] == res[1][2]} |", flush=True)
