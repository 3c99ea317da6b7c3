#!/usr/bin/env python3
"""Times the REFERENCE's own CPU path (gym-electric-motor 3.0.2, unmodified, imported from /root/reference) on this box's host cores.

    MPLBACKEND=Agg python oracle/cpu_reference_bench.py [--steps 10000] [--out profiles/cpu_reference.json]

BASELINE.json configs[0] / SURVEY.md 8(d) "CPU baseline beside it": for each of the three configured env ids and each solver
{reference default scipy dopri5, EulerSolver, ScipySolveIvpSolver() (solve_ivp RK45, solvers.py:187-219)}

  * 1 core:     `for a in actions: env.step(a)` with reset-on-done (episodic, default constraints) and with constraints=() (free run),
                K random-action steps after env.reset(seed=0), time.perf_counter around the loop;
  * all cores:  one process per host core (multiprocessing, one env each, independent action streams), aggregate env-steps/s =
                total steps / (latest end - earliest start) of the stepping loops.

The reference needs `gymnasium`, which is not installed here: oracle/gymnasium_standin (test infrastructure) provides the four
space classes it imports.  This script runs ONLY where /root/reference exists (the build container); the GPU box reads the
recorded JSON (bench.py `cpu_baseline.reference`).  Dashboards stay at the env default (utils.py:6-7 instantiates a MotorDashboard
even for visualization=None; it is never rendered).
"""
import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GEM_REFERENCE", "/root/reference")
ENVS = {"Cont-CC-PermExDc-v0": "box1", "Finite-CC-PMSM-v0": "disc8", "Cont-SC-SCIM-v0": "box3"}
SOLVERS = ("dopri5", "euler", "solve_ivp")


def _import_reference():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, os.path.join(REPO, "oracle", "gymnasium_standin"))
    sys.path.insert(0, os.path.join(REF, "src"))
    import gym_electric_motor as gem
    from gym_electric_motor.physical_systems import solvers

    return gem, solvers


def _run(env_id, solver, K, seed, episodic):
    """-> (steps, terminations, t_start, t_end) of one env's stepping loop (wall clock, time.time for cross-process comparison)."""
    import numpy as np

    gem, solvers = _import_reference()
    sol = {"dopri5": solvers.ScipyOdeSolver, "euler": solvers.EulerSolver, "solve_ivp": solvers.ScipySolveIvpSolver}[solver]()
    kw = dict(ode_solver=sol)
    if not episodic:
        kw["constraints"] = ()
    env = gem.make(env_id, **kw)
    env.reset(seed=0)
    rng = np.random.default_rng(seed)
    kind = ENVS[env_id]
    actions = rng.integers(0, 8, K) if kind == "disc8" else rng.uniform(-1, 1, (K, int(kind[3:])))
    n_term = 0
    for k in range(50):  # warm-up (first-call allocations, scipy integrator set-up)
        _, _, term, _, _ = env.step(int(actions[k]) if kind == "disc8" else actions[k])
        if term:
            env.reset()
    env.reset()
    t0w = time.time()
    t0 = time.perf_counter()
    for k in range(K):
        _, _, term, _, _ = env.step(int(actions[k]) if kind == "disc8" else actions[k])
        if term:
            n_term += 1
            env.reset()
    dt = time.perf_counter() - t0
    return K, n_term, t0w, t0w + dt


def _run_star(args):
    return _run(*args)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "cpu_reference.json"))
    ap.add_argument("--cores", type=int, default=os.cpu_count())
    args = ap.parse_args()
    assert os.path.isdir(os.path.join(REF, "src", "gym_electric_motor")), f"{REF} not found: this tool runs in the build container only"
    K, cores = args.steps, args.cores
    import numpy
    import scipy

    out = {"host": {"cpu_model": cpu_model(), "cores": os.cpu_count(), "cores_used_all_cores_leg": cores, "python": platform.python_version(),
                    "numpy": numpy.__version__, "scipy": scipy.__version__},
           "reference": "gym-electric-motor 3.0.2 from " + REF + " (unmodified; gymnasium stand-in oracle/gymnasium_standin)",
           "steps_per_env": K, "unit": "env-steps/s", "results": {}}
    ctx = mp.get_context("spawn")
    for env_id in ENVS:
        out["results"][env_id] = {}
        for solver in SOLVERS:
            r = {}
            for mode, episodic in (("episodic", True), ("free_run", False)):
                with ctx.Pool(1) as pool:  # a fresh process per measurement: no state shared between configurations
                    n, n_term, t0, t1 = pool.apply(_run, (env_id, solver, K, 1234, episodic))
                r[f"{mode}_1core"] = n / (t1 - t0)
                if episodic:
                    r["episodic_terminations"] = n_term
                with ctx.Pool(cores) as pool:
                    res = pool.map(_run_star, [(env_id, solver, K, 1234 + i, episodic) for i in range(cores)])
                r[f"{mode}_all_cores"] = sum(x[0] for x in res) / (max(x[3] for x in res) - min(x[2] for x in res))
            out["results"][env_id][solver] = r
            print(env_id, solver, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
