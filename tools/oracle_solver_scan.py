#!/usr/bin/env python3
"""CPU (fp64 oracle): error of the oracle's fixed-step restatements against every recorded run of the reference's default solver whose
load is a PolynomialStaticLoad (tests/golden/*_dopri5.npz).  The CPU-side counterpart of tests/solver_scan.py, used to design the
device's kink handling without a GPU (the oracle restates the device algorithm in fp64; the fp32 figure is the GPU scan's).

    python tools/oracle_solver_scan.py [solver ...]        # default: rk4 rk4_kink rk4_kink1

TEST INFRASTRUCTURE (imports oracle/)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from oracle import oracle as orc  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def run(name, solver, nsteps=1):
    d, meta = T._load(name)
    p = orc.params_from_meta(meta, solver=solver)
    p.nsteps = nsteps
    env = orc.OracleEnv(p)
    env.reset()
    obs, done = env.rollout(d["actions"], auto_reset=True)
    try:
        rel, _, col, dmsg = T.compare_trajectory(meta, d, obs, done)
    except AssertionError as e:
        return float("nan"), str(e)[:40]
    return rel, col + ("*" if "flip" in dmsg else "")


def main():
    solvers = sys.argv[1:] or ["rk4", "rk4_kink", "rk4_kink1"]
    names = [c for c in T.CASES if c.endswith("dopri5")]
    names = [n for n in names if T._load(n)[1]["load"] != "ConstantSpeedLoad"]
    print("| fixture | " + " | ".join(solvers) + " |")
    print("|---|" + "---|" * len(solvers))
    worst = {s: 0.0 for s in solvers}
    for name in names:
        row = []
        for s in solvers:
            ns = 1
            sol = s
            if "x" in s:
                sol, ns = s.split("x")
                ns = int(ns)
            rel, col = run(name, sol, ns)
            worst[s] = max(worst[s], rel if rel == rel else 1.0)
            row.append(f"{rel:.1e} {col}")
        print(f"| {name[:-7]} | " + " | ".join(row) + " |")
        sys.stdout.flush()
    print("| **worst** | " + " | ".join(f"{worst[s]:.1e}" for s in solvers) + " |")


if __name__ == "__main__":
    main()
