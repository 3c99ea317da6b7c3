#!/usr/bin/env python3
"""Observed worst parity error per BASELINE config (GPU fp32 through the C ABI vs the reference's recorded trajectories).

    python tools/parity_report.py [--all] > profiles/<round>_parity.md        (on the GPU box)

For every golden fixture of the three BASELINE envs (Cont-CC-PermExDc-v0, Finite-CC-PMSM-v0, Cont-SC-SCIM-v0): the fixture's own
solver where the device has it (Euler), and every device solver against the reference's DEFAULT solver (scipy dopri5) fixtures.
Error = max over columns of max|got - ref| / max(max|ref| of the column, 1e-3) on normalised states, angle on the circle
(tests/test_gpu_parity.py:_rel_err); episodic fixtures are compared PER EPISODE (each episode restarts from the reset state on both
sides, so a done flip near the constraint boundary ends the comparison of that episode only).  TEST INFRASTRUCTURE: imports oracle/.
"""
import argparse
import glob
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true", help="every fixture, not only the three BASELINE envs")
    ap.add_argument("--solvers", default="rk4,dp5")
    args = ap.parse_args()
    import test_gpu_parity as T

    base = ("Cont-CC-PermExDc-v0", "Finite-CC-PMSM-v0", "Cont-SC-SCIM-v0")
    print("| fixture | env | K | reference solver | device solver | worst rel err | column | max abs err | done masks |")
    print("|---|---|---|---|---|---|---|---|---|")
    worst = {}
    for name in T.CASES:
        d, meta = T._load(name)
        if not args.all and meta["env_id"] not in base:
            continue
        if name.startswith("rw_") or name.startswith("rc_"):
            continue
        if meta["solver"] in ("euler", "euler4"):
            solvers = [meta["solver"]]
        else:
            solvers = args.solvers.split(",")
            if meta["env_id"].endswith("SC-SynRM-v0"):
                solvers = [s + "x8" for s in solvers]
        for solver in solvers:
            d, meta, obs, done = T._run_golden(name, "float32", solver=solver)
            rel, ab, col, dmsg = T.compare_trajectory(meta, d, obs, done)
            print(f"| {name} | {meta['env_id']} | {len(d['terminated'])} | {meta['solver']} | {solver} | {rel:.2e} | {col} | {ab:.2e} | {dmsg} |", flush=True)
            key = (meta["env_id"], "same solver" if meta["solver"].startswith("euler") else "vs default dopri5 / solve_ivp")
            if rel > worst.get(key, (0, ""))[0]:
                worst[key] = (rel, f"{name} / {solver} / {col}")
    print()
    print("| env | comparison | worst rel err | where |")
    print("|---|---|---|---|")
    for (env_id, kind), (rel, where) in sorted(worst.items()):
        print(f"| {env_id} | {kind} | {rel:.2e} | {where} |")


if __name__ == "__main__":
    main()
