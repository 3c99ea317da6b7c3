#!/usr/bin/env python3
"""A number on "latency-bound" (round 5 verdict, item 6): for the throughput-matrix rows below half the HBM roofline at 16384 envs -- one
workgroup per CU, the integrator wave alone on its SIMD -- how much of the launch IS the integrator wave's own instruction stream?

Needs timing builds (-DGEMX_TIMING: per-wave-role clock64 deltas) of the units involved, one (load, solver, dead time) combination each:

    python tools/latency_bound_rows.py --build          # in the build container: variants/lat_a, variants/lat_b (~1 min)
    python tools/latency_bound_rows.py > profiles/r06_latency_bound_rows.md      # on the GPU box

Per row: the integrator wave's cycles per control step between its block barriers (compute), its wait for the loader's actions (vmwait)
and at the barrier, the launch time (HIP events, timing build) and
    chain_bound  = (K x compute cycles per step / shader clock + t_fixed) / launch time,   t_fixed = 10 us (dispatch, prologue, release: DESIGN 4.2)
i.e. the fraction of the launch that K sequential steps of ONE wave's instruction stream account for.  >= 0.85: the row is closed -- only
fewer instructions in the step make it faster (more envs per CU cannot: each workgroup's integrator is its own chain); below: the lever is named."""
import argparse
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# label, env id, make kwargs (as source text), variant dir, unit, (load, solver, il)
ROWS = [
    ("SeriesDc cont SC", "Cont-SC-SeriesDc-v0", "{}", "lat_a", "3_0_0", (1, 1, 0)),
    ("ShuntDc finite", "Finite-CC-ShuntDc-v0", "{}", "lat_a", "4_3_0", (0, 1, 0)),
    ("ExtExDc finite (2x4QC)", "Finite-CC-ExtExDc-v0", "{}", "lat_a", "5_5_0", (0, 1, 0)),
    ("PMSM cont SC (poly load)", "Cont-SC-PMSM-v0", "{}", "lat_a", "1_2_0", (1, 1, 0)),
    ("SCIM cont SC", "Cont-SC-SCIM-v0", "{}", "lat_a", "2_2_0", (1, 1, 0)),
    ("SCIM cont SC + fused reward", "Cont-SC-SCIM-v0", "{'reward': True}", "lat_a", "2_2_0", (1, 1, 0)),
    ("PMSM finite (headline, for scale)", "Finite-CC-PMSM-v0", "{}", "lat_a", "1_1_0", (0, 1, 0)),
    ("PMSM finite + RC supply", "Finite-CC-PMSM-v0", "{'rc': True}", "lat_a", "1_1_0", (0, 1, 0)),
    ("PMSM finite + random initial states", "Finite-CC-PMSM-v0", "{'rinit': 'PMSM'}", "lat_a", "1_1_0", (0, 1, 0)),
    ("PMSM cont SC + random initial states", "Cont-SC-PMSM-v0", "{'rinit': 'PMSM', 'rinit_load': True}", "lat_a", "1_2_0", (1, 1, 0)),
    ("PMSM finite + dead time 1us", "Finite-CC-PMSM-v0", "{'converter': dict(interlocking_time=1e-6)}", "lat_b", "1_1_0", (0, 1, 1)),
    ("SCIM cont + random initial states", "Cont-CC-SCIM-v0", "{'rinit': 'SCIM'}", "lat_b", "2_2_0", (0, 1, 0)),
]
N, K = 16384, 500
T_FIXED_US = 10.0


def build():
    jobs = {}
    for _, _, _, var, unit, only in ROWS:
        jobs[(var, unit)] = only
    procs = []
    for (var, unit), only in jobs.items():
        tmp = os.path.join(REPO, "variants", f"_{var}_{unit}")
        procs.append((var, unit, tmp, subprocess.Popen([sys.executable, os.path.join(REPO, "tools", "dev_build.py"), "--units", unit, "--defs", "GEMX_TIMING",
                                                        "--only", ",".join(str(x) for x in only), "--out", tmp])))
    import shutil

    for var, unit, tmp, p in procs:
        assert p.wait() == 0, (var, unit)
    for var in sorted({j[0] for j in jobs}):
        d = os.path.join(REPO, "variants", var)
        shutil.rmtree(d, ignore_errors=True)
        first = True
        for v2, unit, tmp, _ in procs:
            if v2 != var:
                continue
            if first:
                shutil.copytree(tmp, d, symlinks=True)
                first = False
            else:
                name = f"libgemx_u{unit}.so"
                os.remove(os.path.join(d, name))
                shutil.copy2(os.path.join(tmp, name), os.path.join(d, name))
        print("built", d)
    for _, _, tmp, _ in procs:
        shutil.rmtree(tmp, ignore_errors=True)


CHILD = r'''
import ctypes as C, os, sys, json
sys.path.insert(0, %(repo)r)
import torch
import gym_electric_motor_amd as ga
from gym_electric_motor_amd import _lib
env_id, kw, n, K = %(env_id)r, %(kw)s, %(n)d, %(K)d
mk = {}
if kw.get("rc"):
    mk["supply"] = ga.RCVoltageSupply(u_nominal=420.0, supply_parameter=dict(R=0.5, C=2e-3))
if kw.get("rinit"):
    cls = {"PMSM": ga.PermanentMagnetSynchronousMotor, "SCIM": ga.SquirrelCageInductionMotor}[kw["rinit"]]
    mk["motor"] = cls(motor_initializer=dict(random_init="uniform")); mk["seed"] = 3
    if kw.get("rinit_load"):
        mk["load"] = ga.PolynomialStaticLoad(load_initializer=dict(random_init="uniform"))
if "converter" in kw:
    mk["converter"] = kw["converter"]
env = ga.make(env_id, n_envs=n, tau=1e-4, **mk)
ps = env.physical_system
g = torch.Generator(device="cuda").manual_seed(1)
act = torch.randint(0, 8 if ps._n_act == 1 and not hasattr(ps.action_space, "nvec") else 16, (K, n), dtype=torch.uint8, device="cuda", generator=g) if ps._discrete else torch.rand((K, n, ps._n_act), device="cuda", generator=g) * 2 - 1
refs = rew = None
if kw.get("reward"):
    ps.set_reward(reward_weights=dict(omega=1.0), referenced_states=("omega",))
    refs = torch.rand((K, n, 1), device="cuda", generator=g) * 2 - 1
    rew = torch.empty((K, n), device="cuda")
env.reset()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 12
for r in range(reps):
    if r == reps // 2:
        e0.record()
    if refs is None:
        ps.rollout(act)
    else:
        ps.rollout(act, references=refs, reward_out=rew)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / (reps - reps // 2)
L = _lib.load()
L.gemx_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 32)()
L.gemx_debug_read(ps._handle, buf, 32)
tv, tc, tw, tot, wall, nb = buf[0:6]
nb &= 0xFFFFFFFF
desc = L.gemx_last_launch(ps._handle).decode()
print(json.dumps(dict(us=us, vmwait=tv / max(nb, 1), compute=tc / max(nb, 1), barrier=tw / max(nb, 1), nb=nb, clock_ghz=tot / (wall * 10 + 1e-9), desc=desc)))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    args = ap.parse_args()
    if args.build:
        build()
        return
    import json
    import re

    print("# Matrix rows below half the HBM roofline at 16384 envs: how much of the launch is the integrator wave's own instruction stream\n")
    print(f"Timing builds (-DGEMX_TIMING) of the units involved, {N} envs (one workgroup per CU), {K} control steps per launch; cycles per control step of the "
          f"integrator wave of workgroup 0; chain_bound = (K x compute / clock + {T_FIXED_US:.0f} us) / launch.\n")
    print("| row | kernel shape | launch µs | integrator: compute / step | wait for actions / step | at the barrier / step | clock GHz | chain_bound | verdict |")
    print("|---|---|---|---|---|---|---|---|---|")
    for label, env_id, kw, var, unit, only in ROWS:
        env = dict(os.environ, GEMX_UNIT_DIR=os.path.join(REPO, "variants", var))
        code = CHILD % dict(repo=REPO, env_id=env_id, kw=kw, n=N, K=K)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(f"| {label} | failed: {r.stderr.strip().splitlines()[-1][:120] if r.stderr.strip() else '?'} | | | | | | | |")
            continue
        d = json.loads(line[-1])
        m = re.search(r"D=(\d+)", d["desc"])
        D = int(m.group(1)) if m else 1
        comp, vmw, bar = d["compute"] / D, d["vmwait"] / D, d["barrier"] / D
        clock = d["clock_ghz"]
        bound = (K * comp / (clock * 1e3) + T_FIXED_US) / d["us"]
        if bound >= 0.85:
            verdict = "closed: the launch is K sequential steps of one wave; only fewer instructions per step help"
        elif bar > 0.3 * comp:
            verdict = "the integrator WAITS at the block barrier: output / loader waves are the longer pole"
        else:
            verdict = "fixed costs / hand-off dominate the gap"
        shape = re.search(r"<.*?>", d["desc"]).group(0) if "<" in d["desc"] else d["desc"][:40]
        print(f"| {label} | `{shape}` | {d['us']:.1f} | {comp:.0f} | {vmw:.0f} | {bar:.0f} | {clock:.2f} | {bound:.2f} | {verdict} |")


if __name__ == "__main__":
    main()
