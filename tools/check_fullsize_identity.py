#!/usr/bin/env python3
"""Full-size bit-identity of the rate limiter: BASELINE-sized batches x 1000 steps, GEMX_PACE_GBPS=0 (which also changes the shape the launcher picks)
against the default: python tools/check_fullsize_identity.py > profiles/<round>_fullsize_identity.txt"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import gym_electric_motor_amd as ga
print("full-size bit-identity of the rate limiter (GEMX_PACE_GBPS=0 against the default), 1000 steps per launch, two consecutive launches")
for env_id, n in (("Finite-CC-PMSM-v0", 131072), ("Finite-CC-PMSM-v0", 32768), ("Cont-SC-SCIM-v0", 65536), ("Cont-CC-PMSM-v0", 65536), ("Finite-CC-ShuntDc-v0", 49152)):
    outs = []
    for pace in ("0", None):
        if pace is None: os.environ.pop("GEMX_PACE_GBPS", None)
        else: os.environ["GEMX_PACE_GBPS"] = pace
        env = ga.make(env_id, n_envs=n)
        ps = env.physical_system
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(5)
        acts = (torch.randint(0, int(ps.action_space.n), (1000, n), device="cuda", generator=g, dtype=torch.uint8) if ps._discrete
                else torch.rand((1000, n, ps._n_act), device="cuda", generator=g) * 2 - 1)
        ps.rollout(acts)
        o, d = ps.rollout(acts)
        outs.append((o.view(torch.int32).sum(dtype=torch.int64).item(), d.sum(dtype=torch.int64).item(), o[-1].clone(), ps.last_launch().split("K=1000")[1]))
        env.close(); del o, d, acts; torch.cuda.empty_cache()
    same = outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1] and torch.equal(outs[0][2], outs[1][2])
    print(f"{env_id} {n} envs: identical={same}  bit-sum {outs[1][0]}  done {outs[1][1]}  | off:[{outs[0][3].strip()}] default:[{outs[1][3].strip()}]", flush=True)
