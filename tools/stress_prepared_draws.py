"""Stress of the prepared-draw protocol (RINIT instantiation of the pipelined kernel) against the single-wave kernel, which always draws
inline: observations, done bytes, states and the checkpoint blob (reset counters) of launches with millions of resets must be identical.
Exit status 1 on any mismatch.  python tools/stress_prepared_draws.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gym_electric_motor_amd as ga
def run(pipe, env_id, kwf, n, K, seed):
    os.environ["GEMX_PIPE"] = pipe; os.environ["GEMX_QUIET"] = "1"
    env = ga.make(env_id, n_envs=n, ode_solver=ga.RK4Solver(), tau=1e-4, seed=seed, **kwf())
    ps = env.physical_system
    g = torch.Generator(device="cuda").manual_seed(seed)
    acts = torch.randint(0, 8, (K, n), device="cuda", dtype=torch.uint8, generator=g) if ps._discrete else torch.rand((K, n, ps._n_act), device="cuda", generator=g) * 2 - 1
    obs, done = ps.rollout(acts)
    obs2, done2 = ps.rollout(acts[: K // 3])
    r = (obs.clone(), done.clone(), obs2.clone(), done2.clone(), ps.get_state(), ps.get_checkpoint()["aux"].clone())
    env.close()
    return r
CASES = (("Finite-CC-PMSM-v0", lambda: dict(motor=ga.PermanentMagnetSynchronousMotor(motor_initializer=dict(random_init="uniform")))),
         ("Cont-CC-SCIM-v0", lambda: dict(motor=ga.SquirrelCageInductionMotor(motor_initializer=dict(random_init="uniform")))),
         ("Cont-SC-PMSM-v0", lambda: dict(motor=ga.PermanentMagnetSynchronousMotor(motor_initializer=dict(random_init="gaussian")), load=ga.PolynomialStaticLoad(load_initializer=dict(random_init="gaussian")))))
bad = 0
for rep in range(4):
    for env_id, kwf in CASES:
        for n, K in ((16384, 600), (65536 + 70, 300), (1000, 2000)):
            a = run("1", env_id, kwf, n, K, 100 + rep); b = run("0", env_id, kwf, n, K, 100 + rep)
            ok = all(torch.equal(x, y) for x, y in zip(a, b))
            bad += not ok
            print(rep, env_id, n, K, "identical" if ok else "MISMATCH", int(a[1].sum()), "terminations", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
