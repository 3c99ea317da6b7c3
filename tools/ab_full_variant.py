"""A/B of the FULL pipelined instantiation (RC supply / random initialisers) against the single-wave kernel over the batch size.

    python tools/ab_full_variant.py > gpurun_out/full_variant.txt"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gym_electric_motor_amd as ga
K = 500
CASES = (
    ("plain", "Finite-CC-PMSM-v0", lambda: dict()),
    ("rc", "Finite-CC-PMSM-v0", lambda: dict(supply=ga.RCVoltageSupply(u_nominal=420.0, supply_parameter=dict(R=0.5, C=2e-3)))),
    ("rinit", "Finite-CC-PMSM-v0", lambda: dict(motor=ga.PermanentMagnetSynchronousMotor(motor_initializer=dict(random_init="uniform")), seed=3)),
    ("rinit_sc", "Cont-SC-PMSM-v0", lambda: dict(motor=ga.PermanentMagnetSynchronousMotor(motor_initializer=dict(random_init="uniform")),
                                                 load=ga.PolynomialStaticLoad(load_initializer=dict(random_init="uniform")), seed=3)),
    ("scim_plain", "Cont-CC-SCIM-v0", lambda: dict()),
    ("rinit_scim", "Cont-CC-SCIM-v0", lambda: dict(motor=ga.SquirrelCageInductionMotor(motor_initializer=dict(random_init="uniform")), seed=3)),
    ("rinit_gauss", "Cont-CC-PermExDc-v0", lambda: dict(motor=ga.DcPermanentlyExcitedMotor(motor_initializer=dict(random_init="gaussian", random_params=(30.0, 40.0))), seed=3)),
)
only = sys.argv[1:]
for n in (16384, 32768, 65536, 131072):
    for label, env_id, kwf in CASES:
        if only and label not in only:
            continue
        for pipe in ("1", "0") + (("2",) if label.startswith("rinit") and n > 65536 else ()):  # (2: the FULL kernel beyond 4 workgroups per CU)
            os.environ["GEMX_PIPE"] = pipe
            os.environ["GEMX_QUIET"] = "1"
            env = ga.make(env_id, n_envs=n, ode_solver=ga.RK4Solver(), tau=1e-4, **kwf())
            ps = env.physical_system
            g = torch.Generator(device="cuda").manual_seed(1)
            if ps._discrete:
                acts = torch.randint(0, int(ps.action_space.n), (K, n), device="cuda", dtype=torch.uint8, generator=g)
            else:
                acts = torch.rand((K, n, ps._n_act), device="cuda", generator=g) * 2 - 1
            obs = torch.empty((K, n, ps._n_out), device="cuda"); done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
            for _ in range(2): ps.rollout(acts, obs_out=obs, done_out=done)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ps.rollout(acts, obs_out=obs, done_out=done)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            dr = float(done.float().mean())  # terminations per env-step; per wave-step: 1 - (1 - dr)^64
            nb = ps._n_out * 4 + 1 + (1 if ps._discrete else 4 * ps._n_act)
            print(f"{label:11s} N={n:7d} GEMX_PIPE={pipe}: {n*K/ms/1e6:7.1f} G env-steps/s = {n*K*nb/ms/1e6/8000:.3f} of the roofline; done rate {dr:.4f} (a wave resets in {1 - (1 - dr) ** 64:.2f} of its steps)  "
                  f"{ps.last_launch().split(' grid')[0].replace('gemx::', '')}", flush=True)
            env.close()
