"""A/B of the FULL pipelined instantiation (RC supply / random initialisers) against the single-wave kernel over the batch size."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gym_electric_motor_amd as ga
K=500
for n in (16384, 32768, 65536, 131072):
    for pipe in ("1","0"):
        os.environ["GEMX_PIPE"]=pipe
        for label, kw in (("rc", dict(supply=ga.RCVoltageSupply(u_nominal=420.0, supply_parameter=dict(R=0.5, C=2e-3)))),
                          ("rinit", dict(motor=ga.PermanentMagnetSynchronousMotor(motor_initializer=dict(random_init="uniform")), seed=3))):
            env = ga.make("Finite-CC-PMSM-v0", n_envs=n, ode_solver=ga.RK4Solver(), tau=1e-4, **kw)
            ps = env.physical_system
            acts = torch.randint(0, 8, (K, n), device="cuda", dtype=torch.uint8)
            obs = torch.empty((K, n, 14), device="cuda"); done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
            for _ in range(2): ps.rollout(acts, obs_out=obs, done_out=done)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ps.rollout(acts, obs_out=obs, done_out=done)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)/5
            dr = float(done.float().mean())  # terminations per env-step; per wave-step: 1 - (1 - dr)^64
            print(f"{label:6s} N={n:7d} GEMX_PIPE={pipe}: {n*K/ms/1e6:7.1f} G env-steps/s  done rate {dr:.4f} (a wave resets in {1 - (1 - dr) ** 64:.2f} of its steps)  {ps.last_launch().split(' grid')[0]}", flush=True)
            env.close()
