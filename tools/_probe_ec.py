import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import gym_electric_motor_amd as ga
from oracle import oracle as orc
orc.build()
L = orc.lib()
L.orc_dev_set_atol_omega_scaled(1)
for gname, env_id in (("scim_epi_uniform_euler", "Cont-SC-SCIM-v0"), ("pmsm_sc_free_held_dopri5", "Cont-SC-PMSM-v0")):
    try:
        _, meta = orc.load_golden(gname)
    except Exception as e:
        print(gname, "missing", e); continue
    meta = dict(meta, tau=1e-4)
    n, K = 8, 1500
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, (K, n, 3))
    env = ga.make(env_id, n_envs=n, tau=1e-4, dtype="float64", ode_solver=ga.ScipyOdeSolver())
    obs, done = env.rollout(torch.as_tensor(a, device="cuda"))
    obs = obs.cpu().numpy(); done = done.cpu().numpy()
    p = orc.params_from_meta(meta, solver="dev_adaptive_kink", episodic=True)
    worst = 0.0
    for j in range(n):
        e = orc.OracleEnv(p); e.reset()
        ro, rd = e.rollout(a[:, j], auto_reset=True)
        d = np.abs(obs[:, j] - ro); d[:, 12] = np.minimum(d[:, 12], 2 - d[:, 12])
        first = int(np.argmax(d.max(axis=1) > 1e-9)) if (d.max(axis=1) > 1e-9).any() else -1
        worst = max(worst, d.max())
        print(env_id, "lane", j, "max |diff| %.3e" % d.max(), "done equal", bool((rd == done[:, j].astype(bool)).all()), "first step > 1e-9:", first)
    print(env_id, "worst", worst)
    env.close()
