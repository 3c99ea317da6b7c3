"""GPU box: is the error of one RK4 step per control step on a recorded dopri5 run the scheme's (GEMX_LINMAP=0 shows the same) or the
one-step map's (it goes away without the map)?  python tools/probe_map_bias.py [fixture ...]"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_parity as T
import gym_electric_motor_amd as ga
names = sys.argv[1:] or ["dfim_fin_free_held_dopri5", "dfim_cont_free_held_dopri5", "eesm_fin_free_held_dopri5", "eesm_cont_free_held_dopri5",
                         "pmsm_free_held_dopri5", "scim_constspeed_free_held_dopri5"]
for name in names:
    row = []
    for ns in (1, 2):
        for lm in ("1", "0"):
            os.environ["GEMX_LINMAP"] = lm
            d, meta, obs, done = T._run_golden(name, "float32", solver=ga.RK4Solver(nsteps=ns))
            rel, _, col, dmsg = T.compare_trajectory(meta, d, obs, done)
            row.append(f"ns={ns} map={lm}: {rel:.1e} ({col})")
    d, meta, obs, done = T._run_golden(name, "float64", solver=ga.RK4Solver())
    rel, _, col, dmsg = T.compare_trajectory(meta, d, obs, done)
    print(name, "tau", meta["tau"], " | ".join(row), f"| fp64 ns=1: {rel:.1e} ({col})")
