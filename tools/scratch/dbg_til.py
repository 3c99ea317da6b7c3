import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_parity as T
import gym_electric_motor_amd as ga
for name in ("pmsm_free_uniform_til_dopri5", "pmsm_free_held_til_dopri5"):
    for ns in (1, 2, 8):
        for lm in ("1", "0"):
            os.environ["GEMX_LINMAP"] = lm
            d, meta, obs, done = T._run_golden(name, "float32", solver=ga.RK4Solver(nsteps=ns))
            rel, _, col, dmsg = T.compare_trajectory(meta, d, obs, done)
            print(name, "nsteps", ns, "LINMAP", lm, f"{rel:.2e}", col, dmsg[:60])
