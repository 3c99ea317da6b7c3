"""GPU box: converter dead time through the per-segment one-step maps (GEMX_LINMAP=1) against the stage-by-stage solver (=0), plain and
sub-stepped RK4, on the recorded dopri5 runs with interlocking (what found the fp32 map-precision bug of round 3)."""
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
