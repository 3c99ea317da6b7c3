"""How often a reset finds its prepared draw (needs a -DGEMX_TIMING build: debug words 500 / 501 = resets served from the loader wave's
queue / drawn inline, summed over the launch)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gym_electric_motor_amd as ga
from gym_electric_motor_amd import _lib

K = 500
CASES = (("PMSM finite CC", "Finite-CC-PMSM-v0", lambda: dict(motor=ga.PermanentMagnetSynchronousMotor(motor_initializer=dict(random_init="uniform")))),
         ("PMSM cont SC", "Cont-SC-PMSM-v0", lambda: dict(motor=ga.PermanentMagnetSynchronousMotor(motor_initializer=dict(random_init="uniform")),
                                                          load=ga.PolynomialStaticLoad(load_initializer=dict(random_init="uniform")))),
         ("SCIM cont CC", "Cont-CC-SCIM-v0", lambda: dict(motor=ga.SquirrelCageInductionMotor(motor_initializer=dict(random_init="uniform")))))
L = _lib.load()
L.gemx_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for label, env_id, kwf in CASES:
    for n in (16384, 65536):
        env = ga.make(env_id, n_envs=n, ode_solver=ga.RK4Solver(), tau=1e-4, seed=3, **kwf())
        ps = env.physical_system
        g = torch.Generator(device="cuda").manual_seed(1)
        acts = torch.randint(0, 8, (K, n), device="cuda", dtype=torch.uint8, generator=g) if ps._discrete else torch.rand((K, n, ps._n_act), device="cuda", generator=g) * 2 - 1
        buf = (C.c_ulonglong * 504)()
        L.gemx_debug_read(ps._handle, buf, 504)
        p0, i0 = buf[500], buf[501]
        obs, done = ps.rollout(acts)
        torch.cuda.synchronize()
        L.gemx_debug_read(ps._handle, buf, 504)
        p, i = buf[500] - p0, buf[501] - i0
        print(f"{label:15s} N={n:6d}: {int(done.sum())} terminations in {n * K} env-steps; resets from the queue {p}, inline {i} ({100.0 * i / max(1, p + i):.1f} %)", flush=True)
        env.close()
