import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import gym_electric_motor_amd as ga
from gym_electric_motor_amd import _lib
n, K = 16384, 1008
for label, kw in (("control_space=dq", dict(control_space="dq")),
                  ("DqToAbc", dict(physical_system_wrappers=(ga.DqToAbcActionProcessor.make("PMSM"),))),
                  ("DqToAbc+DeadTime(1)", dict(physical_system_wrappers=(ga.DeadTimeProcessor(1), ga.DqToAbcActionProcessor.make("PMSM")))),
                  ("DeadTime(1) + control_space=dq", dict(control_space="dq", physical_system_wrappers=(ga.DeadTimeProcessor(1),)))):
    env = ga.make("Cont-CC-PMSM-v0", n_envs=n, ode_solver=ga.RK4Solver(), tau=1e-4, **kw)
    ps = env.physical_system
    act = torch.rand((K, n, 2), device="cuda") * 2 - 1
    for _ in range(4): ps.rollout(act)
    torch.cuda.synchronize()
    L = _lib.load(); L.gemx_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * 32)(); L.gemx_debug_read(ps._handle, buf, 32)
    tv, tc, tw, tot, wall, nb = buf[0:6]
    print(f"{label:32s}: integrator compute={tc/nb:.0f} barrier={tw/nb:.0f} cycles per block; out0 process={buf[6]/nb:.0f}; loader={buf[12]/nb:.0f}; wall {wall*10/1000:.1f} us  [{ps.last_launch().split(' grid')[0]}]")
    env.close()
