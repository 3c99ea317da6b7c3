"""Scratch probe (needs a -DGEMX_TIMING build): on which SIMD of which CU the integrator wave of every workgroup of a pipelined launch ran."""
import ctypes as C, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gym_electric_motor_amd as ga
from gym_electric_motor_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = ga.make("Finite-CC-PMSM-v0", n_envs=n, device="cuda:0", ode_solver=ga.RK4Solver(), tau=1e-4)
ps = env.physical_system
env.reset()
act = torch.randint(0, 8, (200, n), dtype=torch.uint8, device="cuda:0")
for _ in range(3):
    ps.rollout(act)
torch.cuda.synchronize()
L = _lib.load()
L.gemx_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 400)()
L.gemx_debug_read(ps._handle, buf, 400)
raw = bytes(buf)[1024:1024 + 2 * min(1024, n // 64)]
print(L.gemx_last_launch(ps._handle))
per_cu = collections.defaultdict(list)
for i in range(len(raw) // 2):
    v = raw[2 * i] | (raw[2 * i + 1] << 8)
    simd, cu, sh, se, xcc = v & 3, (v >> 2) & 15, (v >> 6) & 1, (v >> 7) & 7, (v >> 10) & 15
    per_cu[(xcc, se, sh, cu)].append((i, simd))
hist = collections.Counter()
for k, v in per_cu.items():
    hist[tuple(sorted(collections.Counter(s for _, s in v).values(), reverse=True))] += 1
print(f"{len(per_cu)} CUs; integrator waves per SIMD on a CU (sorted counts) -> number of CUs: {dict(hist)}")
for k in list(sorted(per_cu))[:6]:
    print(k, per_cu[k])
