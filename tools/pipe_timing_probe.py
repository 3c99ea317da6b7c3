"""Scratch probe (needs a -DGEMX_TIMING build): per-wave cycle breakdown of the pipelined kernel."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gym_electric_motor_amd as ga
from gym_electric_motor_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 500
env_id = sys.argv[3] if len(sys.argv) > 3 else "Finite-CC-PMSM-v0"
solver = ga.EulerSolver() if os.environ.get("PROBE_SOLVER", "rk4") == "euler" else ga.RK4Solver()
extra = {}
if os.environ.get("PROBE_RINIT"):  # random initial states (the RINIT instantiation: prepared draws)
    extra = dict(motor=ga.PermanentMagnetSynchronousMotor(motor_initializer=dict(random_init="uniform")), seed=3)
if os.environ.get("PROBE_RINIT") == "SCIM":
    extra = dict(motor=ga.SquirrelCageInductionMotor(motor_initializer=dict(random_init="uniform")), seed=3)
if os.environ.get("PROBE_RINIT_LOAD"):  # ... and the load's omega drawn as well (bench_matrix.py's "PMSM cont SC + random initial states")
    extra["load"] = ga.PolynomialStaticLoad(load_initializer=dict(random_init="uniform"))
if os.environ.get("PROBE_RC"):
    extra = dict(supply=ga.RCVoltageSupply(u_nominal=420.0, supply_parameter=dict(R=0.5, C=2e-3)))
env = ga.make(env_id, n_envs=n, device="cuda:0", **extra) if os.environ.get("PROBE_SOLVER") == "default" else ga.make(env_id, n_envs=n, device="cuda:0", ode_solver=solver, tau=1e-4, **extra)
ps = env.physical_system
env.reset()
if "Finite" in env_id:
    act = torch.randint(0, 8, (K, n), dtype=torch.uint8, device="cuda:0")
else:
    act = torch.rand((K, n, ps._n_act), device="cuda:0") * 2 - 1
refs = rew = None
if os.environ.get("PROBE_REWARD"):
    ps.set_reward(reward_weights=dict(i_sd=0.5, i_sq=0.5), referenced_states=("i_sd", "i_sq"))
    refs = torch.rand((K, n, 2), device="cuda:0") * 2 - 1
    rew = torch.empty((K, n), device="cuda:0")
reps = int(os.environ.get("PROBE_REPS", "5"))  # (a long train of launches shows the power-limited steady state of bench.py)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(reps):
    if r == reps // 2:
        e0.record()
    if refs is None:
        ps.rollout(act)
    else:
        ps.rollout(act, references=refs, reward_out=rew)
e1.record()
torch.cuda.synchronize()
print(f"HIP events: {1e3 * e0.elapsed_time(e1) / (reps - reps // 2):.1f} us per launch over the last {reps - reps // 2} of {reps} launches")
L = _lib.load()
L.gemx_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 32)()
L.gemx_debug_read(ps._handle, buf, 32)
print(L.gemx_last_launch(ps._handle))
for base in (0, 16):
    tv, tc, tw, tot, wall, nb = buf[base:base + 6]
    nlong, nb = nb >> 32, nb & 0xFFFFFFFF
    if nb == 0:  # (dc_stream_kernel instruments workgroup 0 only)
        continue
    outs = buf[base + 6:base + 12]
    print(f"blk{'0' if base == 0 else '37'}: nb={nb} total={tot} cyc wall={wall} (100MHz ticks => {wall*10} ns, clock={tot/(wall*10+1e-9):.3f} GHz)")
    print(f"   integrator per block: vmwait={tv/nb:.0f} compute={tc/nb:.0f} barrier={tw/nb:.0f} cycles")
    for w in range(3):  # (first three output waves)
        print(f"   out wave {w}: process={outs[2*w]/nb:.0f} barrier={outs[2*w+1]/nb:.0f}")
    print(f"   out wave 0: of which reward + flush={buf[base+14]/nb:.0f}")
    v = buf[base + 15]
    print(f"   integrator: {nlong} of {nb} barrier waits > 1000 cycles; out wave 0: longest block {v >> 40} cycles, {v & 0xFFFFF} blocks > 2000, {(v >> 20) & 0xFFFFF} > 3000")
    print(f"   loader: stage+wait={buf[base+12]/nb:.0f} barrier={buf[base+13]/nb:.0f}")

if os.environ.get("PROBE_RINIT"):  # loader: cycles per prepared-draw phase (scan | block 0 | block 1 | previous block 0 | previous block 1 | finish)
    ph = (C.c_ulonglong * 504)()
    L.gemx_debug_read(ps._handle, ph, 504)
    print("   loader prepared-draw phases (cycles per execution x executions): " + "  ".join(f"{i}: {ph[484 + i] / max(1, ph[490 + i]):.0f} x {ph[490 + i]}" for i in range(6)))
if b"dc_stream" in L.gemx_last_launch(ps._handle):  # every wave of workgroup 0 (dc_stream_kernel, -DGEMX_TIMING): work / barrier cycles per block
    allw = (C.c_ulonglong * 96)()
    L.gemx_debug_read(ps._handle, allw, 96)
    nb0 = buf[5] & 0xFFFFFFFF
    for w in range(16):
        wk, br = allw[32 + 2 * w], allw[33 + 2 * w]
        if wk or br:
            print(f"   wave {w:2d}: work={wk / nb0:7.0f} barrier={br / nb0:7.0f} cycles per block")
    if allw[64] or allw[65]:
        print(f"   first output wave, per block: observe + staging writes={allw[64] / nb0:.0f}  read back + wait={allw[65] / nb0:.0f}  stores={allw[66] / nb0:.0f}")

if os.environ.get("PROBE_TRACE"):  # barrier trace of workgroup 0, blocks 100..147: per wave (arrive, release), relative to the first arrival
    big = (C.c_ulonglong * 448)()
    L.gemx_debug_read(ps._handle, big, 448)
    tr = [[(big[64 + w * 96 + 2 * i], big[64 + w * 96 + 2 * i + 1]) for i in range(48)] for w in range(4)]
    t00 = min(t[0][0] for t in tr if t[0][0])
    print("block: integrator(arrive,release) out0 out1 loader   [cycles since the trace began]")
    for i in range(48):
        print(f"{100 + i}: " + "  ".join(f"({tr[w][i][0] - t00:6d},{tr[w][i][1] - t00:6d})" for w in range(4)))

if os.environ.get("PROBE_STEP"):  # K = 1 (gemx_step -> advance_kernel): cycles per phase of workgroup 0 / the last workgroup, wall span
    for _ in range(20):
        ps.simulate(act[0])
    torch.cuda.synchronize()
    buf2 = (C.c_ulonglong * 48)()
    L.gemx_debug_read(ps._handle, buf2, 48)
    print(L.gemx_last_launch(ps._handle))
    for name, base in (("first wg", 32), ("last wg ", 40)):
        ld, comp, flush, drain, w0, w1 = buf2[base:base + 6]
        print(f"   {name}: load+sync={ld} compute+sync={comp} flush+state stores issued={flush} store drain={drain} cycles; in-kernel wall {10*(w1-w0)} ns")
    print(f"   first wg entry -> last wg exit: {10 * (buf2[45] - buf2[36])} ns; last wg entered {10 * (buf2[44] - buf2[36])} ns after the first")
