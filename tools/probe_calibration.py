import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, time, torch, gym_electric_motor_amd as ga
env_id, n = sys.argv[1], int(sys.argv[2])
for trial in range(3):
    env = ga.make(env_id, n_envs=n, ode_solver=ga.RK4Solver())
    ps = env.physical_system
    K = 1000
    acts = torch.randint(0, 32, (K, n), device="cuda", dtype=torch.uint8) if ps._discrete else torch.rand((K, n, ps._n_act), device="cuda") * 2 - 1
    obs = torch.empty((K, n, ps._n_out), device="cuda"); done = torch.empty((K, n), dtype=torch.uint8, device="cuda")
    hist = []
    for i in range(60):
        for _ in range(8): ps.rollout(acts, obs_out=obs, done_out=done)
        torch.cuda.synchronize()
        ll = ps.last_launch()
        tag = ll[ll.find("limiter"):] if "limiter" in ll else "none"
        if not hist or hist[-1] != tag: hist.append(tag)
        if "calibrated" in tag: break
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ps.rollout(acts, obs_out=obs, done_out=done)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    b = (1 if ps._discrete else 4 * ps._n_act) + 4 * ps._n_out + 1
    print(f"{env_id} {n}: {n*K*b/dt/8e12:.3f} after {i*8} launches; path: {hist[-4:]}")
    env.close()
