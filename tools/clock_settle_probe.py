"""Scratch probe: per-launch duration of the headline rollout over a long train of back-to-back launches (how long the clock governor
takes to settle), and after idle gaps of different length."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gym_electric_motor_amd as ga

n, K = 16384, 1000
env = ga.make("Finite-CC-PMSM-v0", n_envs=n, device="cuda:0", ode_solver=ga.RK4Solver(), tau=1e-4)
ps = env.physical_system
env.reset()
act = torch.randint(0, 8, (K, n), dtype=torch.uint8, device="cuda:0")
obs = torch.empty((K, n, 14), device="cuda:0")
done = torch.empty((K, n), dtype=torch.uint8, device="cuda:0")


def train(m):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(m + 1)]
    ev[0].record()
    for i in range(m):
        ps.rollout(act, obs_out=obs, done_out=done)
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [1e3 * ev[i].elapsed_time(ev[i + 1]) for i in range(m)]


def show(tag, d):
    g = [sum(d[i:i + 10]) / len(d[i:i + 10]) for i in range(0, len(d), 10)]
    print(tag, " ".join(f"{x:.0f}" for x in g), flush=True)


ps.rollout(act, obs_out=obs, done_out=done)
torch.cuda.synchronize()
time.sleep(1.0)
show("from idle (1 s), means of 10 launches [us]:", train(400))
for gap in (0.001, 0.01, 0.1, 1.0):
    time.sleep(gap)
    show(f"after a {gap * 1e3:.0f} ms gap:", train(100))
