#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched SCML stepper (BASELINE.json metric), one JSON line on rank 0.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one control step (PhysicalSystem.simulate) of ALL envs of the job.  Default workload = BASELINE.json
configs[2], the config the metric ("batched PMSM") is quoted on: Finite-CC-PMSM-v0 (Finite-B6C + dq transform),
16384 envs per GPU, RK4, fp32, default SquaredConstraint(i_sd, i_sq) done mask with in-kernel auto-reset,
uniformly random discrete actions resident in HBM (synthetic).  tau = 1e-4 as the metric line states (the env's
own default is 1e-5; the arithmetic, hence the throughput, does not depend on tau).

Timed region: exactly K steps = ceil(K / chunk) launches of the fused advance kernel (`gemx_rollout`, chunk
steps per launch, every step's observation row [N, S_out] and done byte written to HBM), bracketed by barrier +
torch.cuda.synchronize() on both sides; wall time = max over ranks; value = total envs * K / wall time.
Multi-GPU: envs are independent -> each rank steps its own shard, no data-path collective ("scaling": "weak").

Extra objects in the JSON line:
  roofline     : HBM roofline of the dominant kernel (advance_kernel): algorithmic bytes per launch / mean launch
                 duration measured here with HIP events on the launch stream; peak = 8000 GB/s (MI355X spec).
  cpu_baseline : oracle/gemx_oracle.c (scalar fp64 restatement of the reference algorithm, "port") timed on ONE host
                 core on a bounded sample of the same workload.
  single_step  : the same workload advanced by one launch per control step (closed-loop RL usage), informational.
  at_scale     : the same kernel at a larger batch (what the chip does when it is full), informational.
"""
import argparse
import json
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md); 6290 GB/s measured float4 copy

WORKLOADS = {
    # name: env_id, envs/GPU, solver, tau, action bytes per env-step, S_ode, S_out
    "pmsm": dict(env_id="Finite-CC-PMSM-v0", envs=16384, solver="rk4", tau=1e-4, a_bytes=1, s_ode=4, s_out=14,
                 desc="Finite-CC-PMSM-v0 (Finite-B6C + dq transform), RK4, fp32, tau=1e-4, default constraint + auto-reset"),
    "permexdc": dict(env_id="Cont-CC-PermExDc-v0", envs=4096, solver="euler", tau=1e-4, a_bytes=4, s_ode=2, s_out=5,
                     desc="Cont-CC-PermExDc-v0 (Cont-4QC), Euler, fp32, tau=1e-4, default constraint + auto-reset"),
    "scim": dict(env_id="Cont-SC-SCIM-v0", envs=65536, solver="rk4", tau=1e-4, a_bytes=12, s_ode=6, s_out=14,
                 desc="Cont-SC-SCIM-v0 (Cont-B6C, PolynomialStaticLoad), RK4, fp32, tau=1e-4, default constraint + auto-reset"),
}


def bytes_per_env_step_fused(w):
    """action in + observation row out + done byte (state stays in registers)."""
    return w["a_bytes"] + 4 * w["s_out"] + 1


def bytes_per_env_step_single(w):
    """+ ODE state read and written every step."""
    return bytes_per_env_step_fused(w) + 2 * 4 * w["s_ode"]


def make_env(ga, w, n_envs, device):
    sol = ga.EulerSolver() if w["solver"] == "euler" else ga.RK4Solver()
    return ga.make(w["env_id"], n_envs=n_envs, device=device, ode_solver=sol, tau=w["tau"])


def make_actions(torch, ps, K, n, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    if ps._discrete:
        return torch.randint(0, 8, (K, n), device=device, generator=g, dtype=torch.uint8)
    return torch.rand((K, n, ps._n_act), device=device, generator=g, dtype=torch.float32) * 2 - 1


def run_fused(torch, env, acts, obs, done, K, chunk, events=False):
    """K steps as ceil(K/chunk) launches.  With events: ONE pair of HIP events on the launch stream around the run of full-chunk
    launches (an event pair per launch adds two marker packets, ~10 us, to every 140-us kernel) -> (e0, e1, n_full_launches)."""
    n_full = K // chunk
    e0 = e1 = None
    k = i = 0
    while k < K:
        c = min(chunk, K - k)
        if events and i == 0:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        env.rollout(acts[k % acts.shape[0] : k % acts.shape[0] + c], obs_out=obs[:c], done_out=done[:c])
        i += 1
        if events and i == n_full:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
        k += c
    return (e0, e1, n_full) if events else None


def measure(torch, dist, env, w, n_local, K, W, chunk, device, world, seed):
    ps = env.physical_system
    chunk = max(1, min(chunk, K))
    acts = make_actions(torch, ps, chunk * max(1, min(4, math.ceil(K / chunk))), n_local, device, seed)
    obs = torch.empty((chunk, n_local, ps._n_out), dtype=torch.float32, device=device)
    done = torch.empty((chunk, n_local), dtype=torch.uint8, device=device)
    env.reset()
    run_fused(torch, env, acts, obs, done, W, chunk)  # warmup (untimed)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev = run_fused(torch, env, acts, obs, done, K, chunk, events=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    e0, e1, n_full = ev
    launch_ms = e0.elapsed_time(e1) / n_full if n_full else dt * 1e3 / max(1, math.ceil(K / chunk))
    assert torch.isfinite(obs).all()
    return dt, launch_ms, chunk


def measure_single_step(torch, env, w, n_local, K, W, device, seed):
    ps = env.physical_system
    Ka = min(K, 256)
    acts = make_actions(torch, ps, Ka, n_local, device, seed)
    env.reset()
    for k in range(W):
        ps.simulate(acts[k % Ka])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for k in range(K):
        ps.simulate(acts[k % Ka])
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, e0.elapsed_time(e1) / K


def cpu_baseline(w, budget_s=12.0):
    """The oracle (scalar fp64 C restatement of the reference algorithm) on ONE host core, bounded sample."""
    import numpy as np

    from oracle import oracle as orc

    golden = {"pmsm": "pmsm_epi_held_tau1e-4_euler", "permexdc": "permexdc_epi_held_euler", "scim": "scim_epi_uniform_euler"}[w["key"]]
    _, meta = orc.load_golden(golden)
    meta = dict(meta, tau=w["tau"])
    p = orc.params_from_meta(meta, solver=w["solver"], episodic=True)
    rng = np.random.default_rng(1234)
    n_env, K = 64, 1000

    def acts(n_env, K):
        if w["a_bytes"] == 1:
            return rng.integers(0, 8, (K, n_env, 1)).astype(np.float64)
        return rng.uniform(-1, 1, (K, n_env, w["a_bytes"] // 4))

    a = acts(n_env, K)
    t0 = time.perf_counter()
    orc.rollout_many(p, a)
    t_probe = time.perf_counter() - t0
    scale = max(1, int(budget_s / max(t_probe, 1e-3)))
    n_env2 = min(n_env * scale, 65536)
    a = acts(n_env2, K)
    t0 = time.perf_counter()
    orc.rollout_many(p, a)
    dt = time.perf_counter() - t0
    return dict(value=n_env2 * K / dt, unit="env-steps/s", cores=1, kind="port",
                sample=f"{n_env2} envs x {K} steps of the same workload ({w['env_id']}, {w['solver']}, episodic) through "
                       f"oracle/gemx_oracle.c (fp64, gcc -O2), {dt:.1f} s on 1 of {os.cpu_count()} host cores; the reference's own "
                       "Python path measured 8.7e3 (dopri5) / 1.2e4 (Euler) env-steps/s on 1 core (BASELINE.md)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="pmsm")
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--chunk", type=int, default=1000, help="control steps fused into one launch")
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline / single_step / at_scale legs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import gym_electric_motor_amd as ga
    from gym_electric_motor_amd import distributed as gd

    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    rank, world, local_rank = gd.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    w = dict(WORKLOADS[args.workload], key=args.workload)
    n_local = args.envs_per_gpu or w["envs"]
    n_total = n_local * world
    K, W = args.steps, args.warmup

    env = make_env(ga, w, n_local, local_rank)
    dt, launch_ms, chunk = measure(torch, dist, env, w, n_local, K, W, args.chunk, device, world, seed=1234 + rank)
    kernel_desc = env.physical_system.last_launch()
    env.close()

    if rank == 0:
        b_step = bytes_per_env_step_fused(w)
        launch_bytes = n_local * (chunk * b_step + 2 * 4 * w["s_ode"])
        achieved = launch_bytes / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{args.workload}:{n_local}:{chunk}")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec (batched PMSM, tau=1e-4)" if args.workload == "pmsm" else f"env-steps/sec ({args.workload})",
            "value": n_total * K / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{w['desc']}; {n_local} envs/GPU x {world} GPU(s); fused rollout, {chunk} steps/launch, "
                                   "obs [K,N,14] rows + done bytes written every step",
                       "env_id": w["env_id"], "envs_per_gpu": n_local, "solver": w["solver"], "tau": w["tau"],
                       "steps_per_launch": chunk, "parallelism": f"env-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "kernel": kernel_desc, "launch_ms": launch_ms,
                         "algorithmic_bytes_per_launch": launch_bytes, "bytes_per_env_step": b_step},
        }
        if not args.no_extras and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
            # closed-loop usage: one launch per control step
            env1 = make_env(ga, w, n_local, local_rank)
            Ks = min(K, 2000)
            dts, ms1 = measure_single_step(torch, env1, w, n_local, Ks, min(W, 100), device, seed=99)
            env1.close()
            b1 = bytes_per_env_step_single(w)
            out["single_step"] = {"value": n_local * Ks / dts, "unit": "env-steps/s", "ms_per_step": dts / Ks * 1e3,
                                  "device_ms_per_step": ms1, "achieved_GBps": n_local * b1 / (ms1 * 1e-3) / 1e9,
                                  "bytes_per_env_step": b1, "note": "one gemx_step launch per control step (launch-bound at this N)"}
            # the same kernel with the chip full
            n_big = 2 ** 20
            c_big = 100
            envb = make_env(ga, w, n_big, local_rank)
            dtb, msb, c_big = measure(torch, dist, envb, w, n_big, 2 * c_big, c_big, c_big, device, 1, seed=7)
            envb.close()
            bb = n_big * (c_big * b_step + 2 * 4 * w["s_ode"])
            out["at_scale"] = {"envs": n_big, "steps_per_launch": c_big, "value": n_big * 2 * c_big / dtb, "unit": "env-steps/s",
                               "launch_ms": msb, "achieved_GBps": bb / (msb * 1e-3) / 1e9, "frac_of_peak": bb / (msb * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        elif not args.no_extras:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
