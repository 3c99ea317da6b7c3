#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched SCML stepper (BASELINE.json metric), one JSON line on rank 0.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One bench "step" = ONE launch of the fused hot path (`gemx_rollout`) over one batch of synthetic input: `--steps-per-launch`
(default 1000) control steps (PhysicalSystem.simulate) of ALL envs of the job, every step's observation row [N, S_out] and done byte
written to HBM.  So `--steps 20 --warmup 5` = 20 timed launches after 5 untimed ones, `ms_per_step` = milliseconds per launch and
`value` = total envs x steps_per_launch x steps / wall time.  (Round 1 mapped `--steps` to control steps, which made the driver's
`--steps 20` a single 20-step launch: launch-latency bound, 5 % of the HBM roofline; the kernel needs ~200 us launches to show its
steady state.)

Default workload = BASELINE.json configs[2], the config the metric ("batched PMSM") is quoted on: Finite-CC-PMSM-v0 (Finite-B6C +
dq transform), 16384 envs per GPU, RK4, fp32, default SquaredConstraint(i_sd, i_sq) done mask with in-kernel auto-reset, uniformly
random discrete actions resident in HBM (synthetic).  tau = 1e-4 as the metric line states (the env's own default is 1e-5; the
arithmetic, hence the throughput, does not depend on tau).

Timed region: exactly K launches bracketed by barrier + torch.cuda.synchronize() on both sides; wall time = max over ranks.
Clock settling: an MI355X coming out of idle runs the first ~20 ms of a load below its sustained clock (measured per launch, headline
kernel, `tools/clock_settle_probe.py` -> profiles/r02g_clock_settle.md: 170 us for the first ten launches, 195-205 us for the next
twenty, 153 us from the ~100th launch on; an idle gap of >= 10 ms starts this over) -- and `--warmup 5 --steps 20` is 5 ms of work.
Every leg therefore runs `--settle-ms` (default 60) of the SAME launches untimed before its W warm-up launches, so that `value` is the
sustained rate a training run sees; the figure for the same W / K straight from an idle GPU is reported beside it as `cold_start`
(`--settle-ms 0` makes it the headline again).
Multi-GPU: envs are independent -> each rank steps its own shard, no data-path collective ("scaling": "weak").  Without WORLD_SIZE in
the environment `--gpus N` (N > 1) spawns its own N ranks (torch.multiprocessing, one per GPU, RCCL on 127.0.0.1).
`--gather chunk|step` additionally times the batched-return path: one RCCL all-gather of each launch's [K, n_local, S_out]
observation chunk (+ done bytes) / of every step's [n_local, S_out] rows; reported under "gather" beside the gather-off `value`.

Extra objects in the JSON line (rank 0):
  roofline            HBM roofline of the dominant kernel: algorithmic bytes per launch / mean launch duration measured HERE with HIP
                      events on the launch stream around the K timed launches; peak = 8000 GB/s (MI355X spec); traffic = PMC-measured
                      HBM bytes per launch of this exact (workload, envs, steps_per_launch) from profiles/hbm_traffic.json.
  cpu_baseline        oracle/gemx_oracle.c (scalar fp64 restatement, "port") timed on ONE host core on a bounded sample of the same
                      workload (`all_cores`: the same port on up to 32 host cores), plus the REFERENCE's own Python path as
                      recorded by oracle/cpu_reference_bench.py (fields, not prose).
  headline_no_linmap  the same launches with GEMX_LINMAP=0 (RK4 evaluated stage by stage instead of through the one-step affine map).
  single_step / single_step_graph   one launch per control step (closed-loop RL usage), eager and replayed from a HIP graph.
  configs             BASELINE configs 2 and 4 (PermExDc 4096 envs Euler; SCIM 65536 envs RK4 with the env's PolynomialStaticLoad -- also
                      with split_kinks -- and with the ConstantSpeedLoad BASELINE.json names) through the same measurement.
  at_scale            the headline kernel with the chip full (1M envs).
"""
import argparse
import json
import math
import os
import socket
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
    # BASELINE.json configs[3] as written: "5-state induction motor + ConstantSpeedLoad" (the env id's own default load is the
    # PolynomialStaticLoad of the "scim" workload above; SURVEY.md 8(d) C4 names both)
    "scim_constspeed": dict(env_id="Cont-SC-SCIM-v0", envs=65536, solver="rk4", tau=1e-4, a_bytes=12, s_ode=6, s_out=14, const_speed=100.0,
                            desc="Cont-SC-SCIM-v0 (Cont-B6C) with ConstantSpeedLoad(omega_fixed=100), RK4, fp32, tau=1e-4, default constraint + auto-reset"),
}


def bytes_per_env_step_fused(w):
    """action in + observation row out + done byte (state stays in registers)."""
    return w["a_bytes"] + 4 * w["s_out"] + 1


def bytes_per_env_step_single(w):
    """+ ODE state read and written every step."""
    return bytes_per_env_step_fused(w) + 2 * 4 * w["s_ode"]


def make_env(ga, w, n_envs, device, split_kinks=False):
    sol = ga.EulerSolver() if w["solver"] == "euler" else ga.RK4Solver(split_kinks=split_kinks)
    kw = {}
    if w.get("const_speed") is not None:
        kw["load"] = ga.ConstantSpeedLoad(omega_fixed=w["const_speed"])
    return ga.make(w["env_id"], n_envs=n_envs, device=device, ode_solver=sol, tau=w["tau"], **kw)


def make_actions(torch, ps, K, n, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    if ps._discrete:
        return torch.randint(0, 8, (K, n), device=device, generator=g, dtype=torch.uint8)
    return torch.rand((K, n, ps._n_act), device=device, generator=g, dtype=torch.float32) * 2 - 1


class Timed:
    """wall seconds (max over ranks), mean device ms per launch (HIP events on the launch stream)."""

    def __init__(self, wall, launch_ms):
        self.wall, self.launch_ms = wall, launch_ms


def measure(torch, dist, env, n_local, steps, warmup, spl, device, world, seed, gather="off", gd=None, settle_ms=0.0):
    """`warmup` untimed + `steps` timed launches of `spl` control steps each, after `settle_ms` of the same launches (clock governor, see
    module docstring).  gather: off | chunk | step."""
    ps = env.physical_system
    n_act_bufs = max(1, min(4, steps))
    acts = make_actions(torch, ps, spl * n_act_bufs, n_local, device, seed)
    obs = torch.empty((spl, n_local, ps._n_out), dtype=torch.float32, device=device)
    done = torch.empty((spl, n_local), dtype=torch.uint8, device=device)
    env.reset()

    def launch(i):
        a0 = (i % n_act_bufs) * spl
        if gather == "step":  # batched return after EVERY control step: one launch + one all-gather per step
            for k in range(spl):
                o = ps.simulate(acts[a0 + k])
                gd.gather_observations(o, ps.done)
        else:
            env.rollout(acts[a0 : a0 + spl], obs_out=obs, done_out=done)
            if gather == "chunk":
                gd.gather_rollout(obs, done)

    if settle_ms > 0:
        t_end, i = time.perf_counter() + settle_ms * 1e-3, 0
        while time.perf_counter() < t_end:
            for _ in range(4):  # (plain launches: a wall-clock-bounded loop must not contain collectives)
                a0 = (i % n_act_bufs) * spl
                env.rollout(acts[a0 : a0 + spl], obs_out=obs, done_out=done)
                i += 1
            torch.cuda.synchronize()
    for i in range(warmup):
        launch(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        launch(i)
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        if dist.get_backend() == "gloo":
            t = torch.tensor([dt], dtype=torch.float64)
        else:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if gather != "step":
        assert torch.isfinite(obs).all()
    return Timed(dt, e0.elapsed_time(e1) / steps)


def measure_single_step(torch, env, n_local, K, W, device, seed, graph_steps=0, settle_ms=0.0):
    """One gemx_step launch per control step.  graph_steps > 0: `graph_steps` launches captured into ONE HIP graph and replayed."""
    ps = env.physical_system
    Ka = 256
    acts_all = make_actions(torch, ps, Ka, n_local, device, seed)
    acts = [acts_all[k] for k in range(Ka)]  # views made once: indexing a tensor costs ~2 us of host time per call
    env.reset()
    t_end = time.perf_counter() + settle_ms * 1e-3
    while time.perf_counter() < t_end:
        for k in range(256):
            ps.simulate(acts[k])
        torch.cuda.synchronize()
    for k in range(W):
        ps.simulate(acts[k % Ka])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graph_steps:
        S = min(graph_steps, Ka)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for k in range(3):
                ps.simulate(acts[k])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for k in range(S):
                ps.simulate(acts[k])
        torch.cuda.synchronize()
        reps = max(1, K // S)
        graph.replay()
        torch.cuda.synchronize()
        t_end = time.perf_counter() + settle_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(16):
                graph.replay()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = reps * S
    else:
        t0 = time.perf_counter()
        e0.record()
        for k in range(K):
            ps.simulate(acts[k % Ka])
        e1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = K
    assert torch.isfinite(ps._obs).all()
    return dt / n, e0.elapsed_time(e1) / n, n


def cpu_baseline(w, budget_s=12.0):
    """The oracle (scalar fp64 C restatement of the reference algorithm) on ONE host core, bounded sample; beside it the reference's
    own Python path as oracle/cpu_reference_bench.py recorded it in the build container (profiles/cpu_reference.json)."""
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
    # the same port on many host cores: one thread per core, at most 32 (ctypes releases the GIL; orc_rollout_many keeps its state on the
    # stack), each with its own slice of envs sized for ~0.5 s at the single-core rate -- a container whose CPU quota is below its
    # visible core count (the GPU boxes show 256 cores) then costs seconds, not minutes, and shows up as a poor speed-up
    import concurrent.futures as cf

    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    cores = max(1, min(visible, 32))
    per = max(64, int(n_env2 * K / dt * 0.5 / K))
    slices = [acts(per, K) for _ in range(min(cores, 4))]  # (a few distinct action tensors, reused: host memory stays small)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(lambda i: orc.rollout_many(p, slices[i % len(slices)]), range(cores)))
    dt_all = time.perf_counter() - t0
    all_cores = dict(value=cores * per * K / dt_all, unit="env-steps/s", cores=cores, visible_cores=visible,
                     sample=f"{cores} threads x {per} envs x {K} steps, {dt_all:.1f} s")
    out = dict(value=n_env2 * K / dt, unit="env-steps/s", cores=1, kind="port", host_cores=os.cpu_count(), all_cores=all_cores,
               sample=f"{n_env2} envs x {K} steps of the same workload ({w['env_id']}, {w['solver']}, episodic) through "
                      f"oracle/gemx_oracle.c (fp64, gcc -O2), {dt:.1f} s on 1 of {os.cpu_count()} host cores")
    ref_path = os.path.join(REPO, "profiles", "cpu_reference.json")
    if os.path.exists(ref_path):
        try:
            ref = json.load(open(ref_path))
            out["reference"] = {"source": "profiles/cpu_reference.json (oracle/cpu_reference_bench.py: the reference's own Python path, "
                                          "timed in the build container, which has /root/reference; the GPU box has not)",
                                "host": ref.get("host"), "env_steps_per_s": ref.get("results", {}).get(w["env_id"])}
        except Exception as e:  # a malformed record must not kill the bench line
            out["reference"] = {"error": repr(e)}
    return out


def roofline_of(w, n_local, spl, launch_ms, kernel_desc, workload_key):
    b_step = bytes_per_env_step_fused(w)
    launch_bytes = n_local * (spl * b_step + 2 * 4 * w["s_ode"])
    achieved = launch_bytes / (launch_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{workload_key}:{n_local}:{spl}")
        except Exception:
            traffic = None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "kernel": kernel_desc, "launch_ms": launch_ms, "algorithmic_bytes_per_launch": launch_bytes,
            "bytes_per_env_step": b_step}


def worker(args, rank, world, local_rank, backend):
    import torch
    import torch.distributed as dist

    import gym_electric_motor_amd as ga
    from gym_electric_motor_amd import distributed as gd

    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev if args.oversubscribe else local_rank
    if dev_index >= ndev:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but only {ndev} are visible; run with --gpus <= {ndev}, or pass "
                         "--oversubscribe to put several ranks on one GPU (gloo control plane; a functional check, not a measurement)")
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)

    w = dict(WORKLOADS[args.workload], key=args.workload)
    n_local = args.envs_per_gpu or (32768 if (args.workload == "pmsm" and world == 8) else w["envs"])  # BASELINE config 5: 8 x 32768
    n_total = n_local * world
    K, W, spl = args.steps, args.warmup, args.steps_per_launch

    S = args.settle_ms
    env = make_env(ga, w, n_local, dev_index)
    t_cold = measure(torch, dist, env, n_local, K, W, spl, device, world, seed=1234 + rank) if S > 0 else None  # straight from idle
    t = measure(torch, dist, env, n_local, K, W, spl, device, world, seed=1234 + rank, settle_ms=S)
    kernel_desc = env.physical_system.last_launch()
    same_shard = None
    if n_local != w["envs"] and args.envs_per_gpu is None:  # --gpus 8 default = BASELINE config 5 (8 x 32768): also the N=1 shard size
        env_s = make_env(ga, w, w["envs"], dev_index)
        ts = measure(torch, dist, env_s, w["envs"], K, W, spl, device, world, seed=1234 + rank, settle_ms=S)
        env_s.close()
        same_shard = {"envs_per_gpu": w["envs"], "value": w["envs"] * world * spl * K / ts.wall, "unit": "env-steps/s",
                      "ms_per_step": ts.wall / K * 1e3, "note": "same per-GPU shard as the --gpus 1/2/4 lines (strict weak scaling)"}
    gathered = None
    if args.gather != "off" and world > 1 and backend == "gloo":
        gathered = {"skipped": "oversubscribed ranks run a gloo control plane; the device all-gather needs RCCL (one rank per GPU)"}
    elif args.gather != "off":
        modes = ["chunk", "step"] if args.gather == "both" else [args.gather]
        gathered = {}
        for mode in modes:
            Kg = K if mode == "chunk" else max(1, min(K, 2))
            spl_g = spl if mode == "chunk" else min(spl, 200)
            tg = measure(torch, dist, env, n_local, Kg, min(W, 2), spl_g, device, world, seed=4321 + rank, gather=mode, gd=gd, settle_ms=S)
            per_call = n_local * (spl_g if mode == "chunk" else 1) * (4 * w["s_out"] + 1)
            gathered[mode] = {"value": n_total * spl_g * Kg / tg.wall, "unit": "env-steps/s", "steps": Kg, "steps_per_launch": spl_g,
                              "ms_per_step": tg.wall / Kg * 1e3, "bytes_gathered_per_rank_per_call": per_call * world,
                              "collective": f"{backend} all_gather_into_tensor, world {world}"}
    env.close()

    if rank == 0:
        out = {
            "metric": "env-steps/sec (batched PMSM, tau=1e-4)" if args.workload == "pmsm" else f"env-steps/sec ({args.workload})",
            "value": n_total * spl * K / t.wall,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": t.wall / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{w['desc']}; {n_local} envs/GPU x {world} GPU(s); one bench step = one fused launch of {spl} control "
                                   "steps, obs [K,N,S_out] rows + done bytes written for every control step",
                       "env_id": w["env_id"], "envs_per_gpu": n_local, "solver": w["solver"], "tau": w["tau"],
                       "steps_per_launch": spl, "control_steps_timed": spl * K, "clock_settle_ms": S,
                       "parallelism": f"env-sharded x{world}, no data-path collective", "world_size": world,
                       "backend": (backend if world > 1 else None), "oversubscribed": bool(args.oversubscribe and world > ndev)},
            "roofline": roofline_of(w, n_local, spl, t.launch_ms, kernel_desc, args.workload),
        }
        if t_cold is not None:
            rcold = roofline_of(w, n_local, spl, t_cold.launch_ms, kernel_desc, args.workload)
            out["cold_start"] = {"value": n_total * spl * K / t_cold.wall, "unit": "env-steps/s", "ms_per_step": t_cold.wall / K * 1e3,
                                 "roofline_frac": rcold["frac"],
                                 "note": f"the same {W} + {K} launches straight from an idle GPU, no clock settling (module docstring)"}
        if gathered is not None:
            out["gather"] = gathered
        if same_shard is not None:
            out["same_shard_as_n1"] = same_shard
        if not args.no_extras and world == 1:
            extras(torch, dist, ga, args, w, n_local, spl, device, dev_index, out)
        elif not args.no_extras:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def extras(torch, dist, ga, args, w, n_local, spl, device, dev_index, out):
    """Rank 0, N = 1 only: the CPU baseline and the informational legs (each bounded to a few seconds)."""
    out["cpu_baseline"] = cpu_baseline(w)
    # the same launches without the one-step affine map (general stage-by-stage RK4)
    os.environ["GEMX_LINMAP"] = "0"
    try:
        env0 = make_env(ga, w, n_local, dev_index)
        t0 = measure(torch, dist, env0, n_local, min(args.steps, 10), 2, spl, device, 1, seed=77, settle_ms=args.settle_ms)
        desc0 = env0.physical_system.last_launch()
        env0.close()
    finally:
        del os.environ["GEMX_LINMAP"]
    r0 = roofline_of(w, n_local, spl, t0.launch_ms, desc0, args.workload + "/nolinmap")
    out["headline_no_linmap"] = {"value": n_local * spl * min(args.steps, 10) / t0.wall, "unit": "env-steps/s", "launch_ms": t0.launch_ms,
                                 "achieved_GBps": r0["achieved"], "frac_of_peak": r0["frac"], "env": "GEMX_LINMAP=0"}
    # closed-loop usage: one launch per control step, eager and from a HIP graph
    b1 = bytes_per_env_step_single(w)
    env1 = make_env(ga, w, n_local, dev_index)
    host_s, dev_ms, n1 = measure_single_step(torch, env1, n_local, 2000, 100, device, seed=99, settle_ms=args.settle_ms)
    out["single_step"] = {"value": n_local / host_s, "unit": "env-steps/s", "ms_per_step": host_s * 1e3, "device_ms_per_step": dev_ms,
                          "achieved_GBps": n_local * b1 / (dev_ms * 1e-3) / 1e9, "bytes_per_env_step": b1, "steps": n1,
                          "note": "one gemx_step launch per control step, eager"}
    host_s, dev_ms, n1 = measure_single_step(torch, env1, n_local, 4096, 100, device, seed=99, graph_steps=64, settle_ms=args.settle_ms)
    out["single_step_graph"] = {"value": n_local / host_s, "unit": "env-steps/s", "ms_per_step": host_s * 1e3, "device_ms_per_step": dev_ms,
                                "achieved_GBps": n_local * b1 / (dev_ms * 1e-3) / 1e9, "bytes_per_env_step": b1, "steps": n1,
                                "note": "64 gemx_step launches captured into one HIP graph (torch.cuda.CUDAGraph) and replayed"}
    env1.close()
    # BASELINE configs 2 and 4 through the same measurement
    out["configs"] = {}
    for key in ("permexdc", "scim", "scim_constspeed"):
        if key == args.workload:
            continue
        wc = dict(WORKLOADS[key], key=key)
        envc = make_env(ga, wc, wc["envs"], dev_index)
        tc = measure(torch, dist, envc, wc["envs"], 10, 3, spl, device, 1, seed=5, settle_ms=args.settle_ms)
        rc = roofline_of(wc, wc["envs"], spl, tc.launch_ms, envc.physical_system.last_launch(), key)
        envc.close()
        out["configs"][key] = {"workload": wc["desc"], "envs": wc["envs"], "steps_per_launch": spl, "value": wc["envs"] * spl * 10 / tc.wall,
                               "unit": "env-steps/s", "roofline": rc}
        if key == "scim":  # the same config with RK4Solver(split_kinks=True): steps cut at the PolynomialStaticLoad's kinks (accuracy option)
            envk = make_env(ga, wc, wc["envs"], dev_index, split_kinks=True)
            tk = measure(torch, dist, envk, wc["envs"], 10, 3, spl, device, 1, seed=5, settle_ms=args.settle_ms)
            rk = roofline_of(wc, wc["envs"], spl, tk.launch_ms, envk.physical_system.last_launch(), key + "/split_kinks")
            envk.close()
            out["configs"]["scim_split_kinks"] = {"workload": wc["desc"] + ", RK4Solver(split_kinks=True)", "envs": wc["envs"], "steps_per_launch": spl,
                                                  "value": wc["envs"] * spl * 10 / tk.wall, "unit": "env-steps/s", "roofline": rk}
    # the headline kernel with the chip full
    n_big, c_big = 2 ** 20, 100
    envb = make_env(ga, w, n_big, dev_index)
    tb = measure(torch, dist, envb, n_big, 4, 2, c_big, device, 1, seed=7, settle_ms=args.settle_ms)
    envb.close()
    bb = n_big * (c_big * bytes_per_env_step_fused(w) + 2 * 4 * w["s_ode"])
    out["at_scale"] = {"envs": n_big, "steps_per_launch": c_big, "value": n_big * 4 * c_big / tb.wall, "unit": "env-steps/s",
                       "launch_ms": tb.launch_ms, "achieved_GBps": bb / (tb.launch_ms * 1e-3) / 1e9,
                       "frac_of_peak": bb / (tb.launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}


def _spawned(local_rank, args, world, port, backend):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    worker(args, local_rank, world, local_rank, backend)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed launches (one bench step = one fused launch)")
    ap.add_argument("--warmup", type=int, default=5, help="untimed launches")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="pmsm")
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--steps-per-launch", "--chunk", dest="steps_per_launch", type=int, default=1000,
                    help="control steps fused into one launch")
    ap.add_argument("--gather", choices=["off", "chunk", "step", "both"], default="off",
                    help="also time the batched-return path: all-gather of each launch's observation chunk / of every step's rows")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than visible GPUs (ranks share GPUs, gloo control plane): functional check only")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed launches of the same workload for this long before each leg's warm-up (clock governor; 0 = off)")
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline / single_step / configs / at_scale legs")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0 or args.steps_per_launch < 2 or args.settle_ms < 0:
        raise SystemExit("bench.py: --steps >= 1, --warmup >= 0, --steps-per-launch >= 2, --settle-ms >= 0")

    world_env = os.environ.get("WORLD_SIZE")
    backend = "gloo" if args.oversubscribe else "nccl"
    if world_env is not None:  # launched by torchrun / the driver: one rank per process already
        world = int(world_env)
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
        worker(args, int(os.environ.get("RANK", "0")), world, int(os.environ.get("LOCAL_RANK", "0")), backend)
    elif args.gpus == 1:
        worker(args, 0, 1, 0, backend)
    else:  # self-spawn: one process per GPU
        import torch
        import torch.multiprocessing as mp

        ndev = torch.cuda.device_count()
        if args.gpus > ndev and not args.oversubscribe:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible; pass --oversubscribe to share GPUs between "
                             "ranks (gloo control plane; a functional check, not a measurement)")
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_spawned, args=(args, args.gpus, port, backend), nprocs=args.gpus, join=True)


if __name__ == "__main__":
    main()
