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
`--gather` times the batched-return path beside the gather-off `value`: `chunk` = one RCCL all-gather of each launch's [K, n_local,
S_out] observation chunk (+ done bytes) into preallocated buffers, `step` = of every step's [n_local, S_out] rows.  Default `auto`
(round 4) = a bounded `chunk` leg (<= 5 launches) whenever the world has more than one rank, so that every multi-GPU line carries the
only collective the north star names: `gather.chunk` (rollout + gather rate) and `rccl` = the collective by itself on a launch's real
outputs {world_seen, backend, version, bytes_per_rank, ms, GB_per_s = (W - 1) x bytes_per_rank / time, own_slot_bit_identical}; `config5`
(W x 32768 envs, BASELINE config 5's shard size) rides every N > 1 line too.  The headline is assembled BEFORE any optional leg runs and
every optional leg is wrapped: a failing gather / config5 / extras leg becomes an "error" field of the line, never a lost line.
WORLD_SIZE in the environment -- also WORLD_SIZE=1 -- or --force-dist initialises the process group, so one rank on one GPU runs the
exact code path of the scaling run (tests/test_gpu_parity.py::test_bench_multi_gpu_code_path_through_rccl_in_a_world_of_one).
One clock (round 4): `roofline.achieved / frac / launch_ms` are priced with the WALL time of the timed region / launches -- the clock
`value` and `ms_per_step` use -- so value x bytes per env-step == roofline.achieved; the HIP-event mean is `launch_ms_hip_events`.
Legs whose launches are short (BASELINE config 2: 20 us) time >= 3 ms of launches per region for the same reason.

Robustness to the box (round 3): the timed region is run `--repeats` (3) times back to back and the MEDIAN region is what `value`,
`ms_per_step` and `roofline` report (min / max under `repeats`); core clock, memory clock, socket power and temperature are read from
the GPU's hwmon files before and after every leg (`telemetry`); `sustained_1s` legs run >= `--sustain-s` seconds of back-to-back
launches with those sensors sampled every 20 ms -- a slower line can be told from a slower or power-limited box.
Multi-process robustness: `init_process_group` gets `--dist-timeout` (a dead rank is an exception, not a hung barrier), a world > 1
without MASTER_PORT is refused, and the self-spawner picks a free port.

Every fused launch of every leg goes through `PhysicalSystem.bind_rollout()` (round 4): the action / observation / done tensors of a leg
are fixed, so they are checked and their pointers taken once and a bench step is the `gemx_rollout` FFI call -- `rollout()`'s per-call
argument handling, 8-12 us of Python, is more than half of what a launch of BASELINE config 2 takes on the device.

Round 5: `roofline` carries both roofs SURVEY.md 8(d) names (`frac` against the 8 TB/s spec, `frac_of_measured` against the guide's measured
6.29 TB/s copy rate); `cpu_baseline` rides the N > 1 line as well (timed on rank 0's host on a shorter sample while the other ranks wait);
launches of more workgroups than CUs run under the CLOSED-LOOP rate limiter (each handle calibrates its interval with HIP events during
its first ~30 paced launches: every leg's `--settle-ms` covers that, and `roofline.kernel` says `limiter calibrated at x`).

Round 6: stdout carries ONE COMPACT line (< 6 KB: the contract's keys, `roofline`, `cpu_baseline`, `legs` = {name: {frac, launch_ms}} for
BASELINE configs 2 / 4 / 4 with the ConstantSpeedLoad / 5's shard / 1M envs / the error-controlled solver, and for N > 1 `gather`, `rccl`,
`config5`), printed LAST; the full record described below goes to `bench_extras.json` beside this script (`--extras-file`; a copy under
gpurun_out/ when that directory exists).  Round 5 printed the full record as the line, it grew to 20 KB and the driver could not parse it.

Objects of the full record (rank 0; the stdout line keeps the numbers of `roofline`, `cpu_baseline`, `legs` only):
  roofline            HBM roofline of the dominant kernel: algorithmic bytes per launch / mean launch duration measured HERE with HIP
                      events on the launch stream around the K timed launches; peak = 8000 GB/s (MI355X spec); traffic = HBM bytes
                      per launch MEASURED IN THIS RUN: two child runs of this script under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`
                      (separate passes; FETCH_SIZE doubled per the guide's gfx950 note) -- `traffic_source` says so, or names
                      profiles/hbm_traffic.json (an earlier collection's passes) when rocprofv3 is not available / `--no-pmc`.
  repeats, telemetry  see above.
  cpu_baseline        oracle/gemx_oracle.c (scalar fp64 restatement, "port") timed on ONE host core on a bounded sample of the same
                      workload (`all_cores`: the same port on up to 32 host cores), plus the REFERENCE's own Python path as
                      recorded by oracle/cpu_reference_bench.py (fields, not prose).
  headline_no_linmap  the same launches with GEMX_LINMAP=0 (RK4 evaluated stage by stage instead of through the one-step affine map).
  sustained_1s        >= 1 s of back-to-back launches of the headline workload with clocks / power sampled; since round 4 EVERY leg under
                      `configs` has one, and its `frac` = min(minimum of the repeated windows, sustained second) is the figure to quote.
  overrides           the GEMX_* environment switches active in this run (normally none); gemx_last_launch() names them too.
  single_step / single_step_bound / single_step_graph   one launch per control step (closed-loop RL usage): eager
                      `PhysicalSystem.simulate()` on a device tensor, the pre-bound `bind_step()` call, and 64 steps replayed from a HIP graph.
  configs             BASELINE config 2 (PermExDc 4096 envs Euler, with `launch_model`: t = t_fixed + K t_step fitted over launches of
                      250 ... 2000 steps, and `frac_of_latency_bound` = (K x the integrator's dependency chain + t_fixed) / measured),
                      config 4 (`scim`: SCIM 65536 envs with the env's PolynomialStaticLoad and the solver `make(env_id)` hands out = RK4 + kink
                      correction; `scim_plain_rk4` beside it; `scim_error_controlled` = ScipyOdeSolver(), the device's error-controlled
                      Dormand-Prince; `scim_constspeed` = with the ConstantSpeedLoad BASELINE.json names) and config 5's per-GPU shard
                      (PMSM 32768 envs), each through the same measurement with 3 repeats.
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

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_MEASURED_COPY_GBPS = 6290.0  # the guide's MEASURED float4 copy rate (MI355X_MICROARCH.md:35): SURVEY.md 8(d) asks for both roofs

class Telemetry:
    """Shader clock / memory clock / socket power / temperature of ONE GPU from the amdgpu hwmon files in sysfs (microseconds per read),
    located through the device's PCI bus id (hipDeviceGetPCIBusId).  Everything is best effort: a box that hides sysfs yields
    {"available": False, "reason": ...} and the bench goes on."""

    def __init__(self, dev_index):
        self.dir, self.reason = None, None
        try:
            import ctypes as C
            import glob

            import torch  # noqa: F401  (loads the HIP runtime)

            hip = None
            for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
                try:
                    hip = C.CDLL(name)
                    break
                except OSError:
                    continue
            bus = None
            if hip is not None:
                buf = C.create_string_buffer(64)
                if hip.hipDeviceGetPCIBusId(buf, 64, int(dev_index)) == 0:
                    bus = buf.value.decode().lower()
            cands = []
            if bus:
                cands = glob.glob(f"/sys/bus/pci/devices/{bus}/hwmon/hwmon*")
            if not cands:  # fall back: the dev_index-th card that exposes a shader clock (PCI order)
                cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
                if len(cards) > dev_index:
                    cands = glob.glob(os.path.join(os.path.dirname(cards[dev_index]), "hwmon", "hwmon*"))
                    self.reason = "PCI bus id lookup failed: card chosen by order"
            if cands:
                self.dir = cands[0]
                self.bus = bus
            else:
                self.reason = "no amdgpu hwmon directory visible in sysfs"
        except Exception as e:  # pragma: no cover
            self.reason = repr(e)

    def _read(self, name, scale):
        try:
            with open(os.path.join(self.dir, name)) as fh:
                return float(fh.read().strip()) * scale
        except Exception:
            return None

    def sample(self):
        if self.dir is None:
            return None
        return {"sclk_mhz": self._read("freq1_input", 1e-6), "mclk_mhz": self._read("freq2_input", 1e-6),
                "power_w": self._read("power1_input", 1e-6) or self._read("power1_average", 1e-6), "temp_c": self._read("temp2_input", 1e-3)}

    def describe(self):
        return {"available": self.dir is not None, "source": self.dir, "pci_bus_id": getattr(self, "bus", None), "note": self.reason}


class Sampler:
    """Background sampling of Telemetry while a leg runs (period ~20 ms): median / min / max of every quantity."""

    def __init__(self, tele, period_s=0.02):
        import threading

        self.tele, self.period, self.rows, self._stop = tele, period_s, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            r = self.tele.sample()
            if r:
                self.rows.append(r)
            self._stop.wait(self.period)

    def __enter__(self):
        if self.tele.dir is not None:
            self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._t.is_alive():
            self._t.join(timeout=1.0)

    def summary(self):
        if not self.rows:
            return None
        out = {"samples": len(self.rows)}
        for k in self.rows[0]:
            v = sorted(x[k] for x in self.rows if x[k] is not None)
            if v:
                out[k] = {"median": v[len(v) // 2], "min": v[0], "max": v[-1]}
        return out


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


def make_env(ga, w, n_envs, device, split_kinks=None, error_controlled=False):
    """split_kinks None: the solver `make(env_id)` hands out when the caller names none (envs.default_ode_solver: RK4, with the
    PolynomialStaticLoad's kinks corrected for wherever the env id's own load has them) -- except Euler for BASELINE config 2, which
    names it; True / False: RK4Solver(split_kinks=...) by name."""
    kw = {}
    if w.get("const_speed") is not None:
        kw["load"] = ga.ConstantSpeedLoad(omega_fixed=w["const_speed"])
    if error_controlled:
        kw["ode_solver"] = ga.ScipyOdeSolver()
    elif w["solver"] == "euler":
        kw["ode_solver"] = ga.EulerSolver()
    elif split_kinks is not None:
        kw["ode_solver"] = ga.RK4Solver(split_kinks=split_kinks)
    return ga.make(w["env_id"], n_envs=n_envs, device=device, tau=w["tau"], **kw)


def make_actions(torch, ps, K, n, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    if ps._discrete:
        return torch.randint(0, 8, (K, n), device=device, generator=g, dtype=torch.uint8)
    return torch.rand((K, n, ps._n_act), device=device, generator=g, dtype=torch.float32) * 2 - 1


class Timed:
    """wall seconds (max over ranks), mean device ms per launch (HIP events on the launch stream)."""

    def __init__(self, wall, launch_ms):
        self.wall, self.launch_ms = wall, launch_ms


def median_of(ts):
    """the Timed with the median wall time of a list of repeats"""
    return sorted(ts, key=lambda t: t.wall)[len(ts) // 2]


def measure(torch, dist, env, n_local, steps, warmup, spl, device, world, seed, gather="off", gd=None, settle_ms=0.0, repeats=1):
    """`warmup` untimed + `steps` timed launches of `spl` control steps each, after `settle_ms` of the same launches (clock governor, see
    module docstring).  gather: off | chunk | step.  repeats > 1: the timed region (exactly `steps` launches between barrier +
    synchronize on both sides) is run that many times back to back; returns the list of Timed (repeats == 1: the one Timed)."""
    ps = env.physical_system
    n_act_bufs = max(1, min(4, steps))
    acts = make_actions(torch, ps, spl * n_act_bufs, n_local, device, seed)
    obs = torch.empty((spl, n_local, ps._n_out), dtype=torch.float32, device=device)
    done = torch.empty((spl, n_local), dtype=torch.uint8, device=device)
    gbuf = None
    if gather == "chunk":  # the gathered chunk [W, K, n_local, S_out] (+ done bytes), allocated once
        wg = dist.get_world_size()
        gbuf = (torch.empty((wg,) + tuple(obs.shape), dtype=obs.dtype, device=device), torch.empty((wg,) + tuple(done.shape), dtype=done.dtype, device=device))
    env.reset()
    # one pre-bound launcher per action chunk (PhysicalSystem.bind_rollout: the tensors are checked and their pointers taken ONCE; a
    # call is the gemx_rollout FFI call).  rollout()'s per-call argument handling is 8-12 us of Python -- more than half of what a
    # 20-us launch of BASELINE config 2 takes on the device, and the host set that leg's pace (round 4: a kernel 1.7 us shorter did not
    # move it, profiles/r04t_tail_pipe2.txt)
    bound = [env.bind_rollout(acts[j * spl : (j + 1) * spl], obs, done) for j in range(n_act_bufs)]

    def launch(i):
        if gather == "step":  # batched return after EVERY control step: one launch + one all-gather per step
            a0 = (i % n_act_bufs) * spl
            for k in range(spl):
                o = ps.simulate(acts[a0 + k])
                gd.gather_observations(o, ps.done, force=True)
        else:
            bound[i % n_act_bufs]()
            if gather == "chunk":
                gd.gather_rollout(obs, done, force=True, out=gbuf)

    if settle_ms > 0:
        t_end, i = time.perf_counter() + settle_ms * 1e-3, 0
        while time.perf_counter() < t_end:
            for _ in range(4):  # (plain launches: a wall-clock-bounded loop must not contain collectives)
                bound[i % n_act_bufs]()
                i += 1
            torch.cuda.synchronize()
    # the closed-loop rate limiter calibrates during a handle's first paced launches (30 per bracket, up to five brackets): the timed
    # region starts when it has settled (bounded: 400 more launches)
    for _ in range(50):
        if "limiter calibrating" not in ps.last_launch():
            break
        for j in range(8):
            bound[j % n_act_bufs]()
        torch.cuda.synchronize()
    for i in range(warmup):
        launch(i)
    out = []
    pg = dist.is_available() and dist.is_initialized()  # (also a world of ONE under torchrun: the same barriers / all-reduce as any N)
    for _ in range(max(1, repeats)):
        torch.cuda.synchronize()
        if pg:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(steps):
            launch(i)
        e1.record()
        torch.cuda.synchronize()
        if pg:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if pg:
            if dist.get_backend() == "gloo":
                t = torch.tensor([dt], dtype=torch.float64)
            else:
                t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        out.append(Timed(dt, e0.elapsed_time(e1) / steps))
    if gather != "step":
        assert torch.isfinite(obs).all()
    return out if repeats > 1 else out[0]


def measure_sustained(torch, env, n_local, spl, device, seed, seconds, tele, launch_ms_hint):
    """>= `seconds` of back-to-back launches (no host synchronisation inside), clocks / power sampled meanwhile: does the rate of the
    3-ms timed window hold when the chip has time to reach its power / thermal limits?"""
    ps = env.physical_system
    acts = make_actions(torch, ps, spl * 2, n_local, device, seed)
    obs = torch.empty((spl, n_local, ps._n_out), dtype=torch.float32, device=device)
    done = torch.empty((spl, n_local), dtype=torch.uint8, device=device)
    env.reset()
    n = max(8, int(math.ceil(seconds / (launch_ms_hint * 1e-3))))
    bound = [env.bind_rollout(acts[j * spl : (j + 1) * spl], obs, done) for j in range(2)]  # (see measure())
    for i in range(8):
        bound[i % 2]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with Sampler(tele) as smp:
        t0 = time.perf_counter()
        e0.record()
        for i in range(n):
            bound[i % 2]()
        e1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt, e0.elapsed_time(e1) / n, n, smp.summary()


def measure_traffic_pmc(args, workload, n_local, timeout_s=150):
    """HBM bytes per launch of the dominant kernel, MEASURED IN THIS RUN: two child runs of this script under `rocprofv3 --pmc`
    (FETCH_SIZE and WRITE_SIZE in separate passes -- they do not fit one: 3 + 2 TCC slots), per the guide's HBM section:
    counters in KiB, FETCH_SIZE doubled on gfx950 (it tallies 128-byte requests at 64 bytes), WRITE_SIZE as reported.
    Returns (bytes per launch | None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="gemx_pmc_", dir="/tmp")
        cmd = [rp, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--no-extras", "--no-pmc", "--workload", workload, "--steps", "3", "--warmup", "1", "--settle-ms", "0", "--repeats", "1",
               "--steps-per-launch", str(args.steps_per_launch), "--envs-per-gpu", str(n_local), "--gather", "off"]
        try:
            cenv = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
            subprocess.run(cmd, cwd="/tmp", env=dict(cenv, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            acc = {}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    kn = r["Kernel_Name"]
                    if ("advance" in kn or "dc_stream" in kn) and r["Counter_Name"] == counter:
                        a = acc.setdefault(kn, [0.0, set()])
                        a[0] += float(r["Counter_Value"])
                        a[1].add(r["Dispatch_Id"])
            if not acc:
                return None, f"no {counter} rows for the stepping kernels (rocprofv3 pass failed?)"
            kn = max(acc, key=lambda k: acc[k][0])  # the dominant kernel
            vals[counter] = acc[kn][0] / max(1, len(acc[kn][1]))
        except Exception as e:
            return None, f"{counter} pass failed: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate "
            "passes, 4 launches each), (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch")


def measure_single_step(torch, env, n_local, K, W, device, seed, graph_steps=0, settle_ms=0.0, bound=False):
    """One gemx_step launch per control step.  graph_steps > 0: `graph_steps` launches captured into ONE HIP graph and replayed.
    bound: through PhysicalSystem.bind_step() -- one action buffer the 'policy' writes into (here: not at all), a pre-bound FFI call."""
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
    elif bound:
        step, _, _ = ps.bind_step(acts_all[0])
        for _ in range(64):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(K):
            step()
        e1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = K
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
                                "same_host": False, "host": ref.get("host"), "env_steps_per_s": ref.get("results", {}).get(w["env_id"])}
        except Exception as e:  # a malformed record must not kill the bench line
            out["reference"] = {"error": repr(e)}
    return out


def roofline_of(w, n_local, spl, launch_ms, kernel_desc, workload_key, traffic=None, traffic_source=None, events_ms=None):
    """launch_ms: the time per launch the roofline fraction is priced with.  Since round 4 that is the SAME clock `value` uses -- wall time
    of the timed region / launches (max over ranks) -- so that value x bytes_per_env_step == roofline.achieved; the mean from the HIP
    events on the launch stream is reported beside it (`launch_ms_hip_events`; it excludes the host's barrier / synchronise tail, and
    the two differ by ~1 % on a 3-ms region)."""
    b_step = bytes_per_env_step_fused(w)
    launch_bytes = n_local * (spl * b_step + 2 * 4 * w["s_ode"])
    achieved = launch_bytes / (launch_ms * 1e-3) / 1e9
    if traffic is None:  # not measured in this run: the last builder-run collection (tools/update_hbm_traffic.py)
        tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"{workload_key}:{n_local}:{spl}")
                traffic_source = "profiles/hbm_traffic.json (rocprofv3 --pmc passes of an earlier collection)" if traffic is not None else None
            except Exception:
                traffic = None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "peak_measured_copy": HBM_MEASURED_COPY_GBPS, "frac_of_measured": achieved / HBM_MEASURED_COPY_GBPS,
            "traffic": traffic, "traffic_source": traffic_source, "kernel": kernel_desc, "launch_ms": launch_ms,
            "launch_ms_hip_events": events_ms, "clock": "wall time of the timed region / launches (the clock `value` uses)",
            "algorithmic_bytes_per_launch": launch_bytes, "bytes_per_env_step": b_step}


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
    dist_on = world > 1 or args.force_dist
    if dist_on and not dist.is_initialized():
        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise SystemExit("bench.py: WORLD_SIZE > 1 without MASTER_PORT (torchrun sets it; --gpus N without torchrun picks a free port)")
            from gym_electric_motor_amd.distributed import free_port

            os.environ["MASTER_PORT"] = str(free_port())  # a world of one talks to itself
        # a rank that dies leaves the others with an exception after --dist-timeout seconds instead of a hang in a barrier
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.dist_timeout))

    w = dict(WORKLOADS[args.workload], key=args.workload)
    # the same shard on every GPU at every N (weak scaling: per-GPU work fixed); BASELINE config 5 (8 x 32768 envs) is measured beside it
    # at every N > 1 as `config5` (rounds 1-2 made it the N = 8 default, which mixed a change of shard size into the driver's scaling curve)
    n_local = args.envs_per_gpu or w["envs"]
    n_total = n_local * world
    K, W, spl = args.steps, args.warmup, args.steps_per_launch

    S = args.settle_ms
    tele = Telemetry(dev_index)
    env = make_env(ga, w, n_local, dev_index)  # the solver make(env_id) hands out (Finite-CC-PMSM-v0: plain RK4; --workload scim: RK4 + kink correction)
    t_cold = measure(torch, dist, env, n_local, K, W, spl, device, world, seed=1234 + rank) if S > 0 else None  # straight from idle
    tele_before = tele.sample()
    reps = measure(torch, dist, env, n_local, K, W, spl, device, world, seed=1234 + rank, settle_ms=S, repeats=max(1, args.repeats))
    reps = reps if isinstance(reps, list) else [reps]
    tele_after = tele.sample()
    t = median_of(reps)
    kernel_desc = env.physical_system.last_launch()

    def rl(tm, traffic=None, source=None):
        return roofline_of(w, n_local, spl, tm.wall / K * 1e3, kernel_desc, args.workload, traffic=traffic, traffic_source=source, events_ms=tm.launch_ms)

    out = None
    if rank == 0:  # the headline is complete BEFORE any optional leg runs: nothing below can lose `value`
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
                       "backend": (backend if dist_on else None), "oversubscribed": bool(args.oversubscribe and world > ndev)},
            "roofline": rl(t),
            "repeats": {"n": len(reps), "note": f"the timed region ({K} launches) run {len(reps)} times back to back; value / ms_per_step / roofline = the MEDIAN region",
                        "values": [n_total * spl * K / r.wall for r in reps], "launch_ms": [r.wall / K * 1e3 for r in reps],
                        "launch_ms_hip_events": [r.launch_ms for r in reps],
                        "roofline_frac_min": min(rl(r)["frac"] for r in reps), "roofline_frac_max": max(rl(r)["frac"] for r in reps)},
            "telemetry": dict(tele.describe(), before=tele_before, after=tele_after),
        }
        if t_cold is not None:
            out["cold_start"] = {"value": n_total * spl * K / t_cold.wall, "unit": "env-steps/s", "ms_per_step": t_cold.wall / K * 1e3,
                                 "roofline_frac": rl(t_cold)["frac"],
                                 "note": f"the same {W} + {K} launches straight from an idle GPU, no clock settling (module docstring)"}

    # ---- optional legs.  Every one is wrapped: a failure becomes an "error" field of the line, never a lost line.
    def guarded(name, fn):
        try:
            return fn()
        except BaseException as e:  # (incl. SystemExit / KeyboardInterrupt from a library: the line must still be printed)
            if isinstance(e, KeyboardInterrupt):
                raise
            import traceback

            return {"error": repr(e), "leg": name, "trace": traceback.format_exc(limit=4)}

    if args.workload == "pmsm" and args.envs_per_gpu is None and (args.config5 == "on" or (args.config5 == "auto" and world > 1)):
        def c5():
            env_s = make_env(ga, w, 32768, dev_index)
            try:
                ts = measure(torch, dist, env_s, 32768, K, W, spl, device, world, seed=1234 + rank, settle_ms=S)
            finally:
                env_s.close()
            r5 = roofline_of(w, 32768, spl, ts.wall / K * 1e3, "", args.workload, events_ms=ts.launch_ms)
            return {"envs_per_gpu": 32768, "envs_total": 32768 * world, "value": 32768 * world * spl * K / ts.wall, "unit": "env-steps/s",
                    "ms_per_step": ts.wall / K * 1e3, "roofline_frac_per_gpu": r5["frac"],
                    "note": "BASELINE.json configs[4] is 8 x 32768 envs: this is world x 32768 (twice the per-GPU shard of the scaling lines; "
                            "the single-GPU figure for that shard is configs.pmsm_c5_shard of the --gpus 1 line)"}
        c5r = guarded("config5", c5)
        if out is not None:
            out["config5"] = c5r

    gmode = args.gather
    if gmode == "auto":  # the batched-return path is part of every multi-GPU line (bounded: a few launches), off on a lone GPU
        gmode = "chunk" if world > 1 else "off"
    if gmode != "off":
        if not dist_on:
            raise SystemExit("bench.py: --gather needs a process group: run under torchrun, with --gpus N > 1, or add --force-dist")
        modes = ["chunk", "step"] if gmode == "both" else [gmode]
        gathered = {}
        for mode in modes:
            def gleg(mode=mode):
                Kg = max(1, min(K, 5)) if mode == "chunk" else max(1, min(K, 2))
                spl_g = spl if mode == "chunk" else min(spl, 200)
                tg = measure(torch, dist, env, n_local, Kg, min(W, 2), spl_g, device, world, seed=4321 + rank, gather=mode, gd=gd, settle_ms=S)
                per_call = n_local * (spl_g if mode == "chunk" else 1) * (4 * w["s_out"] + 1)
                res = {"value": n_total * spl_g * Kg / tg.wall, "unit": "env-steps/s", "steps": Kg, "steps_per_launch": spl_g,
                       "ms_per_step": tg.wall / Kg * 1e3, "bytes_gathered_per_rank_per_call": per_call * world,
                       "collective": f"{backend} all_gather_into_tensor, world {world}"}
                if mode == "chunk":  # the collective alone, on this launch's real outputs
                    res["rccl"] = collective_alone(torch, dist, gd, env, n_local, spl_g, device, world, backend, per_call)
                return res
            gathered[mode] = guarded("gather:" + mode, gleg)
        if out is not None:
            out["gather"] = gathered
            ch = gathered.get("chunk")
            out["rccl"] = ch.get("rccl") if isinstance(ch, dict) and "rccl" in ch else {"error": (ch or {}).get("error", "chunk gather leg did not run")}
    env.close()

    if rank == 0:
        if not args.no_pmc and world == 1:
            tr, note = guarded_pair(lambda: measure_traffic_pmc(args, args.workload, n_local))
            if tr is not None:
                out["roofline"] = rl(t, traffic=tr, source=note)
            else:
                out["roofline"]["traffic_source"] = f"{out['roofline'].get('traffic_source')}; in-run PMC pass unavailable: {note}"
        if not args.no_extras and world == 1:
            try:
                extras(torch, dist, ga, args, w, n_local, spl, device, dev_index, out, tele)
            except Exception as e:  # an informational leg must not cost the headline
                import traceback

                out["extras_error"] = {"error": repr(e), "trace": traceback.format_exc(limit=6)}
        elif not args.no_extras:
            # N > 1: the CPU figure does not depend on the number of GPUs -- the same port on rank 0's host cores, on a shorter sample (the
            # other ranks wait at the shutdown barrier meanwhile; round 4 left this field None on every multi-GPU line)
            cb = guarded("cpu_baseline", lambda: cpu_baseline(w, budget_s=4.0))
            if isinstance(cb, dict) and "error" not in cb:
                cb["note"] = f"timed on rank 0's host while the other {world - 1} rank(s) wait; identical at every N"
            out["cpu_baseline"] = cb
        out["overrides"] = {k: v for k, v in os.environ.items() if k.startswith("GEMX_") and k != "GEMX_COVERAGE_FILE"}  # A/B switches active in THIS run (normally none)
        line = emit(out, args, defer=dist_on)

    if dist_on:
        # RCCL writes to the C library's stdout ("Librccl path : ..."), block-buffered on a pipe and flushed when the process EXITS -- i.e.
        # after a line printed here: the driver's "last stdout line" of every multi-GPU run would be RCCL's.  So with a process group the
        # line goes out AFTER the teardown, behind a flush of the C streams; a watchdog prints it anyway if the teardown hangs.
        timer = None
        if rank == 0:
            import threading

            def late():
                flush_c_streams()
                print(line, flush=True)
                os._exit(0)

            timer = threading.Timer(60.0, late)
            timer.daemon = True
            timer.start()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:
            print(f"bench.py: rank {rank}: shutdown barrier failed: {e!r}", file=sys.stderr)
        if rank == 0:
            timer.cancel()
            flush_c_streams()
            print(line, flush=True)


LINE_LIMIT = 6000  # bytes of the ONE stdout line (round 5's 20-KB line was more than the driver parses: BENCH_r05.json parsed = null)
LEGS = ("permexdc", "scim", "scim_constspeed", "scim_plain_rk4", "scim_error_controlled", "scim_device_actions", "pmsm_c5_shard")


def _r(x, nd=4):
    """numbers of the stdout line: 4-5 significant digits are what the measurement carries"""
    if isinstance(x, float):
        return float(f"{x:.{nd + 1}g}") if x == x and abs(x) != float("inf") else None
    return x


def _cut(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def compact_line(out, extras_file=None):
    """The stdout line: the contract's keys + `roofline` + `cpu_baseline` + `legs` {name: {frac, launch_ms}}, numbers only, < LINE_LIMIT
    bytes whatever the legs returned (tests/test_host_cpu.py asserts it on a full record).  Everything else -- repeats, telemetry,
    sustained seconds, cold start, single-step legs, notes, tracebacks of failed legs -- is the side file `extras_file`."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out.get(k) for k in keep}  # (full precision: value x bytes per env-step == roofline.achieved is checked to 1e-9)
    c = out.get("config") or {}
    line["config"] = {"workload": _cut(c.get("workload", ""), 300), **{k: c.get(k) for k in ("env_id", "envs_per_gpu", "solver", "tau", "steps_per_launch",
                                                                                           "parallelism", "world_size", "backend", "oversubscribed") if k in c}}
    ro = out.get("roofline") or {}
    line["roofline"] = {k: ro.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_of_measured", "launch_ms",
                                               "launch_ms_hip_events", "algorithmic_bytes_per_launch", "bytes_per_env_step")}
    line["roofline"]["kernel"] = _cut(ro.get("kernel", ""), 200)
    rp = out.get("repeats") or {}
    if "roofline_frac_min" in rp:
        line["roofline"]["frac_min"], line["roofline"]["frac_max"] = _r(rp["roofline_frac_min"]), _r(rp["roofline_frac_max"])
    s1 = out.get("sustained_1s") or {}
    if "roofline_frac" in s1:
        line["roofline"]["frac_sustained_1s"] = _r(s1["roofline_frac"])
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict) and "error" not in cb:
        cl = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"), "sample": _cut(cb.get("sample", ""), 200)}
        ac = cb.get("all_cores") or {}
        if "value" in ac:
            cl["all_cores"] = {"value": _r(ac["value"]), "cores": ac.get("cores")}
        ref = (cb.get("reference") or {}).get("env_steps_per_s") or {}
        if ref:  # the reference's own Python path, env-steps/s on ONE core (episodic), recorded on the build host: another machine
            cl["reference"] = {"same_host": False, "unit": "env-steps/s on 1 core", **{k: _r(v.get("episodic_1core")) for k, v in ref.items() if isinstance(v, dict)},
                               "all_cores_dopri5": _r((ref.get("dopri5") or {}).get("episodic_all_cores")),
                               "cores": ((cb.get("reference") or {}).get("host") or {}).get("cores")}
        line["cpu_baseline"] = cl
    elif cb is not None:
        line["cpu_baseline"] = {"error": _cut((cb or {}).get("error", "failed"), 200)}
    legs = {}
    for name in LEGS:
        v = (out.get("configs") or {}).get(name)
        if not isinstance(v, dict):
            continue
        if "error" in v:
            legs[name] = {"error": _cut(v["error"], 120)}
            continue
        lm = (v.get("roofline") or {}).get("launch_ms", v.get("launch_ms"))
        legs[name] = {"frac": _r(v.get("frac")), "launch_ms": _r(lm)}
        fl = (v.get("launch_model") or {}).get("frac_of_latency_bound")
        if fl is not None:
            legs[name]["frac_of_latency_bound"] = _r(fl)
    a = out.get("at_scale")
    if isinstance(a, dict) and "frac_of_peak" in a:
        legs["at_scale"] = {"frac": _r(min(a.get("frac_of_peak_repeats") or [a["frac_of_peak"]])), "launch_ms": _r(a.get("launch_ms")), "envs": a.get("envs")}
    if legs:
        line["legs"] = legs
    # multi-GPU lines: the one collective of the path and config 5, numbers only
    g = (out.get("gather") or {}).get("chunk")
    if isinstance(g, dict):
        line["gather"] = {"chunk": {"value": _r(g.get("value"), 6), "ms_per_step": _r(g.get("ms_per_step"))} if "error" not in g else {"error": _cut(g["error"], 120)}}
    rc = out.get("rccl")
    if isinstance(rc, dict):
        line["rccl"] = {k: _r(rc.get(k)) for k in ("world_seen", "backend", "bytes_per_rank", "ms", "GB_per_s", "own_slot_bit_identical") if k in rc} or {"error": _cut(rc.get("error"), 120)}
    c5 = out.get("config5")
    if isinstance(c5, dict):
        line["config5"] = {k: _r(c5.get(k), 6) for k in ("envs_per_gpu", "envs_total", "value", "ms_per_step", "roofline_frac_per_gpu") if k in c5} or {"error": _cut(c5.get("error"), 120)}
    cs = out.get("cold_start")
    if isinstance(cs, dict):
        line["cold_start_value"] = _r(cs.get("value"), 6)
    if out.get("extras_error"):
        line["extras_error"] = _cut(out["extras_error"].get("error"), 160)
    line["overrides"] = {k: _cut(v, 40) for k, v in list((out.get("overrides") or {}).items())[:8]}  # (normally {})
    if extras_file:
        line["extras_file"] = extras_file
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) >= LINE_LIMIT:  # cannot happen with the fields above (every string is cut); if it ever does, the contract's keys still go out
        for k in ("legs", "gather", "rccl", "config5", "overrides", "extras_error", "cold_start_value"):
            line.pop(k, None)
        txt = json.dumps(line, separators=(",", ":"))
    return txt


def flush_c_streams():
    """fflush(NULL): whatever native libraries (RCCL) have buffered on the C stdout goes out NOW, ahead of the line"""
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def emit(out, args, defer=False):
    """Full record -> the side file (`--extras-file`, default bench_extras.json beside this script; a copy under gpurun_out/ when that
    directory exists, so that a gpurun call brings it home); ONE compact line -> stdout, last (defer: returned, the caller prints it
    after the process group's teardown)."""
    path = args.extras_file or os.path.join(REPO, "bench_extras.json")
    written = None
    for p in [path] + ([os.path.join(REPO, "gpurun_out", os.path.basename(path))] if os.path.isdir(os.path.join(REPO, "gpurun_out")) else []):
        try:
            with open(p, "w") as fh:
                json.dump(out, fh, indent=1)
            written = written or os.path.relpath(p, REPO)
        except OSError as e:  # a read-only checkout: the line still goes out
            print(f"bench.py: could not write {p}: {e!r}", file=sys.stderr)
    line = compact_line(out, written)
    if not defer:
        flush_c_streams()
        sys.stdout.flush()
        print(line, flush=True)
    return line


def guarded_pair(fn):
    try:
        return fn()
    except Exception as e:
        return None, repr(e)


def collective_alone(torch, dist, gd, env, n_local, spl, device, world, backend, per_call):
    """The batched-return collective by itself: all-gather of ONE launch's real outputs ([K, n_local, S_out] rows + done bytes) into
    preallocated buffers, timed over a few calls between barriers.  bytes_per_rank = what each rank contributes per call;
    GB_per_s = bus bandwidth per rank, (W - 1) x bytes_per_rank / time (what each rank receives over its xGMI links; a world of one
    moves bytes_per_rank through a device-local copy and reports that instead)."""
    ps = env.physical_system
    obs = torch.empty((spl, n_local, ps._n_out), dtype=torch.float32, device=device)
    done = torch.empty((spl, n_local), dtype=torch.uint8, device=device)
    acts = make_actions(torch, ps, spl, n_local, device, 99)
    env.reset()
    env.rollout(acts, obs_out=obs, done_out=done)
    wg = dist.get_world_size()
    gbuf = (torch.empty((wg,) + tuple(obs.shape), dtype=obs.dtype, device=device), torch.empty((wg,) + tuple(done.shape), dtype=done.dtype, device=device))
    for _ in range(2):
        gd.gather_rollout(obs, done, force=True, out=gbuf)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        gd.gather_rollout(obs, done, force=True, out=gbuf)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    rank = dist.get_rank()
    ok = bool(torch.equal(gbuf[0][rank], obs)) and bool(torch.equal(gbuf[1][rank], done))  # this rank's own slot, bit for bit
    try:
        ver = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception:
        ver = None
    moved = (wg - 1) * per_call if wg > 1 else per_call
    return {"world_seen": wg, "backend": backend, "version": ver, "bytes_per_rank": per_call, "ms": dt * 1e3, "GB_per_s": moved / dt / 1e9,
            "own_slot_bit_identical": ok, "calls": n,
            "note": "all_gather_into_tensor of one launch's observation chunk + done bytes; GB_per_s = (W - 1) x bytes_per_rank / time"
                    + (" (world of one: bytes_per_rank / time, a device-local copy)" if wg == 1 else "")}


def leg(torch, dist, ga, args, key, device, dev_index, spl, tele, envs=None, split_kinks=None, error_controlled=False, steps=10, repeats=3, seed=5,
        sustained=True):
    """One informational leg through the same measurement as the headline: `repeats` timed regions of `steps` launches, MEDIAN reported,
    min / max beside it, clocks / power before and after; then (round 4, every leg) >= --sustain-s seconds of back-to-back launches with
    the sensors sampled -- `frac` = min(window, sustained) is the figure to quote: a 16-ms window can sit above what the chip holds once
    it reaches its power limit.  split_kinks: see make_env (None = the solver make(env_id) hands out)."""
    wc = dict(WORKLOADS[key], key=key)
    n = envs or wc["envs"]
    env = make_env(ga, wc, n, dev_index, split_kinks=split_kinks, error_controlled=error_controlled)
    try:
        # the timed region is priced by the wall clock (barrier / synchronise tail included): make it >= 3 ms of launches, as the headline's
        # 20 x 0.15 ms is -- ten 20-us launches of BASELINE config 2 would be a 0.2-ms region, a tenth of it host synchronisation.  Launches
        # shorter than 50 us: >= 10 ms.  With the pre-bound launcher the host queues such launches five times faster than the device runs
        # them, and the region's start and end (first submission, the wake-up of the final synchronise with ~150 launches queued) came
        # to 185 us of a 2.9-ms region of config 2: 19.6 us per launch on the wall clock, 18.3 between the HIP events (r04t_bench2.txt)
        probe = measure(torch, dist, env, n, 10, 3, spl, device, 1, seed=seed, settle_ms=args.settle_ms, repeats=1)
        steps = max(steps, int(math.ceil((10.0 if probe.launch_ms < 0.05 else 3.0) / max(probe.launch_ms, 1e-3))))
        before = tele.sample()
        reps = measure(torch, dist, env, n, steps, 3, spl, device, 1, seed=seed, settle_ms=args.settle_ms, repeats=repeats)
        reps = reps if isinstance(reps, list) else [reps]
        after = tele.sample()
        desc = env.physical_system.last_launch()
        solver_name = type(env.physical_system._ode_solver).__name__ + ("(split_kinks=True)" if getattr(env.physical_system._ode_solver, "_split_kinks", False) else "")
        tm = median_of(reps)
        ro = lambda r: roofline_of(wc, n, spl, r.wall / steps * 1e3, desc, key, events_ms=r.launch_ms)  # noqa: E731
        fr = [ro(r)["frac"] for r in reps]
        how = (", the solver make(env_id) hands out: " if split_kinks is None and not error_controlled and wc["solver"] != "euler" else ", ") + solver_name
        if error_controlled:
            how += " = error-controlled Dormand-Prince 5(4), rtol 1e-6"
        res = {"workload": wc["desc"] + how, "envs": n, "steps_per_launch": spl, "launches_per_region": steps, "solver": solver_name,
               "value": n * spl * steps / tm.wall, "unit": "env-steps/s", "roofline": ro(tm),
               "repeats": {"n": len(reps), "roofline_frac": fr, "roofline_frac_min": min(fr), "roofline_frac_max": max(fr)},
               "telemetry": {"before": before, "after": after}}
        res["frac"] = min(fr)
        if sustained and args.sustain_s > 0:
            dt, lms, nl, smp = measure_sustained(torch, env, n, spl, device, seed + 8, args.sustain_s, tele, tm.launch_ms)
            fs = roofline_of(wc, n, spl, dt / nl * 1e3, "", key)["frac"]
            res["sustained_1s"] = {"value": n * spl * nl / dt, "launches": nl, "seconds": dt, "launch_ms": dt / nl * 1e3, "launch_ms_hip_events": lms,
                                   "roofline_frac": fs, "telemetry_during": smp}
            res["frac"] = min(res["frac"], fs)
        res["frac_note"] = "min(minimum of the repeated windows, sustained second)"
    finally:
        env.close()
    return res, tm


def extras(torch, dist, ga, args, w, n_local, spl, device, dev_index, out, tele):
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
    r0 = roofline_of(w, n_local, spl, t0.wall / min(args.steps, 10) * 1e3, desc0, args.workload + "/nolinmap", events_ms=t0.launch_ms)
    out["headline_no_linmap"] = {"value": n_local * spl * min(args.steps, 10) / t0.wall, "unit": "env-steps/s", "launch_ms": t0.launch_ms,
                                 "achieved_GBps": r0["achieved"], "frac_of_peak": r0["frac"], "env": "GEMX_LINMAP=0"}
    # does the 3-ms window hold for a second?  (clocks / power sampled every 20 ms while the launches run)
    envs_ = make_env(ga, w, n_local, dev_index)
    dt, lms, n, smp = measure_sustained(torch, envs_, n_local, spl, device, 11, args.sustain_s, tele, out["roofline"]["launch_ms"])
    envs_.close()
    rs = roofline_of(w, n_local, spl, dt / n * 1e3, out["roofline"]["kernel"], args.workload)
    out["sustained_1s"] = {"value": n_local * spl * n / dt, "unit": "env-steps/s", "launches": n, "seconds": dt, "launch_ms": dt / n * 1e3,
                           "launch_ms_hip_events": lms, "roofline_frac": rs["frac"], "telemetry_during": smp,
                           "note": f">= {args.sustain_s} s of back-to-back launches of the headline workload, no host synchronisation inside"}
    # closed-loop usage: one launch per control step, eager and from a HIP graph
    b1 = bytes_per_env_step_single(w)
    env1 = make_env(ga, w, n_local, dev_index)
    host_s, dev_ms, n1 = measure_single_step(torch, env1, n_local, 2000, 100, device, seed=99, settle_ms=args.settle_ms)
    out["single_step"] = {"value": n_local / host_s, "unit": "env-steps/s", "ms_per_step": host_s * 1e3, "device_ms_per_step": dev_ms,
                          "achieved_GBps": n_local * b1 / (dev_ms * 1e-3) / 1e9, "bytes_per_env_step": b1, "steps": n1,
                          "note": "one gemx_step launch per control step, eager (PhysicalSystem.simulate on a device tensor)"}
    host_s, dev_ms, n1 = measure_single_step(torch, env1, n_local, 2000, 100, device, seed=99, settle_ms=args.settle_ms, bound=True)
    out["single_step_bound"] = {"value": n_local / host_s, "unit": "env-steps/s", "ms_per_step": host_s * 1e3, "device_ms_per_step": dev_ms, "steps": n1,
                                "note": "one gemx_step launch per control step through PhysicalSystem.bind_step(): the policy writes into one "
                                        "action buffer, a call is the pre-bound FFI call and nothing else"}
    host_s, dev_ms, n1 = measure_single_step(torch, env1, n_local, 4096, 100, device, seed=99, graph_steps=64, settle_ms=args.settle_ms)
    out["single_step_graph"] = {"value": n_local / host_s, "unit": "env-steps/s", "ms_per_step": host_s * 1e3, "device_ms_per_step": dev_ms,
                                "achieved_GBps": n_local * b1 / (dev_ms * 1e-3) / 1e9, "bytes_per_env_step": b1, "steps": n1,
                                "note": "64 gemx_step launches captured into one HIP graph (torch.cuda.CUDAGraph) and replayed"}
    env1.close()
    # BASELINE configs 2, 4 and 5's per-GPU shard through the same measurement (3 repeats each: median, min, max; then a sustained second)
    out["configs"] = {}
    for key in ("permexdc", "scim", "scim_constspeed"):
        if key == args.workload:
            continue
        # `scim` = BASELINE config 4 with the solver make("Cont-SC-SCIM-v0") hands out (RK4 + kink correction); plain RK4 beside it
        out["configs"][key], tm = leg(torch, dist, ga, args, key, device, dev_index, spl, tele)
        if key == "scim":
            out["configs"]["scim_plain_rk4"], _ = leg(torch, dist, ga, args, key, device, dev_index, spl, tele, split_kinks=False)
            out["configs"]["scim_split_kinks"] = {"same_as": "configs.scim", "frac": out["configs"]["scim"]["frac"],
                                                  "roofline": out["configs"]["scim"]["roofline"],
                                                  "note": "rounds 2-3 reported make()'s solver under this key and plain RK4 as `scim`; since round 4 "
                                                          "`scim` IS make()'s solver and plain RK4 is `scim_plain_rk4`"}
            # ... and with the reference default's semantics on the device (GEMX_SOLVER_ADAPTIVE: every lane cuts its own steps)
            out["configs"]["scim_error_controlled"], _ = leg(torch, dist, ga, args, key, device, dev_index, spl, tele, error_controlled=True, steps=5, sustained=False)
        if key == "permexdc":
            out["configs"][key]["launch_model"] = launch_model(torch, dist, ga, args, key, device, dev_index, out["configs"][key])
    # "actions ... can be generated on-device".  Config 4's launches once more with the loader wave GENERATING the random
    # duty cycles (gemx_rollout_synthetic: no action tensor is read; 57 instead of 69 algorithmic bytes per env-step).  An extra beside
    # `configs.scim`, whose actions stay a tensor in HBM as a policy's would be.
    def device_actions_leg():
        wc = dict(WORKLOADS["scim"], key="scim")
        env_s = make_env(ga, wc, wc["envs"], dev_index)
        try:
            ps = env_s.physical_system
            obs = torch.empty((spl, wc["envs"], ps._n_out), dtype=torch.float32, device=device)
            done = torch.empty((spl, wc["envs"]), dtype=torch.uint8, device=device)
            t_end = time.perf_counter() + args.settle_ms * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(4):
                    ps.rollout_synthetic(spl, seed=11, step0=0, obs_out=obs, done_out=done)
                torch.cuda.synchronize()
            walls = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(10):
                    ps.rollout_synthetic(spl, seed=11, step0=i * spl, obs_out=obs, done_out=done)
                torch.cuda.synchronize()
                walls.append((time.perf_counter() - t0) / 10)
            assert torch.isfinite(obs).all()
            wall = sorted(walls)[1]
            b_step = 4 * wc["s_out"] + 1  # no action bytes
            gbs = wc["envs"] * (spl * b_step + 2 * 4 * wc["s_ode"]) / wall / 1e9
            return {"workload": wc["desc"] + ", actions generated on the device (gemx_rollout_synthetic)", "envs": wc["envs"], "steps_per_launch": spl,
                    "value": wc["envs"] * spl / wall, "unit": "env-steps/s", "launch_ms": wall * 1e3, "bytes_per_env_step": b_step,
                    "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBPS, "frac_of_measured": gbs / HBM_MEASURED_COPY_GBPS, "kernel": ps.last_launch(),
                    "note": "informational: BASELINE config 4 stays `configs.scim` (actions = a tensor resident in HBM)"}
        finally:
            env_s.close()
    try:
        out["configs"]["scim_device_actions"] = device_actions_leg()
    except Exception as e:  # (an informational leg must not cost the line)
        out["configs"]["scim_device_actions"] = {"error": repr(e)}
    if args.workload == "pmsm":  # BASELINE config 5 = 8 x 32768 envs: its shard on this one GPU
        out["configs"]["pmsm_c5_shard"], _ = leg(torch, dist, ga, args, "pmsm", device, dev_index, spl, tele, envs=32768)
    # the headline kernel with the chip full
    n_big, c_big = 2 ** 20, 100
    envb = make_env(ga, w, n_big, dev_index)
    n_reg = 12  # launches per timed region: ~12 ms, four rotating action chunks of 100 MB each (more than the 256 MB Infinity Cache holds: the
    #             actions are read from HBM, as they would be behind a policy; one re-read chunk runs 4-5 % faster, tools/bench_matrix.py)
    tb = measure(torch, dist, envb, n_big, n_reg, 2, c_big, device, 1, seed=7, settle_ms=args.settle_ms, repeats=3)
    envb.close()
    bb = n_big * (c_big * bytes_per_env_step_fused(w) + 2 * 4 * w["s_ode"])
    tbm = median_of(tb)
    fr = [bb / (r.wall / n_reg) / 1e9 / HBM_PEAK_GBPS for r in tb]
    out["at_scale"] = {"envs": n_big, "steps_per_launch": c_big, "launches_per_region": n_reg, "value": n_big * n_reg * c_big / tbm.wall, "unit": "env-steps/s",
                       "launch_ms": tbm.wall / n_reg * 1e3, "launch_ms_hip_events": tbm.launch_ms, "achieved_GBps": bb / (tbm.wall / n_reg) / 1e9,
                       "frac_of_peak": bb / (tbm.wall / n_reg) / 1e9 / HBM_PEAK_GBPS, "frac_of_peak_repeats": fr,
                       "telemetry_after": tele.sample()}


def launch_model(torch, dist, ga, args, key, device, dev_index, leg_out):
    """BASELINE config 2 is not bandwidth bound: 4096 envs are 64 workgroups on 256 CUs, and a launch lasts K times what ONE wave needs
    per control step plus a fixed part.  Fit t_launch(K) = t_fixed + K t_step over four launch lengths and put the
    result beside two floors: the step-to-step dependency chain of the integrator wave (3 dependent VALU instructions, 14.1 cycles at
    2.4 GHz: tools/microbench_chain.hip) and the time the algorithmic bytes take at the HBM peak."""
    wc = dict(WORKLOADS[key], key=key)
    pts = []
    for k in (250, 500, 1000, 2000):
        env = make_env(ga, wc, wc["envs"], dev_index)
        r = measure(torch, dist, env, wc["envs"], 10, 3, k, device, 1, seed=5, settle_ms=args.settle_ms, repeats=3)
        env.close()
        pts.append((k, median_of(r).launch_ms * 1e3))
    n = len(pts)
    sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
    sxx, sxy = sum(p[0] * p[0] for p in pts), sum(p[0] * p[1] for p in pts)
    t_step = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    t_fixed = (sy - t_step * sx) / n
    spl = leg_out["steps_per_launch"]
    t_meas = (leg_out["roofline"]["launch_ms_hip_events"] or leg_out["roofline"]["launch_ms"]) * 1e3  # (HIP events, like the fit's points)
    chain_ns = 14.1 / 2.4
    t_chain = spl * chain_ns * 1e-3 + t_fixed
    t_hbm = leg_out["roofline"]["algorithmic_bytes_per_launch"] / (HBM_PEAK_GBPS * 1e3)
    return {"fit_us": {"t_fixed": t_fixed, "t_step": t_step, "points": pts}, "unit": "microseconds (t_launch = t_fixed + K * t_step, least squares)",
            "floor_dependency_chain_us": t_chain, "floor_hbm_peak_us": t_hbm, "measured_us": t_meas,
            "frac_of_latency_bound": t_chain / t_meas,
            "note": "frac_of_latency_bound = (K x 5.9 ns of the integrator's 3-instruction dependency chain + the fitted fixed part) / measured launch time"}


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
    ap.add_argument("--gather", choices=["auto", "off", "chunk", "step", "both"], default="auto",
                    help="also time the batched-return path: all-gather of each launch's observation chunk / of every step's rows; "
                         "auto (default) = a bounded `chunk` leg whenever the world is larger than one rank, off on a lone GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even for ONE rank (implied when WORLD_SIZE is in the environment): the barriers, the "
                         "max-over-ranks all-reduce and --gather then run through RCCL exactly as in a multi-GPU run")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than visible GPUs (ranks share GPUs, gloo control plane): functional check only")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed launches of the same workload for this long before each leg's warm-up (clock governor; 0 = off)")
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline / single_step / configs / at_scale legs")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps launches each; the median is reported, min / max beside it")
    ap.add_argument("--sustain-s", type=float, default=1.0, help="length of the sustained legs (back-to-back launches, clocks / power sampled)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc child runs (N = 1 only)")
    ap.add_argument("--config5", choices=["auto", "on", "off"], default="auto",
                    help="also measure BASELINE config 5's shard (32768 envs per GPU) on every rank and report it as `config5` (auto: at --gpus 8)")
    ap.add_argument("--extras-file", default=None, help="where the full record goes (default: bench_extras.json beside this script); stdout carries "
                                                         "one compact line (< 6 KB)")
    ap.add_argument("--dist-timeout", type=float, default=300.0, help="torch.distributed timeout in seconds (rendezvous, barriers, collectives)")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0 or args.steps_per_launch < 2 or args.settle_ms < 0:
        raise SystemExit("bench.py: --steps >= 1, --warmup >= 0, --steps-per-launch >= 2, --settle-ms >= 0")

    world_env = os.environ.get("WORLD_SIZE")
    backend = "gloo" if args.oversubscribe else "nccl"
    if world_env is not None:  # launched by torchrun / the driver: one rank per process already
        world = int(world_env)
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
        args.force_dist = True  # a launcher-made world, even of one rank, gets its process group
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
        from gym_electric_motor_amd.distributed import free_port

        port = free_port()  # (not a fixed default: two jobs on one node would meet on it)
        mp.spawn(_spawned, args=(args, args.gpus, port, backend), nprocs=args.gpus, join=True)


if __name__ == "__main__":
    main()
