"""EVERY compiled stepping-kernel instantiation is launched and compared, bit for bit, with another kernel of the same handle configuration.

Round 5's verdict: 2218 kernels compiled, no record of which any test launches.  The coverage log (GEMX_COVERAGE_FILE, tools/
instantiation_coverage.py) showed 473 of 2455 -- so the library was pruned to the instantiations the dispatcher can reach
(csrc/gemx_kernels.hpp: pipe_kernel_of, launch_advance_unit) and this file walks ALL of them: for each of the 19 (system, converter)
units x 2 loads x 3 solvers x with / without converter dead time, in fp32

    advance_kernel          single-wave kernel (GEMX_PIPE=0)                    = the reference result of the configuration
    advance_pipe_kernel     <4, 2>; <12, 3>, <12, 6>, <2, 2> where built (GEMX_PIPE_SHAPE); FULL (RC supply); FULL + SLOW (solver
                            sub-steps: compared with the single-wave kernel at the same sub-steps); FULL + RINIT (random initial states)
    step_kernel             one launch per control step
    dc_stream_kernel        DC machines behind a constant-speed load, 32 and 64 envs per workgroup
    linmap_kernel           whenever a one-step map is built

and in fp64 advance_kernel against step_kernel.  Observation rows, done bytes and the final ODE state must be identical (the documented
property of the library: every kernel / shape / chunking gives the same bits).  The coverage record of this file alone reaches every
kernel of every unit library: profiles/r06_instantiation_coverage.md."""
import os

import pytest

pytestmark = pytest.mark.gpu

# unit -> (env id, make kwargs).  Units 10 / 11 are the internal dq kinds of the continuous B6 converters (control_space='dq', the
# DqToAbcActionProcessor folded into the kernel)
UNITS = {
    "0_0": ("Cont-CC-PermExDc-v0", {}), "0_3": ("Finite-CC-PermExDc-v0", {}),
    "1_1": ("Finite-CC-PMSM-v0", {}), "1_2": ("Cont-CC-PMSM-v0", {}), "1_10": ("Cont-CC-PMSM-v0", {"control_space": "dq"}),
    "2_1": ("Finite-CC-SCIM-v0", {}), "2_2": ("Cont-CC-SCIM-v0", {}), "2_10": ("Cont-CC-SCIM-v0", {"control_space": "dq"}),
    "3_0": ("Cont-CC-SeriesDc-v0", {}), "3_3": ("Finite-CC-SeriesDc-v0", {}),
    "4_0": ("Cont-CC-ShuntDc-v0", {}), "4_3": ("Finite-CC-ShuntDc-v0", {}),
    "5_4": ("Cont-CC-ExtExDc-v0", {}), "5_5": ("Finite-CC-ExtExDc-v0", {}),
    "6_6": ("Cont-CC-EESM-v0", {}), "6_7": ("Finite-CC-EESM-v0", {}), "6_11": ("Cont-CC-EESM-v0", {"action_frame": "dq_processor"}),
    "7_8": ("Cont-CC-DFIM-v0", {}), "7_9": ("Finite-CC-DFIM-v0", {}),
}
N, K = 192, 26  # three workgroups; 26 = two blocks of twelve + a partial one = six of four + a partial one


def _solver(ga, name, nsteps=1):
    return {"euler": ga.EulerSolver, "rk4": ga.RK4Solver, "dp5": ga.DormandPrince5Solver}[name](nsteps=nsteps)


def _load(ga, kind):
    return ga.ConstantSpeedLoad(omega_fixed=60.0) if kind == "const" else ga.PolynomialStaticLoad(load_parameter=dict(a=0.01, b=0.01, c=0.0))


def _make(ga, unit, load, solver, il, dtype="float32", nsteps=1, **extra):
    env_id, kw = UNITS[unit]
    kw = dict(kw, **extra)
    if il:
        kw["converter"] = dict(interlocking_time=1e-6)
    return ga.make(env_id, n_envs=N, tau=1e-4, load=_load(ga, load), ode_solver=_solver(ga, solver, nsteps), dtype=dtype, **kw)


def _actions(torch, ps, seed=5):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if ps._discrete:
        n_act = int(getattr(ps.action_space, "n", 0) or 0)
        if not n_act:  # MultiDiscrete: the flat index
            n_act = 1
            for v in ps.action_space.nvec:
                n_act *= int(v)
        return torch.randint(0, n_act, (K, N), device="cuda", generator=g, dtype=torch.uint8)
    return (torch.rand((K, N, ps._n_act), device="cuda", generator=g, dtype=torch.float64) * 2 - 1).to(ps._tdtype)


def _run(torch, env, acts, stepwise=False):
    ps = env.physical_system
    env.reset()
    if stepwise:
        rows, dones = [], []
        for k in range(acts.shape[0]):
            rows.append(ps.simulate(acts[k]).clone())
            dones.append(ps.done.clone())
        obs, done = torch.stack(rows), torch.stack(dones)
    else:
        obs, done = env.rollout(acts)
        obs, done = obs.clone(), done.clone()
    return obs, done, ps.get_state().clone(), ps.last_launch()


def _same(torch, a, b, what):
    assert torch.equal(a[0], b[0]), (what, "observations", a[3], b[3], float((a[0].double() - b[0].double()).abs().max()))
    assert torch.equal(a[1].to(torch.uint8), b[1].to(torch.uint8)), (what, "done", a[3], b[3])
    assert torch.equal(a[2], b[2]), (what, "final state", a[3], b[3])


@pytest.mark.parametrize("unit", sorted(UNITS))
@pytest.mark.timeout(900)
def test_every_fp32_instantiation_of_a_unit_is_launched_and_bit_identical(unit, monkeypatch):
    import torch

    import gym_electric_motor_amd as ga

    for k in [k for k in os.environ if k.startswith("GEMX_") and k != "GEMX_COVERAGE_FILE"]:
        monkeypatch.delenv(k)
    sys_kind = int(unit.split("_")[0])
    dc = sys_kind in (0, 3, 4, 5)
    for load in ("const", "poly"):
        for solver in ("euler", "rk4", "dp5"):
            for il in (False, True):
                if il and sys_kind == 6:  # (the EESM refuses converter dead time, as the reference's branch for it cannot execute: no IL code in its units)
                    with pytest.raises(ValueError):
                        _make(ga, unit, load, solver, True)
                    continue
                tag = f"{unit} load={load} solver={solver} il={il}"
                # -- the single-wave kernel: the reference of this configuration
                monkeypatch.setenv("GEMX_PIPE", "0")
                env = _make(ga, unit, load, solver, il)
                acts = _actions(torch, env.physical_system)
                ref = _run(torch, env, acts)
                assert "advance_kernel" in ref[3], (tag, ref[3])
                env.close()
                monkeypatch.delenv("GEMX_PIPE")
                # -- one launch per control step
                env = _make(ga, unit, load, solver, il)
                st = _run(torch, env, acts, stepwise=True)
                assert "step_kernel" in st[3], (tag, st[3])
                _same(torch, ref, st, tag + " step_kernel")
                env.close()
                # -- the pipelined shapes (the dispatcher ignores a forced shape that is not built and takes <4, 2>)
                monkeypatch.setenv("GEMX_DC_STREAM", "0")
                deep = solver == "rk4" and not il and sys_kind != 7  # (the DFIM's 24-value rows leave no LDS for twelve-step blocks)
                built = {1: "D=4", 0: "D=12" if deep else None, 3: "D=12" if deep else None,
                         2: "D=2" if (solver != "euler" and not il) else None}
                for shape, want in built.items():
                    if want is None:
                        continue
                    monkeypatch.setenv("GEMX_PIPE_SHAPE", str(shape))
                    env = _make(ga, unit, load, solver, il)
                    got = _run(torch, env, acts)
                    env.close()
                    assert "advance_pipe_kernel" in got[3], (tag, shape, got[3])
                    if want not in got[3]:  # a deep shape whose LDS footprint does not fit this system takes <4, 2>: it must then not be compiled either
                        pytest.fail(f"{tag}: forced shape {shape} ran {got[3]}")
                    if shape == 3:
                        assert "x 512 threads" in got[3], (tag, got[3])
                    _same(torch, ref, got, f"{tag} shape {shape}")
                monkeypatch.delenv("GEMX_PIPE_SHAPE")
                monkeypatch.delenv("GEMX_DC_STREAM")
                # -- dc_stream_kernel: DC machine, constant speed, no dead time; 32 and 64 envs per workgroup
                if dc and load == "const" and not il:
                    for epw in ("32", "64"):
                        monkeypatch.setenv("GEMX_DCS_EPW", epw)
                        env = _make(ga, unit, load, solver, il)
                        got = _run(torch, env, acts)
                        env.close()
                        assert "dc_stream_kernel" in got[3], (tag, got[3])
                        _same(torch, ref, got, f"{tag} dc_stream epw {epw}")
                    monkeypatch.delenv("GEMX_DCS_EPW")
                # -- FULL: RC supply (finite EESM converter: not available behind an RC supply)
                if unit != "6_7":
                    outs = []
                    for pipe in ("0", "1"):
                        monkeypatch.setenv("GEMX_PIPE", pipe)
                        env = _make(ga, unit, load, solver, il, supply=ga.RCVoltageSupply(supply_parameter=dict(R=0.05, C=2e-3)))
                        outs.append(_run(torch, env, acts))
                        env.close()
                    monkeypatch.delenv("GEMX_PIPE")
                    assert "advance_kernel" in outs[0][3] and "advance_pipe_kernel" in outs[1][3], (tag, outs[0][3], outs[1][3])
                    _same(torch, outs[0], outs[1], tag + " FULL (RC supply)")
                # -- FULL + SLOW: solver sub-steps
                outs = []
                for pipe in ("0", "1"):
                    monkeypatch.setenv("GEMX_PIPE", pipe)
                    env = _make(ga, unit, load, solver, il, nsteps=2)
                    outs.append(_run(torch, env, acts))
                    env.close()
                monkeypatch.delenv("GEMX_PIPE")
                assert "advance_kernel" in outs[0][3] and "advance_pipe_kernel" in outs[1][3], (tag, outs[0][3], outs[1][3])
                _same(torch, outs[0], outs[1], tag + " FULL + SLOW (two sub-steps)")
                # -- FULL + RINIT: random initial states
                outs = []
                for pipe in ("0", "1"):
                    monkeypatch.setenv("GEMX_PIPE", pipe)
                    env = _make(ga, unit, load, solver, il, motor=dict(motor_initializer=dict(random_init="uniform")), seed=3)
                    outs.append(_run(torch, env, acts))
                    env.close()
                monkeypatch.delenv("GEMX_PIPE")
                assert "advance_kernel" in outs[0][3] and "advance_pipe_kernel" in outs[1][3], (tag, outs[0][3], outs[1][3])
                _same(torch, outs[0], outs[1], tag + " FULL + RINIT (random initial states)")


@pytest.mark.parametrize("unit", sorted(UNITS))
def test_every_fp64_instantiation_of_a_unit_is_launched_and_bit_identical(unit):
    """The fp64 diagnostic units: the single-wave kernel against one launch per control step, 2 loads x 3 solvers (the dead-time code is
    always compiled in there)."""
    import torch

    import gym_electric_motor_amd as ga

    for load in ("const", "poly"):
        for solver in ("euler", "rk4", "dp5"):
            env = _make(ga, unit, load, solver, False, dtype="float64")
            acts = _actions(torch, env.physical_system)
            a = _run(torch, env, acts)
            b = _run(torch, env, acts, stepwise=True)
            env.close()
            assert "advance_kernel" in a[3] and "f64" in a[3] and "step_kernel" in b[3], (a[3], b[3])
            _same(torch, a, b, f"{unit} fp64 load={load} solver={solver}")


def test_the_fp64_forms_of_the_small_kernels():
    """libgemx.so's own kernels in their fp64 instantiations (the diagnostic build's state access, synthetic action stream and Wiener
    reference generators): get_state / set_state round trip, the synthetic stream equal to the fp32 one value for value (the stream is
    exact in both types), chunked reference generation equal to one-shot generation."""
    import torch

    import gym_electric_motor_amd as ga

    env = ga.make("Cont-CC-PMSM-v0", n_envs=N, tau=1e-4, dtype="float64")
    ps = env.physical_system
    acts = _actions(torch, ps)
    env.rollout(acts)
    y = ps.get_state().clone()
    twin = ga.make("Cont-CC-PMSM-v0", n_envs=N, tau=1e-4, dtype="float64")
    twin.physical_system.set_state(y)
    assert torch.equal(twin.physical_system.get_state(), y)
    a = env.rollout(acts[:7])
    b = twin.rollout(acts[:7])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    s64 = ps.synthetic_actions(K, seed=4, step0=9)
    e32 = ga.make("Cont-CC-PMSM-v0", n_envs=N, tau=1e-4)
    s32 = e32.physical_system.synthetic_actions(K, seed=4, step0=9)
    assert s64.dtype == torch.float64 and torch.equal(s64, s32.double())
    g1 = ga.BatchedWienerProcessReferenceGenerator(reference_states=("i_sd", "i_sq"), seed=2).set_modules(ps)
    g2 = ga.BatchedWienerProcessReferenceGenerator(reference_states=("i_sd", "i_sq"), seed=2).set_modules(ps)
    g1.reset()
    g2.reset()
    whole = g1.rollout(60)
    parts = torch.cat([g2.rollout(1), g2.rollout(25), g2.rollout(34)])
    assert whole.dtype == torch.float64 and torch.equal(whole, parts)
    for e in (env, twin, e32):
        e.close()


@pytest.mark.parametrize("env_id, n, kw", [("Cont-CC-PMSM-v0", 200, {}), ("Cont-CC-PMSM-v0", 65536, {}), ("Cont-SC-SCIM-v0", 32768 + 64 + 7, {}),
                                            ("Cont-CC-EESM-v0", 4096, {}), ("Cont-CC-DFIM-v0", 1000, {}), ("Cont-CC-PermExDc-v0", 16384, {}),
                                            ("Cont-CC-PMSM-v0", 4096, {"control_space": "dq"}), ("Cont-CC-PMSM-v0", 16384, {"action_delay": 2})])
def test_half_action_tensor_gives_the_bits_of_its_values_fed_as_fp32(env_id, n, kw):
    """gemx_rollout_half (ABI 7): a [K, N, A] float16 action tensor is widened while it is staged -- observations, done bytes and the final
    state are those of `rollout(actions.float())`, bit for bit, in every pipelined shape incl. partial workgroups, the DeadTimeProcessor's
    delayed reads and control_space='dq'; through `rollout()` and through `bind_rollout()`.  What a half costs against unrounded duty
    cycles is the caller's quantisation (2^-11 relative on [-1, 1]); the kernel adds nothing to it."""
    import torch

    import gym_electric_motor_amd as ga

    Kh = 53
    env = ga.make(env_id, n_envs=n, tau=1e-4, **kw)
    ps = env.physical_system
    g = torch.Generator(device="cuda").manual_seed(17)
    a16 = (torch.rand((Kh, n, ps._n_act), device="cuda", generator=g) * 2.2 - 1.1).to(torch.float16)  # (beyond [-1, 1] too: clipped by the converter)
    env.reset()
    ref = env.rollout(a16.float())
    ref = (ref[0].clone(), ref[1].clone(), ps.get_state().clone(), ps.last_launch())
    env.reset()
    got = env.rollout(a16)
    got = (got[0].clone(), got[1].clone(), ps.get_state().clone(), ps.last_launch())
    assert "advance_pipe_kernel" in got[3], got[3]
    _same(torch, ref, got, f"{env_id} n={n} half action tensor")
    obs = torch.empty_like(ref[0])
    done = torch.empty_like(ref[1])
    env.reset()
    launch = env.bind_rollout(a16, obs, done)
    launch()
    assert torch.equal(obs, ref[0]) and torch.equal(done, ref[1])
    with pytest.raises(ValueError):  # a single step has no fused launch to widen in
        env.rollout(a16[:1])
    env.close()


def test_half_action_tensor_is_refused_where_it_has_no_meaning():
    """The C entry point refuses what it cannot serve, with a message: discrete converters (one byte per action already), fp64 handles,
    a single step (no fused launch to widen in)."""
    import ctypes as C

    import torch

    import gym_electric_motor_amd as ga
    from gym_electric_motor_amd import _lib

    L = _lib.load()
    buf = torch.zeros(8 * 128 * 3, dtype=torch.float16, device="cuda")
    obs = torch.zeros(8 * 128 * 16, dtype=torch.float64, device="cuda")
    done = torch.zeros(8 * 128, dtype=torch.uint8, device="cuda")

    def call(env, K):
        return L.gemx_rollout_half(env.physical_system._handle, C.c_void_p(buf.data_ptr()), K, C.c_void_p(obs.data_ptr()), C.c_void_p(done.data_ptr()), None)

    e = ga.make("Finite-CC-PMSM-v0", n_envs=128)
    assert call(e, 8) == -1 and b"continuous converters only" in L.gemx_last_error()
    e.close()
    e = ga.make("Cont-CC-PMSM-v0", n_envs=128, dtype="float64")
    assert call(e, 8) == -1 and b"fp32 handles only" in L.gemx_last_error()
    e.close()
    e = ga.make("Cont-CC-PMSM-v0", n_envs=128)
    assert call(e, 1) == -1 and b"K must be >= 2" in L.gemx_last_error()
    assert call(e, 8) == 0
    e.close()
