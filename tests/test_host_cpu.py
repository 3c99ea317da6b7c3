"""CPU-only tests of the product's host side: parameter extraction vs the reference (via golden meta), the C-ABI
library loading with every symbol of include/gemx.h, loud failure without a GPU, error behaviour."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import gym_electric_motor_amd as ga
from gym_electric_motor_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _meta(name):
    return json.loads(str(np.load(os.path.join(HERE, "golden", name + ".npz"))["meta"]))


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as g

    g.build()


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "gemx.h")).read()
    declared = set(re.findall(r"\b(gemx_[a-z_0-9]+)\s*\(", header))
    declared -= {"gemx_config", "gemx_handle"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.load()
    for sym in declared:
        assert hasattr(L, sym), sym
    assert L.gemx_abi_version() == _lib.ABI_VERSION
    assert L.gemx_sizeof_config() == C.sizeof(_lib.GemxConfig)


def _check_limits(ps, meta):
    """limits / nominal values against the reference's, and the config the library gets against the REFERENCE's numbers too (not against
    `ps.limits` itself).  Every limit the reference takes from a parameter table is held to 1e-15; only the torque limit, which this
    package derives in closed form (a few ulp off the reference's operation order), gets 1e-13."""
    ref_lim, ref_nom = np.asarray(meta["limits"]), np.asarray(meta["nominal_state"])
    tq = meta["state_names"].index("torque")
    rt = np.full(len(ref_lim), 1e-15)
    rt[tq] = 1e-13
    assert (np.abs(ps.limits - ref_lim) <= rt * np.abs(ref_lim)).all(), (ps.limits, ref_lim)
    assert (np.abs(ps.nominal_state - ref_nom) <= rt * np.abs(ref_nom)).all()
    cfg_lim = np.asarray(list(ps._cfg.limits)[: len(ref_lim)])
    assert (np.abs(cfg_lim - ref_lim) <= rt * np.abs(ref_lim)).all()


@pytest.mark.parametrize("env_id, golden", [
    ("Cont-CC-PermExDc-v0", "permexdc_free_held_euler"),
    ("Cont-SC-PermExDc-v0", "permexdc_sc_free_held_euler"),
    ("Finite-CC-PMSM-v0", "pmsm_free_held_euler"),
    ("Finite-SC-PMSM-v0", "pmsm_sc_free_held_dopri5"),
    ("Cont-SC-SCIM-v0", "scim_free_held_euler"),
    ("Finite-CC-SynRM-v0", "synrm_fin_free_held_euler"),
    ("Cont-SC-SynRM-v0", "synrm_cont_sc_epi_held_euler"),
    ("Finite-CC-PermExDc-v0", "permexdc_fin_free_held_euler"),
    ("Cont-CC-SeriesDc-v0", "series_cont_free_held_euler"),
    ("Cont-SC-SeriesDc-v0", "series_cont_sc_free_held_euler"),
    ("Finite-CC-SeriesDc-v0", "series_fin_free_held_til_euler"),
    ("Cont-CC-ShuntDc-v0", "shunt_cont_free_held_euler"),
    ("Cont-SC-ShuntDc-v0", "shunt_cont_sc_free_held_euler"),
    ("Finite-CC-ShuntDc-v0", "shunt_fin_free_held_til_euler"),
    ("Cont-CC-ExtExDc-v0", "extex_cont_free_held_euler"),
    ("Cont-SC-ExtExDc-v0", "extex_cont_sc_free_held_euler"),
    ("Finite-CC-ExtExDc-v0", "extex_fin_free_held_euler"),
    ("Cont-CC-EESM-v0", "eesm_cont_free_held_euler"),
    ("Cont-SC-EESM-v0", "eesm_cont_sc_epi_held_euler"),
    ("Finite-CC-EESM-v0", "eesm_fin_free_held_euler"),
    ("Cont-CC-DFIM-v0", "dfim_cont_free_held_euler"),
    ("Cont-SC-DFIM-v0", "dfim_cont_sc_free_held_euler"),
    ("Finite-CC-DFIM-v0", "dfim_fin_free_held_euler"),
    ("Finite-SC-DFIM-v0", "dfim_fin_sc_free_uniform_euler"),
])
def test_host_metadata_matches_reference(env_id, golden):
    """limits / nominal_state / model constants / names / j_total as the live reference reported them."""
    meta = _meta(golden)
    ps = ga.make(env_id, n_envs=8, _defer_create=True).physical_system
    assert list(ps.state_names) == meta["state_names"]
    _check_limits(ps, meta)
    assert np.allclose(np.asarray(ps.electrical_motor._model_constants), np.asarray(meta["model_constants"]), rtol=1e-15, atol=0)
    assert ps.mechanical_load.j_total == pytest.approx(meta["j_total"], rel=1e-15)
    assert ps.tau == meta["tau"] and ps.supply.u_nominal == meta["u_nominal"]
    assert ps.state_positions == {n: i for i, n in enumerate(meta["state_names"])}
    assert ps.state_space.low.shape == (len(meta["state_names"]),)
    cfg = ps._cfg
    assert cfg.struct_size == C.sizeof(_lib.GemxConfig)
    assert list(cfg.limits)[: len(meta["limits"])] == [float(x) for x in ps.limits]  # the config carries the system's limits verbatim


ALL_ENV_IDS = [f"{a}-{c}-{m}-v0" for m in ("PermExDc", "SeriesDc", "ShuntDc", "ExtExDc", "PMSM", "SynRM", "SCIM", "EESM", "DFIM")
               for c in ("CC", "TC", "SC") for a in ("Cont", "Finite")]


@pytest.mark.parametrize("env_id", ALL_ENV_IDS)
def test_every_env_id_is_built_like_the_reference_builds_it(env_id):
    """All 54 ids: what `make(env_id)` assembles (names, limits, nominal values, model constants, inertia, load parameters, control step,
    supply, converter and its dead time, the load's speed / initial speed) against what the reference's `gem.make(env_id)` reported when
    oracle/make_golden.py:main_defaults recorded it.  (Round 3 found Cont-TC-ShuntDc-v0's omega_fixed = 230 this way.)"""
    meta = _meta("default_" + env_id[:-3].replace("-", "_").lower() + "_dopri5")
    assert meta["env_id"] == env_id
    ps = ga.make(env_id, n_envs=8, _defer_create=True).physical_system
    assert list(ps.state_names) == meta["state_names"]
    _check_limits(ps, meta)
    assert np.allclose(np.asarray(ps.electrical_motor._model_constants), np.asarray(meta["model_constants"]), rtol=1e-15, atol=0)
    assert ps.mechanical_load.j_total == pytest.approx(meta["j_total"], rel=1e-15)
    assert ps.tau == meta["tau"] and ps.supply.u_nominal == meta["u_nominal"]
    assert type(ps.mechanical_load).__name__ == meta["load"] and type(ps.electrical_motor).__name__ == meta["motor"]
    assert type(ps.converter).__name__ == meta["converter"].split("[")[0] and ps._interlocking_time() == meta["interlocking_time"]
    cfg = ps._cfg
    if meta["load"] == "ConstantSpeedLoad":
        assert cfg.init_state[0] == meta["omega_fixed"]
    else:
        lp = meta["load_parameter"]
        assert (cfg.load_a, cfg.load_b, cfg.load_c, cfg.tau_decay) == (lp["a"], lp["b"], lp["c"], meta["tau_decay"]) and cfg.init_state[0] == 0.0
    assert type(ps).__name__ == "Batched" + meta["system"]
    # the solver a user gets without naming one (envs.default_ode_solver): classical RK4, kink splitting exactly for the speed-dependent loads
    assert cfg.solver_kind == _lib.SOLVER_RK4 and cfg.solver_nsteps == 1
    assert cfg.solver_flags == (0 if meta["load"] == "ConstantSpeedLoad" else _lib.SOLVER_SPLIT_KINKS)


def test_default_constraints_become_masks():
    dc = ga.make("Cont-CC-PermExDc-v0", n_envs=2, _defer_create=True).physical_system
    assert dc._cfg.limit_mask == 1 << dc.state_positions["i"] and dc._cfg.squared_mask == 0 and dc._cfg.auto_reset == 1
    pm = ga.make("Finite-CC-PMSM-v0", n_envs=2, _defer_create=True).physical_system
    assert pm._cfg.squared_mask == (1 << pm.state_positions["i_sd"]) | (1 << pm.state_positions["i_sq"])
    one = ga.make("Finite-CC-PMSM-v0", n_envs=1, _defer_create=True).physical_system
    assert one._cfg.auto_reset == 0  # n_envs == 1: the reference env shell decides when to reset
    allc = ga.make("Cont-CC-PermExDc-v0", n_envs=2, constraints=("all_states",), _defer_create=True).physical_system
    assert allc._cfg.limit_mask == 0b11111
    sh = ga.make("Finite-CC-ShuntDc-v0", n_envs=2, _defer_create=True).physical_system
    assert sh._cfg.limit_mask == 0b1100 and sh._cfg.system_kind == 4 and sh._cfg.converter_kind == 3 and sh.action_space.n == 4
    assert list(sh._cfg.init_state)[:3] == [100.0, 0.0, 0.0] and sh.state_names == ["omega", "torque", "i_a", "i_e", "u", "u_sup"]


def test_pmsm_initialiser_dict_order_quirk():
    """Reference quirk (permanent_magnet_synchronous_motor.py:98 + synchronous_motor.py:125-131): the VALUES of the
    initialiser dict go into [i_sd, i_sq, epsilon] in dict order."""
    m = ga.PermanentMagnetSynchronousMotor(motor_initializer={"states": {"i_sq": 1.0, "i_sd": 2.0, "epsilon": 0.5}})
    ps = ga.make("Finite-CC-PMSM-v0", n_envs=2, motor=m, _defer_create=True).physical_system
    assert list(ps._cfg.init_state)[:4] == [100.0, 1.0, 2.0, 0.5]


def test_unsupported_pieces_raise():
    with pytest.raises(KeyError):
        ga.make("Cont-CC-PMSM-v1", n_envs=2, _defer_create=True)   # unknown env id
    with pytest.raises(KeyError):
        ga.make("Cont-XC-DFIM-v0", n_envs=2, _defer_create=True)
    with pytest.raises(ValueError):  # a DC system needs a one-voltage DC motor
        ga.BatchedDcMotorSystem(converter=ga.ContFourQuadrantConverter(), motor=ga.PermanentMagnetSynchronousMotor(),
                                load=ga.ConstantSpeedLoad(100.0), supply=ga.IdealVoltageSupply(60.0), ode_solver=ga.EulerSolver(),
                                _defer_create=True)

    class ScipyOdeSolver:  # stands for the reference's scipy-backed solver classes
        pass

    with pytest.raises(ValueError):
        ga.make("Cont-CC-PermExDc-v0", n_envs=2, ode_solver=ScipyOdeSolver(), _defer_create=True)
    lsoda = ScipyOdeSolver()
    lsoda._integrator, lsoda._solver_args = "lsoda", {}
    with pytest.raises(ValueError):
        ga.make("Cont-CC-PermExDc-v0", n_envs=2, ode_solver=lsoda, _defer_create=True)
    with pytest.raises(KeyError):  # update_parameter_dict semantics, utils.py:73-94
        ga.DcPermanentlyExcitedMotor(motor_parameter=dict(r_x=1.0))
    with pytest.raises(AssertionError, match="only available for Continuous"):  # physical_systems.py:431-434
        ga.make("Finite-CC-PMSM-v0", n_envs=2, control_space="dq", _defer_create=True)
    with pytest.raises(ValueError, match="action_frame"):  # a DC system has no dq frame
        ga.make("Cont-CC-ExtExDc-v0", n_envs=2, control_space="dq", _defer_create=True)
    with pytest.raises(NotImplementedError, match="flux observer"):
        ga.DqToAbcActionProcessor.make("SCIM")
    with pytest.raises(ValueError, match="INSIDE"):
        ga.make("Cont-CC-PMSM-v0", n_envs=2, _defer_create=True,
                physical_system_wrappers=(ga.DqToAbcActionProcessor.make("PMSM"), ga.DeadTimeProcessor(1)))


def test_error_controlled_solver_config():
    """ga.ScipyOdeSolver() and the REFERENCE's own ScipyOdeSolver('dopri5', **kwargs) instance (solvers.py:139-184: its default solver)
    both select the device's error-controlled Dormand-Prince: GEMX_SOLVER_DP5 + GEMX_SOLVER_ADAPTIVE with the caller's tolerances (scipy's
    atol 1e-12 is raised to the device's floor 1e-9); the flag is refused with another scheme."""
    cfg = ga.make("Cont-SC-SCIM-v0", n_envs=2, ode_solver=ga.ScipyOdeSolver(), _defer_create=True).physical_system._cfg
    AK = _lib.SOLVER_ADAPTIVE | _lib.SOLVER_SPLIT_KINKS  # (round 6: the load's kinks in closed form by default; split_kinks=False opts out)
    assert (cfg.solver_kind, cfg.solver_nsteps, cfg.solver_flags) == (_lib.SOLVER_DP5, 1, AK)
    assert (cfg.solver_rtol, cfg.solver_atol) == (1e-6, 1e-9)
    cfg = ga.make("Cont-SC-SCIM-v0", n_envs=2, ode_solver=ga.ScipyOdeSolver(rtol=1e-5, atol=1e-7, nsteps=500, split_kinks=False), _defer_create=True).physical_system._cfg
    assert (cfg.solver_flags, cfg.solver_rtol, cfg.solver_atol) == (_lib.SOLVER_ADAPTIVE, 1e-5, 1e-7)
    with pytest.raises(ValueError):
        ga.ScipyOdeSolver("lsoda")

    class ScipyOdeSolver:  # the shape of the reference's class (duck-typed by name, like every reference component)
        def __init__(self, integrator="dopri5", **kwargs):
            self._integrator, self._solver_args = integrator, kwargs

    cfg = ga.make("Cont-SC-SCIM-v0", n_envs=2, ode_solver=ScipyOdeSolver(), _defer_create=True).physical_system._cfg
    assert (cfg.solver_kind, cfg.solver_flags, cfg.solver_rtol, cfg.solver_atol) == (_lib.SOLVER_DP5, AK, 1e-6, 1e-9)
    cfg = ga.make("Cont-SC-SCIM-v0", n_envs=2, ode_solver=ScipyOdeSolver(rtol=1e-4, atol=1e-5), _defer_create=True).physical_system._cfg
    assert (cfg.solver_rtol, cfg.solver_atol) == (1e-4, 1e-5)
    if os.path.isdir("/root/reference/src"):  # the live class, in a child process (its imports stay out of this session)
        import subprocess
        import sys

        code = ("import os, sys; os.environ['MPLBACKEND'] = 'Agg'; sys.path[:0] = [%r, %r, %r]\n"
                "from gym_electric_motor.physical_systems.solvers import ScipyOdeSolver\n"
                "import gym_electric_motor_amd as ga\nfrom gym_electric_motor_amd import _lib\n"
                "c = ga.make('Cont-SC-SCIM-v0', n_envs=2, ode_solver=ScipyOdeSolver(), _defer_create=True).physical_system._cfg\n"
                "assert (c.solver_kind, c.solver_flags, c.solver_rtol) == (_lib.SOLVER_DP5, _lib.SOLVER_ADAPTIVE | _lib.SOLVER_SPLIT_KINKS, 1e-6)\nprint('OK')\n"
                % (os.path.join(REPO, "oracle", "gymnasium_standin"), "/root/reference/src", REPO))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]
    L = _lib.load()
    h = C.c_void_p()
    bad = _lib.GemxConfig.from_buffer_copy(cfg)
    bad.solver_kind = _lib.SOLVER_RK4
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"GEMX_SOLVER_ADAPTIVE" in L.gemx_last_error()
    bad = _lib.GemxConfig.from_buffer_copy(cfg)
    bad.solver_rtol = 0.5
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"solver_rtol" in L.gemx_last_error()


def test_c_abi_argument_validation_without_gpu():
    """gemx_create validates before touching the device; error text through gemx_last_error()."""
    L = _lib.load()
    ps = ga.make("Cont-CC-PermExDc-v0", n_envs=2, _defer_create=True).physical_system
    h = C.c_void_p()
    cfg = ps._cfg
    bad = _lib.GemxConfig.from_buffer_copy(cfg)
    bad.struct_size = 12
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"ABI" in L.gemx_last_error()
    bad = _lib.GemxConfig.from_buffer_copy(cfg)
    bad.interlocking_time = 1.0
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"interlocking" in L.gemx_last_error()
    bad = _lib.GemxConfig.from_buffer_copy(cfg)
    bad.model[5] = 1.0  # outside the DC sparsity pattern
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"sparsity" in L.gemx_last_error()
    assert L.gemx_create(C.byref(cfg), 0, 0, C.byref(h)) == -1
    assert L.gemx_destroy(None) == 0
    # system / converter pairing, and no dead time for the EESM (the reference's branch for it cannot execute)
    bad = _lib.GemxConfig.from_buffer_copy(cfg)
    bad.converter_kind = _lib.CONV_CONT_2X4QC
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"combination" in L.gemx_last_error()
    eesm = ga.make("Finite-CC-EESM-v0", n_envs=2, _defer_create=True).physical_system._cfg
    bad = _lib.GemxConfig.from_buffer_copy(eesm)
    bad.interlocking_time = 1e-6
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"EESM" in L.gemx_last_error()
    bad = _lib.GemxConfig.from_buffer_copy(eesm)
    bad.model[2 * _lib.MODEL_COLS + 2] = 1.0  # d(i_e)/dt has no i_q term
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"sparsity" in L.gemx_last_error()


def test_action_stage_config():
    """control_space='dq', DqToAbcActionProcessor and DeadTimeProcessor fold into gemx_config.action_frame / action_delay."""
    ps = ga.make("Cont-CC-PMSM-v0", n_envs=2, control_space="dq", _defer_create=True).physical_system
    assert ps._cfg.action_frame == _lib.ACT_DQ_SPACE and ps._cfg.action_delay == 0 and ps.action_space.shape == (2,)
    ps = ga.make("Cont-SC-SCIM-v0", n_envs=2, control_space="dq", _defer_create=True).physical_system
    assert ps._cfg.action_frame == _lib.ACT_DQ_SPACE and ps.action_space.shape == (2,)
    ps = ga.make("Cont-CC-EESM-v0", n_envs=2, _defer_create=True,
                 physical_system_wrappers=(ga.DeadTimeProcessor(steps=2), ga.DqToAbcActionProcessor.make("EESM"))).physical_system
    assert ps._cfg.action_frame == _lib.ACT_DQ_PROCESSOR and ps._cfg.action_delay == 2 and ps.action_space.shape == (3,)
    assert ps.dead_time == 2
    ps = ga.make("Finite-CC-DFIM-v0", n_envs=2, _defer_create=True, physical_system_wrappers=(ga.DeadTimeProcessor(3),)).physical_system
    assert ps._cfg.action_frame == _lib.ACT_ABC and ps._cfg.action_delay == 3 and list(ps.action_space.nvec) == [8, 8]
    L = _lib.load()
    h = C.c_void_p()
    bad = _lib.GemxConfig.from_buffer_copy(ps._cfg)
    bad.action_delay = 99
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"action_delay" in L.gemx_last_error()
    bad = _lib.GemxConfig.from_buffer_copy(ps._cfg)
    bad.action_frame = _lib.ACT_DQ_PROCESSOR
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"action_frame" in L.gemx_last_error()


@pytest.mark.parametrize("name, kwargs", [
    ("rw_pmsm_cont_cc_epi_held_euler", dict(reward_weights=dict(i_sd=0.5, i_sq=0.5))),
    ("rw_scim_cont_sc_epi_held_euler", dict(reward_weights=dict(omega=1.0))),
    ("rw_permexdc_cont_tc_epi_held_euler", dict(reward_weights=dict(torque=1.0))),
    ("rw_pmsm_cont_cc_pow2_epi_held_euler", dict(reward_weights=dict(i_sd=0.3, i_sq=0.6, omega=0.1), reward_power=2, bias="positive",
                                                violation_reward=-7.5, normed_reward_weights=True)),
    ("rw_eesm_cont_cc_pow_mixed_epi_held_euler", dict(reward_weights=dict(i_sd=0.4, i_sq=0.4, i_e=0.2),
                                                     reward_power=dict(i_sd=1, i_sq=2, i_e=0.5), gamma=0.95)),
])
def test_reward_config_derivation_matches_reference(name, kwargs):
    """set_reward() with WeightedSumOfErrors' own arguments derives the arrays the live reference's reward function held
    (weights after normalisation, powers, state lengths, bias, default violation reward = min(range[0] / (1 - gamma), 0))."""
    meta = _meta(name)
    rw = meta["reward"]
    ps = ga.make(meta["env_id"], n_envs=2, _defer_create=True).physical_system
    refd = [n for n, r in zip(meta["state_names"], rw["referenced_states"]) if r]
    rc = ps.set_reward(referenced_states=refd, **kwargs)
    n = len(meta["state_names"])
    assert np.allclose(list(rc.weight)[:n], rw["weights"], rtol=1e-15, atol=0)
    assert np.allclose(list(rc.state_length)[:n], rw["state_length"], rtol=0, atol=0)
    w = np.array(rw["weights"])
    assert np.allclose(np.array(list(rc.power)[:n])[w != 0], np.array(rw["powers"])[w != 0])
    assert rc.bias == pytest.approx(rw["bias"], abs=1e-15) and rc.violation_reward == pytest.approx(rw["violation_reward"], rel=1e-14)
    assert [rc.ref_index[j] for j in range(rc.n_ref)] == [i for i, r in enumerate(rw["referenced_states"]) if r]


def test_rc_supply_config():
    """RCVoltageSupply (voltage_supplies.py:75-123) -> gemx_config.supply_kind / supply_r / supply_c; state space low 0 for u_sup."""
    sup = ga.RCVoltageSupply(u_nominal=420.0, supply_parameter=dict(R=0.5, C=2e-3))
    ps = ga.make("Finite-CC-PMSM-v0", n_envs=2, supply=sup, _defer_create=True).physical_system
    assert ps._cfg.supply_kind == _lib.SUPPLY_RC and ps._cfg.supply_r == 0.5 and ps._cfg.supply_c == 2e-3 and ps._cfg.u_nominal == 420.0
    assert ps.state_space.low[ps.state_positions["u_sup"]] == 0.0 and sup.supply_range == (0, 420.0)
    with pytest.raises(AssertionError):
        ga.RCVoltageSupply(supply_parameter=dict(R=1.0))
    L = _lib.load()
    h = C.c_void_p()
    bad = _lib.GemxConfig.from_buffer_copy(ps._cfg)
    bad.supply_c = 0.0
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"supply_r" in L.gemx_last_error()
    eesm = ga.make("Finite-CC-EESM-v0", n_envs=2, supply=ga.RCVoltageSupply(420.0), _defer_create=True).physical_system._cfg
    assert L.gemx_create(C.byref(eesm), 4, 0, C.byref(h)) == -1 and b"finite EESM" in L.gemx_last_error()

    class AC1PhaseSupply:  # stands for the reference's AC supplies
        u_nominal = 230.0
        supply_range = (-325.0, 325.0)

    with pytest.raises(ValueError, match="not on the accelerated path"):
        ga.make("Cont-CC-PermExDc-v0", n_envs=2, supply=AC1PhaseSupply(), _defer_create=True)


def test_random_initialiser_bounds_match_reference_support():
    """motor_initializer / load_initializer with random_init: the per-ODE-state sampling bounds (nominal value x state-space low,
    clipped to `interval`) cover exactly the support of 4000 resets of the live reference (tests/golden/init_samples.npz)."""
    d = np.load(os.path.join(GOLDEN, "init_samples.npz"))
    cls = {"PermanentMagnetSynchronousMotor": ga.PermanentMagnetSynchronousMotor, "DcExternallyExcitedMotor": ga.DcExternallyExcitedMotor,
           "ExternallyExcitedSynchronousMotor": ga.ExternallyExcitedSynchronousMotor, "DcPermanentlyExcitedMotor": ga.DcPermanentlyExcitedMotor}
    for case in ("pmsm_sc_uniform", "extex_cc_uniform_interval", "eesm_sc_uniform"):
        meta = json.loads(str(d[case + "_meta"]))
        kw = dict(motor=cls[meta["motor"]](motor_initializer=meta["motor_initializer"]), n_envs=2, seed=5, _defer_create=True)
        if meta["load_initializer"] is not None:
            kw["load"] = ga.PolynomialStaticLoad(load_parameter=meta["load_parameter"], load_initializer=meta["load_initializer"])
        ps = ga.make(meta["env_id"], **kw).physical_system
        c, y = ps._cfg, d[case + "_y"]
        assert c.init_kind == _lib.INIT_UNIFORM and c.seed == 5
        for j in range(y.shape[1]):
            lo, hi = c.init_lo[j], c.init_hi[j]
            if hi == lo:
                assert np.ptp(y[:, j]) == 0 and y[0, j] == c.init_state[j]
            else:
                assert lo <= y[:, j].min() < lo + 0.01 * (hi - lo) and hi - 0.01 * (hi - lo) < y[:, j].max() <= hi
    # induction machines (round 4): static bounds for currents / angle, the flux slots carry the interval only, and init_flux what the
    # per-reset flux bounds need (induction_motor.py:250-285); the draw RULE, restated in numpy, passes KS tests against the reference
    from scipy import stats

    for case in ("scim_sc_uniform", "scim_cc_negspeed_uniform", "dfim_cc_negspeed_interval_uniform"):
        meta, y = json.loads(str(d[case + "_meta"])), d[case + "_y"]
        mcls = ga.SquirrelCageInductionMotor if "scim" in case else ga.DoublyFedInductionMotor
        kw = dict(motor=mcls(motor_initializer=meta["motor_initializer"]), n_envs=2, seed=5, _defer_create=True)
        if meta["load"] == "ConstantSpeedLoad":
            kw["load"] = ga.ConstantSpeedLoad(omega_fixed=float(y[0, 0]))
        c = ga.make(meta["env_id"], **kw).physical_system._cfg
        mp = meta["motor_parameter"]
        l_r = mp["l_m"] + mp["l_sigr"]
        assert c.init_flux_mode == 1 and c.init_flux[1] == mp["p"] and abs(c.init_flux[5] - mp["l_m"] / l_r) < 1e-15 and c.init_flux[6] == mp["l_m"]
        lo, hi, const, fl = (np.array(x[:8]) for x in (c.init_lo, c.init_hi, c.init_state, c.init_flux))
        rng, prev, got = np.random.default_rng(3), None, np.zeros((4000, 6))
        for k in range(4000):  # gemx_common.hpp:init_draw_all, operation by operation
            u = rng.uniform(size=8)
            v = np.array([lo[j] + (hi[j] - lo[j]) * u[j] if lo[j] < hi[j] and np.isfinite(lo[j]) else const[j] for j in range(8)])
            eps = 2 * np.pi * u[7] - np.pi
            ce, se, psi = np.cos(eps), np.sin(eps), fl[0]
            if v[0] != 0:
                ia, ib = (const[1], const[2]) if prev is None else (prev[1], prev[2])
                i_d, i_q = ce * ia + se * ib, -se * ia + ce * ib
                psi = 0.9 * min(max((fl[1] * v[0] * fl[2] * i_d + fl[3] * i_q + fl[4]) / (-fl[1] * v[0] * fl[5]), 0.0), abs(fl[6] * i_d))
            for j, h in ((3, abs(psi * ce)), (4, abs(psi * se))):
                a, b = max(-h, lo[j]), min(h, hi[j])
                v[j] = a + (b - a) * u[j] if a < b else a
            got[k], prev = v[:6], v
        for j in range(6):
            if np.ptp(y[:, j]) == 0:
                assert np.ptp(got[:, j]) == 0 and got[0, j] == y[0, j]
            else:
                assert stats.ks_2samp(got[:, j], y[:, j]).pvalue > 1e-3, (case, j)
        assert stats.ks_2samp(np.hypot(got[:, 3], got[:, 4]), np.hypot(y[:, 3], y[:, 4])).pvalue > 1e-3, case
    # validation in gemx_create: induction-motor states need the flux mode, and a constant-speed omega cannot be random
    L = _lib.load()
    h = C.c_void_p()
    scim = ga.make("Cont-CC-SCIM-v0", n_envs=2, _defer_create=True).physical_system._cfg
    bad = _lib.GemxConfig.from_buffer_copy(scim)
    bad.init_kind, bad.init_lo[1], bad.init_hi[1] = _lib.INIT_UNIFORM, -1.0, 1.0
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"induction" in L.gemx_last_error()
    bad = _lib.GemxConfig.from_buffer_copy(scim)
    bad.init_kind, bad.init_lo[0], bad.init_hi[0] = _lib.INIT_UNIFORM, 0.0, 10.0
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"ConstantSpeedLoad" in L.gemx_last_error()
    with pytest.raises(NotImplementedError):
        ga.DcPermanentlyExcitedMotor(motor_initializer=dict(random_init="cauchy"))


def test_wiener_generator_margins_match_reference():
    """BatchedWienerProcessReferenceGenerator.set_modules derives limit margin / initial range like subepisoded_reference_generator.py:66-84."""
    w = np.load(os.path.join(GOLDEN, "wiener_samples.npz"))
    ps = ga.make("Cont-CC-PMSM-v0", n_envs=2, _defer_create=True).physical_system
    gen = ga.BatchedWienerProcessReferenceGenerator(reference_states=("i_sd", "i_sq"), seed=1).set_modules(ps, _defer_create=True)
    c = gen._cfg
    assert np.allclose([[c.margin_lo[j], c.margin_hi[j]] for j in range(2)], w["margins"], rtol=1e-14)
    assert np.allclose([[c.initial_lo[j], c.initial_hi[j]] for j in range(2)], w["margins"], rtol=1e-14)
    assert [c.sigma_lo[0], c.sigma_hi[0]] == list(w["sigma_range"]) and [c.episode_len_lo, c.episode_len_hi] == [int(x) for x in w["episode_lengths"]]
    g2 = ga.BatchedWienerProcessReferenceGenerator(reference_states="omega", limit_margin=(0.2, 0.5)).set_modules(
        ga.make("Cont-SC-SCIM-v0", n_envs=2, _defer_create=True).physical_system, _defer_create=True)
    assert (g2._cfg.margin_lo[0], g2._cfg.margin_hi[0]) == (-0.2, 0.5)


def test_multi_converter_holders():
    """Cont/FiniteMultiConverter mirrors (converters.py:498-740): spaces, tau propagation, per-sub-converter dead time."""
    c = ga.ContMultiConverter(subconverters=[ga.ContB6BridgeConverter, ga.ContFourQuadrantConverter], tau=2e-4)
    assert c.action_space.shape == (4,) and c.currents.shape == (4,) and c.voltages.shape == (4,)
    assert [sc.tau for sc in c.sub_converters] == [2e-4, 2e-4]
    c.tau = 1e-4
    assert [sc.tau for sc in c.sub_converters] == [1e-4, 1e-4]
    f = ga.FiniteMultiConverter(subconverters=[ga.FiniteFourQuadrantConverter(interlocking_time=1e-6),
                                               ga.FiniteFourQuadrantConverter(interlocking_time=1e-6)])
    assert list(f.action_space.nvec) == [4, 4]
    ps = ga.make("Finite-CC-ExtExDc-v0", n_envs=2, converter=f, _defer_create=True).physical_system
    assert ps._cfg.interlocking_time == 1e-6 and ps._cfg.converter_kind == _lib.CONV_FINITE_2X4QC
    mixed = ga.FiniteMultiConverter(subconverters=[ga.FiniteFourQuadrantConverter(interlocking_time=1e-6), ga.FiniteFourQuadrantConverter()])
    with pytest.raises(ValueError, match="share one interlocking_time"):
        ga.make("Finite-CC-ExtExDc-v0", n_envs=2, converter=mixed, _defer_create=True)
    with pytest.raises(ValueError, match="not on the accelerated path"):
        ga.make("Finite-CC-ExtExDc-v0", n_envs=2, _defer_create=True,
                converter=ga.FiniteMultiConverter(subconverters=[ga.FiniteB6BridgeConverter, ga.FiniteB6BridgeConverter]))


def test_dead_time_processor_reset_action_is_folded_into_the_config():
    """DeadTimeProcessor(steps, reset_action=callable) (dead_time_processor.py:27-50): `steps` copies of ONE action travel as
    gemx_config.action_delay_reset (a MultiDiscrete action as its flat index); different actions per slot are refused with a message;
    the reference's default (None / zeros) leaves the row at zero; gemx_create validates a discrete index."""
    ps = ga.make("Cont-CC-PMSM-v0", n_envs=2, _defer_create=True,
                 physical_system_wrappers=(ga.DeadTimeProcessor(2, reset_action=lambda: [np.array([0.4, -0.3, 0.1])] * 2),)).physical_system
    assert list(ps._cfg.action_delay_reset)[:4] == [0.4, -0.3, 0.1, 0.0] and ps._cfg.action_delay == 2
    ps = ga.make("Finite-CC-ExtExDc-v0", n_envs=2, _defer_create=True,
                 physical_system_wrappers=(ga.DeadTimeProcessor(3, reset_action=lambda: [[2, 1]] * 3),)).physical_system
    assert ps._cfg.action_delay_reset[0] == 2 + 4 * 1 and ps._cfg.action_delay_reset[1] == 0.0
    ps = ga.make("Finite-CC-PMSM-v0", n_envs=2, _defer_create=True, physical_system_wrappers=(ga.DeadTimeProcessor(2, reset_action=lambda: [5, 5]),)).physical_system
    assert ps._cfg.action_delay_reset[0] == 5.0
    ps = ga.make("Finite-CC-PMSM-v0", n_envs=2, _defer_create=True, physical_system_wrappers=(ga.DeadTimeProcessor(2),)).physical_system
    assert not any(ps._cfg.action_delay_reset)
    with pytest.raises(NotImplementedError, match="ONE action"):
        ga.make("Finite-CC-PMSM-v0", n_envs=2, _defer_create=True, physical_system_wrappers=(ga.DeadTimeProcessor(2, reset_action=lambda: [5, 3]),))
    with pytest.raises(ValueError, match="for a dead time of 2 steps"):
        ga.make("Finite-CC-PMSM-v0", n_envs=2, _defer_create=True, physical_system_wrappers=(ga.DeadTimeProcessor(2, reset_action=lambda: [5]),))
    # round 5 (advisor finding): the row is validated against the WRAPPED system's action space -- length, bounds, per-component ranges
    with pytest.raises(ValueError, match="not an element"):  # a short continuous row used to be zero-padded in silence
        ga.make("Cont-CC-PMSM-v0", n_envs=2, _defer_create=True, physical_system_wrappers=(ga.DeadTimeProcessor(2, reset_action=lambda: [np.array([0.4, -0.3])] * 2),))
    with pytest.raises(ValueError, match="not an element"):  # beyond the bounds
        ga.make("Cont-CC-PMSM-v0", n_envs=2, _defer_create=True, physical_system_wrappers=(ga.DeadTimeProcessor(2, reset_action=lambda: [np.array([0.4, -1.3, 0.0])] * 2),))
    with pytest.raises(ValueError, match="not an element"):  # [4, 0] of MultiDiscrete([4, 4]) used to alias the flat index of [0, 1]
        ga.make("Finite-CC-ExtExDc-v0", n_envs=2, _defer_create=True, physical_system_wrappers=(ga.DeadTimeProcessor(3, reset_action=lambda: [[4, 0]] * 3),))
    with pytest.raises(ValueError, match="not an element"):  # Discrete(8)
        ga.make("Finite-CC-PMSM-v0", n_envs=2, _defer_create=True, physical_system_wrappers=(ga.DeadTimeProcessor(2, reset_action=lambda: [8, 8]),))
    # behind a DqToAbcActionProcessor the DeadTimeProcessor wraps the abc system: three duty cycles, not the outer (u_d, u_q)
    ps3 = ga.make("Cont-CC-PMSM-v0", n_envs=2, _defer_create=True,
                  physical_system_wrappers=(ga.DeadTimeProcessor(2, reset_action=lambda: [np.array([0.1, 0.2, -0.3])] * 2), ga.DqToAbcActionProcessor.make("PMSM"))).physical_system
    assert list(ps3._cfg.action_delay_reset)[:3] == [0.1, 0.2, -0.3]
    L = _lib.load()
    h = C.c_void_p()
    bad = _lib.GemxConfig.from_buffer_copy(ps._cfg)
    bad.action_delay_reset[0] = 8.0  # Finite-B6C has 8 switching states: 0 .. 7
    assert L.gemx_create(C.byref(bad), 4, 0, C.byref(h)) == -1 and b"action_delay_reset" in L.gemx_last_error()


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.GemxError, match="no CPU fallback"):
        ga.make("Cont-CC-PermExDc-v0", n_envs=4)


def test_parity_contract_constants_match_the_design_document():
    """The parity contract's numbers live in tests/parity_contract.py (imported by the GPU parity tests) and, in prose, in DESIGN.md section 2:
    this test pins BOTH -- changing a tolerance, the flux floor of the induction machines' weighting or the dead-time sign margin in one
    place fails here until the other says the same (round 5 verdict, weak #1: the two carve-out constants used to live in the GPU test file only)."""
    import re
    import sys

    sys.path.insert(0, os.path.join(REPO, "tests"))
    import parity_contract as pc

    assert (pc.TOL_FP32, pc.REL_FLOOR, pc.TOL_FP64_SAME_INTEGRATOR, pc.DONE_MARGIN) == (1e-4, 1e-3, 1e-9, 1e-5)
    assert (pc.FLUX_FLOOR, pc.SIGN_MARGIN, pc.DEAD_TIME_MIN_COVER) == (0.05, 2e-5, 0.5)
    design = open(os.path.join(REPO, "DESIGN.md")).read()
    sec2 = design[design.index("## 2. Parity"):design.index("## 3. ")]
    flat = re.sub(r"\s+", " ", sec2)
    assert "**1e-4 relative per column**" in flat and "max(max|x_ref|, 1e-3)" in flat        # TOL_FP32, REL_FLOOR
    assert "same integrator in fp64: 1e-9 absolute" in flat                                   # TOL_FP64_SAME_INTEGRATOR
    assert "margin is < 1e-5" in flat                                                         # DONE_MARGIN
    assert "min(1, |ψ_r| / (0.05 max|ψ_r|))" in flat and "above 5 % of its range" in flat      # FLUX_FLOOR
    assert "< 2e-5 of the limit" in flat and "cover half the run" in flat                     # SIGN_MARGIN, DEAD_TIME_MIN_COVER
    gpu = open(os.path.join(REPO, "tests", "test_gpu_parity.py")).read()
    assert "from parity_contract import FLUX_FLOOR, SIGN_MARGIN" in gpu and not re.search(r"^(FLUX_FLOOR|SIGN_MARGIN)\s*=", gpu, re.M)


def test_unit_libraries_are_complete_small_and_self_contained():
    """Round 6: one shared object per kernel unit, loaded by gemx_create.  All 38 exist beside libgemx.so, each exports gemx_unit_init /
    gemx_unit_launch (and nothing of libgemx.so is needed to load one: no DT_NEEDED on it), what a handle maps -- the C ABI + ONE unit --
    stays far below 20 MB on disk, and the whole set below half of round 5's 144 MB library."""
    import subprocess

    from gym_electric_motor_amd import build as b

    libs = b.all_libs()
    assert len(libs) == 39 and all(os.path.exists(p) for p in libs), [p for p in libs if not os.path.exists(p)]
    sizes = {os.path.basename(p): os.path.getsize(p) for p in libs}
    assert sizes["libgemx.so"] < 2 << 20 and max(v for k, v in sizes.items() if k != "libgemx.so") + sizes["libgemx.so"] < 20 << 20, sizes
    assert sum(sizes.values()) < 72 << 20, sum(sizes.values())
    for p in libs[1:4] + libs[-2:]:
        dyn = subprocess.run(["nm", "-D", "--defined-only", p], capture_output=True, text=True).stdout
        assert " T gemx_unit_init" in dyn and " T gemx_unit_launch" in dyn, p
        needed = subprocess.run(["readelf", "-d", p], capture_output=True, text=True).stdout
        assert "libgemx.so" not in needed, p
    # a unit compiled against another handle layout / ABI refuses to serve (what gemx_create reports as "rebuild the package")
    u = C.CDLL(libs[1])
    u.gemx_unit_init.argtypes = [C.c_ulonglong, C.c_int, C.c_void_p]
    assert u.gemx_unit_init(1, _lib.ABI_VERSION, None) == -1 and u.gemx_unit_init(0, _lib.ABI_VERSION + 1, None) == -1


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "gym_electric_motor_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "gemx_oracle" not in src, f


REF_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference tree only exists in the build container")
def test_drops_into_unmodified_reference_env_shell_with_reference_components():
    """reference component INSTANCES + the reference's ElectricMotorEnvironment around the batched
    system (host-side wiring only here: the build container has no GPU, the GPU box has no reference).  Runs in a
    subprocess so the reference / gymnasium stand-in imports do not leak into this test session."""
    import subprocess
    import sys

    code = r'''
import os, sys
os.environ["MPLBACKEND"] = "Agg"
sys.path[:0] = [%r, %r, %r]
import numpy as np
import gym_electric_motor as gem
from gym_electric_motor.core import ElectricMotorEnvironment, PhysicalSystem
from gym_electric_motor import physical_systems as ps, reference_generators as rg, reward_functions as rf
from gym_electric_motor.constraints import SquaredConstraint
import gym_electric_motor_amd as ga

ref_sys = ps.SynchronousMotorSystem(converter=ps.FiniteB6BridgeConverter(), motor=ps.PermanentMagnetSynchronousMotor(),
    load=ps.ConstantSpeedLoad(omega_fixed=100.0), supply=ps.IdealVoltageSupply(u_nominal=420.0), ode_solver=ps.EulerSolver(), tau=1e-5)
system = ga.BatchedSynchronousMotorSystem(converter=ps.FiniteB6BridgeConverter(interlocking_time=1e-6),
    motor=ps.PermanentMagnetSynchronousMotor(), load=ps.ConstantSpeedLoad(omega_fixed=100.0),
    supply=ps.IdealVoltageSupply(u_nominal=420.0), ode_solver=ps.EulerSolver(nsteps=2), tau=1e-5, n_envs=1, _defer_create=True)
assert isinstance(system, PhysicalSystem)
env = ElectricMotorEnvironment(physical_system=system,
    reference_generator=rg.WienerProcessReferenceGenerator(reference_state="i_sq"),
    reward_function=rf.WeightedSumOfErrors(reward_weights=dict(i_sq=1.0)),
    constraints=(SquaredConstraint(("i_sq", "i_sd")),), visualization=())
assert env.physical_system is system and env.action_space == ref_sys.action_space
assert list(system.state_names) == list(ref_sys.state_names)
assert np.array_equal(system.limits, ref_sys.limits) and np.array_equal(system.nominal_state, ref_sys.nominal_state)
assert np.array_equal(system.state_space.low, ref_sys.state_space.low) and np.array_equal(system.state_space.high, ref_sys.state_space.high)
cfg = system._cfg
assert cfg.solver_kind == 0 and cfg.solver_nsteps == 2 and cfg.interlocking_time == 1e-6 and cfg.converter_kind == 1
assert np.allclose(np.array(cfg.model).reshape(5, 11)[:3, :7], ref_sys.electrical_motor._model_constants)
# the three configured ids built from REFERENCE components give the same config as from this package's mirrors
for eid in ("Cont-CC-PermExDc-v0", "Finite-CC-PMSM-v0", "Cont-SC-SCIM-v0"):
    renv = gem.make(eid, ode_solver=ps.EulerSolver())
    rs = renv.physical_system
    mine = ga.make(eid, n_envs=1, ode_solver=ga.EulerSolver(), _defer_create=True).physical_system
    cls = type(mine)
    theirs = cls(converter=rs.converter, motor=rs.electrical_motor, load=rs.mechanical_load, supply=rs.supply,
                 ode_solver=ps.EulerSolver(), tau=rs.tau, n_envs=1, constraints=mine._constraints, _defer_create=True)
    assert bytes(theirs._cfg) == bytes(mine._cfg), eid
print("OK")
''' % (os.path.join(REPO, "oracle", "gymnasium_standin"), REF_SRC, REPO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]


def test_config_struct_header_binding_and_docs_agree():
    """include/gemx.h `gemx_config` == gym_electric_motor_amd._lib.GemxConfig == the binding sketch in INTEGRATION.md: field names,
    order, C types and array lengths (a maintainer who copies the sketch must pass gemx_create's struct_size / ABI check)."""
    import ctypes as C
    import importlib.util
    import re

    from gym_electric_motor_amd import _lib

    spec = importlib.util.spec_from_file_location("gen_integration_sketch", os.path.join(REPO, "tools", "gen_integration_sketch.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    header = open(os.path.join(REPO, "include", "gemx.h")).read()
    fields = gen.parse_struct(header)
    ct = {"int32_t": C.c_int32, "uint32_t": C.c_uint32, "uint64_t": C.c_uint64, "int64_t": C.c_int64, "double": C.c_double}
    want = [(name, ct[t] * n if n else ct[t]) for t, name, n in fields]
    got = list(_lib.GemxConfig._fields_)
    assert [g[0] for g in got] == [w[0] for w in want]
    for (gn, gt), (wn, wt) in zip(got, want):
        assert C.sizeof(gt) == C.sizeof(wt) and getattr(gt, "_length_", 0) == getattr(wt, "_length_", 0), gn
    assert gen.header_constants(header)["GEMX_ABI_VERSION"] == _lib.ABI_VERSION
    # the markdown sketch: every ("name", C.c_type[ * n]) pair inside the GemxConfig class of INTEGRATION.md
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    cls = md[md.index("class GemxConfig(C.Structure):"):md.index("class GemxSCMLSystem(PhysicalSystem):")]
    pairs = re.findall(r'\("(\w+)", C\.(c_\w+)(?: \* (\d+))?\)', cls)
    assert [(n, t, int(k) if k else None) for n, t, k in pairs] == [(name, gen.CTYPES[t][2:], n) for t, name, n in fields]
    assert f"abi_version={_lib.ABI_VERSION}" in md
    # reward / refgen structs of the binding against the header too
    for struct, cls_ in (("gemx_reward_config", _lib.GemxRewardConfig), ("gemx_refgen_config", _lib.GemxRefgenConfig)):
        f2 = gen.parse_struct(header, struct)
        assert [g[0] for g in cls_._fields_] == [name for _, name, _ in f2], struct
        for (gn, gt), (t, name, n) in zip(cls_._fields_, f2):
            assert C.sizeof(gt) == C.sizeof(ct[t]) * (n or 1), (struct, gn)


def test_bench_cli_contract_without_a_gpu():
    """bench.py: the driver's flags parse; impossible requests fail with a message, not a traceback; the workload table carries the
    algorithmic bytes per env-step of SURVEY.md 8(d) (25 / 58 / 69 B fused, 41 / 90 / 117 B single step)."""
    import subprocess
    import sys

    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import bench

    assert [bench.bytes_per_env_step_fused(dict(w)) for w in (bench.WORKLOADS["permexdc"], bench.WORKLOADS["pmsm"], bench.WORKLOADS["scim"])] == [25, 58, 69]
    assert [bench.bytes_per_env_step_single(dict(w)) for w in (bench.WORKLOADS["permexdc"], bench.WORKLOADS["pmsm"], bench.WORKLOADS["scim"])] == [41, 90, 117]
    # the stdout line stays small enough for the driver to parse (round 5's 20-KB line was recorded as parsed = null): the compact form
    # of a COMPLETE record (last round's collection: every leg present) and of one whose every optional leg failed with a long message
    import json

    full = json.load(open(os.path.join(REPO, "profiles", "r05_bench.json")))
    txt = bench.compact_line(full, "bench_extras.json")
    assert len(txt) < 4096 < bench.LINE_LIMIT == 6000 and "\n" not in txt
    line = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "legs"):
        assert k in line, k
    assert line["value"] == full["value"] and line["roofline"]["achieved"] == full["roofline"]["achieved"] and line["roofline"]["frac"] == full["roofline"]["frac"]
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and "model" not in line["config"]
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and line["cpu_baseline"]["reference"]["same_host"] is False
    assert set(line["legs"]) >= {"permexdc", "scim", "scim_constspeed", "scim_error_controlled", "pmsm_c5_shard", "at_scale"}
    assert all(set(v) >= {"frac", "launch_ms"} for v in line["legs"].values())
    bad = dict(full, configs={k: {"error": "x" * 5000} for k in bench.LEGS}, extras_error={"error": "y" * 9000}, overrides={f"GEMX_{i}": "z" * 100 for i in range(30)},
               gather={"chunk": {"error": "q" * 3000}}, rccl={"error": "q" * 3000}, config5={"error": "e" * 3000}, cpu_baseline={"error": "c" * 3000})
    txt = bench.compact_line(bad, "bench_extras.json")
    assert len(txt) < bench.LINE_LIMIT and json.loads(txt)["value"] == full["value"]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "0"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "--steps >= 1" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    import torch

    if not torch.cuda.is_available():  # self-spawn without enough GPUs: a clear message
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr


# ------------------------------------------------------------------------------------------------ the reference env shell's transcript
def _decode(v):
    """inverse of oracle/make_shell_transcript.py:encode"""
    if isinstance(v, dict):
        if "f" in v:
            return float(v["f"])
        if "nd" in v:
            return np.array([float(x) for x in v["nd"]], dtype=np.float64).reshape(v["shape"])
        if "seq" in v:
            s = [_decode(x) for x in v["seq"]]
            return tuple(s) if v["tuple"] else s
        if "dict" in v:
            return {k: _decode(x) for k, x in v["dict"].items()}
        return v  # Box / Discrete / MultiDiscrete / object descriptors
    return v


def shell_transcript(name):
    import json

    with open(os.path.join(REPO, "tests", "golden", f"shell_{name}.json")) as fh:
        return json.load(fh)


def check_shell_get(ps, entry):
    """One attribute READ of the reference env shell against the replacement physical system."""
    name, want = entry["name"], _decode(entry["value"])
    got = getattr(ps, name)
    if isinstance(want, dict) and "Box" in want:
        assert np.allclose(np.asarray(got.low, dtype=float), _decode(want["Box"]["low"]), rtol=1e-13, atol=0), name
        assert np.allclose(np.asarray(got.high, dtype=float), _decode(want["Box"]["high"]), rtol=1e-13, atol=0), name
        assert list(got.shape) == want["Box"]["shape"]
    elif isinstance(want, dict) and "Discrete" in want:
        assert int(got.n) == want["Discrete"]
    elif isinstance(want, np.ndarray):
        assert np.allclose(np.asarray(got, dtype=float), want, rtol=1e-13, atol=0), (name, got, want)
    elif isinstance(want, float):
        assert got == pytest.approx(want, rel=1e-15), name
    else:
        assert (list(got) if isinstance(want, list) else got) == want, (name, got, want)


@pytest.mark.parametrize("name", ["Cont-CC-PermExDc-v0", "Finite-CC-PMSM-v0", "Cont-SC-SCIM-v0", "Finite-CC-PMSM-v0_DeadTime2"])
def test_reference_env_shell_reads_are_answered_identically(name):
    """tests/golden/shell_*.json: every attribute the UNMODIFIED reference env shell (ElectricMotorEnvironment + its reference generator,
    reward function, constraint monitor, dashboards; core.py:197-371) read from its physical system, from construction over a 200-step
    run -- recorded by oracle/make_shell_transcript.py through a proxy.  Host part: the replacement offers every name the shell touched and
    answers every metadata read identically (the calls -- simulate / reset -- are replayed on the GPU: tests/test_gpu_parity.py)."""
    doc = shell_transcript(name)
    wr = (ga.DeadTimeProcessor(steps=2),) if doc["wrappers"] else ()
    ps = ga.make(doc["env_id"], n_envs=1, physical_system_wrappers=wr, _defer_create=True).physical_system
    assert set(doc["attribute_names"]) == {"action_space", "close", "k", "limits", "nominal_state", "reset", "simulate", "state_names",
                                           "state_positions", "state_space", "tau"}  # the whole surface the shell uses
    for n in doc["attribute_names"]:
        assert hasattr(ps, n), n
    n_get = 0
    for e in doc["log"]:
        if e["op"] == "get" and e["name"] != "k":  # (k counts simulate() calls: checked in the GPU replay)
            check_shell_get(ps, e)
            n_get += 1
    assert n_get > 25 and not any(e["op"] == "set" for e in doc["log"])


def test_register_budget_of_the_bench_kernels():
    """The kernels behind bench.py's legs keep their occupancy line and stay out of scratch memory: VGPR count, spilled VGPRs and private
    segment of the built code objects (llvm-readelf on the code objects of gym_electric_motor_amd/libgemx_u*.so, as tools/vgpr_report.py reads them).  Round 4
    lost a leg twice to an edit elsewhere in the kernel: four pinned registers took the error-controlled <2, 2> SCIM kernel across the
    128-register line (0.148 -> 0.076 of the roofline), and a rolled tail loop put 1.2 KB of scratch into it (0.146 -> 0.05)."""
    import importlib.util
    import re

    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    units = {u: os.path.join(REPO, "gym_electric_motor_amd", f"libgemx_u{u}.so") for u in ("0_0_0", "1_1_0", "2_2_0")}
    if not os.path.exists(readelf) or not all(os.path.exists(o) for o in units.values()):
        pytest.skip("no built objects / llvm-readelf here")
    spec = importlib.util.spec_from_file_location("vgpr_report", os.path.join(REPO, "tools", "vgpr_report.py"))
    vr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vr)
    found = {}
    for u, o in units.items():
        for name, vgpr, spilled, scratch, _sg in vr.kernels_of(o):
            found[re.sub(r"(, false)+>$", ">", name.strip("`"))] = (vgpr, spilled, scratch)  # (trailing defaulted template flags dropped)
    budget = {  # kernel -> (VGPRs <=, spilled VGPRs <=, scratch bytes <=)
        "advance_pipe_kernel<1, 1, 0, 1, false, float, 12, 6>": (128, 0, 32),   # headline
        "advance_pipe_kernel<1, 1, 0, 1, false, float, 12, 3>": (168, 0, 32),   # headline, long launches (paced)
        "advance_pipe_kernel<1, 1, 0, 1, false, float, 4, 2>": (128, 0, 0),     # config 5's shard, 1M envs
        "advance_pipe_kernel<2, 2, 1, 1, false, float, 2, 2>": (128, 0, 0),     # config 4
        "advance_pipe_kernel<2, 2, 1, 2, false, float, 2, 2>": (128, 8, 64),    # config 4, error-controlled
        "dc_stream_kernel<0, 0, 0, float, 32>": (128, 0, 0),                           # config 2
    }
    for k, (v_max, sp_max, sc_max) in budget.items():
        assert k in found, (k, sorted(found)[:5])
        v, sp, sc = found[k]
        assert v <= v_max and sp <= sp_max and sc <= sc_max, (k, "VGPRs / spilled / scratch bytes", (v, sp, sc), "budget", (v_max, sp_max, sc_max))
