#!/usr/bin/env python3
"""Generate golden vectors by executing the UNMODIFIED reference (gym-electric-motor 3.0.2).

TEST / ORACLE INFRASTRUCTURE ONLY -- never imported by the product package.

Run in the build container (where /root/reference exists):

    MPLBACKEND=Agg python oracle/make_golden.py

It puts `oracle/gymnasium_standin` (gymnasium is not installed, no network) and
`/root/reference/src` on sys.path, drives the reference envs with seeded action
tensors and writes small `.npz` fixtures to `tests/golden/`.  The GPU box has no
/root/reference; tests there only read the committed fixtures.

Every fixture stores: the actions, the normalised states the reference returned
(`env.step(...)[0][0]`, i.e. `SCMLSystem.simulate()/limits`,
physical_systems.py:203/525/814), the terminated flags, the reset state and a
JSON `meta` blob with every parameter needed to rebuild the case without the
reference (motor/load parameters, limits, tau, interlocking time, model constants).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("GEM_REFERENCE", "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, os.path.join(HERE, "gymnasium_standin"))
sys.path.insert(0, os.path.join(REF, "src"))

import numpy as np  # noqa: E402

import gym_electric_motor as gem  # noqa: E402
from gym_electric_motor import physical_systems as ps  # noqa: E402
from gym_electric_motor.physical_systems import solvers as ref_solvers  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def make_solver(name):
    if name == "euler":
        return ref_solvers.EulerSolver()
    if name == "euler4":
        return ref_solvers.EulerSolver(nsteps=4)
    if name == "dopri5":
        return ref_solvers.ScipyOdeSolver()  # the default of all 54 envs
    if name == "ivp":  # BASELINE config 1: solve_ivp with its defaults (method RK45, rtol 1e-3, atol 1e-6)
        return ref_solvers.ScipySolveIvpSolver()
    if name == "ivp_tight":
        return ref_solvers.ScipySolveIvpSolver(rtol=1e-10, atol=1e-12)
    raise KeyError(name)


def gen_actions(space_kind, K, seed, mode):
    """uniform: iid every step.  held: piecewise-constant for random dwell times."""
    rng = np.random.default_rng(seed)
    if space_kind == "box1":
        shape, disc = (K, 1), False
    elif space_kind == "box3":
        shape, disc = (K, 3), False
    elif space_kind == "box2":
        shape, disc = (K, 2), False
    elif space_kind == "box4":
        shape, disc = (K, 4), False
    elif space_kind == "box6":
        shape, disc = (K, 6), False
    elif space_kind in ("disc8", "disc4"):
        shape, disc = (K,), True
    elif space_kind in ("mdisc44", "mdisc84", "mdisc88"):  # MultiDiscrete of a FiniteMultiConverter (converters.py:546)
        shape, disc = (K, 2), True
    else:
        raise KeyError(space_kind)
    nact = 4 if space_kind == "disc4" else 8
    if space_kind.startswith("mdisc"):
        nact = np.array([int(space_kind[5]), int(space_kind[6])])
    if mode == "uniform":
        return rng.integers(0, nact, shape).astype(np.int64) if disc else rng.uniform(-1, 1, shape)
    # held
    out = np.zeros(shape, dtype=np.int64 if disc else np.float64)
    k = 0
    while k < K:
        dwell = int(rng.integers(1, 40))
        if disc:
            v = rng.integers(0, nact)  # array-valued `nact` -> one draw per sub-converter
        else:
            v = rng.uniform(-1, 1, shape[1:]) * rng.uniform(0.0, 1.0)
        out[k : k + dwell] = v
        k += dwell
    return out


def describe(env):
    """Everything a restatement needs, read from the live reference objects."""
    psys = env.physical_system.unwrapped
    motor, load, conv, sup = psys.electrical_motor, psys.mechanical_load, psys.converter, psys.supply
    meta = dict(
        system=type(psys).__name__,
        motor=type(motor).__name__,
        load=type(load).__name__,
        converter=type(conv).__name__,
        supply=type(sup).__name__,
        state_names=list(psys.state_names),
        limits=[float(x) for x in psys.limits],
        nominal_state=[float(x) for x in psys.nominal_state],
        tau=float(psys.tau),
        interlocking_time=float(conv._interlocking_time),  # multi converters: overwritten below
        u_nominal=float(sup.u_nominal),
        motor_parameter={k: float(v) for k, v in motor.motor_parameter.items()},
        model_constants=np.asarray(motor._model_constants, dtype=float).tolist(),
        j_total=float(load.j_total),
    )
    if isinstance(sup, ps.RCVoltageSupply):
        meta["supply_parameter"] = dict(R=float(sup._r), C=float(sup._c))
    subs = getattr(conv, "_sub_converters", None) if type(conv).__name__.endswith("MultiConverter") else None
    if subs is not None:
        # Cont/FiniteMultiConverter: the holder's own interlocking time is never used; the sub-converters' are
        meta["converter"] += "[" + ",".join(type(sc).__name__ for sc in subs) + "]"
        tils = {float(sc._interlocking_time) for sc in subs}
        assert len(tils) == 1, tils
        meta["interlocking_time"] = tils.pop()
    if isinstance(load, ps.PolynomialStaticLoad):
        meta["load_parameter"] = {k: float(v) for k, v in load.load_parameter.items()}
        meta["tau_decay"] = float(load.tau_decay)
    if isinstance(load, ps.ConstantSpeedLoad):
        meta["omega_fixed"] = float(load.omega_fixed)
    return meta


class _StepRecorder:
    """gym_electric_motor.core.Callback-shaped recorder of the reference the reward was computed against (core.py:346-362)."""

    def __init__(self):
        self.references, self.rewards = [], []

    def set_env(self, env):
        pass

    def on_reset_begin(self):
        pass

    def on_reset_end(self, *_):
        pass

    def on_step_begin(self, *_):
        pass

    def on_step_end(self, k, state, reference, reward, terminated):
        self.references.append(np.array(reference, dtype=float))
        self.rewards.append(float(reward))

    def on_close(self):
        pass


def run_case(name, env_id, solver, K, seed, mode, episodic, space_kind, every=1, action_frame="abc", dead_time_steps=0,
             record_reward=False, dead_time_reset_action=None, **make_kwargs):
    """action_frame: 'abc' | 'dq' (system.control_space = 'dq') | 'dq_processor' (DqToAbcActionProcessor wrapper);
    dead_time_steps > 0: DeadTimeProcessor(steps) wrapped INSIDE the dq processor, as the reference's processors expect."""
    from gym_electric_motor.physical_system_wrappers import DeadTimeProcessor, DqToAbcActionProcessor

    kw = dict(make_kwargs)
    wrappers = []
    if dead_time_steps:
        if dead_time_reset_action is None:
            wrappers.append(DeadTimeProcessor(steps=dead_time_steps))
        else:  # a custom reset action (dead_time_processor.py:27-50): `steps` copies of one action of the inner system's action space
            ra = dead_time_reset_action
            one = int(ra[0]) if space_kind.startswith("disc") else (np.array(ra, dtype=np.int64) if space_kind.startswith("mdisc") else np.array(ra, dtype=np.float64))
            wrappers.append(DeadTimeProcessor(steps=dead_time_steps, reset_action=lambda: [one] * dead_time_steps))
    if action_frame == "dq_processor":
        wrappers.append(DqToAbcActionProcessor.make("EESM" if "EESM" in env_id else "PMSM"))
    if wrappers:
        kw["physical_system_wrappers"] = tuple(wrappers)
    kw["ode_solver"] = make_solver(solver)
    if not episodic:
        kw["constraints"] = ()
    recorder = _StepRecorder()
    if record_reward:
        kw["callbacks"] = (recorder,)
    env = gem.make(env_id, **kw)
    if action_frame == "dq":  # no env class forwards control_space; the attribute is read at simulate() time (lines 491, 777)
        env.physical_system.unwrapped.control_space = "dq"
    if action_frame != "abc":
        env._callbacks = [recorder] if record_reward else []  # the default dashboard's action plots index three abc actions
    (s0, _), _ = env.reset(seed=0)
    # physical-system wrappers that only post-process the observation (e.g. the shunt envs' CurrentSumProcessor, which
    # appends 'i_sum') are outside the path: keep the columns of the unwrapped physical system
    n_keep = len(env.physical_system.unwrapped.state_names)
    s0 = s0[:n_keep]
    actions = gen_actions(space_kind, K, seed, mode)
    states = np.zeros((K, len(s0)))
    term = np.zeros(K, dtype=bool)
    for k in range(K):
        a = actions[k]
        if space_kind.startswith("disc"):
            a = int(a)
        elif space_kind.startswith("mdisc"):
            a = np.array(a, dtype=np.int64)
        (s, _), _, terminated, _, _ = env.step(a)
        states[k] = s[:n_keep]
        term[k] = terminated
        if terminated:
            env.reset()
    meta = describe(env)
    meta.update(name=name, env_id=env_id, solver=solver, K=K, seed=seed, mode=mode, episodic=bool(episodic),
                every=every, constraints=("default" if episodic else "none"), action_frame=action_frame,
                dead_time_steps=int(dead_time_steps))
    if dead_time_reset_action is not None:
        meta["dead_time_reset_action"] = [float(x) for x in dead_time_reset_action]
    idx = np.arange(K)
    keep = idx[(idx % every == every - 1)] if every > 1 else idx
    extra = {}
    if record_reward:
        rf = env._reward_function  # WeightedSumOfErrors (reward_functions/weighted_sum_of_errors.py:88-129)
        meta["reward"] = dict(weights=[float(x) for x in rf._reward_weights], powers=[float(x) for x in np.asarray(rf._n, dtype=float)],
                              state_length=[float(x) for x in rf._state_length], bias=float(rf._bias),
                              violation_reward=float(rf._violation_reward),
                              referenced_states=[bool(x) for x in env.reference_generator.referenced_states])
        extra = dict(references=np.asarray(recorder.references)[:, :n_keep], rewards=np.asarray(recorder.rewards))
        assert len(recorder.rewards) == K
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        **extra,
        actions=actions.astype(np.uint8) if "disc" in space_kind else actions,
        states=states[keep],
        state_index=keep.astype(np.int64),
        terminated=term,
        reset_state=np.asarray(s0, dtype=float),
        meta=np.array(json.dumps(meta)),
    )
    print(f"{name:48s} K={K} terminated={int(term.sum()):5d} max|x|={np.abs(states).max():.3f}")


def replay_ref_data():
    """The reference's only golden trajectory: tests/integration_tests/test_integration.py:18-97
    (Cont-SC-PermExDc-v0, PI cascade, seed 1337, 2001 steps) -> ref_data.npz.  We re-run it to record
    the ACTIONS the controller emitted (ref_data.npz holds states only), check we reproduce the stored
    states, and save actions + the stored states as a fixture the C oracle can be pinned against."""
    sys.path.insert(0, os.path.join(REF, "examples", "classic_controllers"))
    from classic_controllers import Controller
    from gym_electric_motor.reference_generators import SinusoidalReferenceGenerator

    ref = np.load(os.path.join(REF, "tests", "integration_tests", "ref_data.npz"))
    gen = SinusoidalReferenceGenerator(amplitude_range=(1, 1), frequency_range=(5, 5), offset_range=(0, 0),
                                       episode_lengths=(10001, 10001))
    env = gem.make("Cont-SC-PermExDc-v0", reference_generator=gen)
    controller = Controller.make(env)
    (state, reference), _ = env.reset(seed=1337)
    s0 = np.array(state, dtype=float)
    acts, states, terms = [], [], []
    for _ in range(2001):
        action = controller.control(state, reference)
        acts.append(np.array(action, dtype=float).reshape(-1))
        (state, reference), _, terminated, _, _ = env.step(action)
        states.append(state)
        terms.append(terminated)
        if terminated:
            env.reset()
            controller.reset()
    states = np.asarray(states)
    err = np.abs(states - ref["states"]).max()
    assert err < 1e-12, err
    assert np.array_equal(np.asarray(terms), ref["terminations"])
    meta = describe(env)
    meta.update(name="refdata_cont_sc_permexdc_dopri5", env_id="Cont-SC-PermExDc-v0", solver="dopri5", K=2001,
                episodic=True, every=1, constraints="default", repro_err=float(err),
                source="reference tests/integration_tests/ref_data.npz (states, terminations)")
    np.savez_compressed(
        os.path.join(OUT, "refdata_cont_sc_permexdc_dopri5.npz"),
        actions=np.asarray(acts),
        states=ref["states"],
        state_index=np.arange(2001),
        terminated=ref["terminations"],
        reset_state=s0,
        meta=np.array(json.dumps(meta)),
    )
    print(f"ref_data.npz replay: reproduced with max|d|={err:.2e}; terminated={int(np.sum(terms))}")


def converter_kats():
    """Known-answer tables for the converters on the path, produced by the reference classes
    (converters.py:404-495 Cont-4QC, 743-839 Finite-B6C, 842-911 Cont-B6C), with and without interlocking."""
    rng = np.random.default_rng(7)
    out = {}
    # Cont-4QC: convert(i, t) for grids of action x current sign
    acts = np.concatenate([np.linspace(-1.5, 1.5, 31), rng.uniform(-1, 1, 20)])
    curr = np.array([-3.0, 0.0, 2.5])
    for t_il in (0.0, 1e-6, 5e-6):
        conv = ps.ContFourQuadrantConverter(tau=1e-4, interlocking_time=t_il)
        tab = np.zeros((len(acts), len(curr)))
        for i, a in enumerate(acts):
            conv.reset()
            conv.set_action(np.array([a]), 0.0)
            for j, c in enumerate(curr):
                tab[i, j] = conv.convert([c], 0.0)[0]
        out[f"c4qc_til{t_il:g}"] = tab
        out["c4qc_actions"] = acts
        out["c4qc_currents"] = curr
    # Cont-B6C
    acts = rng.uniform(-1.3, 1.3, (40, 3))
    curr = rng.uniform(-1, 1, (40, 3))
    curr[:5] = 0.0
    for t_il in (0.0, 1e-6):
        conv = ps.ContB6BridgeConverter(tau=1e-4, interlocking_time=t_il)
        tab = np.zeros((40, 3))
        for i in range(40):
            conv.reset()
            conv.set_action(acts[i], 0.0)
            tab[i] = conv.convert(curr[i], 0.0)
        out[f"cb6_til{t_il:g}"] = tab
        out["cb6_actions"] = acts
        out["cb6_currents"] = curr
    # Finite-B6C: a persistent converter driven as SynchronousMotorSystem.simulate does
    # (physical_systems.py:494-511): set_action(a, t); for each switching time: convert(i, t_segment_start).
    acts = rng.integers(0, 8, 200)
    curr = rng.uniform(-1, 1, (200, 2, 3))
    for t_il in (0.0, 1e-6):
        tau = 1e-5
        conv = ps.FiniteB6BridgeConverter(tau=tau, interlocking_time=t_il)
        conv.reset()
        nseg = np.zeros(200, dtype=np.int64)
        volt = np.zeros((200, 2, 3))
        t = 0.0
        for k in range(200):
            times = conv.set_action(int(acts[k]), t)
            nseg[k] = len(times)
            t_seg = t
            for s, t_sw in enumerate(times):
                volt[k, s] = conv.convert(curr[k, s], t_seg)
                t_seg = t_sw
            t = t + tau
            if k == 100:
                conv.reset()  # switching state must survive reset() (converters.py:45-54)
        out[f"fb6_til{t_il:g}_nseg"] = nseg
        out[f"fb6_til{t_il:g}_volt"] = volt
        out["fb6_actions"] = acts
        out["fb6_currents"] = curr
    # Finite-4QC (converters.py:313-368) driven as SCMLSystem.simulate does, incl. dead time and reset()
    acts4 = rng.integers(0, 4, 200)
    curr4 = rng.uniform(-1, 1, (200, 2))
    for t_il in (0.0, 1e-6):
        tau = 1e-5
        conv = ps.FiniteFourQuadrantConverter(tau=tau, interlocking_time=t_il)
        conv.reset()
        nseg = np.zeros(200, dtype=np.int64)
        volt = np.zeros((200, 2))
        t = 0.0
        for k in range(200):
            times = conv.set_action(int(acts4[k]), t)
            nseg[k] = len(times)
            t_seg = t
            for sgm, t_sw in enumerate(times):
                volt[k, sgm] = conv.convert([curr4[k, sgm]], t_seg)[0]
                t_seg = t_sw
            t = t + tau
            if k == 100:
                conv.reset()
        out[f"f4qc_til{t_il:g}_nseg"] = nseg
        out[f"f4qc_til{t_il:g}_volt"] = volt
    out["f4qc_actions"] = acts4
    out["f4qc_currents"] = curr4
    np.savez_compressed(os.path.join(OUT, "converter_kats.npz"), **out)
    print("converter KATs written")


def main(only=None):
    """`only`: optional set of groups to (re)generate -- {"multi", "dfim"}; default: everything."""
    os.makedirs(OUT, exist_ok=True)
    if not only:
        main_base()
    if not only or "multi" in only:
        main_multi()
    if not only or "dfim" in only:
        main_dfim()
    if not only or "wrappers" in only:
        main_wrappers()
    if not only or "reward" in only:
        main_reward()
    if not only or "supply" in only:
        main_supply()
    if not only or "init" in only:
        init_samples()
    if only and "reset_action" in only:
        main_reset_action()
    if only and "init_r04" in only:
        init_samples(only=INIT_CASES_R04)
    if not only or "wiener" in only:
        wiener_samples()
    if not only or "r02" in only:
        main_r02()
    if not only or "defaults" in only:
        main_defaults()


ALL_ENV_IDS = [f"{a}-{c}-{m}-v0" for m in ("PermExDc", "SeriesDc", "ShuntDc", "ExtExDc", "PMSM", "SynRM", "SCIM", "EESM", "DFIM")
               for c in ("CC", "TC", "SC") for a in ("Cont", "Finite")]


def default_space_kind(env_id):
    finite, motor = env_id.startswith("Finite"), env_id.split("-")[2]
    if motor in ("PermExDc", "SeriesDc", "ShuntDc"):
        return "disc4" if finite else "box1"
    if motor == "ExtExDc":
        return "mdisc44" if finite else "box2"
    if motor == "EESM":
        return "mdisc84" if finite else "box4"
    if motor == "DFIM":
        return "mdisc88" if finite else "box6"
    return "disc8" if finite else "box3"


def main_defaults():
    """Round 3: every one of the reference's 54 env ids EXACTLY as `gem.make(env_id)` builds it -- default supply, converter, motor, load,
    tau, constraints and the default solver (scipy dopri5) -- driven with piecewise-constant random actions, reset on termination.  The
    device side is tested as `gym_electric_motor_amd.make(env_id)` hands it to a user (its per-id default solver included)."""
    for i, env_id in enumerate(ALL_ENV_IDS):
        slug = env_id[:-3].replace("-", "_").lower()
        run_case(f"default_{slug}_dopri5", env_id, "dopri5", 800, 3000 + i, "held", True, default_space_kind(env_id))


def main_reset_action():
    """Round 4 (SURVEY 8f rank 2 leftover): DeadTimeProcessor(steps, reset_action=...) -- the deque refilled with a NON-zero action at every
    reset (dead_time_processor.py:27-50, 63-72): continuous and discrete actions, episodic (refills inside the run), alone and inside the
    dq processor."""
    K = 1200
    run_case("pmsm_cont_dead2_reset_epi_held_euler", "Cont-CC-PMSM-v0", "euler", K, 1320, "held", True, "box3", dead_time_steps=2,
             dead_time_reset_action=[0.4, -0.3, 0.1])
    run_case("pmsm_fin_dead3_reset_epi_uniform_tau1e-4_euler", "Finite-CC-PMSM-v0", "euler", K, 1321, "uniform", True, "disc8", tau=1e-4,
             dead_time_steps=3, dead_time_reset_action=[5])
    run_case("permexdc_cont_dead1_reset_epi_held_euler", "Cont-CC-PermExDc-v0", "euler", K, 1322, "held", True, "box1", dead_time_steps=1,
             dead_time_reset_action=[-0.03])
    run_case("pmsm_cont_dqproc_dead2_reset_epi_held_euler", "Cont-CC-PMSM-v0", "euler", K, 1323, "held", True, "box2",
             action_frame="dq_processor", dead_time_steps=2, dead_time_reset_action=[0.2, 0.5, -0.4])
    run_case("extex_fin_dead2_reset_epi_uniform_euler", "Finite-CC-ExtExDc-v0", "euler", K, 1324, "uniform", True, "mdisc44", dead_time_steps=2,
             dead_time_reset_action=[2, 1])


def main_r02():
    """Round 2: BASELINE config 1's solver (ScipySolveIvpSolver, solvers.py:187-219) and the bench's own configuration (PMSM,
    tau = 1e-4, uniformly random switching, default constraint, reset on done) under the reference's default solver."""
    dc, pmsm = "Cont-CC-PermExDc-v0", "Finite-CC-PMSM-v0"
    run_case("permexdc_free_uniform_10k_ivp", dc, "ivp", 10000, 1234, "uniform", False, "box1", every=10)
    run_case("permexdc_free_uniform_10k_ivp_tight", dc, "ivp_tight", 10000, 1234, "uniform", False, "box1", every=10)
    run_case("permexdc_epi_held_ivp", dc, "ivp", 2000, 1236, "held", True, "box1")
    run_case("pmsm_free_held_ivp", pmsm, "ivp", 2000, 1235, "held", False, "disc8")
    run_case("scim_free_held_ivp", "Cont-SC-SCIM-v0", "ivp", 2000, 1235, "held", False, "box3")
    run_case("pmsm_epi_uniform_tau1e-4_dopri5", pmsm, "dopri5", 6000, 1290, "uniform", True, "disc8", tau=1e-4)
    run_case("pmsm_epi_uniform_tau1e-4_euler", pmsm, "euler", 6000, 1290, "uniform", True, "disc8", tau=1e-4)
    run_case("scim_epi_held_dopri5", "Cont-SC-SCIM-v0", "dopri5", 4000, 1291, "held", True, "box3")


def main_base():
    replay_ref_data()
    converter_kats()
    dc, pmsm, scim = "Cont-CC-PermExDc-v0", "Finite-CC-PMSM-v0", "Cont-SC-SCIM-v0"
    K = 2000
    # --- config 1/2: Cont-CC-PermExDc-v0 -------------------------------------------------------------
    for solver in ("euler", "dopri5"):
        run_case(f"permexdc_free_uniform_{solver}", dc, solver, K, 1234, "uniform", False, "box1")
        run_case(f"permexdc_free_held_{solver}", dc, solver, K, 1235, "held", False, "box1")
        run_case(f"permexdc_epi_held_{solver}", dc, solver, K, 1236, "held", True, "box1")
    run_case("permexdc_epi_uniform_euler", dc, "euler", K, 1234, "uniform", True, "box1")
    run_case("permexdc_free_held_euler4", dc, "euler4", K, 1235, "held", False, "box1")
    run_case("permexdc_free_held_til_euler", dc, "euler", K, 1235, "held", False, "box1",
             converter=dict(interlocking_time=2e-6))
    run_case("permexdc_free_uniform_10k_euler", dc, "euler", 10000, 1234, "uniform", False, "box1", every=10)
    run_case("permexdc_free_uniform_10k_dopri5", dc, "dopri5", 10000, 1234, "uniform", False, "box1", every=10)
    # Cont-SC-PermExDc (PolynomialStaticLoad) -- the env of ref_data.npz
    run_case("permexdc_sc_free_held_dopri5", "Cont-SC-PermExDc-v0", "dopri5", K, 1237, "held", False, "box1")
    run_case("permexdc_sc_free_held_euler", "Cont-SC-PermExDc-v0", "euler", K, 1237, "held", False, "box1")
    # --- config 3: Finite-CC-PMSM-v0 -----------------------------------------------------------------
    for solver in ("euler", "dopri5"):
        run_case(f"pmsm_free_uniform_{solver}", pmsm, solver, K, 1234, "uniform", False, "disc8")
        run_case(f"pmsm_free_held_{solver}", pmsm, solver, K, 1235, "held", False, "disc8")
        run_case(f"pmsm_free_uniform_tau1e-4_{solver}", pmsm, solver, K, 1234, "uniform", False, "disc8", tau=1e-4)
        run_case(f"pmsm_free_held_til_{solver}", pmsm, solver, K, 1235, "held", False, "disc8",
                 converter=dict(interlocking_time=1e-6))
        run_case(f"pmsm_free_uniform_til_{solver}", pmsm, solver, K, 1234, "uniform", False, "disc8",
                 converter=dict(interlocking_time=1e-6))
        run_case(f"pmsm_epi_held_tau1e-4_{solver}", pmsm, solver, 4000, 1238, "held", True, "disc8", tau=1e-4)
    run_case("pmsm_free_uniform_10k_dopri5", pmsm, "dopri5", 10000, 1234, "uniform", False, "disc8", every=10)
    run_case("pmsm_free_uniform_10k_euler", pmsm, "euler", 10000, 1234, "uniform", False, "disc8", every=10)
    run_case("pmsm_sc_free_held_dopri5", "Finite-SC-PMSM-v0", "dopri5", K, 1239, "held", False, "disc8")
    # --- config 4: Cont-SC-SCIM-v0 -------------------------------------------------------------------
    for solver in ("euler", "dopri5"):
        run_case(f"scim_free_uniform_{solver}", scim, solver, K, 1234, "uniform", False, "box3")
        run_case(f"scim_free_held_{solver}", scim, solver, K, 1235, "held", False, "box3")
        run_case(f"scim_epi_uniform_{solver}", scim, solver, K, 1234, "uniform", True, "box3")
        run_case(f"scim_constspeed_free_held_{solver}", scim, solver, K, 1235, "held", False, "box3",
                 load=ps.ConstantSpeedLoad(omega_fixed=100.0))
    run_case("scim_free_held_til_euler", scim, "euler", K, 1235, "held", False, "box3",
             converter=dict(interlocking_time=2e-6))
    run_case("scim_free_uniform_10k_dopri5", scim, "dopri5", 10000, 1234, "uniform", False, "box3", every=10)
    # --- further motors / converters on the same kernel skeleton ----------------------
    for solver in ("euler", "dopri5"):
        run_case(f"synrm_fin_free_held_{solver}", "Finite-CC-SynRM-v0", solver, K, 1240, "held", False, "disc8")
        # episodic: a free run of this env (J = 0.81e-3 kg m^2) leaves the limits by 20x into a regime where explicit
        # Euler at tau = 1e-4 is oscillatory-unstable and amplifies rounding by ~1e4 -- not a meaningful parity trajectory
        run_case(f"synrm_cont_sc_epi_held_{solver}", "Cont-SC-SynRM-v0", solver, K, 1241, "held", True, "box3")
        run_case(f"permexdc_fin_free_held_{solver}", "Finite-CC-PermExDc-v0", solver, K, 1242, "held", False, "disc4")
        run_case(f"series_cont_free_held_{solver}", "Cont-CC-SeriesDc-v0", solver, K, 1243, "held", False, "box1")
        run_case(f"series_cont_sc_free_held_{solver}", "Cont-SC-SeriesDc-v0", solver, K, 1244, "held", False, "box1")
        run_case(f"shunt_cont_free_held_{solver}", "Cont-CC-ShuntDc-v0", solver, K, 1245, "held", False, "box1")
        run_case(f"shunt_cont_sc_free_held_{solver}", "Cont-SC-ShuntDc-v0", solver, K, 1246, "held", False, "box1")
    run_case("synrm_fin_epi_held_tau1e-4_euler", "Finite-CC-SynRM-v0", "euler", 4000, 1247, "held", True, "disc8", tau=1e-4)
    # uniform actions: with "held" actions the run starts with a long zero-vector phase at exactly zero current, where the
    # reference's freewheeling-diode direction in the dead state is decided by ~1e-18 A of matmul rounding noise
    run_case("synrm_fin_free_uniform_til_euler", "Finite-CC-SynRM-v0", "euler", K, 1240, "uniform", False, "disc8",
             converter=dict(interlocking_time=1e-6))
    run_case("permexdc_fin_free_held_til_euler", "Finite-CC-PermExDc-v0", "euler", K, 1242, "held", False, "disc4",
             converter=dict(interlocking_time=1e-6))
    run_case("permexdc_fin_free_uniform_til_euler", "Finite-CC-PermExDc-v0", "euler", K, 1248, "uniform", False, "disc4",
             converter=dict(interlocking_time=1e-6))
    run_case("permexdc_fin_epi_held_euler", "Finite-CC-PermExDc-v0", "euler", K, 1249, "held", True, "disc4")
    run_case("series_fin_free_held_til_euler", "Finite-CC-SeriesDc-v0", "euler", K, 1250, "held", False, "disc4",
             converter=dict(interlocking_time=1e-6))
    run_case("series_cont_epi_held_euler", "Cont-CC-SeriesDc-v0", "euler", K, 1251, "held", True, "box1")
    run_case("shunt_fin_free_held_til_euler", "Finite-CC-ShuntDc-v0", "euler", K, 1252, "held", False, "disc4",
             converter=dict(interlocking_time=1e-6))
    run_case("shunt_cont_epi_held_euler", "Cont-CC-ShuntDc-v0", "euler", K, 1253, "held", True, "box1")


def main_multi():
    """Multi-converter systems: ExtExDc = 2 x 4QC, EESM = B6 + 4QC."""
    K = 2000
    xc, xf = "Cont-CC-ExtExDc-v0", "Finite-CC-ExtExDc-v0"
    for solver in ("euler", "dopri5"):
        run_case(f"extex_cont_free_held_{solver}", xc, solver, K, 1260, "held", False, "box2")
        run_case(f"extex_cont_sc_free_held_{solver}", "Cont-SC-ExtExDc-v0", solver, K, 1261, "held", False, "box2")
        run_case(f"extex_fin_free_held_{solver}", xf, solver, K, 1262, "held", False, "mdisc44")
        run_case(f"eesm_cont_free_held_{solver}", "Cont-CC-EESM-v0", solver, K, 1270, "held", False, "box4")
        # episodic: the default EESM parameter set has sigma < 0 (an exponentially unstable d/e-axis pair), a free run
        # of the speed-control env overflows
        run_case(f"eesm_cont_sc_epi_held_{solver}", "Cont-SC-EESM-v0", solver, K, 1271, "held", True, "box4")
        run_case(f"eesm_fin_free_held_{solver}", "Finite-CC-EESM-v0", solver, K, 1272, "held", False, "mdisc84")
    run_case("extex_cont_free_uniform_euler", xc, "euler", K, 1263, "uniform", False, "box2")
    run_case("extex_cont_epi_held_euler", xc, "euler", K, 1264, "held", True, "box2")
    run_case("extex_fin_epi_held_euler", xf, "euler", K, 1265, "held", True, "mdisc44")
    run_case("extex_fin_free_uniform_euler", xf, "euler", K, 1266, "uniform", False, "mdisc44")
    # dead time: `converter=dict(interlocking_time=...)` only reaches the holder, so build the sub-converters by hand
    run_case("extex_fin_free_held_til_euler", xf, "euler", K, 1267, "held", False, "mdisc44",
             converter=ps.FiniteMultiConverter(subconverters=[ps.FiniteFourQuadrantConverter(interlocking_time=1e-6),
                                                              ps.FiniteFourQuadrantConverter(interlocking_time=1e-6)]))
    run_case("extex_fin_free_uniform_til_euler", xf, "euler", K, 1268, "uniform", False, "mdisc44",
             converter=ps.FiniteMultiConverter(subconverters=[ps.FiniteFourQuadrantConverter(interlocking_time=1e-6),
                                                              ps.FiniteFourQuadrantConverter(interlocking_time=1e-6)]))
    run_case("extex_cont_free_held_til_euler", xc, "euler", K, 1269, "held", False, "box2",
             converter=ps.ContMultiConverter(subconverters=[ps.ContFourQuadrantConverter(interlocking_time=2e-6),
                                                            ps.ContFourQuadrantConverter(interlocking_time=2e-6)]))
    run_case("eesm_cont_free_uniform_euler", "Cont-CC-EESM-v0", "euler", K, 1273, "uniform", False, "box4")
    run_case("eesm_cont_epi_held_euler", "Cont-CC-EESM-v0", "euler", 4000, 1274, "held", True, "box4")
    run_case("eesm_fin_epi_held_tau1e-4_euler", "Finite-CC-EESM-v0", "euler", 4000, 1275, "held", True, "mdisc84", tau=1e-4)
    run_case("eesm_fin_free_uniform_euler", "Finite-CC-EESM-v0", "euler", K, 1276, "uniform", False, "mdisc84")


def main_wrappers():
    """SURVEY 8f rank 1 (control_space='dq') and rank 2 (DqToAbcActionProcessor, DeadTimeProcessor in front of simulate())."""
    K = 2000
    pm, sc, ee = "Cont-CC-PMSM-v0", "Cont-SC-SCIM-v0", "Cont-CC-EESM-v0"
    for solver in ("euler", "dopri5"):
        run_case(f"pmsm_cont_dqspace_free_held_{solver}", pm, solver, K, 1300, "held", False, "box2", action_frame="dq")
        run_case(f"scim_cont_dqspace_free_held_{solver}", sc, solver, K, 1301, "held", False, "box2", action_frame="dq")
        run_case(f"pmsm_cont_dqproc_free_held_{solver}", pm, solver, K, 1302, "held", False, "box2", action_frame="dq_processor")
        run_case(f"pmsm_cont_dqproc_dead2_free_held_{solver}", pm, solver, K, 1303, "held", False, "box2",
                 action_frame="dq_processor", dead_time_steps=2)
    run_case("pmsm_cont_sc_dqproc_dead1_epi_uniform_euler", "Cont-SC-PMSM-v0", "euler", K, 1304, "uniform", True, "box2",
             action_frame="dq_processor", dead_time_steps=1)
    run_case("synrm_cont_dqspace_free_held_euler", "Cont-CC-SynRM-v0", "euler", K, 1305, "held", False, "box2", action_frame="dq")
    run_case("eesm_cont_dqproc_free_held_euler", ee, "euler", K, 1306, "held", False, "box3", action_frame="dq_processor")
    run_case("eesm_cont_dqproc_dead1_epi_held_euler", ee, "euler", 4000, 1307, "held", True, "box3",
             action_frame="dq_processor", dead_time_steps=1)
    # DeadTimeProcessor alone: continuous and discrete actions (reset action zeros / 0), episodic -> deque refilled on reset
    run_case("pmsm_fin_dead1_free_uniform_euler", "Finite-CC-PMSM-v0", "euler", K, 1308, "uniform", False, "disc8", dead_time_steps=1)
    run_case("pmsm_fin_dead3_epi_held_tau1e-4_euler", "Finite-CC-PMSM-v0", "euler", 4000, 1309, "held", True, "disc8", tau=1e-4,
             dead_time_steps=3)
    run_case("permexdc_cont_dead2_epi_held_euler", "Cont-CC-PermExDc-v0", "euler", K, 1310, "held", True, "box1", dead_time_steps=2)
    run_case("scim_cont_dead1_free_held_dopri5", sc, "dopri5", K, 1311, "held", False, "box3", dead_time_steps=1)
    run_case("pmsm_fin_dead1_til_free_uniform_euler", "Finite-CC-PMSM-v0", "euler", K, 1312, "uniform", False, "disc8",
             dead_time_steps=1, converter=dict(interlocking_time=1e-6))


def main_reward():
    """WeightedSumOfErrors reward against the reference the env's generator produced (recorded as data;
    the generators' numpy PCG64 streams themselves are not reproducible on the device)."""
    K = 2000
    run_case("rw_pmsm_cont_cc_epi_held_euler", "Cont-CC-PMSM-v0", "euler", K, 1400, "held", True, "box3", record_reward=True)
    run_case("rw_pmsm_fin_cc_epi_uniform_tau1e-4_euler", "Finite-CC-PMSM-v0", "euler", K, 1401, "uniform", True, "disc8", tau=1e-4,
             record_reward=True)
    run_case("rw_scim_cont_sc_epi_held_euler", "Cont-SC-SCIM-v0", "euler", K, 1402, "held", True, "box3", record_reward=True)
    run_case("rw_permexdc_cont_tc_epi_held_euler", "Cont-TC-PermExDc-v0", "euler", K, 1403, "held", True, "box1", record_reward=True)
    run_case("rw_dfim_cont_cc_epi_held_euler", "Cont-CC-DFIM-v0", "euler", K, 1404, "held", True, "box6", record_reward=True)
    from gym_electric_motor.reward_functions import WeightedSumOfErrors
    run_case("rw_pmsm_cont_cc_pow2_epi_held_euler", "Cont-CC-PMSM-v0", "euler", K, 1405, "held", True, "box3", record_reward=True,
             reward_function=WeightedSumOfErrors(reward_weights=dict(i_sd=0.3, i_sq=0.6, omega=0.1), reward_power=2, bias="positive",
                                                 violation_reward=-7.5, normed_reward_weights=True))
    run_case("rw_eesm_cont_cc_pow_mixed_epi_held_euler", "Cont-CC-EESM-v0", "euler", K, 1406, "held", True, "box4", record_reward=True,
             reward_function=WeightedSumOfErrors(reward_weights=dict(i_sd=0.4, i_sq=0.4, i_e=0.2), reward_power=dict(i_sd=1, i_sq=2, i_e=0.5),
                                                 gamma=0.95))


INIT_CASES = {
    # name: (env_id, motor_initializer, load_initializer)
    "pmsm_sc_uniform": ("Cont-SC-PMSM-v0", dict(random_init="uniform"), dict(random_init="uniform")),
    "permexdc_sc_gauss": ("Cont-SC-PermExDc-v0", dict(random_init="gaussian", random_params=(30.0, 40.0), interval=[[-50.0, 90.0]]),
                          dict(random_init="gaussian", random_params=(100.0, 60.0))),
    "extex_cc_uniform_interval": ("Cont-CC-ExtExDc-v0", dict(random_init="uniform", interval=[[-20.0, 60.0], [0.0, 10.0]]), None),
    "eesm_sc_uniform": ("Cont-SC-EESM-v0", dict(random_init="uniform"), dict(random_init="uniform", interval=[[-100.0, 300.0]])),
    # round 4: induction machines -- the flux bounds are re-derived at every reset from a random field angle (induction_motor.py:174-185,
    # 250-285).  omega == 0 (the SC envs): psi_d_max = l_m i_sd_nominal; omega != 0: from the PREVIOUS reset's stator currents, clipped
    # (at the CC envs' +100 rad/s the clip leaves 0 -- every flux draw is 0 --, so the live branch is recorded at -100 rad/s)
    "scim_sc_uniform": ("Cont-SC-SCIM-v0", dict(random_init="uniform"), None),
    "scim_cc_uniform": ("Cont-CC-SCIM-v0", dict(random_init="uniform"), None),
    "scim_cc_negspeed_uniform": ("Cont-CC-SCIM-v0", dict(random_init="uniform"), None, dict(omega_fixed=-100.0)),
    "dfim_cc_negspeed_interval_uniform": ("Cont-CC-DFIM-v0", dict(random_init="uniform", interval=[[-5.0, 6.0], [-7.5, 7.5], [-0.8, 1.5], [-2.0, 0.4], [-1.0, 2.0]]),
                                          None, dict(omega_fixed=-60.0)),
}
INIT_CASES_R04 = ("scim_sc_uniform", "scim_cc_uniform", "scim_cc_negspeed_uniform", "dfim_cc_negspeed_interval_uniform")


def init_samples(n=4000, only=None):
    """random initialisers.  The reference's numpy streams cannot be reproduced on a device, so the fixture holds
    SAMPLES of the initial ODE state the reference draws (physical_system.reset() n times) for distributional tests.
    only: regenerate just these cases and keep the file's other entries (round 4 added the induction machines this way)."""
    path = os.path.join(OUT, "init_samples.npz")
    out = {}
    if only is not None and os.path.exists(path):
        old = np.load(path)
        out = {k: old[k] for k in old.files}
    for name, case in INIT_CASES.items():
        if only is not None and name not in only:
            continue
        env_id, mi, li = case[:3]
        kw = dict(motor=dict(motor_initializer=mi))
        if li is not None:
            kw["load"] = dict(load_initializer=li)
        if len(case) > 3:
            kw["load"] = dict(kw.get("load", {}), **case[3])
        np.random.seed(4021)  # InductionMotor._update_initial_limits draws its field angle from the GLOBAL numpy stream
        env = gem.make(env_id, **kw)
        env.reset(seed=123)
        psys = env.physical_system.unwrapped
        ys, obs = [], []
        for _ in range(n):
            o = psys.reset()
            ys.append(np.array(psys._ode_solver.y, dtype=float))
            obs.append(np.array(o, dtype=float))
        out[name + "_y"] = np.asarray(ys)
        out[name + "_obs"] = np.asarray(obs)
        out[name + "_meta"] = np.array(json.dumps(dict(describe(env), env_id=env_id, motor_initializer=mi, load_initializer=li)))
        print(f"init samples {name}: y mean {np.asarray(ys).mean(axis=0).round(3)} min {np.asarray(ys).min(axis=0).round(3)} max {np.asarray(ys).max(axis=0).round(3)}")
    np.savez_compressed(path, **out)


def wiener_samples(T=400000, n_reset=3000):
    """samples of the reference's MultipleReferenceGenerator([Wiener(i_sd), Wiener(i_sq)]) of Cont-CC-PMSM-v0 for
    distributional tests of the device-side generator: sub-episode lengths and sigmas, sigma-normalised increments, margins,
    initial values drawn by reset()."""
    env = gem.make("Cont-CC-PMSM-v0")
    (state, _), _ = env.reset(seed=4242)
    state = env.physical_system.reset()
    rg = env.reference_generator
    subs = rg._sub_generators
    out = dict(names=np.array([sg._reference_state for sg in subs]), margins=np.array([sg._limit_margin for sg in subs], dtype=float),
               sigma_range=np.array(subs[0]._sigma_range, dtype=float), episode_lengths=np.array(subs[0]._episode_len_range, dtype=float))
    rg.reset(state)
    vals = np.zeros((T, len(subs)))
    sig = np.zeros((T, len(subs)))
    lens = np.zeros((T, len(subs)), dtype=np.int64)
    for t in range(T):
        vals[t] = rg.get_reference_observation(state)
        for j, sg in enumerate(subs):
            sig[t, j] = sg._current_sigma
            lens[t, j] = sg._current_episode_length
    out.update(values=vals[:20000], sigma_trace=sig[:20000])
    for j in range(len(subs)):
        change = np.nonzero(np.diff(sig[:, j]) != 0)[0] + 1  # sub-episode starts
        out[f"sub_sigma_{j}"] = sig[change, j]
        out[f"sub_len_{j}"] = lens[change, j]
        dv = np.diff(vals[:, j])
        inside = (vals[1:, j] > out["margins"][j, 0] + 1e-12) & (vals[1:, j] < out["margins"][j, 1] - 1e-12) & (np.diff(sig[:, j]) == 0)
        out[f"z_{j}"] = (dv / sig[1:, j])[inside][:100000]
    init = np.zeros((n_reset, len(subs)))
    for i in range(n_reset):
        ref, _, _ = rg.reset(state)
        init[i] = ref[rg.referenced_states]
    out["initial_values"] = init
    np.savez_compressed(os.path.join(OUT, "wiener_samples.npz"), **out)
    print("wiener samples:", {k: np.asarray(v).shape for k, v in out.items()}, "margins", out["margins"].tolist())


def main_supply():
    """RCVoltageSupply (one Euler state fed by converter.i_sup, which makes i_sup live)."""
    K = 2000
    rc = lambda u, r=1.0, c=4e-3: ps.RCVoltageSupply(u_nominal=u, supply_parameter=dict(R=r, C=c))  # noqa: E731
    for solver in ("euler", "dopri5"):
        run_case(f"rc_permexdc_cont_free_held_{solver}", "Cont-CC-PermExDc-v0", solver, K, 1500, "held", False, "box1", supply=rc(60.0, 0.05))
        run_case(f"rc_pmsm_fin_free_held_{solver}", "Finite-CC-PMSM-v0", solver, K, 1501, "held", False, "disc8", supply=rc(420.0))
        run_case(f"rc_scim_cont_sc_free_held_{solver}", "Cont-SC-SCIM-v0", solver, K, 1502, "held", False, "box3", supply=rc(420.0, 2.0, 1e-3))
    run_case("rc_permexdc_fin_epi_held_euler", "Finite-CC-PermExDc-v0", "euler", K, 1503, "held", True, "disc4", supply=rc(60.0, 0.05))
    run_case("rc_permexdc_cont_til_free_uniform_euler", "Cont-CC-PermExDc-v0", "euler", K, 1504, "uniform", False, "box1",
             supply=rc(60.0, 0.05), converter=dict(interlocking_time=2e-6))
    run_case("rc_pmsm_fin_til_epi_uniform_tau1e-4_euler", "Finite-CC-PMSM-v0", "euler", 4000, 1505, "uniform", True, "disc8", tau=1e-4,
             supply=rc(420.0, 0.5), converter=dict(interlocking_time=1e-6))
    run_case("rc_pmsm_cont_free_uniform_euler", "Cont-CC-PMSM-v0", "euler", K, 1506, "uniform", False, "box3", supply=rc(300.0))
    run_case("rc_extex_fin_free_held_euler", "Finite-CC-ExtExDc-v0", "euler", K, 1507, "held", False, "mdisc44", supply=rc(60.0, 0.05))
    run_case("rc_eesm_cont_epi_held_euler", "Cont-CC-EESM-v0", "euler", K, 1508, "held", True, "box4", supply=rc(300.0))
    run_case("rc_dfim_fin_free_uniform_euler", "Finite-CC-DFIM-v0", "euler", K, 1509, "uniform", False, "mdisc88", supply=rc(420.0, 2.0))
    run_case("rc_series_cont_sc_free_held_euler", "Cont-SC-SeriesDc-v0", "euler", K, 1510, "held", False, "box1", supply=rc(60.0, 0.05))


def main_dfim():
    """Doubly fed induction motor: MultiConverter of two B6 bridges (stator, rotor), 24 system states."""
    K = 2000
    for solver in ("euler", "dopri5"):
        run_case(f"dfim_cont_free_held_{solver}", "Cont-CC-DFIM-v0", solver, K, 1280, "held", False, "box6")
        run_case(f"dfim_cont_sc_free_held_{solver}", "Cont-SC-DFIM-v0", solver, K, 1281, "held", False, "box6")
        run_case(f"dfim_fin_free_held_{solver}", "Finite-CC-DFIM-v0", solver, K, 1282, "held", False, "mdisc88")
    run_case("dfim_cont_free_uniform_euler", "Cont-CC-DFIM-v0", "euler", K, 1283, "uniform", False, "box6")
    run_case("dfim_cont_sc_epi_held_euler", "Cont-SC-DFIM-v0", "euler", K, 1284, "held", True, "box6")
    run_case("dfim_fin_epi_held_tau1e-4_euler", "Finite-CC-DFIM-v0", "euler", 4000, 1285, "held", True, "mdisc88", tau=1e-4)
    run_case("dfim_fin_sc_free_uniform_euler", "Finite-SC-DFIM-v0", "euler", K, 1286, "uniform", False, "mdisc88")
    b6til = lambda cls, til: [cls(interlocking_time=til), cls(interlocking_time=til)]  # noqa: E731
    run_case("dfim_fin_free_uniform_til_euler", "Finite-CC-DFIM-v0", "euler", K, 1287, "uniform", False, "mdisc88",
             converter=ps.FiniteMultiConverter(subconverters=b6til(ps.FiniteB6BridgeConverter, 1e-6)))
    run_case("dfim_fin_free_held_til_dopri5", "Finite-CC-DFIM-v0", "dopri5", K, 1288, "held", False, "mdisc88",
             converter=ps.FiniteMultiConverter(subconverters=b6til(ps.FiniteB6BridgeConverter, 1e-6)))
    run_case("dfim_cont_free_held_til_euler", "Cont-CC-DFIM-v0", "euler", K, 1289, "held", False, "box6",
             converter=ps.ContMultiConverter(subconverters=b6til(ps.ContB6BridgeConverter, 2e-6)))


if __name__ == "__main__":
    main(set(sys.argv[1:]))
