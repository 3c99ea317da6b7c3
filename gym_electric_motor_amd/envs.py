"""Batched env factory for the env ids on the accelerated path.

`make(env_id, n_envs=N, **kwargs)` mirrors `gym_electric_motor.make` (reference __init__.py:27, core.py:291-292)
for the env ids whose physical system is built from supported components, with the same per-id defaults as the
reference env classes (supply voltage, converter, motor, load, tau, constraints):

    {Finite,Cont}-{CC,TC,SC}-{PermExDc,SeriesDc,ShuntDc}-v0   envs/gym_dcm/{permex,series,shunt}_dc_motor_env/*.py
    {Finite,Cont}-{CC,TC,SC}-{PMSM,SynRM}-v0                  envs/gym_pmsm/*.py, envs/gym_synrm/*.py
    {Finite,Cont}-{CC,TC,SC}-SCIM-v0                          envs/gym_im/squirrel_cage_induction_motor_envs/*.py
    {Finite,Cont}-{CC,TC,SC}-ExtExDc-v0                       envs/gym_dcm/extex_dc_motor_env/*.py   (MultiConverter 2 x 4QC)
    {Finite,Cont}-{CC,TC,SC}-EESM-v0                          envs/gym_eesm/*.py                     (MultiConverter B6 + 4QC)
    {Finite,Cont}-{CC,TC,SC}-DFIM-v0                          envs/gym_im/doubly_fed_induction_motor_envs/*.py (MultiConverter 2 x B6)
(all 54 env ids of the reference.)

Only the physical system + constraint monitor (done mask) are device-resident.  Reference generators, reward
functions and visualisation are outside the accelerated path: `step()` returns
`reward=None`.  For a full single-env GEM environment pass a `BatchedSCMLSystem(n_envs=1)` as
`physical_system=` to the reference's own `ElectricMotorEnvironment`.
"""
import re

from . import components as comp
from . import physical_systems as bps

_ID = re.compile(r"^(Finite|Cont)-(CC|TC|SC)-(PermExDc|SeriesDc|ShuntDc|ExtExDc|PMSM|SynRM|SCIM|EESM|DFIM)-v0$")


def _initialize(arg, default_class, default_args):
    """utils.initialize (utils.py:5-16): instance | dict of overrides | None."""
    if arg is None:
        return default_class(**default_args)
    if isinstance(arg, type):
        raise Exception("Need initialization value")
    if type(arg) is str:
        raise Exception("Deprecated in version 3.0.0")
    if type(arg) is dict:
        args = dict(default_args)
        args.update(arg)
        return default_class(**args)
    return arg


# speed-control (SC) envs: PolynomialStaticLoad parameters per env class (e.g. cont_sc_permex_dc_env.py:159,
# finite_sc_permex_dc_env.py:160, cont_sc_series_dc_env.py:157, finite_sc_series_dc_env.py:157, cont_sc_shunt_dc_env.py:159,
# cont_sc_pmsm_env.py:153, cont_sc_synrm_env.py:153, cont_sc_scim_env.py:161)
_SC_LOAD = {
    ("Cont", "PermExDc"): dict(a=0.0, b=0.0, c=0.0, j_load=1e-4), ("Finite", "PermExDc"): dict(a=0.0, b=0.0, c=0.0, j_load=1e-3),
    ("Cont", "SeriesDc"): dict(a=0.01, b=0.05, c=0.0, j_load=1e-4), ("Finite", "SeriesDc"): dict(a=0.15, b=0.05, c=0.0, j_load=1e-4),
    ("Cont", "ShuntDc"): dict(a=0.05, b=0.01, c=0.0, j_load=1e-4), ("Finite", "ShuntDc"): dict(a=0.05, b=0.01, c=0.0, j_load=1e-4),
    # cont_sc_extex_dc_env.py:161, finite_sc_extex_dc_env.py:162; cont_sc_eesm_env.py:165 (-> the fall-through default below),
    # finite_sc_eesm_env.py:160 (PolynomialStaticLoad's own defaults)
    ("Cont", "ExtExDc"): dict(a=0.0, b=0.0, c=0.0, j_load=1e-4), ("Finite", "ExtExDc"): dict(a=0.0, b=0.0, c=0.0, j_load=1e-4),
    ("Finite", "EESM"): dict(),
}
# the one env class whose ConstantSpeedLoad does not turn at 100 rad/s (cont_tc_shunt_dc_env.py:155)
_OMEGA_FIXED = {"Cont-TC-ShuntDc-v0": 230.0}
# supply voltages that differ from the family default (60 V DC motors, 420 V three-phase)
_U_NOMINAL = {"Cont-CC-PMSM-v0": 300.0, "Finite-CC-SeriesDc-v0": 420.0, "Finite-TC-SeriesDc-v0": 420.0, "Cont-CC-EESM-v0": 300.0}


def default_components(env_id):
    """Per-id defaults, read off the reference env classes (e.g. cont_cc_permex_dc_env.py:146-160,
    finite_cc_pmsm_env.py:148-166, cont_sc_scim_env.py:153-170, cont_cc_series_dc_env.py:144-160,
    cont_cc_shunt_dc_env.py:145-161, cont_cc_synrm_env.py:152-160)."""
    m = _ID.match(env_id)
    if not m:
        raise KeyError(f"{env_id!r} is not on the accelerated path; supported: "
                       "(Finite|Cont)-(CC|TC|SC)-(PermExDc|SeriesDc|ShuntDc|ExtExDc|PMSM|SynRM|SCIM|EESM|DFIM)-v0")
    action, control, motor = m.groups()
    finite = action == "Finite"
    dc = motor.endswith("Dc")
    d_conv_args = dict()
    if motor == "ExtExDc":  # cont_cc_extex_dc_env.py:146-160: MultiConverter of two 4QCs
        sub = comp.FiniteFourQuadrantConverter if finite else comp.ContFourQuadrantConverter
        d = dict(system=bps.BatchedDcMotorSystem, supply=dict(u_nominal=60.0), motor=comp.DcExternallyExcitedMotor,
                 converter=comp.FiniteMultiConverter if finite else comp.ContMultiConverter, constraints=("i_a", "i_e"))
        d_conv_args = dict(subconverters=(sub, sub))
    elif motor == "EESM":  # cont_cc_eesm_env.py:153-170: B6 bridge + 4QC
        subs = (comp.FiniteB6BridgeConverter, comp.FiniteFourQuadrantConverter) if finite else (comp.ContB6BridgeConverter, comp.ContFourQuadrantConverter)
        d = dict(system=bps.BatchedExternallyExcitedSynchronousMotorSystem, supply=dict(u_nominal=420.0),
                 motor=comp.ExternallyExcitedSynchronousMotor, converter=comp.FiniteMultiConverter if finite else comp.ContMultiConverter,
                 constraints=(bps.SquaredConstraint(("i_sq", "i_sd")), bps.LimitConstraint(("i_e",))))
        d_conv_args = dict(subconverters=subs)
    elif motor == "DFIM":  # cont_cc_dfim_env.py:160-180: stator and rotor B6 bridges
        sub = comp.FiniteB6BridgeConverter if finite else comp.ContB6BridgeConverter
        d = dict(system=bps.BatchedDoublyFedInductionMotorSystem, supply=dict(u_nominal=420.0), motor=comp.DoublyFedInductionMotor,
                 converter=comp.FiniteMultiConverter if finite else comp.ContMultiConverter,
                 constraints=(bps.SquaredConstraint(("i_sq", "i_sd")),))
        d_conv_args = dict(subconverters=(sub, sub))
    elif dc:
        motor_cls = {"PermExDc": comp.DcPermanentlyExcitedMotor, "SeriesDc": comp.DcSeriesMotor, "ShuntDc": comp.DcShuntMotor}[motor]
        d = dict(system=bps.BatchedDcMotorSystem, supply=dict(u_nominal=60.0), motor=motor_cls,
                 converter=comp.FiniteFourQuadrantConverter if finite else comp.ContFourQuadrantConverter,
                 constraints=("i_a", "i_e") if motor == "ShuntDc" else ("i",))
        # NOTE: the reference's shunt envs additionally wrap the system in a CurrentSumProcessor ('i_sum' observation):
        # observation post-processing, outside the accelerated path
    else:
        motor_cls = {"PMSM": comp.PermanentMagnetSynchronousMotor, "SynRM": comp.SynchronousReluctanceMotor,
                     "SCIM": comp.SquirrelCageInductionMotor}[motor]
        system = bps.BatchedSquirrelCageInductionMotorSystem if motor == "SCIM" else bps.BatchedSynchronousMotorSystem
        d = dict(system=system, supply=dict(u_nominal=420.0), motor=motor_cls,
                 converter=comp.FiniteB6BridgeConverter if finite else comp.ContB6BridgeConverter,
                 constraints=(bps.SquaredConstraint(("i_sq", "i_sd")),))
    if env_id in _U_NOMINAL:
        d["supply"] = dict(u_nominal=_U_NOMINAL[env_id])
    if control == "SC":
        d["load"] = (comp.PolynomialStaticLoad, dict(load_parameter=_SC_LOAD.get((action, motor), dict(a=0.01, b=0.01, c=0.0))))
    else:
        d["load"] = (comp.ConstantSpeedLoad, dict(omega_fixed=_OMEGA_FIXED.get(env_id, 100.0)))
    d["tau"] = 1e-5 if finite else 1e-4
    d["converter_args"] = d_conv_args
    return d


def default_ode_solver(env_id, tau=None, load=None):
    """The solver `make(env_id)` uses when the caller names none.  The reference's default is scipy's ADAPTIVE dopri5 (rtol 1e-6,
    solvers.py:139-184); the device integrates with fixed steps, so the default is chosen per env such that the fp32 trajectories stay
    within the 1e-4 contract of the reference's default-solver runs.  Measured on the GPU over every one of the reference's 54 env ids
    exactly as `gem.make(env_id)` builds them plus ~100 further recorded dopri5 runs (tests/solver_scan.py -> profiles/r04a_solver_scan.md):

    * ConstantSpeedLoad (the CC / TC envs): one classical RK4 step per control step -- the electrical subsystem is linear there and the
      step is the exact one-step map of the scheme: <= 1.8e-5 on the envs as shipped (free runs far beyond the limits: < 1e-4).
      More sub-steps make fp32 WORSE here (rounding accumulates: 8 sub-steps reach 2.7e-3 on a free-running EESM), so none;
    * PolynomialStaticLoad (the SC envs: omega is a state, the load torque has kinks at |omega| = a tau_decay / J): RK4 with every step
      corrected for those kinks in closed form (split_kinks; the adaptive reference solver rejects and splits such steps): <= 6.3e-6 on
      every recorded run with such a load (plain RK4: up to 7.9e-5 -- Cont-SC-ShuntDc-v0 --, 6.8e-5 on the SCIM), ONE pass of the
      scheme (rounds 2-3: up to three), at plain RK4's rate wherever the launch is bandwidth bound (BASELINE config 4).

    `tau` / `load` (instance, class or class name): what the env is actually built with, when it differs from the env id's defaults."""
    d = default_components(env_id)
    if load is None:
        load = d["load"][0]
    lname = load if isinstance(load, str) else (load.__name__ if isinstance(load, type) else type(load).__name__)
    return comp.RK4Solver(split_kinks=lname != "ConstantSpeedLoad")


class BatchedElectricMotorEnv:
    """Vector-env style shell around a batched physical system (physics + done mask only)."""

    def __init__(self, physical_system):
        self.physical_system = physical_system
        self.action_space = physical_system.action_space
        self.state_space = physical_system.state_space
        self.n_envs = physical_system.n_envs

    @property
    def unwrapped(self):
        return self

    def reset(self, seed=None, options=None):
        """All envs to the initial state; returns (observations, {})."""
        return self.physical_system.reset(), {}

    def step(self, actions, references=None):
        """-> (obs [N, S_out], reward, terminated [N] uint8, truncated=False, {}).  reward: [N] device tensor when a reward function is
        installed (`physical_system.set_reward`) and `references [N, n_ref]` are passed, else None.  With auto_reset (default for
        n_envs > 1) an env that terminated restarts from the reset state on its next step; the state it shows
        right after that restart is `physical_system.reset_observation`."""
        if references is not None:
            obs = self.physical_system.simulate(actions, references=references)
            return obs, self.physical_system.reward, self.physical_system.done, False, {}
        obs = self.physical_system.simulate(actions)
        return obs, None, self.physical_system.done, False, {}

    def rollout(self, actions, **kw):
        return self.physical_system.rollout(actions, **kw)

    def rollout_synthetic(self, K, **kw):
        """K fused steps on random actions generated on the device (PhysicalSystem.rollout_synthetic)."""
        return self.physical_system.rollout_synthetic(K, **kw)

    def bind_rollout(self, actions, obs_out, done_out, stream=None):
        """-> zero-argument launch(): the pre-bound `gemx_rollout` call for fixed tensors (PhysicalSystem.bind_rollout)."""
        return self.physical_system.bind_rollout(actions, obs_out, done_out, stream=stream)

    def close(self):
        self.physical_system.close()


def make(env_id, n_envs=1, device=0, supply=None, converter=None, motor=None, load=None, ode_solver=None, tau=None,
         constraints=None, dtype="float32", auto_reset=None, obs_layout="aos", physical_system_wrappers=(), **kwargs):
    """Build a batched env.  Component arguments follow the reference's env-arg convention (instance | dict | None).
    physical_system_wrappers: reference-style tuple (innermost first) of DeadTimeProcessor / DqToAbcActionProcessor holders
    (or the reference's own instances); they are folded into the kernel's action stage."""
    from .physical_system_wrappers import fold_wrappers

    if physical_system_wrappers:
        kwargs = dict(kwargs, **fold_wrappers(physical_system_wrappers))
    d = default_components(env_id)
    tau = d["tau"] if tau is None else tau
    conv_cls = d["converter"]
    load = _initialize(load, d["load"][0], d["load"][1])
    if ode_solver is None:
        ode_solver = default_ode_solver(env_id, tau=tau, load=load)
    system = d["system"](
        supply=_initialize(supply, comp.IdealVoltageSupply, d["supply"]),
        converter=_initialize(converter, conv_cls, d["converter_args"]),
        motor=_initialize(motor, d["motor"], dict()),
        load=load,
        ode_solver=_initialize(ode_solver, comp.RK4Solver, dict()),
        tau=tau,
        n_envs=n_envs,
        device=device,
        dtype=dtype,
        constraints=d["constraints"] if constraints is None else constraints,
        auto_reset=auto_reset,
        obs_layout=obs_layout,
        **kwargs,
    )
    return BatchedElectricMotorEnv(system)
