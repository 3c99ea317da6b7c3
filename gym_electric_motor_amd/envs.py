"""Batched env factory for the env ids on the accelerated path.

`make(env_id, n_envs=N, **kwargs)` mirrors `gym_electric_motor.make` (reference __init__.py:27, core.py:291-292)
for the env ids whose physical system is built from supported components, with the same per-id defaults as the
reference env classes (supply voltage, converter, motor, load, tau, constraints):

    Cont-{CC,TC,SC}-PermExDc-v0   envs/gym_dcm/permex_dc_motor_env/cont_*_permex_dc_env.py
    {Finite,Cont}-{CC,TC,SC}-PMSM-v0   envs/gym_pmsm/*.py
    {Finite,Cont}-{CC,TC,SC}-SCIM-v0   envs/gym_im/squirrel_cage_induction_motor_envs/*.py

Only the physical system + constraint monitor (done mask) are device-resident.  Reference generators, reward
functions and visualisation are outside the accelerated path: `step()` returns
`reward=None`.  For a full single-env GEM environment pass a `BatchedSCMLSystem(n_envs=1)` as
`physical_system=` to the reference's own `ElectricMotorEnvironment`.
"""
import re

from . import components as comp
from . import physical_systems as bps

_ID = re.compile(r"^(Finite|Cont)-(CC|TC|SC)-(PermExDc|PMSM|SCIM)-v0$")


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


def default_components(env_id):
    """Per-id defaults, read off the reference env classes (e.g. cont_cc_permex_dc_env.py:146-160,
    finite_cc_pmsm_env.py:148-166, cont_sc_scim_env.py:153-170)."""
    m = _ID.match(env_id)
    if not m:
        raise KeyError(f"{env_id!r} is not on the accelerated path; supported: (Finite|Cont)-(CC|TC|SC)-(PermExDc|PMSM|SCIM)-v0 "
                       "(Finite-*-PermExDc needs the Finite-4QC converter, not built yet)")
    action, control, motor = m.groups()
    finite = action == "Finite"
    speed_control = control == "SC"
    if motor == "PermExDc":
        if finite:
            raise KeyError("Finite-*-PermExDc-v0 uses FiniteFourQuadrantConverter, which is not on the accelerated path yet")
        d = dict(system=bps.BatchedDcMotorSystem, supply=dict(u_nominal=60.0), converter=comp.ContFourQuadrantConverter,
                 motor=comp.DcPermanentlyExcitedMotor, constraints=("i",))
    else:
        conv = comp.FiniteB6BridgeConverter if finite else comp.ContB6BridgeConverter
        d = dict(supply=dict(u_nominal=420.0), converter=conv, constraints=(bps.SquaredConstraint(("i_sq", "i_sd")),))
        if motor == "PMSM":
            d.update(system=bps.BatchedSynchronousMotorSystem, motor=comp.PermanentMagnetSynchronousMotor)
        else:
            d.update(system=bps.BatchedSquirrelCageInductionMotorSystem, motor=comp.SquirrelCageInductionMotor)
    if env_id == "Cont-CC-PMSM-v0":
        d["supply"] = dict(u_nominal=300.0)  # cont_cc_pmsm_env.py:154 (all other PMSM/SCIM ids: 420 V)
    if speed_control and motor == "PermExDc":
        d["load"] = (comp.PolynomialStaticLoad, dict(load_parameter=dict(a=0.0, b=0.0, c=0.0, j_load=1e-4)))  # cont_sc_permex_dc_env.py:159
    elif speed_control:
        d["load"] = (comp.PolynomialStaticLoad, dict(load_parameter=dict(a=0.01, b=0.01, c=0.0)))
    else:
        d["load"] = (comp.ConstantSpeedLoad, dict(omega_fixed=100.0))
    d["tau"] = 1e-5 if finite else 1e-4
    return d


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

    def step(self, actions):
        """-> (obs [N, S_out], reward=None, terminated [N] uint8, truncated=False, {}).  With auto_reset (default for
        n_envs > 1) an env that terminated restarts from the reset state on its next step; the state it shows
        right after that restart is `physical_system.reset_observation`."""
        obs = self.physical_system.simulate(actions)
        return obs, None, self.physical_system.done, False, {}

    def rollout(self, actions, **kw):
        return self.physical_system.rollout(actions, **kw)

    def close(self):
        self.physical_system.close()


def make(env_id, n_envs=1, device=0, supply=None, converter=None, motor=None, load=None, ode_solver=None, tau=None,
         constraints=None, dtype="float32", auto_reset=None, obs_layout="aos", **kwargs):
    """Build a batched env.  Component arguments follow the reference's env-arg convention (instance | dict | None)."""
    d = default_components(env_id)
    tau = d["tau"] if tau is None else tau
    conv_cls = d["converter"]
    system = d["system"](
        supply=_initialize(supply, comp.IdealVoltageSupply, d["supply"]),
        converter=_initialize(converter, conv_cls, dict()),
        motor=_initialize(motor, d["motor"], dict()),
        load=_initialize(load, d["load"][0], d["load"][1]),
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
