"""Parameter-holder mirrors of the reference components that sit on the accelerated path.

These classes keep the reference's names, constructor arguments, default parameters and the derivation of
limits / nominal values / model constants, so that `BatchedSCMLSystem(converter=..., motor=..., load=...,
supply=..., ode_solver=...)` reads like `SCMLSystem(...)` (reference physical_systems/physical_systems.py:54).
They contain NO physics: the right-hand sides, converters, solvers and constraints run only in the HIP kernels
(csrc/gemx.hip).  Instances of the reference's own component classes are accepted as well -- the batched system
reads the same attributes from either (`motor_parameter`, `limits`, `nominal_values`, `_model_constants`,
`j_total`, `load_parameter`, `omega_fixed`, `_interlocking_time`, `u_nominal`, `_nsteps`).

Citations are relative to /root/reference/src/gym_electric_motor/physical_systems/.
"""
import math

import numpy as np

from .spaces import Box, Discrete, MultiDiscrete


def update_parameter_dict(source, update):
    """utils.py:73-94 -- unknown keys raise KeyError."""
    for key in update:
        if key not in source:
            raise KeyError(f'Cannot update_dict the source_dict. The key "{key}" is not available.')
    out = dict(source)
    out.update(update)
    return out


# ------------------------------------------------------------------------------------------------- supply
class IdealVoltageSupply:
    """voltage_supplies.py:60-72: constant u_nominal."""

    voltage_len = 1

    def __init__(self, u_nominal=600.0):
        self._u_nominal = float(u_nominal)
        self.supply_range = (u_nominal, u_nominal)

    @property
    def u_nominal(self):
        return self._u_nominal


class RCVoltageSupply(IdealVoltageSupply):
    """voltage_supplies.py:75-123: a capacitor C fed through R from an ideal source u_nominal; the kernels advance its voltage by
    one explicit Euler step per control step with the current the converter draws (converter.i_sup)."""

    def __init__(self, u_nominal=600.0, supply_parameter=None):
        super().__init__(u_nominal)
        supply_parameter = supply_parameter or {"R": 1, "C": 4e-3}
        assert "R" in supply_parameter.keys(), "Pass key 'R' for Resistance in your dict"
        assert "C" in supply_parameter.keys(), "Pass key 'C' for Capacitance in your dict"
        self.supply_range = (0, u_nominal)
        self._r = supply_parameter["R"]
        self._c = supply_parameter["C"]


# ------------------------------------------------------------------------------------------------- converters
class _Converter:
    """converters.py:5-111 (tau, interlocking_time)."""

    voltages = currents = action_space = None

    def __init__(self, tau, interlocking_time=0.0):
        self._tau = float(tau)
        self._interlocking_time = float(interlocking_time)

    @property
    def tau(self):
        return self._tau

    @tau.setter
    def tau(self, value):
        self._tau = float(value)


class ContFourQuadrantConverter(_Converter):
    """Key 'Cont-4QC', converters.py:438-495."""

    voltages = Box(-1, 1, shape=(1,), dtype=np.float64)
    currents = Box(-1, 1, shape=(1,), dtype=np.float64)
    action_space = Box(-1, 1, shape=(1,), dtype=np.float64)

    def __init__(self, tau=1e-4, interlocking_time=0.0):
        super().__init__(tau, interlocking_time)


class ContB6BridgeConverter(_Converter):
    """Key 'Cont-B6C', converters.py:842-911."""

    voltages = Box(-1, 1, shape=(3,), dtype=np.float64)
    currents = Box(-1, 1, shape=(3,), dtype=np.float64)
    action_space = Box(-1, 1, shape=(3,), dtype=np.float64)

    def __init__(self, tau=1e-4, interlocking_time=0.0):
        super().__init__(tau, interlocking_time)


class FiniteB6BridgeConverter(_Converter):
    """Key 'Finite-B6C', converters.py:743-839."""

    voltages = Box(-1, 1, shape=(3,), dtype=np.float64)
    currents = Box(-1, 1, shape=(3,), dtype=np.float64)
    action_space = Discrete(8)

    def __init__(self, tau=1e-5, interlocking_time=0.0):
        super().__init__(tau, interlocking_time)


class FiniteFourQuadrantConverter(_Converter):
    """Key 'Finite-4QC', converters.py:313-368 (actions 0..3: T2T4 / T1T4 / T2T3 / T1T3)."""

    voltages = Box(-1, 1, shape=(1,), dtype=np.float64)
    currents = Box(-1, 1, shape=(1,), dtype=np.float64)
    action_space = Discrete(4)

    def __init__(self, tau=1e-5, interlocking_time=0.0):
        super().__init__(tau, interlocking_time)


class _MultiConverter(_Converter):
    """Cont/FiniteMultiConverter, converters.py:498-740: a list of sub-converters whose actions, currents and voltages
    are concatenated.  As in the reference, sub-converters given as classes are instantiated with the holder's kwargs;
    the holder's own `interlocking_time` is never used -- the sub-converters' values are."""

    def __init__(self, subconverters, tau, **kwargs):
        super().__init__(tau, kwargs.get("interlocking_time", 0.0))
        self._sub_converters = []
        for sc in subconverters:
            assert not isinstance(sc, str)
            if isinstance(sc, type):
                sc = sc(**kwargs)
            self._sub_converters.append(sc)
        self.subsignal_current_space_dims = np.array([int(np.squeeze(sc.currents.shape) or 1) for sc in self._sub_converters])
        self.subsignal_voltage_space_dims = np.array([int(np.squeeze(sc.voltages.shape) or 1) for sc in self._sub_converters])
        self.currents = Box(np.concatenate([sc.currents.low for sc in self._sub_converters]),
                            np.concatenate([sc.currents.high for sc in self._sub_converters]), dtype=np.float64)
        self.voltages = Box(np.concatenate([sc.voltages.low for sc in self._sub_converters]),
                            np.concatenate([sc.voltages.high for sc in self._sub_converters]), dtype=np.float64)
        self.tau = tau

    @property
    def sub_converters(self):
        return self._sub_converters

    @property
    def tau(self):
        return self._tau

    @tau.setter
    def tau(self, value):  # converters.py:504-508 / 608-612
        self._tau = float(value)
        for sc in getattr(self, "_sub_converters", ()):
            sc.tau = value


class ContMultiConverter(_MultiConverter):
    """Key 'Cont-Multi', converters.py:598-740."""

    def __init__(self, subconverters, tau=1e-4, **kwargs):
        super().__init__(subconverters, tau, **kwargs)
        self.action_space = Box(np.concatenate([sc.action_space.low for sc in self._sub_converters]),
                                np.concatenate([sc.action_space.high for sc in self._sub_converters]), dtype=np.float64)


class FiniteMultiConverter(_MultiConverter):
    """Key 'Finite-Multi', converters.py:498-595 (action space MultiDiscrete of the sub-converters' action counts)."""

    def __init__(self, subconverters, tau=1e-5, **kwargs):
        super().__init__(subconverters, tau, **kwargs)
        self.action_space = MultiDiscrete([sc.action_space.n for sc in self._sub_converters])


# ------------------------------------------------------------------------------------------------- solvers
class EulerSolver:
    """solvers.py:79-136."""

    def __init__(self, nsteps=1):
        self._nsteps = int(nsteps)


class RK4Solver:
    """Classical 4th-order Runge-Kutta with `nsteps` sub-steps per integration segment.  The reference has no
    RK4; its like-for-like CPU counterpart is the default scipy dopri5 path.
    split_kinks: correct every step for the kinks of a PolynomialStaticLoad's torque (|omega| = a tau_decay / J), where the reference's
    adaptive default solver rejects and splits its steps (include/gemx.h: GEMX_SOLVER_SPLIT_KINKS).  Since round 4 that is ONE pass of the
    scheme on a smooth extension of the load torque plus the defect integrated in closed form along the step's own omega path (rounds
    2-3 cut the step at the predicted crossings: up to three passes); the name of the option is kept."""

    def __init__(self, nsteps=1, split_kinks=False):
        self._nsteps = int(nsteps)
        self._split_kinks = bool(split_kinks)


class DormandPrince5Solver:
    """One fixed Dormand-Prince 5th-order step per sub-step: what the reference's default
    scipy.integrate.ode('dopri5') (solvers.py:139-184) computes whenever its trial step is accepted.  split_kinks: see RK4Solver."""

    def __init__(self, nsteps=1, split_kinks=False):
        self._nsteps = int(nsteps)
        self._split_kinks = bool(split_kinks)


class ScipyOdeSolver(DormandPrince5Solver):
    """The device's counterpart of the reference's DEFAULT solver, ScipyOdeSolver('dopri5') (solvers.py:139-184: scipy's adaptive DOPRI5,
    rtol 1e-6, atol 1e-12): error-controlled Dormand-Prince 5(4) -- every integration segment is tried as one step and cut, lane by
    lane, where the embedded error estimate exceeds the tolerance in scipy's norm (include/gemx.h: GEMX_SOLVER_ADAPTIVE).  Same
    tolerance semantics, not scipy's step sequence.  Keyword arguments as `scipy.integrate.ode.set_integrator('dopri5', ...)` takes
    them: `rtol`, `atol` (absolute, in state units; default 1e-9 -- scipy's 1e-12 is below fp32 resolution and changes nothing); others
    (`nsteps`, `first_step`, `safety`, ...) are accepted and ignored.  A step at the floor of 1/1024 of a segment that still misses the
    tolerance raises a warning through `check_errors()`.
    split_kinks (default True; round 6): behind a PolynomialStaticLoad every attempt integrates the smooth model system of the kink
    correction (RK4Solver.split_kinks) and the kink's defect is added in closed form -- the error estimate then never sees the kink of the
    load torque at |omega| = a tau_decay / J, whose crossing costs the plain controller five to eight attempts, and the 64 lanes of a wave
    wait for the one that crosses: BASELINE config 4 under random actions, 4.7 -> 1.8 attempts per control step and wave, same accuracy
    against the reference's dopri5 runs (profiles/r06_wave_step_statistics.md).  False: the plain controller of rounds 4-5."""

    def __init__(self, integrator="dopri5", rtol=1e-6, atol=1e-9, split_kinks=True, atol_omega=None, **kwargs):
        """atol_omega: absolute tolerance of omega in rad/s; None = atol x the speed limit, i.e. `atol` in NORMALISED units (1e-9 of the
        speed range instead of 1e-9 rad/s: a speed-control episode starts at omega = 0, where a physical-unit atol makes lanes cut steps
        that no observation can show, and the 64 lanes of a wave wait for them -- include/gemx.h: solver_atol_omega)."""
        if integrator != "dopri5":
            raise ValueError(f"integrator {integrator!r}: the accelerated path restates 'dopri5' (the reference's default) only")
        super().__init__(nsteps=1, split_kinks=split_kinks)
        self._adaptive = True
        self._rtol, self._atol = float(rtol), float(atol)
        self._atol_omega = 0.0 if atol_omega is None else float(atol_omega)
        self._ignored = dict(kwargs)


# ------------------------------------------------------------------------------------------------- motors
class _ElectricMotor:
    """electric_motors/electric_motor.py:9-325 (parameter / limit / nominal bookkeeping only)."""

    CURRENTS = []
    VOLTAGES = []
    _default_motor_parameter = {}
    _default_nominal_values = {}
    _default_limits = {}
    _default_initializer = {"states": {}, "interval": None, "random_init": None, "random_params": None}

    def __init__(self, motor_parameter=None, nominal_values=None, limit_values=None, motor_initializer=None):
        self._motor_parameter = update_parameter_dict(self._default_motor_parameter, motor_parameter or {})
        self._limits = update_parameter_dict(self._default_limits, limit_values or {})
        self._nominal_values = update_parameter_dict(self._default_nominal_values, nominal_values or {})
        self._initializer = update_parameter_dict(self._default_initializer, motor_initializer or {})
        if self._initializer.get("random_init") not in (None, "uniform", "normal", "gaussian"):
            raise NotImplementedError(f"random_init {self._initializer.get('random_init')!r} (electric_motor.py:229-257 knows uniform / gaussian)")
        self._initial_states = dict(self._initializer["states"] or {})

    motor_parameter = property(lambda self: self._motor_parameter)
    limits = property(lambda self: self._limits)
    nominal_values = property(lambda self: self._nominal_values)
    initializer = property(lambda self: self._initializer)

    def _update_limits(self, limits_d=None, nominal_d=None):
        """Completes `limits` / `nominal_values` the way the reference does after a motor computed its derived quantities
        (behaviour of electric_motor.py:296-317): a quantity the user left unset (absent or 0) takes the derived limit -- omega always
        falls back to the class default -- and an unset nominal value takes the derived nominal value, else the (final) limit."""
        derived = {**(limits_d or {}), "omega": self._default_limits["omega"]}
        self._limits.update({q: v for q, v in derived.items() if not self._limits.get(q)})
        fallback = nominal_d or {}
        self._nominal_values.update({q: fallback.get(q, lim) for q, lim in self._limits.items() if not self._nominal_values.get(q)})


class DcPermanentlyExcitedMotor(_ElectricMotor):
    """electric_motors/dc_permanently_excited_motor.py:6-120 (PMG-132 defaults, lines 49-63)."""

    CURRENTS = ["i"]
    VOLTAGES = ["u"]
    _default_motor_parameter = {"r_a": 16e-3, "l_a": 19e-6, "psi_e": 0.165, "j_rotor": 0.025}
    _default_nominal_values = dict(omega=300, torque=16.0, i=97, u=60)
    _default_limits = dict(omega=400, torque=38.0, i=210, u=60)
    _default_initializer = {"states": {"i": 0.0}, "interval": None, "random_init": None, "random_params": (None, None)}

    def __init__(self, motor_parameter=None, nominal_values=None, limit_values=None, motor_initializer=None):
        super().__init__(motor_parameter, nominal_values, limit_values, motor_initializer)
        mp = self._motor_parameter
        # _update_model, lines 71-75
        self._model_constants = np.array([[-mp["psi_e"], -mp["r_a"], 1.0]]) / mp["l_a"]
        # _update_limits, lines 95-105 + DcMotor._update_limits (dc_motor.py:153-160)
        r_a = 1 if mp["r_a"] == 0 else mp["r_a"]
        agenda = {"u": self._default_limits["u"], "i": self._limits["u"] / r_a}
        agenda["torque"] = mp["psi_e"] * self._limits["i"]
        self._update_limits(agenda)

    def torque_coefficients(self):
        return [self._motor_parameter["psi_e"], 0.0]

    def initial_motor_state(self):
        return [float(self._initial_states.get("i", 0.0))]

    def get_state_space(self, input_currents, input_voltages):
        """dc_permanently_excited_motor.py:107-120."""
        lc, lv = input_currents.low[0] == -1, input_voltages.low[0] == -1
        low = {"omega": -1 if lv else 0, "torque": -1 if lc else 0, "i": -1 if lc else 0, "u": -1 if lv else 0}
        return low, {"omega": 1, "torque": 1, "i": 1, "u": 1}


class DcSeriesMotor(_ElectricMotor):
    """electric_motors/dc_series_motor.py:6-140: one circuit, di/dt = (-(r_a + r_e) i - l_e' omega i + u) / (l_a + l_e)."""

    CURRENTS = ["i"]
    VOLTAGES = ["u"]
    _default_motor_parameter = {"r_a": 16e-3, "r_e": 48e-3, "l_a": 19e-6, "l_e_prime": 1.7e-3, "l_e": 5.4e-3, "j_rotor": 0.0025}
    _default_nominal_values = dict(omega=300, torque=16.0, i=97, i_a=97, i_e=97, u=60, u_a=60, u_e=60)
    _default_limits = dict(omega=400, torque=38.0, i=210, i_a=210, i_e=210, u=60, u_a=60, u_e=60)
    _default_initializer = {"states": {"i": 0.0}, "interval": None, "random_init": None, "random_params": (None, None)}

    def __init__(self, motor_parameter=None, nominal_values=None, limit_values=None, motor_initializer=None):
        super().__init__(motor_parameter, nominal_values, limit_values, motor_initializer)
        mp = self._motor_parameter
        # _update_model, lines 68-72: features [i, omega * i, u]
        self._model_constants = np.array([[-mp["r_a"] - mp["r_e"], -mp["l_e_prime"], 1.0]]) / (mp["l_a"] + mp["l_e"])
        # _update_limits, lines 89-99 + DcMotor._update_limits (dc_motor.py:153-160)
        r_a = 1 if mp["r_a"] == 0 else mp["r_a"]
        agenda = {"u": self._default_limits["u"], "i": self._limits["u"] / (r_a + mp["r_e"])}
        agenda["torque"] = mp["l_e_prime"] * self._limits["i"] * self._limits["i"]
        self._update_limits(agenda)

    def torque_coefficients(self):
        return [self._motor_parameter["l_e_prime"], 0.0]

    def initial_motor_state(self):
        return [float(self._initial_states.get("i", 0.0))]

    def get_state_space(self, input_currents, input_voltages):
        """dc_series_motor.py:101-116."""
        low = {"omega": 0, "torque": 0, "i": -1 if input_currents.low[0] == -1 else 0, "u": -1 if input_voltages.low[0] == -1 else 0}
        return low, {"omega": 1, "torque": 1, "i": 1, "u": 1}


class DcShuntMotor(_ElectricMotor):
    """electric_motors/dc_shunt_motor.py:6-150 over dc_motor.py: armature and exciting circuit fed by the SAME voltage."""

    CURRENTS = ["i_a", "i_e"]
    VOLTAGES = ["u"]
    _default_motor_parameter = {"r_a": 16e-3, "r_e": 4e-1, "l_a": 19e-6, "l_e_prime": 1.7e-3, "l_e": 5.4e-3, "j_rotor": 0.0025}
    _default_nominal_values = dict(omega=300, torque=16.0, i=97, i_a=97, i_e=97, u=60, u_a=60, u_e=60)
    _default_limits = dict(omega=400, torque=38.0, i=210, i_a=210, i_e=210, u=60, u_a=60, u_e=60)
    _default_initializer = {"states": {"i_a": 0.0, "i_e": 0.0}, "interval": None, "random_init": None, "random_params": (None, None)}

    def __init__(self, motor_parameter=None, nominal_values=None, limit_values=None, motor_initializer=None):
        super().__init__(motor_parameter, nominal_values, limit_values, motor_initializer)
        mp = self._motor_parameter
        # DcMotor._update_model, dc_motor.py:96-104: features [i_a, i_e, omega * i_e, u_a, u_e]
        self._model_constants = np.array([[-mp["r_a"], 0, -mp["l_e_prime"], 1, 0], [0, -mp["r_e"], 0, 0, 1]], dtype=float)
        self._model_constants[0] = self._model_constants[0] / mp["l_a"]
        self._model_constants[1] = self._model_constants[1] / mp["l_e"]
        # _update_limits, dc_shunt_motor.py:137-150 + dc_motor.py:153-160
        r_a = 1 if mp["r_a"] == 0 else mp["r_a"]
        agenda = {"u": self._default_limits["u"], "i_a": self._limits.get("i", None) or self._limits["u"] / r_a,
                  "i_e": self._limits.get("i", None) or self._limits["u"] / mp["r_e"]}
        agenda["torque"] = mp["l_e_prime"] * self._limits["i_a"] * self._limits["i_e"]
        self._update_limits(agenda)

    def torque_coefficients(self):
        return [self._motor_parameter["l_e_prime"], 0.0]

    def initial_motor_state(self):
        vals = [float(v) for v in self._initial_states.values()]
        return vals if len(vals) == 2 else [0.0, 0.0]

    def get_state_space(self, input_currents, input_voltages):
        """dc_shunt_motor.py:104-135."""
        lc = input_currents.low[0] == -1
        low = {"omega": 0, "torque": -1 if lc else 0, "i_a": -1 if lc else 0, "i_e": -1 if lc else 0,
               "u": -1 if input_voltages.low[0] == -1 else 0}
        return low, {"omega": 1, "torque": 1, "i_a": 1, "i_e": 1, "u": 1}


class DcExternallyExcitedMotor(_ElectricMotor):
    """electric_motors/dc_externally_excited_motor.py:6-120 over dc_motor.py: armature and exciting circuit fed by two
    separate converters (u_a, u_e)."""

    CURRENTS = ["i_a", "i_e"]
    VOLTAGES = ["u_a", "u_e"]
    _default_motor_parameter = {"r_a": 16e-3, "r_e": 16e-2, "l_a": 19e-6, "l_e_prime": 1.7e-3, "l_e": 5.4e-3, "j_rotor": 0.0025}
    _default_nominal_values = dict(omega=300, torque=16.0, i=97, i_a=97, i_e=97, u=60, u_a=60, u_e=60)
    _default_limits = dict(omega=400, torque=38.0, i=210, i_a=210, i_e=210, u=60, u_a=60, u_e=60)
    _default_initializer = {"states": {"i_a": 0.0, "i_e": 0.0}, "interval": None, "random_init": None, "random_params": (None, None)}

    def __init__(self, motor_parameter=None, nominal_values=None, limit_values=None, motor_initializer=None):
        super().__init__(motor_parameter, nominal_values, limit_values, motor_initializer)
        mp = self._motor_parameter
        # DcMotor._update_model, dc_motor.py:96-104: features [i_a, i_e, omega * i_e, u_a, u_e]
        self._model_constants = np.array([[-mp["r_a"], 0, -mp["l_e_prime"], 1, 0], [0, -mp["r_e"], 0, 0, 1]], dtype=float)
        self._model_constants[0] = self._model_constants[0] / mp["l_a"]
        self._model_constants[1] = self._model_constants[1] / mp["l_e"]
        # _update_limits, dc_externally_excited_motor.py:107-120 + dc_motor.py:153-160
        r_a = 1 if mp["r_a"] == 0 else mp["r_a"]
        agenda = {"u_a": self._default_limits["u"], "u_e": self._default_limits["u"],
                  "i_a": self._limits.get("i", None) or self._limits["u"] / r_a,
                  "i_e": self._limits.get("i", None) or self._limits["u"] / mp["r_e"]}
        agenda["torque"] = mp["l_e_prime"] * self._limits["i_a"] * self._limits["i_e"]
        self._update_limits(agenda)

    def torque_coefficients(self):
        return [self._motor_parameter["l_e_prime"], 0.0]

    def initial_motor_state(self):
        vals = [float(v) for v in self._initial_states.values()]
        return vals if len(vals) == 2 else [0.0, 0.0]

    def get_state_space(self, input_currents, input_voltages):
        """dc_motor.py:129-151."""
        ca, ce = input_currents.low[0] == -1, input_currents.low[1] == -1
        va, ve = input_voltages.low[0] == -1, input_voltages.low[1] == -1
        low = {"omega": -1 if va or ve else 0, "torque": -1 if ca or ce else 0, "i_a": -1 if ca else 0, "i_e": -1 if ce else 0,
               "u_a": -1 if va else 0, "u_e": -1 if ve else 0}
        return low, {"omega": 1, "torque": 1, "i_a": 1, "i_e": 1, "u_a": 1, "u_e": 1}


class _ThreePhaseMotor(_ElectricMotor):
    IO_VOLTAGES = []
    IO_CURRENTS = []

    def _three_phase_limits(self):
        """Phase-quantity limits of a three-phase machine behind a B6 bridge (behaviour of synchronous_motor.py:173-189 /
        squirrel_cage_induction_motor.py:131-144 + three_phase_motor.py:125-131): every phase / dq voltage is bounded by half the
        DC-link voltage, every current by the user's 'i' or, failing that, by voltage over stator resistance; then the torque limit."""
        r_s = self._motor_parameter["r_s"]

        def derive(table):
            u_half = 0.5 * table["u"]
            out = {u: u_half for u in self.IO_VOLTAGES}
            # (the current bound reads the table's own voltage entry -- set by the user or not -- exactly as the reference does)
            out.update({i: table.get("i") or table[u] / r_s for u, i in zip(self.IO_VOLTAGES, self.IO_CURRENTS)})
            return out

        self._update_limits(derive(self._limits), derive(self._nominal_values))
        self._update_limits({"torque": self._torque_limit()})


def _max_torque_d_current(flux, delta_l, i_max, root):
    """d-axis current of the stationary torque point on the current circle |i| = i_max for T ~ (flux + delta_l * i_d) * i_q:
    2 delta_l i_d^2 + flux i_d - delta_l i_max^2 = 0  ->  i_d = (-flux + root * sqrt(flux^2 + 8 delta_l^2 i_max^2)) / (4 delta_l)."""
    return (-flux + root * math.sqrt(flux * flux + 8.0 * (delta_l * i_max) ** 2)) / (4.0 * delta_l)


class PermanentMagnetSynchronousMotor(_ThreePhaseMotor):
    """electric_motors/permanent_magnet_synchronous_motor.py:8-173 (defaults lines 86-103)."""

    CURRENTS = ["i_sd", "i_sq"]
    VOLTAGES = ["u_sd", "u_sq"]
    IO_VOLTAGES = ["u_a", "u_b", "u_c", "u_sd", "u_sq"]
    IO_CURRENTS = ["i_a", "i_b", "i_c", "i_sd", "i_sq"]
    _default_motor_parameter = {"p": 3, "l_d": 0.37e-3, "l_q": 1.2e-3, "j_rotor": 0.03883, "r_s": 18e-3, "psi_p": 66e-3}
    _default_limits = dict(omega=4e3 * np.pi / 30, torque=0.0, i=400, epsilon=math.pi, u=300)
    _default_nominal_values = dict(omega=3e3 * np.pi / 30, torque=0.0, i=240, epsilon=math.pi, u=300)
    # NOTE the key order i_sq, i_sd, epsilon of the reference (line 98): reset() returns the VALUES in dict
    # order into the ODE slots [i_sd, i_sq, epsilon], i.e. user-supplied i_sd / i_sq are swapped.  Reproduced.
    _default_initializer = {"states": {"i_sq": 0.0, "i_sd": 0.0, "epsilon": 0.0}, "interval": None,
                            "random_init": None, "random_params": (None, None)}

    def __init__(self, motor_parameter=None, nominal_values=None, limit_values=None, motor_initializer=None):
        super().__init__(motor_parameter, nominal_values, limit_values, motor_initializer)
        mp = self._motor_parameter
        # _update_model, lines 107-119
        m = np.array([
            [0, -mp["r_s"], 0, 1, 0, 0, mp["l_q"] * mp["p"]],
            [-mp["psi_p"] * mp["p"], 0, -mp["r_s"], 0, 1, -mp["l_d"] * mp["p"], 0],
            [mp["p"], 0, 0, 0, 0, 0, 0],
        ], dtype=float)
        m[0] = m[0] / mp["l_d"]
        m[1] = m[1] / mp["l_q"]
        self._model_constants = m
        self._three_phase_limits()

    def torque(self, currents):
        mp = self._motor_parameter
        return 1.5 * mp["p"] * (mp["psi_p"] + (mp["l_d"] - mp["l_q"]) * currents[0]) * currents[1]

    def _torque_limit(self):
        """Torque at the nominal-current operating point with the most torque (behaviour of permanent_magnet_synchronous_motor.py:
        121-132).  Non-salient machine: all current in q at its LIMIT.  Salient: the stationary point of T(i_d) on the nominal-current
        circle; the reference takes the root on the negative-d side for l_d < l_q and keeps the same expression for l_d > l_q."""
        mp = self._motor_parameter
        delta_l = mp["l_d"] - mp["l_q"]
        if delta_l == 0:
            return self.torque([0, self._limits["i_sq"], 0])
        i_n = self._nominal_values["i"]
        i_d = _max_torque_d_current(mp["psi_p"], delta_l, i_n, root=1.0 if delta_l < 0 else -1.0)
        return self.torque([i_d, math.sqrt(i_n * i_n - i_d * i_d), 0])

    def torque_coefficients(self):
        mp = self._motor_parameter
        return [1.5 * mp["p"] * mp["psi_p"], 1.5 * mp["p"] * (mp["l_d"] - mp["l_q"])]

    def initial_motor_state(self):
        # synchronous_motor.py:125-131: np.asarray(list(self._initial_states.values())) -> dict ORDER, not names
        vals = [float(v) for v in self._initial_states.values()]
        return vals if len(vals) == 3 else [0.0, 0.0, 0.0]


class ExternallyExcitedSynchronousMotor(_ThreePhaseMotor):
    """electric_motors/externally_excited_synchronous_motor.py:7-180 (defaults lines 27-50, DOI 10.1109/ICELMACH.2014.6960287)."""

    CURRENTS = ["i_sd", "i_sq", "i_e"]
    VOLTAGES = ["u_sd", "u_sq", "u_e"]
    IO_VOLTAGES = ["u_a", "u_b", "u_c", "u_sd", "u_sq", "u_e"]
    IO_CURRENTS = ["i_a", "i_b", "i_c", "i_sd", "i_sq", "i_e"]
    _default_motor_parameter = {"p": 3, "l_d": 1.66e-3, "l_q": 0.35e-3, "l_m": 1.589e-3, "l_e": 1.74e-3, "j_rotor": 0.3883,
                                "r_s": 15.55e-3, "r_e": 7.2e-3, "k": 65.21}
    _default_limits = dict(omega=12e3 * np.pi / 30, torque=0.0, i=150, i_e=150, epsilon=math.pi, u=320)
    _default_nominal_values = dict(omega=4.3e3 * np.pi / 30, torque=0.0, i=120, i_e=150, epsilon=math.pi, u=320)
    # dict order i_sq, i_sd, i_e, epsilon as in the reference (line 45): reset() fills the ODE slots
    # [i_sd, i_sq, i_e, epsilon] with the VALUES in dict order (synchronous_motor.py:125-131)
    _default_initializer = {"states": {"i_sq": 0.0, "i_sd": 0.0, "i_e": 0.0, "epsilon": 0.0}, "interval": None,
                            "random_init": None, "random_params": (None, None)}

    def __init__(self, motor_parameter=None, nominal_values=None, limit_values=None, motor_initializer=None):
        super().__init__(motor_parameter, nominal_values, limit_values, motor_initializer)
        mp = self._motor_parameter
        # _update_model, lines 69-93: rotor quantities referred to the stator side (upper-case index)
        mp["r_E"] = mp["k"] ** 2 * 3 / 2 * mp["r_e"]
        mp["l_M"] = mp["k"] * 3 / 2 * mp["l_m"]
        mp["l_E"] = mp["k"] ** 2 * 3 / 2 * mp["l_e"]
        mp["i_k_rs"] = 2 / 3 / mp["k"]
        mp["sigma"] = 1 - mp["l_M"] ** 2 / (mp["l_d"] * mp["l_E"])
        sg, ik = mp["sigma"], mp["i_k_rs"]
        # features [omega, i_d, i_q, i_e, u_d, u_q, u_e, omega*i_d, omega*i_q, omega*i_e]
        m = np.array([
            [0, -mp["r_s"] / sg, 0, mp["l_M"] * mp["r_E"] / (sg * mp["l_E"]) * ik, 1 / sg, 0, -mp["l_M"] * mp["k"] / (sg * mp["l_E"]), 0,
             mp["l_q"] * mp["p"] / sg, 0],
            [0, 0, -mp["r_s"], 0, 0, 1, 0, -mp["l_d"] * mp["p"], 0, -mp["p"] * mp["l_M"] * ik],
            [0, mp["l_M"] * mp["r_s"] / (sg * mp["l_d"]), 0, -mp["r_E"] / sg * ik, -mp["l_M"] / (sg * mp["l_d"]), 0, mp["k"] / sg, 0,
             -mp["p"] * mp["l_M"] * mp["l_q"] / (sg * mp["l_d"]), 0],
            [mp["p"], 0, 0, 0, 0, 0, 0, 0, 0, 0],
        ], dtype=float)
        m[0] = m[0] / mp["l_d"]
        m[1] = m[1] / mp["l_q"]
        m[2] = m[2] / mp["l_E"] / ik
        self._model_constants = m
        self._three_phase_limits()

    def torque(self, currents):
        """lines 133-136; currents = [i_sd, i_sq, i_e, ...]."""
        mp = self._motor_parameter
        return 1.5 * mp["p"] * (mp["l_M"] * currents[2] * mp["i_k_rs"] + (mp["l_d"] - mp["l_q"]) * currents[0]) * currents[1]

    def _torque_limit(self):
        """As for the PMSM with the excitation at its limit: flux linkage l_M * i_n, always the physically meaningful root
        (behaviour of externally_excited_synchronous_motor.py:115-131)."""
        mp = self._motor_parameter
        delta_l = mp["l_d"] - mp["l_q"]
        i_e = self._limits["i_e"]
        if delta_l == 0:
            return self.torque([0, self._limits["i_sq"], i_e, 0])
        i_n = self._nominal_values["i"]
        i_d = _max_torque_d_current(mp["l_M"] * i_n, delta_l, i_n, root=1.0)
        return self.torque([i_d, math.sqrt(i_n * i_n - i_d * i_d), i_e, 0])

    def torque_coefficients(self):
        mp = self._motor_parameter
        return [1.5 * mp["p"] * mp["l_M"] * mp["i_k_rs"], 1.5 * mp["p"] * (mp["l_d"] - mp["l_q"])]

    def initial_motor_state(self):
        vals = [float(v) for v in self._initial_states.values()]
        return vals if len(vals) == 4 else [0.0] * 4


class SynchronousReluctanceMotor(_ThreePhaseMotor):
    """electric_motors/synchronous_reluctance_motor.py:8-190 (defaults lines 85-113): a synchronous motor without magnets."""

    CURRENTS = ["i_sd", "i_sq"]
    VOLTAGES = ["u_sd", "u_sq"]
    IO_VOLTAGES = ["u_a", "u_b", "u_c", "u_sd", "u_sq"]
    IO_CURRENTS = ["i_a", "i_b", "i_c", "i_sd", "i_sq"]
    _default_motor_parameter = {"p": 4, "l_d": 10.1e-3, "l_q": 4.1e-3, "j_rotor": 0.8e-3, "r_s": 0.57}
    _default_nominal_values = {"i": 10, "torque": 0, "omega": 3e3 * np.pi / 30, "epsilon": np.pi, "u": 80}
    _default_limits = {"i": 18, "torque": 0, "omega": 4.3e3 * np.pi / 30, "epsilon": np.pi, "u": 80}
    _default_initializer = {"states": {"i_sq": 0.0, "i_sd": 0.0, "epsilon": 0.0}, "interval": None,
                            "random_init": None, "random_params": (None, None)}

    def __init__(self, motor_parameter=None, nominal_values=None, limit_values=None, motor_initializer=None):
        super().__init__(motor_parameter, nominal_values, limit_values, motor_initializer)
        mp = self._motor_parameter
        # _update_model, lines 117-129
        m = np.array([
            [0, -mp["r_s"], 0, 1, 0, 0, mp["l_q"] * mp["p"]],
            [0, 0, -mp["r_s"], 0, 1, -mp["l_d"] * mp["p"], 0],
            [mp["p"], 0, 0, 0, 0, 0, 0],
        ], dtype=float)
        m[0] = m[0] / mp["l_d"]
        m[1] = m[1] / mp["l_q"]
        self._model_constants = m
        self._three_phase_limits()

    def torque(self, currents):
        mp = self._motor_parameter
        return 1.5 * mp["p"] * ((mp["l_d"] - mp["l_q"]) * currents[0]) * currents[1]

    def _torque_limit(self):
        """Pure reluctance torque ~ i_d * i_q peaks at 45 degrees of current angle: both limits over sqrt(2)
        (behaviour of synchronous_reluctance_motor.py:131-133)."""
        k = 1.0 / math.sqrt(2.0)
        return self.torque([k * self._limits["i_sd"], k * self._limits["i_sq"], 0])

    def torque_coefficients(self):
        mp = self._motor_parameter
        return [0.0, 1.5 * mp["p"] * (mp["l_d"] - mp["l_q"])]

    def initial_motor_state(self):
        vals = [float(v) for v in self._initial_states.values()]
        return vals if len(vals) == 3 else [0.0, 0.0, 0.0]


class SquirrelCageInductionMotor(_ThreePhaseMotor):
    """electric_motors/squirrel_cage_induction_motor.py:8-157 over induction_motor.py:7-364."""

    CURRENTS = ["i_salpha", "i_sbeta"]
    FLUXES = ["psi_ralpha", "psi_rbeta"]
    IO_VOLTAGES = ["u_sa", "u_sb", "u_sc", "u_salpha", "u_sbeta", "u_sd", "u_sq"]
    IO_CURRENTS = ["i_sa", "i_sb", "i_sc", "i_salpha", "i_sbeta", "i_sd", "i_sq"]
    _default_motor_parameter = {"p": 2, "l_m": 143.75e-3, "l_sigs": 5.87e-3, "l_sigr": 5.87e-3, "j_rotor": 1.1e-3,
                                "r_s": 2.9338, "r_r": 1.355}
    _default_limits = dict(omega=4e3 * np.pi / 30, torque=0.0, i=5.5, epsilon=math.pi, u=560)
    _default_nominal_values = dict(omega=3e3 * np.pi / 30, torque=0.0, i=3.9, epsilon=math.pi, u=560)
    _default_initializer = {"states": {"i_salpha": 0.0, "i_sbeta": 0.0, "psi_ralpha": 0.0, "psi_rbeta": 0.0, "epsilon": 0.0},
                            "interval": None, "random_init": None, "random_params": (None, None)}

    def __init__(self, motor_parameter=None, nominal_values=None, limit_values=None, motor_initializer=None):
        super().__init__(motor_parameter, nominal_values, limit_values, motor_initializer)
        mp = self._motor_parameter
        # InductionMotor._update_model, induction_motor.py:287-312
        l_s = mp["l_m"] + mp["l_sigs"]
        l_r = mp["l_m"] + mp["l_sigr"]
        sigma = (l_s * l_r - mp["l_m"] ** 2) / (l_s * l_r)
        tau_r = l_r / mp["r_r"]
        tau_sig = sigma * l_s / (mp["r_s"] + mp["r_r"] * (mp["l_m"] ** 2) / (l_r**2))
        k = mp["l_m"] / (sigma * l_r * l_s)
        g = mp["l_m"] * mp["r_r"] / (sigma * l_s * l_r**2)
        self._model_constants = np.array([
            [0, -1 / tau_sig, 0, g, 0, 0, +k * mp["p"], 1 / (sigma * l_s), 0, -k, 0],
            [0, 0, -1 / tau_sig, 0, g, -k * mp["p"], 0, 0, 1 / (sigma * l_s), 0, -k],
            [0, mp["l_m"] / tau_r, 0, -1 / tau_r, 0, 0, -mp["p"], 0, 0, 1, 0],
            [0, 0, mp["l_m"] / tau_r, 0, -1 / tau_r, mp["p"], 0, 0, 0, 0, 1],
            [mp["p"], 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        ], dtype=float)
        self._three_phase_limits()

    def _torque_limit(self):
        """induction_motor.py:223-234."""
        mp = self._motor_parameter
        return 1.5 * mp["p"] * mp["l_m"] ** 2 / (mp["l_m"] + mp["l_sigr"]) * self._limits["i_sd"] * self._limits["i_sq"] / 2

    def torque_coefficients(self):
        mp = self._motor_parameter
        return [1.5 * mp["p"] * mp["l_m"] / (mp["l_m"] + mp["l_sigr"]), 0.0]

    def initial_motor_state(self):
        vals = [float(v) for v in self._initial_states.values()]
        return vals if len(vals) == 5 else [0.0] * 5


class DoublyFedInductionMotor(SquirrelCageInductionMotor):
    """electric_motors/doubly_fed_induction_motor.py:8-240 over induction_motor.py: same model matrix as the SCIM
    (InductionMotor._update_model, induction_motor.py:287-312) with a live rotor voltage; defaults lines 85-110
    (DOI 10.1016/j.jestch.2016.01.015)."""

    ROTOR_VOLTAGES = ["u_ralpha", "u_rbeta"]
    ROTOR_CURRENTS = ["i_ralpha", "i_rbeta"]
    IO_ROTOR_VOLTAGES = ["u_ra", "u_rb", "u_rc", "u_rd", "u_rq"]
    IO_ROTOR_CURRENTS = ["i_ra", "i_rb", "i_rc", "i_rd", "i_rq"]
    IO_VOLTAGES = SquirrelCageInductionMotor.IO_VOLTAGES + IO_ROTOR_VOLTAGES
    IO_CURRENTS = SquirrelCageInductionMotor.IO_CURRENTS + IO_ROTOR_CURRENTS
    _default_motor_parameter = {"p": 2, "l_m": 297.5e-3, "l_sigs": 25.71e-3, "l_sigr": 25.71e-3, "j_rotor": 13.695e-3,
                                "r_s": 4.42, "r_r": 3.51}
    _default_limits = dict(omega=1800 * np.pi / 30, torque=0.0, i=9, epsilon=math.pi, u=720)
    _default_nominal_values = dict(omega=1650 * np.pi / 30, torque=0.0, i=7.5, epsilon=math.pi, u=720)

    def _three_phase_limits(self):
        """doubly_fed_induction_motor.py:119-141: like the SCIM, over stator AND rotor quantities, with r_r as the fallback."""
        voltage_limit = 0.5 * self._limits["u"]
        voltage_nominal = 0.5 * self._nominal_values["u"]
        limits_agenda, nominal_agenda = {}, {}
        r_r = self._motor_parameter["r_r"]
        for u, i in zip(self.IO_VOLTAGES + self.ROTOR_VOLTAGES, self.IO_CURRENTS + self.ROTOR_CURRENTS):
            limits_agenda[u] = voltage_limit
            nominal_agenda[u] = voltage_nominal
            limits_agenda[i] = self._limits.get("i", None) or self._limits[u] / r_r
            nominal_agenda[i] = self._nominal_values.get("i", None) or self._nominal_values[u] / r_r
        self._update_limits(limits_agenda, nominal_agenda)
        self._update_limits(dict(torque=self._torque_limit()))

    def rotor_current_coefficients(self):
        """calculate_rotor_current (physical_systems.py:931-946): i_r = psi_r / l_r - l_m / l_r * i_s."""
        mp = self._motor_parameter
        l_r = mp["l_m"] + mp["l_sigr"]
        return [1 / l_r, mp["l_m"] / l_r]


# ------------------------------------------------------------------------------------------------- loads
class _MechanicalLoad:
    """mechanical_loads/mechanical_load.py:9-236 (bookkeeping only)."""

    _default_initializer = {}

    def __init__(self, j_load=0.0, load_initializer=None):
        self._j_total = self._j_load = float(j_load)
        self._state_names = ["omega"]
        self._limits = {}
        self._nominal_values = {}
        self._initializer = dict(self._default_initializer)
        self._initializer.update(load_initializer or {})
        if self._initializer.get("random_init") not in (None, "uniform", "normal", "gaussian"):
            raise NotImplementedError(f"random_init {self._initializer.get('random_init')!r} (mechanical_load.py:131-156 knows uniform / gaussian)")
        self._initial_states = dict(self._initializer.get("states", {"omega": 0.0}))

    j_total = property(lambda self: self._j_total)
    state_names = property(lambda self: self._state_names)
    limits = property(lambda self: self._limits)
    nominal_values = property(lambda self: self._nominal_values)
    initializer = property(lambda self: self._initializer)

    def set_j_rotor(self, j_rotor):
        self._j_total = self._j_load + float(j_rotor)

    def initial_omega(self):
        return float(self._initial_states.get("omega", 0.0))

    def get_state_space(self, omega_range):
        """mechanical_load.py:225-236."""
        return {"omega": omega_range[0]}, {"omega": omega_range[1]}


class ConstantSpeedLoad(_MechanicalLoad):
    """mechanical_loads/constant_speed_load.py:6-46."""

    _default_initializer = {"states": {"omega": 0.0}, "interval": None, "random_init": None, "random_params": (None, None)}

    def __init__(self, omega_fixed=0, load_initializer=None, **kwargs):
        super().__init__(load_initializer=load_initializer, **kwargs)
        self._initial_states = dict(self._initial_states)
        self._omega = omega_fixed or self._initial_states["omega"]
        if omega_fixed != 0:
            self._initial_states["omega"] = omega_fixed

    @property
    def omega_fixed(self):
        return self._omega


class PolynomialStaticLoad(_MechanicalLoad):
    """mechanical_loads/polynomial_static_load.py:8-107."""

    _load_parameter = dict(a=0.0, b=0.0, c=0.0, j_load=1e-5)
    _default_initializer = {"states": {"omega": 0.0}, "interval": None, "random_init": None, "random_params": (None, None)}
    tau_decay = 1e-3

    def __init__(self, load_parameter=None, limits=None, load_initializer=None):
        self._load_parameter = update_parameter_dict(self._load_parameter, load_parameter or {})
        super().__init__(j_load=self._load_parameter["j_load"], load_initializer=load_initializer)
        self._limits.update(limits or {})

    @property
    def load_parameter(self):
        return self._load_parameter
