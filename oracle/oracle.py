"""ctypes wrapper around oracle/libgemx_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (gym_electric_motor_amd) never does.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgemx_oracle.so")

SYS_DC, SYS_PMSM, SYS_SCIM, SYS_DC_SERIES, SYS_DC_SHUNT, SYS_DC_EXTEX, SYS_EESM, SYS_DFIM = 0, 1, 2, 3, 4, 5, 6, 7
CONV_C4QC, CONV_FB6, CONV_CB6, CONV_F4QC = 0, 1, 2, 3
CONV_C2X4QC, CONV_F2X4QC, CONV_CB6_4QC, CONV_FB6_4QC, CONV_C2XB6, CONV_F2XB6 = 4, 5, 6, 7, 8, 9  # Cont/FiniteMultiConverter of two sub-converters
LOAD_CONST, LOAD_POLY = 0, 1
SOLVER_EULER, SOLVER_RK4, SOLVER_DOPRI5, SOLVER_DP5_FIXED, SOLVER_IVP_RK45, SOLVER_RK4_KINK, SOLVER_DP5_KINK = 0, 1, 2, 3, 4, 5, 6

MAX_ODE, MAX_OUT = 8, 24


class OrcParams(C.Structure):
    _fields_ = [
        ("system", C.c_int32), ("converter", C.c_int32), ("load", C.c_int32), ("solver", C.c_int32),
        ("nsteps", C.c_int32), ("limit_mask", C.c_int32), ("squared_mask", C.c_int32), ("dq_mode", C.c_int32),
        ("act_delay", C.c_int32), ("rc_supply", C.c_int32),
        ("tau", C.c_double), ("t_il", C.c_double), ("u_sup", C.c_double),
        ("mp", C.c_double * 8),
        ("j_total", C.c_double), ("load_a", C.c_double), ("load_b", C.c_double), ("load_c", C.c_double),
        ("tau_decay", C.c_double),
        ("limits", C.c_double * MAX_OUT),
        ("init", C.c_double * MAX_ODE),
        ("sup_r", C.c_double), ("sup_c", C.c_double),
        ("rtol", C.c_double), ("atol", C.c_double),
        ("act_delay_reset", C.c_double * 6),
    ]


def build():
    """(Re)build the oracle library with gcc if it is missing or stale."""
    src = os.path.join(HERE, "gemx_oracle.c")
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        assert L.orc_sizeof_params() == C.sizeof(OrcParams)
        L.orc_sizeof_env.restype = C.c_size_t
        L.orc_kat_poly_load.restype = C.c_double
        L.orc_kat_poly_load.argtypes = [C.POINTER(OrcParams), C.c_double, C.c_double]
        _lib = L
    return _lib


_SYS = {"DcMotorSystem": SYS_DC, "SynchronousMotorSystem": SYS_PMSM, "SquirrelCageInductionMotorSystem": SYS_SCIM,
        "ExternallyExcitedSynchronousMotorSystem": SYS_EESM, "DoublyFedInductionMotorSystem": SYS_DFIM}
_CONV = {"ContFourQuadrantConverter": CONV_C4QC, "FiniteB6BridgeConverter": CONV_FB6, "ContB6BridgeConverter": CONV_CB6,
         "FiniteFourQuadrantConverter": CONV_F4QC,
         # multi converters: make_golden.describe() appends the sub-converter class names
         "ContMultiConverter[ContFourQuadrantConverter,ContFourQuadrantConverter]": CONV_C2X4QC,
         "FiniteMultiConverter[FiniteFourQuadrantConverter,FiniteFourQuadrantConverter]": CONV_F2X4QC,
         "ContMultiConverter[ContB6BridgeConverter,ContFourQuadrantConverter]": CONV_CB6_4QC,
         "FiniteMultiConverter[FiniteB6BridgeConverter,FiniteFourQuadrantConverter]": CONV_FB6_4QC,
         "ContMultiConverter[ContB6BridgeConverter,ContB6BridgeConverter]": CONV_C2XB6,
         "FiniteMultiConverter[FiniteB6BridgeConverter,FiniteB6BridgeConverter]": CONV_F2XB6}
_DC_MOTOR_SYS = {"DcPermanentlyExcitedMotor": SYS_DC, "DcSeriesMotor": SYS_DC_SERIES, "DcShuntMotor": SYS_DC_SHUNT,
                 "DcExternallyExcitedMotor": SYS_DC_EXTEX}
_LOAD = {"ConstantSpeedLoad": LOAD_CONST, "PolynomialStaticLoad": LOAD_POLY}
_SOLVER = {"euler": (SOLVER_EULER, 1), "euler4": (SOLVER_EULER, 4), "rk4": (SOLVER_RK4, 1), "rk4x4": (SOLVER_RK4, 4), "rk4x8": (SOLVER_RK4, 8),
           "dopri5": (SOLVER_DOPRI5, 1), "dp5_fixed": (SOLVER_DP5_FIXED, 1),
           # scipy.integrate.solve_ivp(method="RK45") as ScipySolveIvpSolver drives it (solvers.py:187-219); tolerances below
           "ivp": (SOLVER_IVP_RK45, 1), "ivp_tight": (SOLVER_IVP_RK45, 1),
           # what the HIP kernels do for a PolynomialStaticLoad: a fixed step corrected for the load's kinks in closed form (no reference counterpart)
           "rk4_kink": (SOLVER_RK4_KINK, 1), "dp5_kink": (SOLVER_DP5_KINK, 1),
           # DIAGNOSTIC: the HIP kernels' error controller (ScipyOdeSolver() on the device) in fp64, for wave statistics on the CPU
           "dev_adaptive": (7, 1), "dev_adaptive_kink": (8, 1)}
_IVP_TOL = {"ivp": (1e-3, 1e-6), "ivp_tight": (1e-10, 1e-12)}  # solve_ivp defaults | oracle/make_golden.py:make_solver("ivp_tight")
_MP_KEYS = {SYS_DC: ("r_a", "l_a", "psi_e"), SYS_PMSM: ("p", "l_d", "l_q", "r_s", "psi_p"),
            SYS_SCIM: ("p", "l_m", "l_sigs", "l_sigr", "r_s", "r_r"),
            SYS_DC_SERIES: ("r_a", "r_e", "l_a", "l_e", "l_e_prime"), SYS_DC_SHUNT: ("r_a", "r_e", "l_a", "l_e", "l_e_prime"),
            SYS_DC_EXTEX: ("r_a", "r_e", "l_a", "l_e", "l_e_prime"),
            SYS_EESM: ("p", "l_d", "l_q", "l_m", "l_e", "r_s", "r_e", "k"),
            SYS_DFIM: ("p", "l_m", "l_sigs", "l_sigr", "r_s", "r_r")}


def default_masks(meta):
    """Default constraints of the configured envs: LimitConstraint('i') for Cont-CC-PermExDc-v0
    (cont_cc_permex_dc_env.py:104), SquaredConstraint(('i_sq','i_sd')) for PMSM / SCIM
    (finite_cc_pmsm_env.py:106, cont_sc_scim_env.py:111)."""
    names = meta["state_names"]
    if meta["system"] == "DcMotorSystem":
        if meta["motor"] in ("DcShuntMotor", "DcExternallyExcitedMotor"):  # constraints=("i_a", "i_e"), cont_cc_shunt_dc_env.py:103
            return (1 << names.index("i_a")) | (1 << names.index("i_e")), 0
        return 1 << names.index("i"), 0
    lim = 1 << names.index("i_e") if "i_e" in names else 0  # EESM: + LimitConstraint(("i_e",)), cont_cc_eesm_env.py:108
    return lim, (1 << names.index("i_sd")) | (1 << names.index("i_sq"))


def params_from_meta(meta, solver=None, episodic=None):
    """Build OrcParams from a golden fixture's meta blob (see oracle/make_golden.py:describe)."""
    if isinstance(meta, (str, bytes, np.ndarray)):
        meta = json.loads(str(meta))
    p = OrcParams()
    p.system = _SYS[meta["system"]]
    if meta["system"] == "DcMotorSystem":
        p.system = _DC_MOTOR_SYS[meta["motor"]]
    p.converter = _CONV[meta["converter"]]
    p.load = _LOAD[meta["load"]]
    p.solver, p.nsteps = _SOLVER[solver or meta["solver"]]
    p.rtol, p.atol = _IVP_TOL.get(solver or meta["solver"], (0.0, 0.0))
    epi = meta.get("episodic", False) if episodic is None else episodic
    if epi:
        p.limit_mask, p.squared_mask = default_masks(meta)
    p.tau, p.t_il, p.u_sup = meta["tau"], meta["interlocking_time"], meta["u_nominal"]
    for i, k in enumerate(_MP_KEYS[p.system]):
        p.mp[i] = meta["motor_parameter"].get(k, 0.0)  # SynchronousReluctanceMotor has no psi_p (-> 0)
    p.j_total = meta["j_total"]
    lp = meta.get("load_parameter", {})
    p.load_a, p.load_b, p.load_c = lp.get("a", 0.0), lp.get("b", 0.0), lp.get("c", 0.0)
    p.tau_decay = meta.get("tau_decay", 1e-3)
    for i, v in enumerate(meta["limits"]):
        p.limits[i] = v
    p.init[0] = meta.get("omega_fixed", 0.0)  # default initialisers: omega_fixed | 0, motor states 0
    # action-side wrappers / control space recorded by make_golden.run_case
    p.dq_mode = {"abc": 0, "dq": 1, "dq_processor": 2}[meta.get("action_frame", "abc")]
    p.act_delay = int(meta.get("dead_time_steps", 0))
    for i, v in enumerate(meta.get("dead_time_reset_action") or []):  # DeadTimeProcessor(reset_action=lambda: [a] * steps): the one action a
        p.act_delay_reset[i] = float(v)
    if meta["supply"] == "RCVoltageSupply":
        p.rc_supply, p.sup_r, p.sup_c = 1, meta["supply_parameter"]["R"], meta["supply_parameter"]["C"]
    return p


class OracleEnv:
    """One fp64 reference-restatement env."""

    def __init__(self, params):
        self.p = params
        self.L = lib()
        self._env = C.create_string_buffer(self.L.orc_sizeof_env())
        self.L.orc_init(C.byref(self.p), self._env)
        self.n_out = self.L.orc_n_out(C.byref(self.p))
        self.n_ode = self.L.orc_n_ode(C.byref(self.p))
        self.n_act = self.L.orc_n_act(C.byref(self.p))

    @property
    def y(self):
        """ODE state [omega, motor states...] (a view into the C struct: orc_env starts with `double y[8]`)."""
        return np.frombuffer(self._env, dtype=np.float64, count=MAX_ODE)

    def reset(self):
        obs = np.zeros(self.n_out)
        self.L.orc_reset(C.byref(self.p), self._env, obs.ctypes.data_as(C.c_void_p))
        return obs

    def step(self, action):
        a = np.ascontiguousarray(np.atleast_1d(action), dtype=np.float64)
        obs = np.zeros(self.n_out)
        self.L.orc_step(C.byref(self.p), self._env, a.ctypes.data_as(C.c_void_p), obs.ctypes.data_as(C.c_void_p))
        return obs

    def rollout(self, actions, auto_reset=True):
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).reshape(len(actions), -1))
        K = a.shape[0]
        obs = np.zeros((K, self.n_out))
        done = np.zeros(K, dtype=np.uint8)
        self.L.orc_rollout(C.byref(self.p), self._env, a.ctypes.data_as(C.c_void_p), C.c_int(a.shape[1]), C.c_int(K),
                           C.c_int(int(auto_reset)), obs.ctypes.data_as(C.c_void_p), done.ctypes.data_as(C.c_void_p))
        return obs, done.astype(bool)

    def rollout_diag(self, actions, auto_reset=True):
        """rollout() plus per-step diagnostics: (obs, done, |psi_r| at the start of each step [Wb], smallest |i| a current-sign
        decision of the step met [A], inf where the step made none)."""
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).reshape(len(actions), -1))
        K = a.shape[0]
        obs = np.zeros((K, self.n_out))
        done = np.zeros(K, dtype=np.uint8)
        diag = np.zeros((K, 2))
        self.L.orc_rollout_diag(C.byref(self.p), self._env, a.ctypes.data_as(C.c_void_p), C.c_int(a.shape[1]), C.c_int(K),
                                C.c_int(int(auto_reset)), obs.ctypes.data_as(C.c_void_p), done.ctypes.data_as(C.c_void_p),
                                diag.ctypes.data_as(C.c_void_p))
        margin = np.where(diag[:, 1] > 1e299, np.inf, diag[:, 1])
        return obs, done.astype(bool), diag[:, 0], margin

    def done(self, obs):
        o = np.ascontiguousarray(obs, dtype=np.float64)
        full = np.zeros(MAX_OUT)
        full[: len(o)] = o
        return bool(self.L.orc_done(C.byref(self.p), full.ctypes.data_as(C.c_void_p)))

    def kat_converter(self, action, t, currents):
        """set_action + per-segment convert; currents [2,n] -> (nseg, volt [2,n])  (n = 3, or 2 / 4 for multi converters)."""
        a = np.ascontiguousarray(np.atleast_1d(action), dtype=np.float64)
        cur = np.ascontiguousarray(currents, dtype=np.float64)
        cur = cur.reshape(2, -1)
        n = cur.shape[1]
        volt = np.zeros((2, n))
        nseg = self.L.orc_kat_converter_n(C.byref(self.p), self._env, a.ctypes.data_as(C.c_void_p), C.c_double(t),
                                          cur.ctypes.data_as(C.c_void_p), volt.ctypes.data_as(C.c_void_p), C.c_int(n))
        return nseg, volt

    def kat_converter_reset(self):
        u = np.zeros(6)
        self.L.orc_kat_converter_reset(C.byref(self.p), self._env, u.ctypes.data_as(C.c_void_p))
        return u

    def model_constants(self):
        Cm = np.zeros((5, 11))
        self.L.orc_model_constants(C.byref(self.p), Cm.ctypes.data_as(C.c_void_p))
        return Cm


def undefined_field_angle_steps(params, actions, auto_reset=True, flux_floor=1e-9, exact=False):
    """Induction-motor systems report dq quantities in the rotor-flux frame, eps_field = arctan2(psi_rbeta, psi_ralpha)
    (physical_systems.py:765-769 / 918-929).  While the rotor flux is still (numerically) zero -- the first steps after a
    reset -- the reference's angle is the arctan2 of matmul rounding noise (~1e-17 Wb), which no restatement can reproduce.
    Returns a bool mask [K]: True where the flux magnitude at the START of step k is below `flux_floor` [Wb], i.e. where the
    dq columns of step k are not comparable (everything else, including |i_dq| and the done mask, is).
    exact=True: returns (mask, zero) with zero[k] True where that flux is EXACTLY zero -- there arctan2(0, 0) = 0 is this build's
    (documented) field angle, so its dq columns must equal the alpha-beta quantities."""
    a = np.asarray(actions, dtype=np.float64).reshape(len(actions), -1)
    env = OracleEnv(params)
    env.reset()
    mask = np.zeros(len(a), dtype=bool)
    zero = np.zeros(len(a), dtype=bool)
    if params.system not in (SYS_SCIM, SYS_DFIM):
        return (mask, zero) if exact else mask
    for k in range(len(a)):
        mask[k] = np.hypot(env.y[3], env.y[4]) < flux_floor
        zero[k] = env.y[3] == 0.0 and env.y[4] == 0.0
        obs = env.step(a[k])
        if auto_reset and env.done(obs):
            env.reset()
    return (mask, zero) if exact else mask


DQ_COLUMNS = ("i_sd", "i_sq", "i_rd", "i_rq", "u_sd", "u_sq", "u_rd", "u_rq")


def rewards(meta, states, references, violated):
    """WeightedSumOfErrors over a trajectory: states / references [K, S_out] (normalised), violated [K] -> rewards [K]."""
    rw = meta["reward"]
    L = lib()
    L.orc_reward.restype = C.c_double
    n = len(rw["weights"])
    arr = lambda x: (C.c_double * n)(*[float(v) for v in x])  # noqa: E731
    w, pw, sl = arr(rw["weights"]), arr(rw["powers"]), arr(rw["state_length"])
    out = np.zeros(len(states))
    for k in range(len(states)):
        out[k] = L.orc_reward(C.c_int(n), w, pw, sl, C.c_double(rw["bias"]), C.c_double(rw["violation_reward"]), arr(states[k]),
                              arr(references[k]), C.c_int(int(violated[k])))
    return out


def rollout_many(params, actions, auto_reset=True):
    """actions [K, n_env, A] -> (last_obs [n_env, S_out], n_done).  Single-threaded; used as cpu_baseline 'port'."""
    a = np.ascontiguousarray(actions, dtype=np.float64)
    if a.ndim == 2:
        a = a[:, :, None]
    K, n_env, A = a.shape
    L = lib()
    n_out = L.orc_n_out(C.byref(params))
    last = np.zeros((n_env, n_out))
    nd = C.c_int64(0)
    L.orc_rollout_many(C.byref(params), C.c_int(n_env), a.ctypes.data_as(C.c_void_p), C.c_int(A), C.c_int(K),
                       C.c_int(int(auto_reset)), last.ctypes.data_as(C.c_void_p), C.byref(nd))
    return last, nd.value


def wave_attempts(params, actions, shared=False):
    """orc_wave_attempts: actions [K, lanes <= 64, A] -> (hist_lane[32], hist_wave[32]) of attempts per control step under the device's
    error controller (params.solver must be "dev_adaptive")."""
    a = np.ascontiguousarray(np.asarray(actions, dtype=np.float64))
    K, lanes, A = a.shape
    hl, hw = np.zeros(32, dtype=np.int64), np.zeros(32, dtype=np.int64)
    lib().orc_wave_attempts(C.byref(params), C.c_int(lanes), a.ctypes.data_as(C.c_void_p), C.c_int(A), C.c_int(K), C.c_int(int(shared)),
                            hl.ctypes.data_as(C.c_void_p), hw.ctypes.data_as(C.c_void_p))
    return hl, hw


def load_golden(name, golden_dir=None):
    golden_dir = golden_dir or os.path.join(os.path.dirname(HERE), "tests", "golden")
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.loads(str(d["meta"]))
    return d, meta
