"""ctypes binding of include/gemx.h.  The product path has NO fallback: if `libgemx.so` is missing or no
HIP device is visible, creating a system raises."""
import ctypes as C
import os

from . import build as _build

MAX_ODE, MAX_OUT, MODEL_ROWS, MODEL_COLS = 8, 24, 5, 11
ABI_VERSION = 7

SYS_DC_PERMEX, SYS_SYNC, SYS_SCIM, SYS_DC_SERIES, SYS_DC_SHUNT, SYS_DC_EXTEX, SYS_EESM, SYS_DFIM = 0, 1, 2, 3, 4, 5, 6, 7
CONV_CONT_4QC, CONV_FINITE_B6, CONV_CONT_B6, CONV_FINITE_4QC = 0, 1, 2, 3
ACT_ABC, ACT_DQ_SPACE, ACT_DQ_PROCESSOR = 0, 1, 2
SUPPLY_IDEAL, SUPPLY_RC = 0, 1
INIT_CONST, INIT_UNIFORM, INIT_GAUSSIAN = 0, 1, 2
MAX_DELAY = 8
CONV_CONT_2X4QC, CONV_FINITE_2X4QC, CONV_CONT_B6_4QC, CONV_FINITE_B6_4QC, CONV_CONT_2XB6, CONV_FINITE_2XB6 = 4, 5, 6, 7, 8, 9
LOAD_CONST_SPEED, LOAD_POLY_STATIC = 0, 1
SOLVER_EULER, SOLVER_RK4, SOLVER_DP5 = 0, 1, 2
SOLVER_SPLIT_KINKS, SOLVER_ADAPTIVE = 1, 2
F32, F64 = 0, 1
OBS_AOS, OBS_SOA = 0, 1
ERRFLAG_ACTION, ERRFLAG_OMEGA_MOVED, ERRFLAG_TOLERANCE = 1, 2, 4  # gemx_error_flags bits (GEMX_ERRFLAG_*)


class GemxConfig(C.Structure):
    """Mirror of `gemx_config` (include/gemx.h)."""

    _fields_ = [
        ("struct_size", C.c_int32), ("abi_version", C.c_int32),
        ("system_kind", C.c_int32), ("converter_kind", C.c_int32), ("load_kind", C.c_int32),
        ("solver_kind", C.c_int32), ("solver_nsteps", C.c_int32), ("solver_flags", C.c_int32),
        ("dtype", C.c_int32), ("obs_layout", C.c_int32), ("auto_reset", C.c_int32),
        ("limit_mask", C.c_uint32), ("squared_mask", C.c_uint32),
        ("action_frame", C.c_int32), ("action_delay", C.c_int32),
        ("supply_kind", C.c_int32), ("init_kind", C.c_int32), ("init_flux_mode", C.c_int32), ("init_flux", C.c_double * 8),
        ("seed", C.c_uint64), ("env_base", C.c_int64),
        ("init_lo", C.c_double * MAX_ODE), ("init_hi", C.c_double * MAX_ODE), ("init_mu", C.c_double * MAX_ODE),
        ("init_sigma", C.c_double * MAX_ODE),
        ("supply_r", C.c_double), ("supply_c", C.c_double), ("action_delay_reset", C.c_double * 6), ("solver_rtol", C.c_double), ("solver_atol", C.c_double), ("solver_atol_omega", C.c_double),
        ("tau", C.c_double), ("interlocking_time", C.c_double), ("u_nominal", C.c_double),
        ("model", C.c_double * (MODEL_ROWS * MODEL_COLS)),
        ("torque_coef", C.c_double * 4),
        ("j_total", C.c_double), ("load_a", C.c_double), ("load_b", C.c_double), ("load_c", C.c_double),
        ("tau_decay", C.c_double),
        ("limits", C.c_double * MAX_OUT),
        ("init_state", C.c_double * MAX_ODE),
    ]


class GemxError(RuntimeError):
    pass


_lib = None

MAX_REF = 4


class GemxRewardConfig(C.Structure):
    """Mirror of `gemx_reward_config` (include/gemx.h)."""

    _fields_ = [
        ("struct_size", C.c_int32), ("n_ref", C.c_int32), ("ref_index", C.c_int32 * MAX_REF),
        ("weight", C.c_double * MAX_OUT), ("power", C.c_double * MAX_OUT), ("state_length", C.c_double * MAX_OUT),
        ("bias", C.c_double), ("violation_reward", C.c_double),
    ]


class GemxRefgenConfig(C.Structure):
    """Mirror of `gemx_refgen_config` (include/gemx.h)."""

    _fields_ = [
        ("struct_size", C.c_int32), ("n_ref", C.c_int32), ("seed", C.c_uint64), ("env_base", C.c_int64),
        ("episode_len_lo", C.c_int32), ("episode_len_hi", C.c_int32),
        ("sigma_lo", C.c_double * MAX_REF), ("sigma_hi", C.c_double * MAX_REF),
        ("margin_lo", C.c_double * MAX_REF), ("margin_hi", C.c_double * MAX_REF),
        ("initial_lo", C.c_double * MAX_REF), ("initial_hi", C.c_double * MAX_REF),
    ]


EXPORTS = (
    "gemx_abi_version", "gemx_sizeof_config", "gemx_last_error", "gemx_device_count", "gemx_create", "gemx_destroy",
    "gemx_n_envs", "gemx_n_ode", "gemx_n_out", "gemx_n_action", "gemx_action_itemsize", "gemx_n_switch_bytes", "gemx_reset_observation", "gemx_set_reward", "gemx_rollout_reward", "gemx_refgen_create", "gemx_refgen_destroy", "gemx_refgen_reset",
    "gemx_refgen_rollout", "gemx_refgen_get_state",
    "gemx_reset", "gemx_step", "gemx_rollout", "gemx_rollout_half", "gemx_get_state", "gemx_set_state", "gemx_get_switch_state",
    "gemx_set_switch_state", "gemx_aux_state_bytes", "gemx_get_aux_state", "gemx_set_aux_state", "gemx_reset_again", "gemx_rollout_synthetic", "gemx_synthetic_actions", "gemx_set_rate_limiter", "gemx_set_steps_per_block", "gemx_last_launch", "gemx_error_flags", "gemx_debug_read",
)


def library_path():
    """In-tree libgemx.so; GEMX_LIBRARY names another build of the SAME library (A/B runs of kernel variants)."""
    return os.environ.get("GEMX_LIBRARY") or _build.LIB


def load():
    """Load libgemx.so (built in-tree by `__graft_entry__.build()` / `gym_electric_motor_amd.build.build_library()`)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise GemxError(
            f"{path} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # torch bundles its own libamdhip64.so.7; libgemx.so depends on the same SONAME.  Import torch FIRST so that
    # one HIP runtime serves both (loading the system runtime first and torch's second breaks device discovery).
    import torch  # noqa: F401

    L = C.CDLL(path)
    L.gemx_last_error.restype = C.c_char_p
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.gemx_create.argtypes = [C.POINTER(GemxConfig), i64, C.c_int, C.POINTER(vp)]
    L.gemx_destroy.argtypes = [vp]
    L.gemx_n_envs.argtypes = [vp, C.POINTER(i64)]
    for f in ("gemx_n_ode", "gemx_n_out", "gemx_n_action", "gemx_action_itemsize"):
        getattr(L, f).argtypes = [vp]
    L.gemx_reset_observation.argtypes = [vp, C.POINTER(C.c_double)]
    L.gemx_reset.argtypes = [vp, vp, vp, vp]
    L.gemx_reset_again.argtypes = [vp, vp, vp, vp]
    L.gemx_aux_state_bytes.argtypes = [vp]
    L.gemx_aux_state_bytes.restype = i64
    L.gemx_rollout_synthetic.argtypes = [vp, C.c_uint64, C.c_uint32, i32, vp, vp, vp]
    L.gemx_set_rate_limiter.argtypes = [vp, i32, C.c_double]
    L.gemx_synthetic_actions.argtypes = [vp, C.c_uint64, C.c_uint32, i32, vp, vp]
    L.gemx_get_aux_state.argtypes = [vp, vp, vp]
    L.gemx_set_aux_state.argtypes = [vp, vp, vp]
    L.gemx_step.argtypes = [vp, vp, vp, vp, vp]
    L.gemx_rollout.argtypes = [vp, vp, i32, vp, vp, i32, vp]
    L.gemx_rollout_half.argtypes = [vp, vp, i32, vp, vp, vp]
    L.gemx_set_reward.argtypes = [vp, C.POINTER(GemxRewardConfig)]
    L.gemx_refgen_create.argtypes = [C.POINTER(GemxRefgenConfig), i64, C.c_int, C.c_int, C.POINTER(vp)]
    L.gemx_refgen_destroy.argtypes = [vp]
    L.gemx_refgen_reset.argtypes = [vp, vp, vp]
    L.gemx_refgen_rollout.argtypes = [vp, vp, i32, vp, vp]
    L.gemx_refgen_get_state.argtypes = [vp, vp, vp, vp, vp]
    L.gemx_rollout_reward.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    L.gemx_get_state.argtypes = [vp, vp, vp]
    L.gemx_set_state.argtypes = [vp, vp, vp]
    L.gemx_get_switch_state.argtypes = [vp, vp, vp]
    L.gemx_set_switch_state.argtypes = [vp, vp, vp]
    L.gemx_set_steps_per_block.argtypes = [vp, i32]
    L.gemx_last_launch.argtypes = [vp]
    L.gemx_last_launch.restype = C.c_char_p
    L.gemx_error_flags.argtypes = [vp, C.POINTER(C.c_uint32), vp]
    if L.gemx_abi_version() != ABI_VERSION or L.gemx_sizeof_config() != C.sizeof(GemxConfig):
        raise GemxError("libgemx.so ABI does not match gym_electric_motor_amd._lib.GemxConfig; rebuild the library")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = load().gemx_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise GemxError(msg)
