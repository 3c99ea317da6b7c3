/*
 * gemx.h -- C ABI of the MI355X-native batched physical-system stepper for gym-electric-motor (GEM).
 *
 * The reference (upb-lea/gym-electric-motor 3.0.2) is pure Python and has no FFI: its replaceable seam is
 * the plugin base class gym_electric_motor.core.PhysicalSystem (core.py:589-705) whose two hot methods are
 *     simulate(action) -> state / limits        (core.py:687-698; SCMLSystem.simulate physical_systems.py:171-203,
 *                                                SynchronousMotorSystem.simulate 487-525,
 *                                                SquirrelCageInductionMotorSystem.simulate 771-814)
 *     reset()          -> state / limits        (core.py:678-685; physical_systems.py:256-287, 527-561, 816-847)
 * This library is what a GEM maintainer binds with ctypes behind that seam (INTEGRATION.md shows the stub):
 * one handle = N independent SCML systems (supply -> converter -> motor ODE + load ODE -> normalised state,
 * constraint-violation done flag) advanced in lockstep by hand-written gfx950 kernels.
 *
 * Conventions
 *  - plain pointers and sizes only; all `*_dev` pointers are DEVICE pointers owned by the caller
 *    (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - every call returns GEMX_OK (0) or a negative gemx_status; the message is in gemx_last_error().
 *    Nothing throws across the boundary.  Calls on one handle are not thread-safe; launches are asynchronous.
 *  - real type R = float (dtype GEMX_F32, default; the product path) or double (GEMX_F64, diagnostic).
 *  - there is NO CPU fallback: without a HIP device gemx_create() fails with GEMX_ERR_DEVICE.
 */
#ifndef GEMX_H
#define GEMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEMX_ABI_VERSION 7 /* 2: GEMX_MAX_OUT 16 -> 24 (DFIM system, 24 states); 3: gemx_config.solver_flags; 4: solver_rtol / solver_atol; 5: init_flux_mode / init_flux;
                            * 6: gemx_get_aux_state / gemx_set_aux_state / gemx_aux_state_bytes, gemx_reset_again; reset counters stay at 1 after gemx_create;
                            *    + gemx_rollout_synthetic / gemx_synthetic_actions, gemx_set_rate_limiter (new entry points only);
                            * 7: gemx_config.env_base / gemx_refgen_config.env_base: every device random stream is keyed by the GLOBAL env index
                            *    env_base + i (shards of one job draw what the unsharded job draws); gemx_rollout_half; gemx_config.solver_atol_omega; GEMX_SOLVER_ADAPTIVE
                            *    honours GEMX_SOLVER_SPLIT_KINKS; the initial-state streams are Threefry-4x32-12 (were Philox4x32-10: other draws from
                            *    the same seed, same distributions); one unit library per (system, converter, dtype), loaded by gemx_create */
#define GEMX_MAX_ODE 8  /* ODE state length incl. omega and the angle      */
#define GEMX_MAX_OUT 24 /* system-state (observation) length               */
#define GEMX_MODEL_ROWS 5
#define GEMX_MODEL_COLS 11

typedef enum gemx_status {
    GEMX_OK = 0,
    GEMX_ERR_ARG = -1,     /* invalid argument / unsupported configuration (cf. reference asserts, converters.py:204-206) */
    GEMX_ERR_DEVICE = -2,  /* no HIP device / HIP runtime error                                                        */
    GEMX_ERR_ALLOC = -3
} gemx_status;

/* SCML system class: DcMotorSystem / SynchronousMotorSystem / SquirrelCageInductionMotorSystem */
typedef enum {
    GEMX_SYS_DC_PERMEX = 0, /* DcMotorSystem + DcPermanentlyExcitedMotor                                   */
    GEMX_SYS_SYNC = 1,      /* SynchronousMotorSystem (PMSM, SynRM)                                         */
    GEMX_SYS_SCIM = 2,      /* SquirrelCageInductionMotorSystem                                             */
    GEMX_SYS_DC_SERIES = 3, /* DcMotorSystem + DcSeriesMotor (electric_motors/dc_series_motor.py)           */
    GEMX_SYS_DC_SHUNT = 4,  /* DcMotorSystem + DcShuntMotor  (electric_motors/dc_shunt_motor.py)            */
    GEMX_SYS_DC_EXTEX = 5,  /* DcMotorSystem + DcExternallyExcitedMotor (dc_externally_excited_motor.py)     */
    GEMX_SYS_EESM = 6,      /* ExternallyExcitedSynchronousMotorSystem (physical_systems.py:564-691)         */
    GEMX_SYS_DFIM = 7       /* DoublyFedInductionMotorSystem (physical_systems.py:850-1113)                  */
} gemx_system_kind;
/* ContFourQuadrantConverter (converters.py:438-495), FiniteB6BridgeConverter (743-839), ContB6BridgeConverter (842-911),
 * FiniteFourQuadrantConverter (313-368; DC systems, actions 0..3).
 * Kinds 4..9 are Cont/FiniteMultiConverter (converters.py:498-740) holding exactly the two sub-converters of the
 * reference's ExtExDc envs (2 x 4QC: armature, excitation), EESM envs (B6 stator + 4QC excitation) and DFIM envs
 * (2 x B6: stator, rotor).
 *   continuous actions: the sub-converters' actions concatenated, A = 2 | 4 | 6;
 *   discrete actions:   ONE uint8 = a_0 + n_0 * a_1, the flat index of the reference's MultiDiscrete([n_0, n_1])
 *                       action [a_0, a_1] (n_0 = 4 | 8), i.e. 0..15 | 0..31 | 0..63.
 * The two sub-converters share interlocking_time; EESM handles refuse interlocking_time > 0 (the reference's
 * dead-time branch for this system cannot execute, physical_systems.py:634). */
typedef enum {
    GEMX_CONV_CONT_4QC = 0, GEMX_CONV_FINITE_B6 = 1, GEMX_CONV_CONT_B6 = 2, GEMX_CONV_FINITE_4QC = 3,
    GEMX_CONV_CONT_2X4QC = 4, GEMX_CONV_FINITE_2X4QC = 5, GEMX_CONV_CONT_B6_4QC = 6, GEMX_CONV_FINITE_B6_4QC = 7,
    GEMX_CONV_CONT_2XB6 = 8, GEMX_CONV_FINITE_2XB6 = 9
} gemx_converter_kind;
/* ConstantSpeedLoad (constant_speed_load.py), PolynomialStaticLoad (polynomial_static_load.py) */
typedef enum { GEMX_LOAD_CONST_SPEED = 0, GEMX_LOAD_POLY_STATIC = 1 } gemx_load_kind;
/* EulerSolver(nsteps) (solvers.py:79-136); classical RK4 (not in the reference); one fixed Dormand-Prince-5
 * step per segment (what the reference's default scipy dopri5 does whenever its trial step is accepted) */
typedef enum { GEMX_SOLVER_EULER = 0, GEMX_SOLVER_RK4 = 1, GEMX_SOLVER_DP5 = 2 } gemx_solver_kind;
/* gemx_config.solver_flags.  GEMX_SOLVER_SPLIT_KINKS: with a PolynomialStaticLoad every (sub-)step is corrected for the kinks of the
 * load torque at |omega| = a * tau_decay / J (polynomial_static_load.py:87-92), where a fixed step loses its order and the reference's
 * default solver (scipy dopri5, solvers.py:139-184) rejects and splits its steps.  ABI >= 5 libraries do it in ONE pass of the scheme: the
 * saturation term is replaced by the affine piece of the region the mid-step omega is predicted in (a smooth system: the scheme keeps
 * its order), and the defect -- the time integral of (true - modelled) saturation along the step's own omega path, a cubic through both
 * ends with the first and last stage's slopes -- is added to omega in closed form.  (Earlier libraries cut the step at the predicted
 * crossings, up to three passes; the flag kept its name.)  Worst observed error against the reference's dopri5 trajectories over every
 * recorded PolynomialStaticLoad run (fp32): 7.9e-5 without, 6.3e-6 with.  Ignored for a ConstantSpeedLoad. */
#define GEMX_SOLVER_SPLIT_KINKS 1
/* GEMX_SOLVER_ADAPTIVE (with GEMX_SOLVER_DP5 only): error-controlled sub-stepping -- what the reference's default solver does
 * (ScipyOdeSolver('dopri5'), solvers.py:139-184: scipy's DOPRI5 with rtol 1e-6, atol 1e-12).  Every integration segment is tried as one
 * Dormand-Prince 5(4) step; where the embedded error estimate, in scipy's norm sqrt(mean((err_i / (solver_atol + solver_rtol
 * max(|y_i|, |y_i new|)))^2)), exceeds 1 the lane cuts its step (h <- h clamp(0.9 err^-1/5, 0.2, 1)) and goes on with steps chosen the same
 * way until the segment is through; the other lanes of its wave wait.  The step size a lane ends a control step with is CARRIED to its next
 * one (a state row, like DOPRI5's WORK(7); cleared by a reset, part of the checkpoint blob).  Still not scipy's step sequence (no PI term,
 * fp32 error estimates), so not its bits -- the same tolerance.  Floor: a step of 1/1024 of the segment is taken whatever its estimate and
 * raises GEMX_ERRFLAG_TOLERANCE.  Takes precedence over solver_nsteps; the one-step map and the small-batch DC kernel are not used.
 * Together with GEMX_SOLVER_SPLIT_KINKS (ABI 7; PolynomialStaticLoad): every attempt integrates the smooth model system of the kink
 * correction above and an accepted sub-step adds the kink's defect to omega in closed form, so that the error estimate never sees the
 * kink.  Without it a lane that crosses |omega| = a tau_decay / J cuts its step five to eight times, and under random actions some lane of
 * a 64-lane wave does in most control steps: 4.7 attempts per control step and wave against 1.8 (profiles/r06_wave_step_statistics.md). */
#define GEMX_SOLVER_ADAPTIVE 2
typedef enum { GEMX_F32 = 0, GEMX_F64 = 1 } gemx_dtype;
/* observation layout: [N, S_out] rows per env (the reference contract) or [S_out, N] */
typedef enum { GEMX_OBS_AOS = 0, GEMX_OBS_SOA = 1 } gemx_obs_layout;
typedef enum { GEMX_INIT_CONST = 0, GEMX_INIT_UNIFORM = 1, GEMX_INIT_GAUSSIAN = 2 } gemx_init_kind;
typedef enum { GEMX_SUPPLY_IDEAL = 0, GEMX_SUPPLY_RC = 1 } gemx_supply_kind;
typedef enum { GEMX_ACT_ABC = 0, GEMX_ACT_DQ_SPACE = 1, GEMX_ACT_DQ_PROCESSOR = 2 } gemx_action_frame;
#define GEMX_MAX_DELAY 8

/* Flat description of ONE SCML system shared by all N envs of a handle (parameters are uniform across envs:
 * they travel as kernel arguments -> SGPRs, 0 bytes of HBM traffic per env). */
typedef struct gemx_config {
    int32_t struct_size; /* = sizeof(gemx_config), ABI check */
    int32_t abi_version; /* = GEMX_ABI_VERSION */
    int32_t system_kind, converter_kind, load_kind;
    int32_t solver_kind, solver_nsteps; /* nsteps: sub-steps per integration segment (EulerSolver(nsteps)); >= 1 */
    int32_t solver_flags;               /* GEMX_SOLVER_* bits, 0 = plain fixed steps */
    int32_t dtype, obs_layout;
    int32_t auto_reset;    /* 1: an env whose step ended `done` restarts from init_state on its next step */
    uint32_t limit_mask;   /* LimitConstraint:   done if any |obs[i]| > 1      over set bits (constraints.py:55-58) */
    uint32_t squared_mask; /* SquaredConstraint: done if sum obs[i]^2 > 1      over set bits (constraints.py:96-98) */
    /* Action path in front of simulate() (continuous B6 converters of SYNC / SCIM / EESM systems only for the dq frames):
     *   GEMX_ACT_ABC          actions are the converter's own (default);
     *   GEMX_ACT_DQ_SPACE     system built with control_space='dq' (physical_systems.py:423-435, 491-492, 777-778):
     *                         A = 2, abc = T32(Q(a_dq, eps)) with the step-start (field) angle; SYNC and SCIM;
     *   GEMX_ACT_DQ_PROCESSOR DqToAbcActionProcessor around the system (physical_system_wrappers/
     *                         dq_to_abc_action_processor.py:100-114; EESM 158-175): A = 2 (SYNC) | 3 (EESM: u_d, u_q, u_e),
     *                         abc = T32(Q(a_dq, eps + (0.5 + action_delay) * tau * omega * p)).
     * action_delay = DeadTimeProcessor(steps) INSIDE the dq processor (dead_time_processor.py:63-85): the converter
     * receives the action submitted action_delay steps earlier; the per-env FIFO is refilled by every reset with
     * action_delay_reset below -- zeros, the reference's default reset_actions, or the ONE action a custom
     * `reset_action` callable returns action_delay copies of (dead_time_processor.py:27-50).  Any system / converter;
     * 0 = none, max GEMX_MAX_DELAY. */
    int32_t action_frame;
    int32_t action_delay;
    /* Supply: GEMX_SUPPLY_IDEAL (IdealVoltageSupply, voltage_supplies.py:60-72: u_sup = u_nominal) or GEMX_SUPPLY_RC
     * (RCVoltageSupply, 75-123): one extra state per env, advanced at the start of every control step by one explicit Euler
     * step of du/dt = (u_nominal - u - supply_r * i_sup) / (supply_r * supply_c) over the time since the previous step, with
     * i_sup = converter.i_sup(i_in) (converters.py:289-298, 429-435, 366-368, 493-495, 837-839, 909-911) evaluated, as the
     * reference does, on the NEW duty cycles of a continuous converter / the PREVIOUS step's final switching state of a
     * finite one.  reset() reloads the capacitor (u = u_nominal).  Not available for the finite EESM converter. */
    int32_t supply_kind;
    /* Initial ODE state (electric_motor.py:150-257, mechanical_load.py:100-160).  GEMX_INIT_CONST: init_state below.
     * GEMX_INIT_UNIFORM / GEMX_INIT_GAUSSIAN: every reset (gemx_reset and the in-kernel auto-reset) draws each ODE state j with
     * init_lo[j] < init_hi[j] anew -- uniformly in [lo, hi], or from a normal(init_mu[j], init_sigma[j]) truncated to [lo, hi]
     * (scipy.stats.truncnorm in the reference) -- from a counter-based stream (Threefry-4x32 with 12 rounds since ABI 7, Philox4x32-10
     * before: Salmon et al., SC'11) keyed by `seed` and indexed by (env_base + env, number of resets of that env, j); states with lo == hi
     * keep init_state[j].  numpy's PCG64 streams of the reference
     * cannot be reproduced on a device: parity is distributional (tests/test_gpu_parity.py), and exact for the state -> reset
     * observation map.  Every system; the load's omega for any system with a PolynomialStaticLoad.
     * Induction machines (SCIM, DFIM; ABI 5): init_flux_mode = 1 re-derives the bounds of the two flux states at EVERY reset as the
     * reference does (induction_motor.py:174-185, 250-285; squirrel_cage_induction_motor.py:146-157; doubly_fed_induction_motor.py:
     * 154-165): a field angle eps_mag ~ U(-pi, pi) is drawn, psi_d_max = init_flux[0] (= l_m * nominal i_sd) if this reset's omega
     * is 0, else 0.9 * clip((p omega sigma l_s i_d + (r_s + r_r l_mr^2) i_q + u_q_max + l_mr u_rq_max) / (-p omega l_mr), 0,
     * |l_m i_d|) with (i_d, i_q) = Q^-1(i_s alpha/beta OF THE PREVIOUS RESET'S DRAW -- the configured constants at the first reset --,
     * eps_mag) (the reference reads the motor's stale `_initial_states` there), and the flux states are drawn from
     * +-psi_d_max * (|cos eps_mag|, |sin eps_mag|) clipped to [init_lo, init_hi] of their slots (the user's `interval`;
     * +-HUGE_VAL = none).  init_flux = [l_m * i_sd_nominal, p, sigma * l_s, r_s + r_r * l_mr^2, u_q_max + l_mr * u_rq_max, l_mr, l_m, 0]. */
    int32_t init_kind;
    int32_t init_flux_mode;
    double init_flux[8];
    uint64_t seed;
    /* GLOBAL index of this handle's env 0 (ABI 7).  Every device-side random stream -- the initialisers above, the synthetic actions of
     * gemx_rollout_synthetic / gemx_synthetic_actions -- is a pure function of (seed, env_base + i, ...): a job sharded over W handles
     * (ranks / GPUs) with env_base = the shard's first env draws, env by env, exactly what ONE handle over all envs draws, as every env
     * object of the reference owns its own branch of the seed sequence (core.py:373-385, physical_systems.py:164-169,
     * random_component.py:60-87).  0 for a lone handle; >= 0. */
    int64_t env_base;
    double init_lo[GEMX_MAX_ODE], init_hi[GEMX_MAX_ODE], init_mu[GEMX_MAX_ODE], init_sigma[GEMX_MAX_ODE];
    double supply_r, supply_c;
    /* DeadTimeProcessor reset action, in the action space of the system the processor wraps: continuous converter actions
     * [0 .. A_conv - 1] (A_conv = 2 for control_space='dq'), or [0] = the flat index of a discrete action */
    double action_delay_reset[6];
    double solver_rtol, solver_atol; /* GEMX_SOLVER_ADAPTIVE: relative / absolute (state units) tolerance; 0 = 1e-6 / 1e-9 */
    /* GEMX_SOLVER_ADAPTIVE, systems whose omega is a state (PolynomialStaticLoad): the absolute tolerance of OMEGA in rad/s (ABI 7).
     * 0 = solver_atol x limits[omega], i.e. solver_atol in NORMALISED units -- 1e-9 of the speed range (4e-7 rad/s), not 1e-9 rad/s: a
     * speed-control episode starts at omega = 0, where the relative term vanishes and an absolute tolerance of 1e-9 rad/s (2e-12 of the
     * range, below anything an observation can show) makes lanes cut steps that no result depends on -- with 64 lock-stepped lanes some lane
     * does so in most control steps (BASELINE config 4: 1.7 attempts per control step and wave instead of 1.0; against scipy's dopri5 runs
     * 4e-6 either way: profiles/r06_wave_step_statistics.md).  solver_atol itself restores the behaviour of ABI <= 6. */
    double solver_atol_omega;
    double tau;               /* control step, PhysicalSystem.tau */
    double interlocking_time; /* converter dead time, converters.py:35-41; must be < tau */
    double u_nominal;         /* IdealVoltageSupply.u_nominal, voltage_supplies.py:60-72 */
    /* motor._model_constants zero-padded to 5x11, row-major; feature order as in the reference:
     * DC    (1x3) [omega, i, u]                                  dc_permanently_excited_motor.py:71-84
     * SERIES(1x3) [i, omega*i, u]                                dc_series_motor.py:68-83
     * SHUNT (2x5) [i_a, i_e, omega*i_e, u_a, u_e] (u_a = u_e = u) dc_motor.py:96-127, dc_shunt_motor.py:72-74
     * EXTEX (2x5) [i_a, i_e, omega*i_e, u_a, u_e]                 dc_motor.py:96-127
     * SYNC  (3x7) [omega, i_d, i_q, u_d, u_q, omega*i_d, omega*i_q]  synchronous_motor.py:143-168
     * EESM  (4x10)[omega, i_d, i_q, i_e, u_d, u_q, u_e, omega*i_d, omega*i_q, omega*i_e]
     *                                                            externally_excited_synchronous_motor.py:69-113
     * SCIM  (5x11)[omega, i_a, i_b, psi_a, psi_b, omega*psi_a, omega*psi_b, u_sa, u_sb, u_ra, u_rb] induction_motor.py:187-217
     *             (SCIM: the u_r columns are ignored -- zero rotor voltage; DFIM: same matrix, u_r columns live) */
    double model[GEMX_MODEL_ROWS * GEMX_MODEL_COLS];
    /* torque: DC T = tc[0]*i ; SYNC T = (tc[0] + tc[1]*i_d)*i_q ; SCIM T = tc[0]*(psi_a*i_b - psi_b*i_a) ;
     * SERIES T = tc[0]*i*i ; SHUNT / EXTEX T = tc[0]*i_a*i_e ; EESM T = (tc[0]*i_e + tc[1]*i_d)*i_q ;
     * DFIM: T as SCIM, plus the rotor-current reconstruction i_r = tc[2]*psi_r - tc[3]*i_s
     *       (tc[2] = 1/l_r, tc[3] = l_m/l_r; calculate_rotor_current, physical_systems.py:931-946) */
    double torque_coef[4];
    double j_total;                    /* load.j_total (j_load + j_rotor), mechanical_load.py:35-41 */
    double load_a, load_b, load_c;     /* PolynomialStaticLoad parameters */
    double tau_decay;                  /* PolynomialStaticLoad.tau_decay (1e-3) */
    double limits[GEMX_MAX_OUT];       /* PhysicalSystem.limits, physical_systems.py:105-113 */
    double init_state[GEMX_MAX_ODE];   /* ODE state after reset: [omega, motor states..., epsilon] */
} gemx_config;

typedef struct gemx_handle gemx_handle;

int gemx_abi_version(void);
int gemx_sizeof_config(void);
const char *gemx_last_error(void);
/* Diagnostics: per-wave cycle counts of the pipelined kernel's last launch (integrator / output / loader waves of workgroups 0 and
 * 37), filled only by a library compiled with -DGEMX_TIMING (all zeros otherwise); read by tools/pipe_timing_probe.py. */
int gemx_debug_read(gemx_handle *h, unsigned long long *out, int n);
/* number of visible HIP devices (0 => the library cannot run; callers must fail loudly) */
int gemx_device_count(void);

/* Create N envs on `device`, all at the reset state.  Replaces SCMLSystem.__init__ (physical_systems.py:54-103). */
int gemx_create(const gemx_config *cfg, int64_t n_envs, int device, gemx_handle **out);
int gemx_destroy(gemx_handle *h);

int gemx_n_envs(const gemx_handle *h, int64_t *n);
int gemx_n_ode(const gemx_handle *h);    /* S_ode: 2 | 3 | 4 | 5 | 6  */
int gemx_n_out(const gemx_handle *h);    /* S_out: 5 | 6 | 7 | 14 | 16 | 24 */
int gemx_n_action(const gemx_handle *h); /* A: 1 | 2 | 3 | 4 | 6 (1 for every discrete converter; dq frames: 2 | 3) */
int gemx_action_itemsize(const gemx_handle *h); /* 1 (uint8 discrete) | sizeof(R) */
/* normalised state returned by reset() for the configured constant initialiser, host doubles [S_out] */
int gemx_reset_observation(const gemx_handle *h, double *obs_host);

/* PhysicalSystem.reset(): envs with mask[i] != 0 (all if mask_dev == NULL) go back to init_state, their step
 * counter to 0; the converter switching state survives, as in the reference (converters.py:45-54).
 * obs_out_dev (optional, layout per config) receives the reset observation for the reset envs. */
int gemx_reset(gemx_handle *h, const uint8_t *mask_dev, void *obs_out_dev, void *stream);
/* The same reset WITHOUT drawing anew: with random initialisers (init_kind != GEMX_INIT_CONST) the masked envs go back to the state
 * their current episode started from (reset counters untouched); with constant initialisers identical to gemx_reset.  gemx_create leaves
 * every env in draw #1 with its counter at 1, so a binding obtains the construction-time observation rows with this call and the first
 * in-kernel auto-reset (or gemx_reset) of an env is draw #2 -- never draw #1 twice. */
int gemx_reset_again(gemx_handle *h, const uint8_t *mask_dev, void *obs_out_dev, void *stream);

/* PhysicalSystem.simulate() for all N envs: one control step.
 *   actions_dev : [N, A] R for continuous converters, [N] uint8 for finite ones (0..7 Finite-B6C, 0..3 Finite-4QC)
 *   obs_out_dev : [N, S_out] R (AoS) or [S_out, N] R (SoA); 16-byte aligned
 *   done_out_dev: [N] uint8 (may be NULL when no constraint is configured)                                    */
int gemx_step(gemx_handle *h, const void *actions_dev, void *obs_out_dev, uint8_t *done_out_dev, void *stream);

/* K fused control steps in ONE launch (ODE state stays in registers between steps).
 *   actions_dev [K, N, A]; obs_out_dev [K, N, S_out] (or [K, S_out, N]); done_out_dev [K, N].
 *   obs_every: 1 = write every step; 0 = write only the last step (obs_out_dev [N,S_out], done_out_dev [N] =
 *   OR over the K steps).                                                                                     */
int gemx_rollout(gemx_handle *h, const void *actions_dev, int32_t K, void *obs_out_dev, uint8_t *done_out_dev,
                 int32_t obs_every, void *stream);

/* Reward fused into the rollout: WeightedSumOfErrors.reward (reward_functions/
 * weighted_sum_of_errors.py:125-129) with the caller's reference tensor,
 *   r = (1 - v) * (bias - sum_i weight[i] * (|s_i - ref_i| / state_length[i]) ** power[i]) + v * violation_reward,
 * v = this step's done flag (ConstraintMonitor violation degree 0 | 1, core.py:348-350); ref_i = refs[.., j] for the n_ref
 * referenced states ref_index[j] (the generator's referenced_states, in ascending state order), 0 for all others.
 * weight / power / state_length are full-length state arrays exactly as the reference's reward function holds them
 * (_reward_weights, _n, _state_length = state_space.high - low). */
#define GEMX_MAX_REF 4
typedef struct gemx_reward_config {
    int32_t struct_size; /* = sizeof(gemx_reward_config) */
    int32_t n_ref;       /* 0..GEMX_MAX_REF columns of the reference tensor */
    int32_t ref_index[GEMX_MAX_REF];
    double weight[GEMX_MAX_OUT], power[GEMX_MAX_OUT], state_length[GEMX_MAX_OUT];
    double bias, violation_reward;
} gemx_reward_config;
/* Install (rc != NULL) or remove (NULL) the reward function of a handle. */
int gemx_set_reward(gemx_handle *h, const gemx_reward_config *rc);
/* SYNTHETIC random-action rollouts without an action tensor ("actions ... can be generated on-device"): the action of env e at
 * stream position t = step0 + k is a pure function of (seed, e, t, component) -- continuous entries uniform on (-1, 1), discrete indices
 * uniform over the converter's action set (the flat index for MultiDiscrete) -- generated inside the launch (fp32, K >= 2, the conditions
 * of the pipelined kernel; GEMX_ERR_ARG otherwise).  gemx_synthetic_actions writes the SAME stream into a tensor ([K, N, A] R | [K, N] uint8),
 * for policies / tests / the paths gemx_rollout_synthetic does not serve: gemx_rollout on that tensor gives the same bits. */
int gemx_rollout_synthetic(gemx_handle *h, uint64_t seed, uint32_t step0, int32_t K, void *obs_out_dev, uint8_t *done_out_dev, void *stream);
int gemx_synthetic_actions(gemx_handle *h, uint64_t seed, uint32_t step0, int32_t K, void *actions_out_dev, void *stream);

/* gemx_rollout with a NARROW action tensor (ABI 7): actions_half_dev [K, N, A] IEEE half (continuous converters, fp32 handles, K >= 2;
 * observations of every step).  A policy's duty cycles need no more than eleven bits; from 32768 envs on the launches of the continuous
 * converters are bound by the write path's tolerance for the action stream read between the row stores, and half the read
 * bytes is the one lever a tensor-fed rollout has.  The values are widened to fp32 while they are staged: the results are, bit for bit, those
 * of gemx_rollout on the same values as fp32 (the reference clips and uses float64 duty cycles: converters.py:144-158, 888-903; the
 * quantisation of a half, 2^-11 relative, is the caller's to accept). */
int gemx_rollout_half(gemx_handle *h, const void *actions_half_dev, int32_t K, void *obs_out_dev, uint8_t *done_out_dev, void *stream);

/* The large-batch RATE LIMITER of the fused rollout (more workgroups than CUs: every workgroup is held to one block of rows per interval,
 * because this chip's write path delivers more when it is offered slightly less than it can take -- DESIGN.md 4.1).  Results never depend
 * on it, only the rate.  mode 0: off; 1: open loop at the built-in target of the launch's family and size; 2: closed loop (the default on a
 * whole gfx950: each handle times its own launches with HIP events and keeps the best of a bracket around that target, or none).
 * target_gbps > 0 replaces the built-in target (algorithmic GB/s, chip-wide; then open loop); <= 0 keeps it.  The environment variables
 * GEMX_PACE_GBPS / GEMX_PACE_CAL set the same two things for every handle at gemx_create; this call overrides them for one handle
 * (e.g. several handles sharing the chip, a partitioned device: mode 0).  Forgets the handle's calibration. */
int gemx_set_rate_limiter(gemx_handle *h, int32_t mode, double target_gbps);

/* gemx_rollout (obs_every = 1) that additionally reads refs_dev [K, N, n_ref] (R) and writes reward_out_dev [K, N] (R). */
int gemx_rollout_reward(gemx_handle *h, const void *actions_dev, int32_t K, const void *refs_dev, void *obs_out_dev,
                        uint8_t *done_out_dev, void *reward_out_dev, void *stream);

/* Device-side reference generation: N x n_ref independent WienerProcessReferenceGenerator streams
 * (reference_generators/wiener_process_reference_generator.py:30-49 over subepisoded_reference_generator.py:66-119, combined as
 * multiple_reference_generator.py:77-92 does): sub-episodes of int(U(episode_len_lo, episode_len_hi)) steps, per sub-episode
 * sigma = 10 ** U(log10 sigma_lo, log10 sigma_hi), value += N(0, sigma) per step clipped to [margin_lo, margin_hi]; a reset draws the
 * initial value from U(initial_lo, initial_hi) and starts a new sub-episode.  Counter-based Philox4x32-10 streams keyed by `seed`:
 * chunked generation == one-shot generation; parity with the reference's numpy streams is distributional.
 *   gemx_refgen_rollout(r, done, K, refs): refs[k, env, j] = reference the reward of control step k is computed against (what the agent
 *   saw as "next reference" before acting); done[k, env] != 0 (optional, the physics rollout's done tensor) resets that env's generators
 *   after step k, as `if terminated: env.reset()` does.  The tensor feeds gemx_rollout_reward directly. */
typedef struct gemx_refgen_config {
    int32_t struct_size; /* = sizeof(gemx_refgen_config) */
    int32_t n_ref;       /* 1..GEMX_MAX_REF sub-generators */
    uint64_t seed;
    int64_t env_base;    /* global index of env 0 (ABI 7; see gemx_config.env_base): streams are keyed by (seed, env_base + i, generator, ...) */
    int32_t episode_len_lo, episode_len_hi; /* episode_lengths, default (500, 2000) */
    double sigma_lo[GEMX_MAX_REF], sigma_hi[GEMX_MAX_REF];     /* sigma_range, default (1e-3, 1e-1) */
    double margin_lo[GEMX_MAX_REF], margin_hi[GEMX_MAX_REF];   /* limit_margin in normalised units */
    double initial_lo[GEMX_MAX_REF], initial_hi[GEMX_MAX_REF]; /* initial_range (default: the limit margin) */
} gemx_refgen_config;
typedef struct gemx_refgen gemx_refgen;
int gemx_refgen_create(const gemx_refgen_config *cfg, int64_t n_envs, int device, int dtype, gemx_refgen **out);
int gemx_refgen_destroy(gemx_refgen *r);
int gemx_refgen_reset(gemx_refgen *r, const uint8_t *mask_dev, void *stream);
int gemx_refgen_rollout(gemx_refgen *r, const uint8_t *done_dev, int32_t K, void *refs_out_dev, void *stream);
int gemx_refgen_get_state(gemx_refgen *r, double *value_out_dev, double *sigma_out_dev, int32_t *left_out_dev, void *stream);

/* Checkpoint / parity access to the ODE state, SoA [S_ode, N] of R in physical units (angle in rad; the fp32 build keeps the
 * angle as a 32-bit fraction of a turn internally, so a get/set round trip rounds it to fp32 radians, ~1e-7 rad), plus the
 * per-env packed converter switching state, 2 bits per half-bridge: [N] uint8, or [2][N] uint8 (row 0 = bits 0..7,
 * row 1 = bits 8..11) for the 6 half-bridges of GEMX_CONV_FINITE_2XB6; gemx_n_switch_bytes() = bytes per env.
 * (There is no per-env step counter: PhysicalSystem.k belongs to the binding; the handle's launched-steps count is in the blob below.)
 * Everything else a resumed handle needs is ONE opaque device blob (gemx_aux_state_bytes() bytes, 16-byte aligned): the
 * RCVoltageSupply's two rows (capacitor voltage, time since its last update; voltage_supplies.py:100-123), the DeadTimeProcessor's
 * action queue and its phase (dead_time_processor.py:63-85), the per-env reset counters of the random initialisers (the counter-based
 * streams continue where they were), the exact angle words (see above: gemx_get_state rounds them) and the launched-steps count.
 * A checkpoint = get_state + get_switch_state + get_aux_state, restored in that order (the blob's angle words overwrite the rounded ones);
 * restored into a fresh handle of the SAME configuration (checked: gemx_set_aux_state fails with GEMX_ERR_ARG otherwise, and
 * synchronises `stream` to read the blob's header) the next launches continue bit for bit. */
int gemx_n_switch_bytes(const gemx_handle *h);
int gemx_get_state(gemx_handle *h, void *soa_out_dev, void *stream);
int gemx_set_state(gemx_handle *h, const void *soa_in_dev, void *stream);
int gemx_get_switch_state(gemx_handle *h, uint8_t *out_dev, void *stream);
int gemx_set_switch_state(gemx_handle *h, const uint8_t *in_dev, void *stream);
int64_t gemx_aux_state_bytes(const gemx_handle *h);
int gemx_get_aux_state(gemx_handle *h, void *blob_out_dev, void *stream);
int gemx_set_aux_state(gemx_handle *h, const void *blob_in_dev, void *stream);

/* Tuning: number of control steps whose observations are staged in LDS between global-memory bursts inside
 * gemx_rollout (0 = heuristic from N, S_out and the LDS size; also settable with env GEMX_STEPS_PER_BLOCK). */
int gemx_set_steps_per_block(gemx_handle *h, int32_t steps);

/* Which kernel instantiation / geometry the most recent gemx_step / gemx_rollout on this handle launched (for
 * profiling reports): e.g. "gemx::advance_pipe_kernel<sys=1,conv=1,load=0,solver=1,il=0,f32,D=8> grid=256 x 192 ...". */
const char *gemx_last_launch(const gemx_handle *h);

/* Sticky device error word (synchronises `stream`), GEMX_ERRFLAG_* bits:
 *   ACTION       a discrete action outside the action space was seen (the reference asserts action_space.contains(action),
 *                converters.py:204-206; the kernel masks it into range);
 *   OMEGA_MOVED  a launch specialised for "omega of every env == its initial value" (dc_stream_kernel, constant-speed loads) found
 *                another omega in device memory -- e.g. enqueued on a different stream than a preceding gemx_set_state: the
 *                observations of that launch are invalid;
 *   TOLERANCE    GEMX_SOLVER_ADAPTIVE took a step at its floor (1/1024 of a segment) whose error estimate exceeded the tolerance. */
#define GEMX_ERRFLAG_ACTION 1u
#define GEMX_ERRFLAG_OMEGA_MOVED 2u
#define GEMX_ERRFLAG_TOLERANCE 4u
int gemx_error_flags(gemx_handle *h, uint32_t *flags_host, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GEMX_H */
