/*
 * gemx_oracle.c -- CPU restatement (plain C, IEEE fp64, one env at a time) of the reference's
 * SCMLSystem.simulate() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call this file.  The product (gym_electric_motor_amd/) never links or imports it and
 * shares no source with it: the HIP kernels are an independent implementation that this file checks.
 *
 * Parity pinning: tests/test_oracle_golden.py checks this restatement against
 *   - the reference's only golden trajectory, tests/integration_tests/ref_data.npz (via the replayed
 *     action sequence in tests/golden/refdata_cont_sc_permexdc_dopri5.npz),
 *   - ~40 trajectories recorded from the live reference by oracle/make_golden.py (Euler / dopri5,
 *     free-run / episodic, with and without interlocking) for the three configured env ids,
 *   - converter known-answer tables produced by the reference converter classes,
 *   - the hand-written KATs in the reference's tests (Euler solver, PolynomialStaticLoad, constraints).
 *
 * All file:line citations are relative to /root/reference/src/gym_electric_motor/.
 *
 * Third-party arithmetic: the reference's default solver is scipy.integrate.ode('dopri5')
 * (physical_systems/solvers.py:139-184; scipy 1.15.3 in this image, requirements.txt pins scipy>=1.4.1).
 * scipy is not part of /root/reference.  ORC_SOLVER_DOPRI5 restates the published algorithm of Hairer's
 * DOPRI5 (Dormand & Prince 1980 tableau; Hairer/Norsett/Wanner "Solving ODEs I", II.4-II.5: step-size
 * control with Lund/PI stabilisation and the HINIT starting step) with scipy's settings for this path:
 * rtol=1e-6, atol=1e-12, safety=0.9, dfactor(fac1)=0.2, ifactor(fac2)=10, beta=0 -> DOPRI5's default 0.04,
 * nsteps=500, max_step=0 -> hmax = t_end - t, first_step=0 -> HINIT.  The predicted step size survives from
 * one integrate() call to the next (DOPRI5 stores it back into WORK(7)) and is cleared by set_initial_value()
 * (= reset).  Usually one accepted step per integrate() call; near the load's kinks steps are rejected and
 * split.  The golden vectors recorded from the live scipy path pin this restatement to ~1e-12.
 * ORC_SOLVER_DP5_FIXED is ONE Dormand-Prince step of size (t_end - t), 5th-order solution, no error control
 * (what the adaptive code does whenever its first trial step is accepted).
 * ORC_SOLVER_IVP_RK45 restates scipy.integrate.solve_ivp(method='RK45') as ScipySolveIvpSolver drives it
 * (physical_systems/solvers.py:187-219: a FRESH solve_ivp call per integrate(), t_eval=[t]): scipy/integrate/_ivp/rk.py
 * RungeKutta._step_impl + common.py select_initial_step (Hairer II.4 starting step, error_estimator_order 4, SAFETY 0.9,
 * MIN_FACTOR 0.2, MAX_FACTOR 10, rms norm, scale = atol + max(|y|, |y_new|) rtol), defaults rtol 1e-3 / atol 1e-6.
 * ORC_SOLVER_RK4_KINK / ORC_SOLVER_DP5_KINK restate what the HIP kernels do for a PolynomialStaticLoad (they have no
 * counterpart in the reference): one fixed step per control step on a smooth extension of the load torque, corrected in closed
 * form for the time omega spends beyond a kink of the load torque (|omega| = omega_lim), see integrate_kink().
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

enum { ORC_SYS_DC_PERMEX = 0, ORC_SYS_PMSM = 1, ORC_SYS_SCIM = 2, ORC_SYS_DC_SERIES = 3, ORC_SYS_DC_SHUNT = 4,
       ORC_SYS_DC_EXTEX = 5, ORC_SYS_EESM = 6, ORC_SYS_DFIM = 7 };
/* 4..9: Cont/FiniteMultiConverter (converters.py:498-740) of exactly two sub-converters:
 * 2 x 4QC (the ExtExDc envs), B6 + 4QC (the EESM envs), 2 x B6 (the DFIM envs: stator, rotor) */
enum { ORC_CONV_CONT_4QC = 0, ORC_CONV_FINITE_B6 = 1, ORC_CONV_CONT_B6 = 2, ORC_CONV_FINITE_4QC = 3,
       ORC_CONV_CONT_2X4QC = 4, ORC_CONV_FINITE_2X4QC = 5, ORC_CONV_CONT_B6_4QC = 6, ORC_CONV_FINITE_B6_4QC = 7,
       ORC_CONV_CONT_2XB6 = 8, ORC_CONV_FINITE_2XB6 = 9 };
#define ORC_IS_DC(s) ((s) == ORC_SYS_DC_PERMEX || (s) == ORC_SYS_DC_SERIES || (s) == ORC_SYS_DC_SHUNT || (s) == ORC_SYS_DC_EXTEX)
enum { ORC_LOAD_CONST_SPEED = 0, ORC_LOAD_POLY_STATIC = 1 };
enum { ORC_SOLVER_EULER = 0, ORC_SOLVER_RK4 = 1, ORC_SOLVER_DOPRI5 = 2, ORC_SOLVER_DP5_FIXED = 3, ORC_SOLVER_IVP_RK45 = 4,
       ORC_SOLVER_RK4_KINK = 5, ORC_SOLVER_DP5_KINK = 6, ORC_SOLVER_DEV_ADAPTIVE = 7, ORC_SOLVER_DEV_ADAPTIVE_KINK = 8 };

#define ORC_MAX_ODE 8
#define ORC_MAX_OUT 24

typedef struct orc_params {
    int32_t system, converter, load, solver, nsteps;
    int32_t limit_mask;   /* bit i set: LimitConstraint observes system-state entry i   (constraints.py:55-58) */
    int32_t squared_mask; /* bit i set: SquaredConstraint sums entry i                   (constraints.py:96-98) */
    /* action path in front of simulate():
     * dq_mode 1: SynchronousMotorSystem / SquirrelCage...System(control_space='dq') (physical_systems.py:491-492, 777-778)
     * dq_mode 2: DqToAbcActionProcessor around the system (physical_system_wrappers/dq_to_abc_action_processor.py:100-114,
     *            EESM variant 158-175): abc = T32(Q(a_dq, eps + (0.5 + act_delay) * tau * omega * p))
     * act_delay: DeadTimeProcessor(steps) INSIDE the dq processor (dead_time_processor.py:63-85), reset action = zeros */
    int32_t dq_mode;
    int32_t act_delay;
    int32_t rc_supply; /* 1: RCVoltageSupply (voltage_supplies.py:75-123) with sup_r, sup_c below; u_sup = u_0 */
    double tau, t_il, u_sup;
    double mp[8]; /* DC permex: r_a,l_a,psi_e | PMSM/SynRM: p,l_d,l_q,r_s,psi_p(0 for SynRM) | SCIM: p,l_m,l_sigs,l_sigr,r_s,r_r
                   * DC series / shunt / extex: r_a,r_e,l_a,l_e,l_e_prime | EESM: p,l_d,l_q,l_m,l_e,r_s,r_e,k */
    double j_total, load_a, load_b, load_c, tau_decay;
    double limits[ORC_MAX_OUT];
    double init[ORC_MAX_ODE]; /* initial ODE state [omega, motor states...] */
    double sup_r, sup_c;      /* RCVoltageSupply supply_parameter R, C */
    double rtol, atol;        /* ORC_SOLVER_IVP_RK45: solve_ivp tolerances (0 -> scipy's defaults 1e-3 / 1e-6) */
    double act_delay_reset[6]; /* DeadTimeProcessor(reset_action=...) returning `steps` copies of ONE action (dead_time_processor.py:27-50): that
                               * action in the inner system's action space (a discrete index as a double); zeros = the reference's default */
} orc_params;

typedef struct orc_env {
    double y[ORC_MAX_ODE];
    double t;           /* SCMLSystem._t == solver.t */
    int32_t k;          /* PhysicalSystem._k */
    /* converter state (converters.py:40-43, 193-197) */
    double action_start;
    double duty[6][2];        /* Cont: per sub-converter clipped duty (ContDynamicallyAveragedConverter.set_action:144-146) */
    int32_t sw_state[6];      /* FiniteTwoQuadrantConverter._switching_state, NOT cleared by reset() (45-54) */
    int32_t sw_pattern[6][2]; /* _switching_pattern */
    int32_t sw_plen[6];
    /* solver f_params */
    double u[4];
    double dp_h; /* dopri5: predicted step size carried between integrate() calls (0 -> HINIT) */
    /* constants */
    double C[5][11];
    /* wrappers */
    /* RCVoltageSupply: its own EulerSolver state and time (voltage_supplies.py:96-123) */
    double sup_u, sup_t;
    double last_state[ORC_MAX_OUT]; /* DqToAbcActionProcessor._state = normalised state * limits (lines 96, 112) */
    double fifo[8][6];              /* DeadTimeProcessor._action_deque (oldest first) */
} orc_env;

static int n_ode(const orc_params *p) {
    switch (p->system) {
        case ORC_SYS_DC_PERMEX: case ORC_SYS_DC_SERIES: return 2;
        case ORC_SYS_DC_SHUNT: case ORC_SYS_DC_EXTEX: return 3;
        case ORC_SYS_PMSM: return 4;
        case ORC_SYS_EESM: return 5; /* [omega, i_sd, i_sq, i_e, epsilon] */
        default: return 6;
    }
}
/* state names: DcMotorSystem._build_state_names (physical_systems.py:295-303): [omega, torque] + CURRENTS + VOLTAGES + [u_sup] */
static int n_out(const orc_params *p) {
    if (p->system == ORC_SYS_DC_EXTEX) return 7;  /* [omega, torque, i_a, i_e, u_a, u_e, u_sup] */
    if (p->system == ORC_SYS_EESM) return 16;     /* physical_systems.py:575-593 */
    if (p->system == ORC_SYS_DFIM) return 24;     /* physical_systems.py:882-909 */
    return p->system == ORC_SYS_DC_SHUNT ? 6 : (ORC_IS_DC(p->system) ? 5 : 14);
}

int orc_n_ode(const orc_params *p) { return n_ode(p); }
int orc_n_out(const orc_params *p) { return n_out(p); }

/* ---------------------------------------------------------------- model constants -------------- */
/* dc_permanently_excited_motor.py:71-75; permanent_magnet_synchronous_motor.py:107-119;
 * induction_motor.py:287-312 */
void orc_model_constants(const orc_params *p, double C[5][11]) {
    memset(C, 0, sizeof(double) * 55);
    const double *mp = p->mp;
    if (p->system == ORC_SYS_DC_PERMEX) {
        double r_a = mp[0], l_a = mp[1], psi_e = mp[2];
        C[0][0] = -psi_e / l_a; C[0][1] = -r_a / l_a; C[0][2] = 1.0 / l_a;
    } else if (p->system == ORC_SYS_DC_SERIES) { /* dc_series_motor.py:68-72, features [i, omega*i, u] */
        double r_a = mp[0], r_e = mp[1], l_a = mp[2], l_e = mp[3], lep = mp[4];
        C[0][0] = (-r_a - r_e) / (l_a + l_e); C[0][1] = -lep / (l_a + l_e); C[0][2] = 1.0 / (l_a + l_e);
    } else if (p->system == ORC_SYS_DC_SHUNT || p->system == ORC_SYS_DC_EXTEX) { /* dc_motor.py:96-104, features [i_a, i_e, omega*i_e, u_a, u_e] */
        double r_a = mp[0], r_e = mp[1], l_a = mp[2], l_e = mp[3], lep = mp[4];
        C[0][0] = -r_a / l_a; C[0][2] = -lep / l_a; C[0][3] = 1.0 / l_a;
        C[1][1] = -r_e / l_e; C[1][4] = 1.0 / l_e;
    } else if (p->system == ORC_SYS_EESM) { /* externally_excited_synchronous_motor.py:69-93 */
        double pp = mp[0], l_d = mp[1], l_q = mp[2], l_m = mp[3], l_e = mp[4], r_s = mp[5], r_e = mp[6], k = mp[7];
        double r_E = k * k * 3 / 2 * r_e, l_M = k * 3 / 2 * l_m, l_E = k * k * 3 / 2 * l_e, ik = 2.0 / 3 / k;
        double sigma = 1 - l_M * l_M / (l_d * l_E);
        /* omega, i_d, i_q, i_e, u_d, u_q, u_e, omega*i_d, omega*i_q, omega*i_e */
        double M[4][10] = {
            {0, -r_s / sigma, 0, l_M * r_E / (sigma * l_E) * ik, 1 / sigma, 0, -l_M * k / (sigma * l_E), 0, l_q * pp / sigma, 0},
            {0, 0, -r_s, 0, 0, 1, 0, -l_d * pp, 0, -pp * l_M * ik},
            {0, l_M * r_s / (sigma * l_d), 0, -r_E / sigma * ik, -l_M / (sigma * l_d), 0, k / sigma, 0,
             -pp * l_M * l_q / (sigma * l_d), 0},
            {pp, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
        for (int j = 0; j < 10; ++j) {
            C[0][j] = M[0][j] / l_d; C[1][j] = M[1][j] / l_q; C[2][j] = M[2][j] / l_E / ik; C[3][j] = M[3][j];
        }
    } else if (p->system == ORC_SYS_PMSM) {
        double pp = mp[0], l_d = mp[1], l_q = mp[2], r_s = mp[3], psi_p = mp[4];
        /*            omega,         i_d,   i_q, u_d, u_q, omega*i_d,  omega*i_q */
        double M[3][7] = {{0, -r_s, 0, 1, 0, 0, l_q * pp},
                          {-psi_p * pp, 0, -r_s, 0, 1, -l_d * pp, 0},
                          {pp, 0, 0, 0, 0, 0, 0}};
        for (int j = 0; j < 7; ++j) { C[0][j] = M[0][j] / l_d; C[1][j] = M[1][j] / l_q; C[2][j] = M[2][j]; }
    } else {
        double pp = mp[0], l_m = mp[1], l_sigs = mp[2], l_sigr = mp[3], r_s = mp[4], r_r = mp[5];
        double l_s = l_m + l_sigs, l_r = l_m + l_sigr;
        double sigma = (l_s * l_r - l_m * l_m) / (l_s * l_r);
        double tau_r = l_r / r_r;
        double tau_sig = sigma * l_s / (r_s + r_r * (l_m * l_m) / (l_r * l_r));
        /* omega, i_a, i_b, psi_a, psi_b, omega*psi_a, omega*psi_b, u_sa, u_sb, u_ra, u_rb */
        C[0][1] = -1 / tau_sig; C[0][3] = l_m * r_r / (sigma * l_s * l_r * l_r);
        C[0][6] = l_m * pp / (sigma * l_r * l_s); C[0][7] = 1 / (sigma * l_s); C[0][9] = -l_m / (sigma * l_r * l_s);
        C[1][2] = -1 / tau_sig; C[1][4] = l_m * r_r / (sigma * l_s * l_r * l_r);
        C[1][5] = -l_m * pp / (sigma * l_r * l_s); C[1][8] = 1 / (sigma * l_s); C[1][10] = -l_m / (sigma * l_r * l_s);
        C[2][1] = l_m / tau_r; C[2][3] = -1 / tau_r; C[2][6] = -pp; C[2][9] = 1;
        C[3][2] = l_m / tau_r; C[3][4] = -1 / tau_r; C[3][5] = pp; C[3][10] = 1;
        C[4][0] = pp;
    }
}

/* ---------------------------------------------------------------- motor ------------------------ */
/* torque: dc_permanently_excited_motor.py:67-69; permanent_magnet_synchronous_motor.py:134-139;
 * induction_motor.py:236-248.  `ms` = motor part of the ODE state. */
static double motor_torque(const orc_params *p, const double *ms) {
    const double *mp = p->mp;
    if (p->system == ORC_SYS_DC_PERMEX) return mp[2] * ms[0];
    if (p->system == ORC_SYS_DC_SERIES) return mp[4] * ms[0] * ms[0]; /* dc_series_motor.py:74-76 -> dc_motor.py:106-108 */
    if (p->system == ORC_SYS_DC_SHUNT || p->system == ORC_SYS_DC_EXTEX) return mp[4] * ms[0] * ms[1]; /* dc_motor.py:106-108 */
    if (p->system == ORC_SYS_EESM) { /* externally_excited_synchronous_motor.py:133-136 */
        double l_M = mp[7] * 3 / 2 * mp[3], ik = 2.0 / 3 / mp[7];
        return 1.5 * mp[0] * (l_M * ms[2] * ik + (mp[1] - mp[2]) * ms[0]) * ms[1];
    }
    if (p->system == ORC_SYS_PMSM) return 1.5 * mp[0] * (mp[4] + (mp[1] - mp[2]) * ms[0]) * ms[1];
    return 1.5 * mp[0] * mp[1] / (mp[1] + mp[3]) * (ms[2] * ms[1] - ms[3] * ms[0]);
}

/* electrical_ode = constant matrix x feature vector: dc_permanently_excited_motor.py:81-84;
 * synchronous_motor.py:143-168; induction_motor.py:187-217 + squirrel_cage_induction_motor.py:121-129 */
static void electrical_ode(const orc_params *p, const orc_env *e, const double *ms, const double *u, double omega,
                           double *out) {
    double f[11];
    int nf, nr;
    if (p->system == ORC_SYS_DC_PERMEX) {
        f[0] = omega; f[1] = ms[0]; f[2] = u[0]; nf = 3; nr = 1;
    } else if (p->system == ORC_SYS_DC_SERIES) { /* dc_series_motor.py:78-83 */
        f[0] = ms[0]; f[1] = omega * ms[0]; f[2] = u[0]; nf = 3; nr = 1;
    } else if (p->system == ORC_SYS_DC_SHUNT) { /* dc_shunt_motor.py:72-74: u_a = u_e = u; dc_motor.py:114-127 */
        f[0] = ms[0]; f[1] = ms[1]; f[2] = omega * ms[1]; f[3] = u[0]; f[4] = u[0]; nf = 5; nr = 2;
    } else if (p->system == ORC_SYS_DC_EXTEX) { /* dc_motor.py:114-127 with separate u_a, u_e */
        f[0] = ms[0]; f[1] = ms[1]; f[2] = omega * ms[1]; f[3] = u[0]; f[4] = u[1]; nf = 5; nr = 2;
    } else if (p->system == ORC_SYS_EESM) { /* externally_excited_synchronous_motor.py:95-113 */
        f[0] = omega; f[1] = ms[0]; f[2] = ms[1]; f[3] = ms[2]; f[4] = u[0]; f[5] = u[1]; f[6] = u[2];
        f[7] = omega * ms[0]; f[8] = omega * ms[1]; f[9] = omega * ms[2]; nf = 10; nr = 4;
    } else if (p->system == ORC_SYS_PMSM) {
        f[0] = omega; f[1] = ms[0]; f[2] = ms[1]; f[3] = u[0]; f[4] = u[1]; f[5] = omega * ms[0]; f[6] = omega * ms[1];
        nf = 7; nr = 3;
    } else {
        f[0] = omega; f[1] = ms[0]; f[2] = ms[1]; f[3] = ms[2]; f[4] = ms[3]; f[5] = omega * ms[2]; f[6] = omega * ms[3];
        f[7] = u[0]; f[8] = u[1];
        if (p->system == ORC_SYS_DFIM) { f[9] = u[2]; f[10] = u[3]; } /* doubly_fed_induction_motor.py: u_sr_alphabeta */
        else { f[9] = 0.0; f[10] = 0.0; }                              /* SCIM: zero rotor voltage */
        nf = 11; nr = 5;
    }
    for (int r = 0; r < nr; ++r) {
        double acc = 0.0;
        for (int j = 0; j < nf; ++j) acc += e->C[r][j] * f[j];
        out[r] = acc;
    }
}

/* ---------------------------------------------------------------- load ------------------------- */
/* constant_speed_load.py:40-42 ; polynomial_static_load.py:62-66, 87-99 */
static double mechanical_ode(const orc_params *p, double omega, double torque) {
    if (p->load == ORC_LOAD_CONST_SPEED) return 0.0;
    double omega_linear_factor = p->j_total / p->tau_decay;
    double omega_lim = p->load_a / p->j_total * p->tau_decay;
    double sign = omega > 0 ? 1.0 : (omega < -0.0 ? -1.0 : 0.0);
    double a = fabs(omega) > omega_lim ? sign * p->load_a : omega_linear_factor * omega;
    double static_torque = sign * p->load_c * omega * omega + p->load_b * omega + a;
    return (torque - static_torque) / p->j_total;
}

/* SCMLSystem._system_equation, physical_systems.py:205-236: [load derivative, motor derivative] */
static void system_equation(const orc_params *p, const orc_env *e, const double *y, double *dy) {
    const double *ms = y + 1;
    dy[0] = mechanical_ode(p, y[0], motor_torque(p, ms));
    electrical_ode(p, e, ms, e->u, y[0], dy + 1);
}

/* ---------------------------------------------------------------- solvers ---------------------- */

/* One Dormand-Prince stage sweep from (y, k1) with step h: fills y1 (5th order), k2 := f(y1) (FSAL),
 * err vector in k4 (scaled by h), ysti unused.  Coefficients: Hairer CDOPRI. */
static void dp5_stages(const orc_params *p, const orc_env *e, int n, const double *y, double h, double *k1, double *k2,
                       double *k3, double *k4, double *k5, double *k6, double *y1) {
    double yt[ORC_MAX_ODE];
    for (int i = 0; i < n; ++i) yt[i] = y[i] + h * (1.0 / 5.0) * k1[i];
    system_equation(p, e, yt, k2);
    for (int i = 0; i < n; ++i) yt[i] = y[i] + h * (3.0 / 40.0 * k1[i] + 9.0 / 40.0 * k2[i]);
    system_equation(p, e, yt, k3);
    for (int i = 0; i < n; ++i) yt[i] = y[i] + h * (44.0 / 45.0 * k1[i] - 56.0 / 15.0 * k2[i] + 32.0 / 9.0 * k3[i]);
    system_equation(p, e, yt, k4);
    for (int i = 0; i < n; ++i)
        yt[i] = y[i] + h * (19372.0 / 6561.0 * k1[i] - 25360.0 / 2187.0 * k2[i] + 64448.0 / 6561.0 * k3[i] -
                            212.0 / 729.0 * k4[i]);
    system_equation(p, e, yt, k5);
    for (int i = 0; i < n; ++i)
        yt[i] = y[i] + h * (9017.0 / 3168.0 * k1[i] - 355.0 / 33.0 * k2[i] + 46732.0 / 5247.0 * k3[i] +
                            49.0 / 176.0 * k4[i] - 5103.0 / 18656.0 * k5[i]);
    system_equation(p, e, yt, k6);
    for (int i = 0; i < n; ++i)
        y1[i] = y[i] + h * (35.0 / 384.0 * k1[i] + 500.0 / 1113.0 * k3[i] + 125.0 / 192.0 * k4[i] -
                            2187.0 / 6784.0 * k5[i] + 11.0 / 84.0 * k6[i]);
    system_equation(p, e, y1, k2);
    for (int i = 0; i < n; ++i)
        k4[i] = (71.0 / 57600.0 * k1[i] - 71.0 / 16695.0 * k3[i] + 71.0 / 1920.0 * k4[i] - 17253.0 / 339200.0 * k5[i] +
                 22.0 / 525.0 * k6[i] - 1.0 / 40.0 * k2[i]) * h;
}

/* scipy.integrate.ode('dopri5').integrate(t_end) as used by ScipyOdeSolver (solvers.py:139-184). */
static __thread long g_dp5_attempts = 0; /* diagnostic: Dormand-Prince steps attempted (accepted + rejected), orc_dp5_attempts() */
long orc_dp5_attempts(void) { return g_dp5_attempts; }
static void dopri5_adaptive(const orc_params *p, orc_env *e, double t_end) {
    const double RTOL = 1e-6, ATOL = 1e-12, SAFE = 0.9, FAC1 = 0.2, FAC2 = 10.0, BETA = 0.04, UROUND = 2.3e-16;
    const int NMAX = 500;
    int n = n_ode(p);
    double k1[ORC_MAX_ODE], k2[ORC_MAX_ODE], k3[ORC_MAX_ODE], k4[ORC_MAX_ODE], k5[ORC_MAX_ODE], k6[ORC_MAX_ODE],
        y1[ORC_MAX_ODE];
    double x = e->t, xend = t_end;
    if (xend == x) return;
    double posneg = xend >= x ? 1.0 : -1.0;
    double facold = 1e-4, expo1 = 0.2 - BETA * 0.75, facc1 = 1.0 / FAC1, facc2 = 1.0 / FAC2;
    double hmax = fabs(xend - x);
    double h = e->dp_h;
    int last = 0, reject = 0, nstep = 0;
    system_equation(p, e, e->y, k1);
    if (h == 0.0) { /* HINIT */
        double dnf = 0.0, dny = 0.0;
        for (int i = 0; i < n; ++i) {
            double sk = ATOL + RTOL * fabs(e->y[i]);
            dnf += (k1[i] / sk) * (k1[i] / sk);
            dny += (e->y[i] / sk) * (e->y[i] / sk);
        }
        h = (dnf <= 1e-10 || dny <= 1e-10) ? 1e-6 : sqrt(dny / dnf) * 0.01;
        h = fmin(h, hmax) * posneg;
        for (int i = 0; i < n; ++i) y1[i] = e->y[i] + h * k1[i];
        system_equation(p, e, y1, k2);
        double der2 = 0.0;
        for (int i = 0; i < n; ++i) {
            double sk = ATOL + RTOL * fabs(e->y[i]);
            der2 += ((k2[i] - k1[i]) / sk) * ((k2[i] - k1[i]) / sk);
        }
        der2 = sqrt(der2) / h;
        double der12 = fmax(fabs(der2), sqrt(dnf));
        double h1 = der12 <= 1e-15 ? fmax(1e-6, fabs(h) * 1e-3) : pow(0.01 / der12, 1.0 / 5.0);
        h = fmin(fmin(100.0 * fabs(h), h1), hmax) * posneg;
    }
    for (;;) {
        if (nstep > NMAX) break;                       /* scipy would warn "larger nsteps is needed" */
        if (0.1 * fabs(h) <= fabs(x) * UROUND) break;  /* step size too small */
        if ((x + 1.01 * h - xend) * posneg > 0.0) { h = xend - x; last = 1; }
        nstep++;
        g_dp5_attempts++;
        dp5_stages(p, e, n, e->y, h, k1, k2, k3, k4, k5, k6, y1);
        double err = 0.0;
        for (int i = 0; i < n; ++i) {
            double sk = ATOL + RTOL * fmax(fabs(e->y[i]), fabs(y1[i]));
            err += (k4[i] / sk) * (k4[i] / sk);
        }
        err = sqrt(err / n);
        double fac11 = pow(err, expo1);
        double fac = fac11 / pow(facold, BETA);
        fac = fmax(facc2, fmin(facc1, fac / SAFE));
        double hnew = h / fac;
        if (err <= 1.0) {
            facold = fmax(err, 1e-4);
            for (int i = 0; i < n; ++i) { k1[i] = k2[i]; e->y[i] = y1[i]; }
            x = x + h;
            if (last) { h = hnew; break; }
            if (fabs(hnew) > hmax) hnew = posneg * hmax;
            if (reject) hnew = posneg * fmin(fabs(hnew), fabs(h));
            reject = 0;
        } else {
            hnew = h / fmin(facc1, fac11 / SAFE);
            reject = 1;
            last = 0;
        }
        h = hnew;
    }
    e->dp_h = h;
    e->t = last ? t_end : x;
}

/* scipy.integrate.solve_ivp(fun, [t, t_end], y, t_eval=[t_end], method='RK45', rtol, atol) as ScipySolveIvpSolver.integrate
 * calls it (solvers.py:207-219): a new RK45 object per call -> select_initial_step every control step. */
static double rms_scaled(int n, const double *x, const double *scale) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (x[i] / scale[i]) * (x[i] / scale[i]);
    return sqrt(s) / sqrt((double)n);
}
static void ivp_rk45(const orc_params *p, orc_env *e, double t_end) {
    const double rtol = p->rtol > 0 ? p->rtol : 1e-3, atol = p->atol > 0 ? p->atol : 1e-6;
    const double SAFETY = 0.9, MIN_FACTOR = 0.2, MAX_FACTOR = 10.0;
    int n = n_ode(p);
    double t = e->t;
    /* REFERENCE QUIRK (reproduced): SCMLSystem._system_equation returns the SAME pre-allocated array on every call
     * (physical_systems.py:219-236, `_system_eq_placeholder`), and scipy's RK45 keeps `self.f = fun(t, y)` WITHOUT copying.
     * `self.f` therefore always reads as the MOST RECENT right-hand-side evaluation: select_initial_step's probe f(y0 + h0 f0)
     * replaces f(y0) before the first step uses it as stage 1 (and makes d2 = |f1 - f0| = 0), and a rejected attempt leaves
     * f(t + h, y_new_rejected) behind as stage 1 of the retry.  `last` is that aliased buffer. */
    double last[ORC_MAX_ODE], k1[ORC_MAX_ODE], k2[ORC_MAX_ODE], k3[ORC_MAX_ODE], k4[ORC_MAX_ODE], k5[ORC_MAX_ODE], k6[ORC_MAX_ODE];
    double y1[ORC_MAX_ODE], yt[ORC_MAX_ODE], scale[ORC_MAX_ODE], tmp[ORC_MAX_ODE];
    double *y = e->y;
    if (t_end == t) return;
    system_equation(p, e, y, last); /* RK45.__init__: self.f = self.fun(self.t, self.y) */
    /* common.py select_initial_step (order = error_estimator_order = 4, max_step = inf, direction = +1) */
    double h_abs;
    {
        double interval = fabs(t_end - t);
        for (int i = 0; i < n; ++i) scale[i] = atol + fabs(y[i]) * rtol;
        double d0 = rms_scaled(n, y, scale), d1 = rms_scaled(n, last, scale);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = fmin(h0, interval);
        for (int i = 0; i < n; ++i) y1[i] = y[i] + h0 * last[i];
        system_equation(p, e, y1, last); /* f1 -- and, through the alias, f0 */
        double d2 = 0.0;                 /* norm((f1 - f0) / scale) / h0 with f0 aliasing f1 */
        double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
        h_abs = fmin(fmin(100.0 * h0, h1), interval);
    }
    while (t != t_end) { /* solve_ivp's loop: solver.step() until t == t_bound */
        double min_step = 10.0 * fabs(nextafter(t, INFINITY) - t);
        if (h_abs < min_step) h_abs = min_step;
        int accepted = 0, rejected = 0;
        double t_new = t, h = 0.0;
        while (!accepted) {
            if (h_abs < min_step) return; /* TOO_SMALL_STEP */
            h = h_abs;
            t_new = t + h;
            if (t_new - t_end > 0) t_new = t_end;
            h = t_new - t;
            h_abs = fabs(h);
            /* rk.py rk_step with the Dormand-Prince tableau of RK45; K[0] = f copies whatever the aliased buffer holds NOW */
            for (int i = 0; i < n; ++i) k1[i] = last[i];
            for (int i = 0; i < n; ++i) yt[i] = y[i] + h * (1.0 / 5.0 * k1[i]);
            system_equation(p, e, yt, k2);
            for (int i = 0; i < n; ++i) yt[i] = y[i] + h * (3.0 / 40.0 * k1[i] + 9.0 / 40.0 * k2[i]);
            system_equation(p, e, yt, k3);
            for (int i = 0; i < n; ++i) yt[i] = y[i] + h * (44.0 / 45.0 * k1[i] - 56.0 / 15.0 * k2[i] + 32.0 / 9.0 * k3[i]);
            system_equation(p, e, yt, k4);
            for (int i = 0; i < n; ++i)
                yt[i] = y[i] + h * (19372.0 / 6561.0 * k1[i] - 25360.0 / 2187.0 * k2[i] + 64448.0 / 6561.0 * k3[i] - 212.0 / 729.0 * k4[i]);
            system_equation(p, e, yt, k5);
            for (int i = 0; i < n; ++i)
                yt[i] = y[i] + h * (9017.0 / 3168.0 * k1[i] - 355.0 / 33.0 * k2[i] + 46732.0 / 5247.0 * k3[i] + 49.0 / 176.0 * k4[i] -
                                    5103.0 / 18656.0 * k5[i]);
            system_equation(p, e, yt, k6);
            for (int i = 0; i < n; ++i)
                y1[i] = y[i] + h * (35.0 / 384.0 * k1[i] + 500.0 / 1113.0 * k3[i] + 125.0 / 192.0 * k4[i] - 2187.0 / 6784.0 * k5[i] +
                                    11.0 / 84.0 * k6[i]);
            system_equation(p, e, y1, last); /* f_new = K[6] */
            for (int i = 0; i < n; ++i) {
                scale[i] = atol + fmax(fabs(y[i]), fabs(y1[i])) * rtol;
                tmp[i] = h * (-71.0 / 57600.0 * k1[i] + 71.0 / 16695.0 * k3[i] - 71.0 / 1920.0 * k4[i] + 17253.0 / 339200.0 * k5[i] -
                              22.0 / 525.0 * k6[i] + 1.0 / 40.0 * last[i]);
            }
            double err = rms_scaled(n, tmp, scale);
            if (err < 1.0) {
                double factor = err == 0.0 ? MAX_FACTOR : fmin(MAX_FACTOR, SAFETY * pow(err, -0.2));
                if (rejected) factor = fmin(1.0, factor);
                h_abs *= factor;
                accepted = 1;
            } else {
                h_abs *= fmax(MIN_FACTOR, SAFETY * pow(err, -0.2));
                rejected = 1;
            }
        }
        t = t_new;
        for (int i = 0; i < n; ++i) y[i] = y1[i];
    }
    e->t = t_end;
}

/* What the HIP kernels do for a PolynomialStaticLoad under GEMX_SOLVER_SPLIT_KINKS (gemx_kernels.hpp integrate<>, KinkPath): the load
 * torque's constant term is the saturation sigma(omega) = clamp(J / tau_decay * omega, -a, a) (polynomial_static_load.py:87-92), i.e. the
 * right-hand side has kinks at |omega| = omega_lim, where a fixed step loses its order (scipy's adaptive solvers split their steps
 * there).  ONE step of the scheme per (sub-)step, on a smooth system: sigma replaced by the affine piece c0 + c1 omega of the region the
 * Euler-predicted mid-step omega lies in; then the defect D = omega_true - omega_model, D' = -(1 / tau_decay) [clamp(omega) - phi_m(omega)],
 * integrated to first order along the model's omega path (cubic Hermite through both ends with the slopes of the scheme's first and
 * last stage) in closed form and
 * added to omega.  Same operations in the same order as the kernels (their fp64 build agrees to 1e-9: tests). */
static void system_equation_m(const orc_params *p, const orc_env *e, const double *y, double *dy, int model, double c0, double c1) {
    if (!model) { system_equation(p, e, y, dy); return; }
    const double w = y[0];
    const double tl = p->load_c * (w * fabs(w)) + p->load_b * w + (c1 * w + c0);
    dy[0] = (motor_torque(p, y + 1) - tl) * (1.0 / p->j_total);
    electrical_ode(p, e, y + 1, e->u, w, dy + 1);
}
/* one step of the scheme for the model system, first stage k1 given */
static __thread double g_last0; /* d omega / dt of the scheme's last stage (at t + h): the end slope of omega's path */
static void fixed_step_m(const orc_params *p, orc_env *e, int dp5, double h, const double *k1, int model, double c0, double c1) {
    int n = n_ode(p);
    double k2[ORC_MAX_ODE], k3[ORC_MAX_ODE], k4[ORC_MAX_ODE], k5[ORC_MAX_ODE], k6[ORC_MAX_ODE], yt[ORC_MAX_ODE];
    if (!dp5) {
        for (int i = 0; i < n; ++i) yt[i] = e->y[i] + 0.5 * h * k1[i];
        system_equation_m(p, e, yt, k2, model, c0, c1);
        for (int i = 0; i < n; ++i) yt[i] = e->y[i] + 0.5 * h * k2[i];
        system_equation_m(p, e, yt, k3, model, c0, c1);
        for (int i = 0; i < n; ++i) yt[i] = e->y[i] + h * k3[i];
        system_equation_m(p, e, yt, k4, model, c0, c1);
        for (int i = 0; i < n; ++i) e->y[i] = e->y[i] + h / 6.0 * (k1[i] + 2.0 * (k2[i] + k3[i]) + k4[i]);
        g_last0 = k4[0];
        return;
    }
    for (int i = 0; i < n; ++i) yt[i] = e->y[i] + h * (1.0 / 5.0) * k1[i];
    system_equation_m(p, e, yt, k2, model, c0, c1);
    for (int i = 0; i < n; ++i) yt[i] = e->y[i] + h * (3.0 / 40.0 * k1[i] + 9.0 / 40.0 * k2[i]);
    system_equation_m(p, e, yt, k3, model, c0, c1);
    for (int i = 0; i < n; ++i) yt[i] = e->y[i] + h * (44.0 / 45.0 * k1[i] - 56.0 / 15.0 * k2[i] + 32.0 / 9.0 * k3[i]);
    system_equation_m(p, e, yt, k4, model, c0, c1);
    for (int i = 0; i < n; ++i)
        yt[i] = e->y[i] + h * (19372.0 / 6561.0 * k1[i] - 25360.0 / 2187.0 * k2[i] + 64448.0 / 6561.0 * k3[i] - 212.0 / 729.0 * k4[i]);
    system_equation_m(p, e, yt, k5, model, c0, c1);
    for (int i = 0; i < n; ++i)
        yt[i] = e->y[i] + h * (9017.0 / 3168.0 * k1[i] - 355.0 / 33.0 * k2[i] + 46732.0 / 5247.0 * k3[i] + 49.0 / 176.0 * k4[i] -
                               5103.0 / 18656.0 * k5[i]);
    system_equation_m(p, e, yt, k6, model, c0, c1);
    g_last0 = k6[0];
    for (int i = 0; i < n; ++i)
        e->y[i] = e->y[i] + h * (35.0 / 384.0 * k1[i] + 500.0 / 1113.0 * k3[i] + 125.0 / 192.0 * k4[i] - 2187.0 / 6784.0 * k5[i] +
                                 11.0 / 84.0 * k6[i]);
}
/* dp5_stages on the model system (system_equation_m): k6[0] keeps the last stage's d omega / dt */
static void dp5_stages_m(const orc_params *p, const orc_env *e, int n, const double *y, double h, double *k1, double *k2,
                       double *k3, double *k4, double *k5, double *k6, double *y1, int model, double c0, double c1) {
    double yt[ORC_MAX_ODE];
    for (int i = 0; i < n; ++i) yt[i] = y[i] + h * (1.0 / 5.0) * k1[i];
    system_equation_m(p, e, yt, k2, model, c0, c1);
    for (int i = 0; i < n; ++i) yt[i] = y[i] + h * (3.0 / 40.0 * k1[i] + 9.0 / 40.0 * k2[i]);
    system_equation_m(p, e, yt, k3, model, c0, c1);
    for (int i = 0; i < n; ++i) yt[i] = y[i] + h * (44.0 / 45.0 * k1[i] - 56.0 / 15.0 * k2[i] + 32.0 / 9.0 * k3[i]);
    system_equation_m(p, e, yt, k4, model, c0, c1);
    for (int i = 0; i < n; ++i)
        yt[i] = y[i] + h * (19372.0 / 6561.0 * k1[i] - 25360.0 / 2187.0 * k2[i] + 64448.0 / 6561.0 * k3[i] -
                            212.0 / 729.0 * k4[i]);
    system_equation_m(p, e, yt, k5, model, c0, c1);
    for (int i = 0; i < n; ++i)
        yt[i] = y[i] + h * (9017.0 / 3168.0 * k1[i] - 355.0 / 33.0 * k2[i] + 46732.0 / 5247.0 * k3[i] +
                            49.0 / 176.0 * k4[i] - 5103.0 / 18656.0 * k5[i]);
    system_equation_m(p, e, yt, k6, model, c0, c1);
    for (int i = 0; i < n; ++i)
        y1[i] = y[i] + h * (35.0 / 384.0 * k1[i] + 500.0 / 1113.0 * k3[i] + 125.0 / 192.0 * k4[i] -
                            2187.0 / 6784.0 * k5[i] + 11.0 / 84.0 * k6[i]);
    system_equation_m(p, e, y1, k2, model, c0, c1);
    for (int i = 0; i < n; ++i)
        k4[i] = (71.0 / 57600.0 * k1[i] - 71.0 / 16695.0 * k3[i] + 71.0 / 1920.0 * k4[i] - 17253.0 / 339200.0 * k5[i] +
                 22.0 / 525.0 * k6[i] - 1.0 / 40.0 * k2[i]) * h;
}

static double clamp3(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
/* KinkPath::ramp: U(c) = int_0^1 (s(th) - c)_+ dth along s = w + V0 th + c2 th^2 + c3 th^3; the crossing instant from the quadratic
 * through both ends with the start slope (U is stationary in it) */
typedef struct { double w, w1, V0, V0sq, q4, c2t, c3q, hV0, M; int up; } kink_path;
static double kink_ramp(const kink_path *k, double c) {
    const double cs = c - k->w;
    const double sq = sqrt(fmax(k->q4 * cs + k->V0sq, 0.0));
    const double th = fmin(fmax((cs + cs) * (1.0 / (k->V0 + (k->up ? sq : -sq))), 0.0), 1.0);
    const double Q = (((k->c3q * th + k->c2t) * th + k->hV0) * th + -cs) * th;
    const double Mc = k->M - c;
    const int a0 = k->w >= c, a1 = k->w1 >= c;
    return (a0 & a1) ? Mc : ((a0 | a1) ? (k->up ? Mc - Q : Q) : 0.0);
}
static void integrate_kink_substep(const orc_params *p, orc_env *e, int dp5, double h) {
    const double lim = p->load == ORC_LOAD_POLY_STATIC ? p->load_a / p->j_total * p->tau_decay : 0.0;
    double k1[ORC_MAX_ODE];
    system_equation(p, e, e->y, k1);
    if (!(lim > 0.0)) { /* no kink at all (gemx_create leaves the flag off): the plain scheme on the true system */
        fixed_step_m(p, e, dp5, h, k1, 0, 0.0, 0.0);
        return;
    }
    const double kap = p->j_total / p->tau_decay, inv_j = 1.0 / p->j_total, a = p->load_a;
    const double w = e->y[0];
    const double wmid = 0.5 * h * k1[0] + w;
    const int band = fabs(wmid) < lim;
    const double c1 = band ? kap : 0.0, c0 = band ? 0.0 : copysign(a, wmid);
    k1[0] = (clamp3(kap * w, -a, a) - (c1 * w + c0)) * inv_j + k1[0]; /* first stage of the model system */
    fixed_step_m(p, e, dp5, h, k1, 1, c0, c1);
    const double w1 = e->y[0], phi_lim = copysign(lim, wmid);
    const int needs = (clamp3(w, -lim, lim) != (band ? w : phi_lim)) | (clamp3(w1, -lim, lim) != (band ? w1 : phi_lim));
    if (!needs) return;
    const double V0 = h * k1[0], V1 = h * g_last0, dl = w1 - w;
    const double c2 = 3.0 * dl - 2.0 * V0 - V1, c3 = V0 + V1 - 2.0 * dl;
    const kink_path kp = {w, w1, V0, V0 * V0, 4.0 * (dl - V0), c2 * (1.0 / 3.0), c3 * 0.25, 0.5 * V0,
                          w + 0.5 * V0 + c2 * (1.0 / 3.0) + c3 * 0.25, w1 > w};
    const double lev = (kp.up ? (w < -lim) : !(w > lim)) ? -lim : lim, oth = -lev;
    const double U1 = kink_ramp(&kp, lev);
    const int o0 = w >= oth, o1 = w1 >= oth;
    double U2 = (o0 & o1) ? kp.M - oth : 0.0;
    if (o0 != o1) U2 = kink_ramp(&kp, oth);
    const double Up = lev > 0.0 ? U1 : U2, Um = lev > 0.0 ? U2 : U1;
    const double D = -(h * (1.0 / p->tau_decay)) * ((Um - Up - lim) - (band ? kp.M : phi_lim));
    e->y[0] = w1 + D;
}
/* `nsteps` equal sub-steps per segment, each corrected on its own (gemx_kernels.hpp integrate<>: `for s < ns`); the env's clock
 * advances to the segment end like in every other solver (the converter's dead-time bookkeeping and the RC supply read it) */
static void integrate_kink(const orc_params *p, orc_env *e, int dp5, double t_end) {
    const int ns = p->nsteps > 1 ? p->nsteps : 1;
    const double hs = (t_end - e->t) / ns;
    for (int s = 0; s < ns; ++s) integrate_kink_substep(p, e, dp5, hs);
    e->t = t_end;
}

/* ORC_SOLVER_DEV_ADAPTIVE -- a DIAGNOSTIC, not a restatement of the reference: the error controller of the HIP kernels' ScipyOdeSolver()
 * (gym_electric_motor_amd/csrc/gemx_kernels.hpp: dp5_adaptive / dp5_first_try) in fp64, one lane at a time, so that the statistics of a
 * 64-lane wave (attempts per control step of every lane, of the slowest lane) can be taken on the CPU: tools/wave_step_statistics.py,
 * orc_wave_attempts below.  Norm over the states without the angle, rtol 1e-6, atol 1e-9 (the device's defaults), the carried proposal in
 * e->dp_h, first try = hs / ceil(0.9 hs / proposal), a rejected step cut by clamp(0.9 err^-1/5, 0.2, 1), floor hs / 1024.
 * g_dev_first_try > 0 replaces the first try of the next segment (the wave-shared proposal under test). */
static __thread double g_dev_first_try = 0.0;
static __thread int g_dev_attempts = 0;
int g_dev_atol_omega_scaled = 0;
void orc_dev_set_atol_omega_scaled(int on) { g_dev_atol_omega_scaled = on; }
static __thread double g_dev_en2_first = 0.0; /* error norm squared of the first attempt of the last segment (diagnostic) */
int orc_dev_attempts(void) { int a = g_dev_attempts; g_dev_attempts = 0; return a; }
double orc_dev_en2_first(void) { return g_dev_en2_first; }
static double dev_first_try(const orc_params *p, double hs, double hc) {
    if (hs < 0.5 * p->tau || !(hc > 0.0) || !(hc < 0.9 * hs)) return hs;
    return hs / fmin(ceil(0.9 * hs / hc), 1024.0);
}
static void dev_adaptive(const orc_params *p, orc_env *e, double t_end) {
    const double RTOL = 1e-6, ATOL = 1e-9;
    /* ORC_SOLVER_DEV_ADAPTIVE_KINK: the same controller on the SMOOTH model system of every attempt (the load's saturation replaced by the
     * affine piece of the region the mid-step omega is predicted in), the kink's defect added to omega in closed form after an accepted
     * sub-step -- the error estimate then never sees the kink, whose crossing costs the plain controller 5-8 attempts (round 6). */
    const int kink = p->solver == ORC_SOLVER_DEV_ADAPTIVE_KINK && p->load == ORC_LOAD_POLY_STATIC && p->load_a / p->j_total * p->tau_decay > 0.0;
    const double lim = kink ? p->load_a / p->j_total * p->tau_decay : 0.0, kap = p->j_total / p->tau_decay, inv_j = 1.0 / p->j_total, la = p->load_a;
    const int n = n_ode(p), nz = ORC_IS_DC(p->system) ? n : n - 1; /* the angle is integrated alongside, outside the error norm */
    double k1[ORC_MAX_ODE], k2[ORC_MAX_ODE], k3[ORC_MAX_ODE], k4[ORC_MAX_ODE], k5[ORC_MAX_ODE], k6[ORC_MAX_ODE], y1[ORC_MAX_ODE];
    const double hs = t_end - e->t, hmin = hs / 1024.0;
    const int carries = !(hs < 0.5 * p->tau);
    double t = 0.0, h = g_dev_first_try > 0.0 ? fmin(g_dev_first_try, hs) : dev_first_try(p, hs, e->dp_h), hprop = 0.0;
    g_dev_first_try = 0.0;
    system_equation(p, e, e->y, k1);
    for (int guard = 0; guard < 4096 && t < hs; ++guard) {
        const int fin = !(h < hs - t);
        const double hh = fin ? hs - t : h;
        g_dev_attempts++;
        double k1m[ORC_MAX_ODE], c0 = 0.0, c1 = 0.0, wmid = 0.0;
        int band = 0;
        for (int i = 0; i < n; ++i) k1m[i] = k1[i];
        if (kink) {
            const double w = e->y[0];
            wmid = 0.5 * hh * k1[0] + w;
            band = fabs(wmid) < lim;
            c1 = band ? kap : 0.0; c0 = band ? 0.0 : copysign(la, wmid);
            k1m[0] = (clamp3(kap * w, -la, la) - (c1 * w + c0)) * inv_j + k1[0];
            dp5_stages_m(p, e, n, e->y, hh, k1m, k2, k3, k4, k5, k6, y1, 1, c0, c1);
        } else
        dp5_stages(p, e, n, e->y, hh, k1, k2, k3, k4, k5, k6, y1); /* k2 <- f(y1) (FSAL), k4 <- the error estimate */
        double e2 = 0.0;
        for (int i = 0; i < nz; ++i) {
            /* g_dev_atol_omega_scaled (experiment): omega's absolute tolerance in NORMALISED units, atol x limits[0] */
            const double at = (i == 0 && g_dev_atol_omega_scaled) ? ATOL * p->limits[0] : ATOL;
            const double r = k4[i] / (at + RTOL * fmax(fabs(e->y[i]), fabs(y1[i])));
            e2 += r * r;
        }
        const double en2 = e2 / nz;
        if (guard == 0) g_dev_en2_first = en2;
        const int floor_hit = !(hh > hmin), accept = !(en2 > 1.0) || floor_hit;
        double fac = en2 > 1e-20 ? 0.9 * pow(en2, -0.1) : 10.0;
        fac = fmin(fmax(fac, 0.2), accept ? 10.0 : 1.0);
        if (accept) {
            const double w = e->y[0], w1 = y1[0];
            for (int i = 0; i < n; ++i) { e->y[i] = y1[i]; k1[i] = k2[i]; }
            if (kink) {
                const double phi_lim = copysign(lim, wmid);
                const int needs = (clamp3(w, -lim, lim) != (band ? w : phi_lim)) | (clamp3(w1, -lim, lim) != (band ? w1 : phi_lim));
                if (needs) { /* integrate_kink_substep's defect, end slope = the model's f(y1) */
                    const double V0 = hh * k1m[0], V1 = hh * k2[0], dl = w1 - w;
                    const double c2 = 3.0 * dl - 2.0 * V0 - V1, c3 = V0 + V1 - 2.0 * dl;
                    const kink_path kp = {w, w1, V0, V0 * V0, 4.0 * (dl - V0), c2 * (1.0 / 3.0), c3 * 0.25, 0.5 * V0,
                                          w + 0.5 * V0 + c2 * (1.0 / 3.0) + c3 * 0.25, w1 > w};
                    const double lev = (kp.up ? (w < -lim) : !(w > lim)) ? -lim : lim, oth = -lev;
                    const double U1 = kink_ramp(&kp, lev);
                    const int o0 = w >= oth, o1 = w1 >= oth;
                    double U2 = (o0 & o1) ? kp.M - oth : 0.0;
                    if (o0 != o1) U2 = kink_ramp(&kp, oth);
                    const double Up = lev > 0.0 ? U1 : U2, Um = lev > 0.0 ? U2 : U1;
                    e->y[0] = w1 - (hh * (1.0 / p->tau_decay)) * ((Um - Up - lim) - (band ? kp.M : phi_lim));
                    system_equation(p, e, e->y, k1); /* the TRUE system's slope at the corrected state (no FSAL across a crossing) */
                }
            }
            t = fin ? hs : t + hh;
        }
        h = fmax(hh * fac, hmin);
        if (accept) hprop = fmax(hprop, h);
    }
    if (carries) e->dp_h = hprop;
    e->t = t_end;
}

static void integrate(const orc_params *p, orc_env *e, double t_end) {
    int n = n_ode(p);
    if (p->solver == ORC_SOLVER_DEV_ADAPTIVE || p->solver == ORC_SOLVER_DEV_ADAPTIVE_KINK) { dev_adaptive(p, e, t_end); return; }
    if (p->solver == ORC_SOLVER_DOPRI5) { dopri5_adaptive(p, e, t_end); return; }
    if (p->solver == ORC_SOLVER_IVP_RK45) { ivp_rk45(p, e, t_end); return; }
    if (p->solver == ORC_SOLVER_RK4_KINK || p->solver == ORC_SOLVER_DP5_KINK) { integrate_kink(p, e, p->solver == ORC_SOLVER_DP5_KINK, t_end); return; }
    double k1[ORC_MAX_ODE], k2[ORC_MAX_ODE], k3[ORC_MAX_ODE], k4[ORC_MAX_ODE], k5[ORC_MAX_ODE], k6[ORC_MAX_ODE],
        yt[ORC_MAX_ODE];
    if (p->solver == ORC_SOLVER_EULER) {
        if (p->nsteps <= 1) { /* solvers.py:124-136 */
            system_equation(p, e, e->y, k1);
            double h = t_end - e->t;
            for (int i = 0; i < n; ++i) e->y[i] = e->y[i] + k1[i] * h;
        } else { /* solvers.py:103-122 */
            double h = (t_end - e->t) / p->nsteps;
            for (int s = 0; s < p->nsteps; ++s) {
                system_equation(p, e, e->y, k1);
                for (int i = 0; i < n; ++i) e->y[i] = e->y[i] + k1[i] * h;
            }
        }
    } else if (p->solver == ORC_SOLVER_RK4) { /* classical RK4 with nsteps sub-steps; the reference has none */
        int ns = p->nsteps > 1 ? p->nsteps : 1;
        double h = (t_end - e->t) / ns;
        for (int sub = 0; sub < ns; ++sub) {
        system_equation(p, e, e->y, k1);
        for (int i = 0; i < n; ++i) yt[i] = e->y[i] + 0.5 * h * k1[i];
        system_equation(p, e, yt, k2);
        for (int i = 0; i < n; ++i) yt[i] = e->y[i] + 0.5 * h * k2[i];
        system_equation(p, e, yt, k3);
        for (int i = 0; i < n; ++i) yt[i] = e->y[i] + h * k3[i];
        system_equation(p, e, yt, k4);
        for (int i = 0; i < n; ++i) e->y[i] = e->y[i] + h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        }
    } else { /* ORC_SOLVER_DP5_FIXED: one Dormand-Prince step, 5th-order solution, no error control */
        double h = t_end - e->t;
        system_equation(p, e, e->y, k1);
        for (int i = 0; i < n; ++i) yt[i] = e->y[i] + h * (1.0 / 5.0) * k1[i];
        system_equation(p, e, yt, k2);
        for (int i = 0; i < n; ++i) yt[i] = e->y[i] + h * (3.0 / 40.0 * k1[i] + 9.0 / 40.0 * k2[i]);
        system_equation(p, e, yt, k3);
        for (int i = 0; i < n; ++i) yt[i] = e->y[i] + h * (44.0 / 45.0 * k1[i] - 56.0 / 15.0 * k2[i] + 32.0 / 9.0 * k3[i]);
        system_equation(p, e, yt, k4);
        for (int i = 0; i < n; ++i)
            yt[i] = e->y[i] + h * (19372.0 / 6561.0 * k1[i] - 25360.0 / 2187.0 * k2[i] + 64448.0 / 6561.0 * k3[i] -
                                   212.0 / 729.0 * k4[i]);
        system_equation(p, e, yt, k5);
        for (int i = 0; i < n; ++i)
            yt[i] = e->y[i] + h * (9017.0 / 3168.0 * k1[i] - 355.0 / 33.0 * k2[i] + 46732.0 / 5247.0 * k3[i] +
                                   49.0 / 176.0 * k4[i] - 5103.0 / 18656.0 * k5[i]);
        system_equation(p, e, yt, k6);
        for (int i = 0; i < n; ++i)
            e->y[i] = e->y[i] + h * (35.0 / 384.0 * k1[i] + 500.0 / 1113.0 * k3[i] + 125.0 / 192.0 * k4[i] -
                                     2187.0 / 6784.0 * k5[i] + 11.0 / 84.0 * k6[i]);
    }
    e->t = t_end;
}

/* ---------------------------------------------------------------- transforms ------------------- */
/* three_phase_motor.py:18-28 (matrices), 31-88 (t_23, t_32, q, q_inv) */
static void t_23(const double *abc, double *ab) {
    const double s = 0.5 * sqrt(3.0);
    ab[0] = 2.0 / 3.0 * (abc[0] - 0.5 * abc[1] - 0.5 * abc[2]);
    ab[1] = 2.0 / 3.0 * (s * abc[1] - s * abc[2]);
}
static void t_32(const double *ab, double *abc) {
    const double s = 0.5 * sqrt(3.0);
    abc[0] = ab[0];
    abc[1] = -0.5 * ab[0] + s * ab[1];
    abc[2] = -0.5 * ab[0] - s * ab[1];
}
static void q_rot(const double *x, double eps, double *out) {
    double c = cos(eps), s = sin(eps);
    double o0 = c * x[0] - s * x[1], o1 = s * x[0] + c * x[1];
    out[0] = o0; out[1] = o1;
}
static void dq_to_abc(const double *dq, double eps, double *abc) { double ab[2]; q_rot(dq, eps, ab); t_32(ab, abc); }
static void abc_to_dq(const double *abc, double eps, double *dq) { double ab[2]; t_23(abc, ab); q_rot(ab, -eps, dq); }

/* ---------------------------------------------------------------- converters ------------------- */
static double clip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
static double sgn(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); } /* np.sign */

/* ContTwoQuadrantConverter via ContDynamicallyAveragedConverter: set_action clips to action_space [0,1]
 * (converters.py:144-146), convert = clip(duty - sign(i)/tau*t_il, 0, 1) (148-158, 177-184, 425-427). */
/* diagnostic for the parity tests (orc_rollout_diag): the smallest |i| [A] a current-SIGN decision of the running step met -- a
 * dead leg's freewheeling diode, a continuous leg's dead-time correction.  An fp32 run may decide such a step the other way. */
static __thread double g_sign_margin = 1e300;
static void note_sign(double i) { if (fabs(i) < g_sign_margin) g_sign_margin = fabs(i); }
static double cont2qc_convert(const orc_params *p, double duty, double i) {
    if (p->t_il > 0.0) note_sign(i);
    return clip(duty - sgn(i) / p->tau * p->t_il, 0.0, 1.0);
}

/* FiniteTwoQuadrantConverter._set_switching_pattern, converters.py:300-310.  Returns #segments (1|2). */
static int fin2qc_set_action(const orc_params *p, orc_env *e, int leg, int action) {
    if (action == 0 || e->sw_state[leg] == 0 || action == e->sw_state[leg] || p->t_il == 0.0) {
        e->sw_pattern[leg][0] = action; e->sw_plen[leg] = 1;
        return 1;
    }
    e->sw_pattern[leg][0] = 0; e->sw_pattern[leg][1] = action; e->sw_plen[leg] = 2;
    return 2;
}
/* FiniteTwoQuadrantConverter.convert, converters.py:270-287 */
static double fin2qc_convert(const orc_params *p, orc_env *e, int leg, double i, double t) {
    if (t - p->tau / 1000.0 > e->action_start + p->t_il)
        e->sw_state[leg] = e->sw_pattern[leg][e->sw_plen[leg] - 1];
    else
        e->sw_state[leg] = e->sw_pattern[leg][0];
    if (e->sw_state[leg] == 0) { note_sign(i); return i < 0 ? 1.0 : 0.0; }
    if (e->sw_state[leg] == 1) return 1.0;
    return 0.0;
}

static const int B6_SUBACTIONS[8][3] = {{2, 2, 2}, {2, 2, 1}, {2, 1, 2}, {2, 1, 1},
                                        {1, 2, 2}, {1, 2, 1}, {1, 1, 2}, {1, 1, 1}}; /* converters.py:788-797 */

/* One plain converter `kind` (0..3) whose first half-bridge / duty slot is `leg0`.  Returns the length of the
 * switching-time list the reference's set_action() returns (1: [t+tau], 2: [t+t_il, t+tau]). */
static int sub_set_action(const orc_params *p, orc_env *e, int kind, int leg0, const double *action) {
    if (kind == ORC_CONV_CONT_4QC) { /* converters.py:485-491 */
        e->duty[leg0][0] = clip(0.5 * (action[0] + 1.0), 0.0, 1.0);
        e->duty[leg0][1] = clip(-0.5 * (action[0] - 1.0), 0.0, 1.0);
        return 1;
    }
    if (kind == ORC_CONV_CONT_B6) { /* converters.py:897-903 */
        for (int l = 0; l < 3; ++l) e->duty[leg0 + l][0] = clip(0.5 * (action[l] + 1.0), 0.0, 1.0);
        return 1;
    }
    int two = 0;
    if (kind == ORC_CONV_FINITE_4QC) { /* FiniteFourQuadrantConverter.set_action, converters.py:354-364 */
        static const int A0[4] = {1, 1, 2, 2}, A1[4] = {1, 2, 1, 2};
        int a4 = (int)action[0];
        if (fin2qc_set_action(p, e, leg0, A0[a4]) == 2) two = 1;
        if (fin2qc_set_action(p, e, leg0 + 1, A1[a4]) == 2) two = 1;
    } else { /* Finite-B6C, converters.py:825-835: union of the legs' switching times, sorted */
        int a = (int)action[0];
        for (int l = 0; l < 3; ++l)
            if (fin2qc_set_action(p, e, leg0 + l, B6_SUBACTIONS[a][l]) == 2) two = 1;
    }
    return two ? 2 : 1;
}

/* convert(i_out, t) of one plain converter: normalised output voltages (1 for 4QC, 3 for B6) */
static void sub_convert(const orc_params *p, orc_env *e, int kind, int leg0, const double *i_in, double t, double *u) {
    if (kind == ORC_CONV_CONT_4QC) { /* converters.py:481-483: both sub-converters see the SAME i_out */
        u[0] = cont2qc_convert(p, e->duty[leg0][0], i_in[0]) - cont2qc_convert(p, e->duty[leg0][1], i_in[0]);
    } else if (kind == ORC_CONV_CONT_B6) { /* converters.py:888-895 */
        for (int l = 0; l < 3; ++l) u[l] = cont2qc_convert(p, e->duty[leg0 + l][0], i_in[l]) - 0.5;
    } else if (kind == ORC_CONV_FINITE_4QC) { /* converters.py:350-352: second leg sees -i_out */
        u[0] = fin2qc_convert(p, e, leg0, i_in[0], t) - fin2qc_convert(p, e, leg0 + 1, -i_in[0], t);
    } else { /* converters.py:816-823 */
        for (int l = 0; l < 3; ++l) u[l] = fin2qc_convert(p, e, leg0 + l, i_in[l], t) - 0.5;
    }
}

static int sub_nsig(int kind) { return (kind == ORC_CONV_CONT_B6 || kind == ORC_CONV_FINITE_B6) ? 3 : 1; }
static int sub_nact(int kind) { return kind == ORC_CONV_CONT_B6 ? 3 : 1; }
static int sub_nleg(int kind) { return kind == ORC_CONV_CONT_4QC ? 1 : (kind == ORC_CONV_FINITE_4QC ? 2 : 3); }

/* Decompose a converter kind into its plain sub-converters (MultiConverter: converters.py:519-548 / 640-676) */
static int conv_subs(const orc_params *p, int *kinds) {
    switch (p->converter) {
        case ORC_CONV_CONT_2X4QC: kinds[0] = kinds[1] = ORC_CONV_CONT_4QC; return 2;
        case ORC_CONV_FINITE_2X4QC: kinds[0] = kinds[1] = ORC_CONV_FINITE_4QC; return 2;
        case ORC_CONV_CONT_B6_4QC: kinds[0] = ORC_CONV_CONT_B6; kinds[1] = ORC_CONV_CONT_4QC; return 2;
        case ORC_CONV_FINITE_B6_4QC: kinds[0] = ORC_CONV_FINITE_B6; kinds[1] = ORC_CONV_FINITE_4QC; return 2;
        case ORC_CONV_CONT_2XB6: kinds[0] = kinds[1] = ORC_CONV_CONT_B6; return 2;
        case ORC_CONV_FINITE_2XB6: kinds[0] = kinds[1] = ORC_CONV_FINITE_B6; return 2;
        default: kinds[0] = p->converter; return 1;
    }
}
static int conv_nsig(const orc_params *p) {
    int kinds[2], n = conv_subs(p, kinds), tot = 0;
    for (int i = 0; i < n; ++i) tot += sub_nsig(kinds[i]);
    return tot;
}
static int conv_nact(const orc_params *p) {
    int kinds[2], n = conv_subs(p, kinds), tot = 0;
    for (int i = 0; i < n; ++i) tot += sub_nact(kinds[i]);
    return tot;
}
int orc_n_act(const orc_params *p) { return p->dq_mode ? conv_nact(p) - 1 : conv_nact(p); } /* (u_d, u_q) replace (u_a, u_b, u_c) */

/* converter.set_action: returns the number of integration segments; seg_end[] = absolute switching times.
 * MultiConverter.set_action (converters.py:566-570 / 678-685): the action is split per sub-converter and the
 * switching times are the sorted UNION of the sub-converters' lists, so one dead-time transition anywhere makes
 * [t + t_il, t + tau] (all sub-converters share tau; the restatement assumes they share t_il as well). */
static int conv_set_action(const orc_params *p, orc_env *e, const double *action, double t, double *seg_end) {
    e->action_start = t; /* converters.py:67 */
    int kinds[2], n = conv_subs(p, kinds), leg0 = 0, two = 0;
    for (int i = 0; i < n; ++i) {
        if (sub_set_action(p, e, kinds[i], leg0, action) == 2) two = 1;
        action += sub_nact(kinds[i]);
        leg0 += sub_nleg(kinds[i]);
    }
    if (two) { seg_end[0] = t + p->t_il; seg_end[1] = t + p->tau; return 2; }
    seg_end[0] = t + p->tau;
    return 1;
}

/* converter.convert(i_out, t); MultiConverter.convert slices i_out by the sub-converters' signal widths
 * (converters.py:550-558 / 693-701) */
static void conv_convert(const orc_params *p, orc_env *e, const double *i_in, double t, double *u) {
    int kinds[2], n = conv_subs(p, kinds), leg0 = 0;
    for (int i = 0; i < n; ++i) {
        sub_convert(p, e, kinds[i], leg0, i_in, t, u);
        i_in += sub_nsig(kinds[i]); u += sub_nsig(kinds[i]);
        leg0 += sub_nleg(kinds[i]);
    }
}

static void conv_reset(const orc_params *p, orc_env *e, double *u) {
    e->action_start = 0.0; /* converters.py:45-54; switching state/pattern intentionally untouched */
    int kinds[2], n = conv_subs(p, kinds);
    for (int i = 0; i < n; ++i) {
        if (sub_nsig(kinds[i]) == 1) *u++ = 0.0;                /* converters.py:344-348, 475-479 */
        else { u[0] = u[1] = u[2] = -0.5; u += 3; }            /* converters.py:808-814, 880-886 */
    }
}

/* converter.i_sup(i_out): current drawn from the supply for the converter's CURRENT internal state, i.e. (as *.simulate()
 * calls it before convert()) the new duty cycles of a continuous converter but the PREVIOUS convert()'s switching state of a
 * finite one.  Cont-2QC converters.py:429-435, Finite-2QC 289-298, 4QC 366-368 / 493-495, B6 837-839 / 909-911, Multi 572-580. */
static double cont2qc_i_sup(const orc_params *p, double duty, double i) {
    double interlocking_current = i < 0 ? 1.0 : 0.0;
    return (duty + p->t_il / p->tau * (interlocking_current - duty)) * i;
}
static double fin2qc_i_sup(const orc_env *e, int leg, double i) {
    if (e->sw_state[leg] == 0) return i < 0 ? i : 0.0;
    if (e->sw_state[leg] == 1) return i;
    return 0.0;
}
static double conv_i_sup(const orc_params *p, const orc_env *e, const double *i_in) {
    int kinds[2], n = conv_subs(p, kinds), leg0 = 0;
    double tot = 0.0;
    for (int k = 0; k < n; ++k) {
        if (kinds[k] == ORC_CONV_CONT_4QC) tot += cont2qc_i_sup(p, e->duty[leg0][0], i_in[0]) + cont2qc_i_sup(p, e->duty[leg0][1], -i_in[0]);
        else if (kinds[k] == ORC_CONV_CONT_B6) for (int l = 0; l < 3; ++l) tot += cont2qc_i_sup(p, e->duty[leg0 + l][0], i_in[l]);
        else if (kinds[k] == ORC_CONV_FINITE_4QC) tot += fin2qc_i_sup(e, leg0, i_in[0]) + fin2qc_i_sup(e, leg0 + 1, -i_in[0]);
        else for (int l = 0; l < 3; ++l) tot += fin2qc_i_sup(e, leg0 + l, i_in[l]);
        i_in += sub_nsig(kinds[k]);
        leg0 += sub_nleg(kinds[k]);
    }
    return tot;
}
/* supply.get_voltage(t, i_sup): Ideal voltage_supplies.py:70-72; RC 116-123 = one explicit Euler step of
 * du/dt = (u_0 - u - R i_sup) / (R C) from the supply solver's own time to t (EulerSolver._integrate_one_step, solvers.py:131-136) */
static double supply_voltage(const orc_params *p, orc_env *e, double t, const double *i_in) {
    if (!p->rc_supply) return p->u_sup;
    double i_sup = conv_i_sup(p, e, i_in);
    e->sup_u = e->sup_u + (p->u_sup - e->sup_u - p->sup_r * i_sup) / (p->sup_r * p->sup_c) * (t - e->sup_t);
    e->sup_t = t;
    return e->sup_u;
}

/* ---------------------------------------------------------------- simulate --------------------- */
static double wrap_eps(double eps) { /* physical_systems.py:520-522 / 809-811 */
    eps = fmod(eps, 2.0 * M_PI);
    if (eps < 0) eps += 2.0 * M_PI; /* python % is non-negative for positive modulus */
    if (eps > M_PI) eps -= 2.0 * M_PI;
    return eps;
}

static void normalise(const orc_params *p, double *obs) {
    int n = n_out(p);
    for (int i = 0; i < n; ++i) obs[i] = obs[i] / p->limits[i];
}

/* SCMLSystem.simulate (DC motors), physical_systems.py:171-203.  i_in = motor.i_in(currents): the current itself
 * (dc_permanently_excited_motor.py:77-79, dc_series_motor.py:85-87), i_a + i_e (dc_shunt_motor.py:68-70) or the
 * list [i_a, i_e] (externally excited: DcMotor.i_in). */
static void dc_i_in(const orc_params *p, const orc_env *e, double *i_in) {
    if (p->system == ORC_SYS_DC_SHUNT) i_in[0] = e->y[1] + e->y[2];
    else if (p->system == ORC_SYS_DC_EXTEX) { i_in[0] = e->y[1]; i_in[1] = e->y[2]; } /* dc_motor.py:110-112 */
    else i_in[0] = e->y[1];
}
static void simulate_dc(const orc_params *p, orc_env *e, const double *action, double *obs) {
    double seg_end[2], i_in[2], u_n[2], u_in[2] = {0};
    double u_sup = p->u_sup;
    int nu = conv_nsig(p);
    dc_i_in(p, e, i_in);
    int nseg = conv_set_action(p, e, action, e->t, seg_end);
    double t0 = e->t;
    for (int s = 0; s < nseg; ++s) {
        u_sup = supply_voltage(p, e, t0, i_in); /* get_voltage(self._t, i_sup): self._t is the STEP start in every segment */
        conv_convert(p, e, i_in, e->t, u_n);
        for (int j = 0; j < nu; ++j) { u_in[j] = u_n[j] * u_sup; e->u[j] = u_in[j]; }
        integrate(p, e, s == nseg - 1 ? t0 + p->tau : seg_end[s]);
        dc_i_in(p, e, i_in);
    }
    e->k += 1;
    int nc = n_ode(p) - 1;
    obs[0] = e->y[0]; obs[1] = motor_torque(p, e->y + 1);
    for (int c = 0; c < nc; ++c) obs[2 + c] = e->y[1 + c];
    for (int j = 0; j < nu; ++j) obs[2 + nc + j] = u_in[j];
    obs[2 + nc + nu] = u_sup;
    normalise(p, obs);
}

/* ExternallyExcitedSynchronousMotorSystem.simulate, physical_systems.py:619-652.  The reference's dead-time loop
 * (lines 628-638) calls abc_to_dq_space(u_in[:2], eps) and cannot run; only the single-segment path exists. */
static void simulate_eesm(const orc_params *p, orc_env *e, const double *action, double *obs) {
    double seg_end[2], i_in[4], u_n[4], u_in[4], u_dq[2], i_abc[3];
    double u_sup = p->u_sup;
    double eps = e->y[4];
    dq_to_abc(e->y + 1, eps, i_in);
    i_in[3] = e->y[3];
    conv_set_action(p, e, action, e->t, seg_end);
    u_sup = supply_voltage(p, e, e->t, i_in);
    conv_convert(p, e, i_in, e->t, u_n);
    for (int l = 0; l < 4; ++l) u_in[l] = u_n[l] * u_sup;
    abc_to_dq(u_in, eps, u_dq);
    e->u[0] = u_dq[0]; e->u[1] = u_dq[1]; e->u[2] = u_in[3];
    integrate(p, e, e->t + p->tau);
    e->k += 1;
    double torque = motor_torque(p, e->y + 1);
    dq_to_abc(e->y + 1, eps, i_abc); /* eps of the step start (line 646) */
    obs[0] = e->y[0]; obs[1] = torque;
    obs[2] = i_abc[0]; obs[3] = i_abc[1]; obs[4] = i_abc[2]; obs[5] = e->y[1]; obs[6] = e->y[2]; obs[7] = e->y[3];
    obs[8] = u_in[0]; obs[9] = u_in[1]; obs[10] = u_in[2]; obs[11] = u_dq[0]; obs[12] = u_dq[1]; obs[13] = u_in[3];
    obs[14] = wrap_eps(e->y[4]); obs[15] = u_sup;
    normalise(p, obs);
}

/* SynchronousMotorSystem.simulate, physical_systems.py:487-525 (control_space == 'abc') */
static void simulate_pmsm(const orc_params *p, orc_env *e, const double *action, double *obs) {
    double seg_end[2], i_in[3], u_n[3], u_in[3] = {0}, u_dq[2] = {0}, i_abc[3];
    double u_sup = p->u_sup;
    double eps = e->y[3], a_abc[3];
    if (p->dq_mode == 1) { dq_to_abc(action, eps, a_abc); action = a_abc; } /* control_space == 'dq', line 491-492 */
    dq_to_abc(e->y + 1, eps, i_in);
    int nseg = conv_set_action(p, e, action, e->t, seg_end);
    double t0 = e->t;
    for (int s = 0; s < nseg; ++s) {
        u_sup = supply_voltage(p, e, t0, i_in);
        conv_convert(p, e, i_in, e->t, u_n);
        for (int l = 0; l < 3; ++l) u_in[l] = u_n[l] * u_sup;
        abc_to_dq(u_in, eps, u_dq);
        e->u[0] = u_dq[0]; e->u[1] = u_dq[1];
        integrate(p, e, s == nseg - 1 ? t0 + p->tau : seg_end[s]);
        if (s < nseg - 1) { eps = e->y[3]; dq_to_abc(e->y + 1, eps, i_in); }
    }
    e->k += 1;
    double torque = motor_torque(p, e->y + 1);
    dq_to_abc(e->y + 1, eps, i_abc); /* eps of the LAST segment start (line 519) */
    obs[0] = e->y[0]; obs[1] = torque;
    obs[2] = i_abc[0]; obs[3] = i_abc[1]; obs[4] = i_abc[2]; obs[5] = e->y[1]; obs[6] = e->y[2];
    obs[7] = u_in[0]; obs[8] = u_in[1]; obs[9] = u_in[2]; obs[10] = u_dq[0]; obs[11] = u_dq[1];
    obs[12] = wrap_eps(e->y[3]); obs[13] = u_sup;
    normalise(p, obs);
}

/* SquirrelCageInductionMotorSystem.simulate, physical_systems.py:771-814 (control_space == 'abc') */
static void simulate_scim(const orc_params *p, orc_env *e, const double *action, double *obs) {
    double seg_end[2], i_in[3], u_n[3], u_in[3] = {0}, u_dq[2] = {0}, u_ab[2], i_dq[2], i_abc[3];
    double u_sup = p->u_sup;
    double eps_fs = atan2(e->y[4], e->y[3]), a_abc[3]; /* calculate_field_angle, 765-769 */
    if (p->dq_mode == 1) { dq_to_abc(action, eps_fs, a_abc); action = a_abc; } /* control_space == 'dq', line 777-778 */
    t_32(e->y + 1, i_in);
    int nseg = conv_set_action(p, e, action, e->t, seg_end);
    double t0 = e->t;
    for (int s = 0; s < nseg; ++s) {
        u_sup = supply_voltage(p, e, t0, i_in);
        conv_convert(p, e, i_in, e->t, u_n);
        for (int l = 0; l < 3; ++l) u_in[l] = u_n[l] * u_sup;
        abc_to_dq(u_in, eps_fs, u_dq);
        t_23(u_in, u_ab);
        e->u[0] = u_ab[0]; e->u[1] = u_ab[1];
        integrate(p, e, s == nseg - 1 ? t0 + p->tau : seg_end[s]);
        if (s < nseg - 1) { eps_fs = atan2(e->y[4], e->y[3]); t_32(e->y + 1, i_in); }
    }
    e->k += 1;
    double torque = motor_torque(p, e->y + 1);
    q_rot(e->y + 1, -eps_fs, i_dq);  /* stale field angle, line 806 */
    dq_to_abc(i_dq, eps_fs, i_abc);  /* line 807 */
    obs[0] = e->y[0]; obs[1] = torque;
    obs[2] = i_abc[0]; obs[3] = i_abc[1]; obs[4] = i_abc[2]; obs[5] = i_dq[0]; obs[6] = i_dq[1];
    obs[7] = u_in[0]; obs[8] = u_in[1]; obs[9] = u_in[2]; obs[10] = u_dq[0]; obs[11] = u_dq[1];
    obs[12] = wrap_eps(e->y[5]); obs[13] = u_sup;
    normalise(p, obs);
}

/* DoublyFedInductionMotorSystem, physical_systems.py:850-1113 (simulate 948-1029).  Rotor current from the states
 * (calculate_rotor_current, 931-946). */
static void dfim_rotor_current(const orc_params *p, const double *y, double *ir) {
    double l_m = p->mp[1], l_r = p->mp[1] + p->mp[3];
    ir[0] = 1 / l_r * y[3] - l_m / l_r * y[1];
    ir[1] = 1 / l_r * y[4] - l_m / l_r * y[2];
}
static void simulate_dfim(const orc_params *p, orc_env *e, const double *action, double *obs) {
    double seg_end[2], i_in[6], u_n[6], u_in[6] = {0}, u_sdq[2] = {0}, u_rdq[2] = {0}, u_sab[2], u_rab[2], ir[2];
    double u_sup = p->u_sup;
    double eps_field = atan2(e->y[4], e->y[3]), eps_el = e->y[5];
    t_32(e->y + 1, i_in);
    dfim_rotor_current(p, e->y, ir);
    t_32(ir, i_in + 3);
    int nseg = conv_set_action(p, e, action, e->t, seg_end);
    double t0 = e->t;
    for (int s = 0; s < nseg; ++s) {
        u_sup = supply_voltage(p, e, t0, i_in);
        conv_convert(p, e, i_in, e->t, u_n);
        for (int l = 0; l < 6; ++l) u_in[l] = u_n[l] * u_sup;
        abc_to_dq(u_in, eps_field, u_sdq);                 /* line 990 (only the last segment's value is reported) */
        abc_to_dq(u_in + 3, eps_field - eps_el, u_rdq);    /* line 972 / 991 */
        t_23(u_in, u_sab);
        q_rot(u_rdq, eps_field, u_rab);                    /* dq_to_alphabeta_space(u_rdq, eps_field), line 974 / 993 */
        e->u[0] = u_sab[0]; e->u[1] = u_sab[1]; e->u[2] = u_rab[0]; e->u[3] = u_rab[1];
        integrate(p, e, s == nseg - 1 ? t0 + p->tau : seg_end[s]);
        if (s < nseg - 1) {
            eps_field = atan2(e->y[4], e->y[3]); eps_el = e->y[5];
            t_32(e->y + 1, i_in);
            dfim_rotor_current(p, e->y, ir);
            t_32(ir, i_in + 3);
        }
    }
    e->k += 1;
    double torque = motor_torque(p, e->y + 1);
    double i_sdq[2], i_sabc[3], i_rdq[2], i_rdef[3];
    q_rot(e->y + 1, -eps_field, i_sdq);                    /* stale field angle, line 1003 */
    dq_to_abc(i_sdq, eps_field, i_sabc);
    dfim_rotor_current(p, e->y, ir);
    q_rot(ir, -eps_field, i_rdq);                          /* line 1005 */
    dq_to_abc(i_rdq, eps_field - eps_el, i_rdef);          /* stale eps_el, line 1006 */
    obs[0] = e->y[0]; obs[1] = torque;
    for (int l = 0; l < 3; ++l) { obs[2 + l] = i_sabc[l]; obs[7 + l] = i_rdef[l]; obs[12 + l] = u_in[l]; obs[17 + l] = u_in[3 + l]; }
    obs[5] = i_sdq[0]; obs[6] = i_sdq[1]; obs[10] = i_rdq[0]; obs[11] = i_rdq[1];
    obs[15] = u_sdq[0]; obs[16] = u_sdq[1]; obs[20] = u_rdq[0]; obs[21] = u_rdq[1];
    obs[22] = wrap_eps(e->y[5]); obs[23] = u_sup;
    normalise(p, obs);
}

/* ---------------------------------------------------------------- public API ------------------- */
void orc_init(const orc_params *p, orc_env *e) {
    memset(e, 0, sizeof(*e));
    orc_model_constants(p, e->C);
}

/* SCMLSystem.reset 256-287 / SynchronousMotorSystem.reset 527-561 / SquirrelCage...reset 816-847,
 * with constant initialisers (electric_motor.py:270-285, mechanical_load.py:169-186). */
void orc_reset(const orc_params *p, orc_env *e, double *obs) {
    int n = n_ode(p);
    for (int i = 0; i < n; ++i) e->y[i] = p->init[i];
    e->t = 0.0; e->k = 0;
    e->dp_h = 0.0; /* ode.set_initial_value() re-creates the integrator work array */
    e->sup_u = p->u_sup; e->sup_t = 0.0; /* RCVoltageSupply.reset: the capacitor is loaded again (voltage_supplies.py:108-114) */
    double u_n[6], u_abc[6], u_dq[2], i_abc[3], i_dq[2];
    double u_sup = p->u_sup;
    conv_reset(p, e, u_n);
    double torque = motor_torque(p, e->y + 1);
    if (ORC_IS_DC(p->system)) {
        int nc = n - 1, nu = conv_nsig(p);
        obs[0] = e->y[0]; obs[1] = torque;
        for (int c = 0; c < nc; ++c) obs[2 + c] = e->y[1 + c];
        for (int j = 0; j < nu; ++j) obs[2 + nc + j] = u_n[j] * u_sup;
        obs[2 + nc + nu] = u_sup;
    } else if (p->system == ORC_SYS_EESM) { /* physical_systems.py:654-691 */
        double eps = e->y[4];
        if (eps > M_PI) eps -= 2.0 * M_PI;
        for (int l = 0; l < 4; ++l) u_abc[l] = u_n[l] * u_sup;
        abc_to_dq(u_abc, eps, u_dq);
        dq_to_abc(e->y + 1, eps, i_abc);
        obs[0] = e->y[0]; obs[1] = torque; obs[2] = i_abc[0]; obs[3] = i_abc[1]; obs[4] = i_abc[2];
        obs[5] = e->y[1]; obs[6] = e->y[2]; obs[7] = e->y[3];
        /* the reference concatenates (u_a,u_b,u_c,u_e) then (u_sd,u_sq) here -- NOT the state_names order
         * (lines 679-690); with the converter's reset voltages (u_e = 0, u_sd ~ u_sq ~ 0) this is invisible */
        obs[8] = u_abc[0]; obs[9] = u_abc[1]; obs[10] = u_abc[2]; obs[11] = u_abc[3]; obs[12] = u_dq[0]; obs[13] = u_dq[1];
        obs[14] = eps; obs[15] = u_sup;
    } else if (p->system == ORC_SYS_DFIM) { /* physical_systems.py:1031-1113 */
        double eps_el = e->y[5], eps_field = atan2(e->y[4], e->y[3]), ir[2], i_rdq[2], i_rdef[3], u_rdq[2];
        if (eps_el > M_PI) eps_el -= 2.0 * M_PI;
        if (eps_field > M_PI) eps_field -= 2.0 * M_PI;
        for (int l = 0; l < 6; ++l) u_abc[l] = u_n[l] * u_sup;
        abc_to_dq(u_abc, eps_field, u_dq);
        abc_to_dq(u_abc + 3, eps_field - eps_el, u_rdq);
        q_rot(e->y + 1, -eps_field, i_dq);
        dq_to_abc(i_dq, eps_field, i_abc);
        dfim_rotor_current(p, e->y, ir);
        q_rot(ir, -(eps_field - eps_el), i_rdq);           /* reset uses eps_field - eps_el here (line 1084), simulate eps_field */
        dq_to_abc(i_rdq, eps_field - eps_el, i_rdef);
        obs[0] = e->y[0]; obs[1] = torque;
        for (int l = 0; l < 3; ++l) { obs[2 + l] = i_abc[l]; obs[7 + l] = i_rdef[l]; obs[12 + l] = u_abc[l]; obs[17 + l] = u_abc[3 + l]; }
        obs[5] = i_dq[0]; obs[6] = i_dq[1]; obs[10] = i_rdq[0]; obs[11] = i_rdq[1];
        obs[15] = u_dq[0]; obs[16] = u_dq[1]; obs[20] = u_rdq[0]; obs[21] = u_rdq[1];
        obs[22] = eps_el; obs[23] = u_sup;
    } else if (p->system == ORC_SYS_PMSM) {
        double eps = e->y[3];
        if (eps > M_PI) eps -= 2.0 * M_PI;
        for (int l = 0; l < 3; ++l) u_abc[l] = u_n[l] * u_sup;
        abc_to_dq(u_abc, eps, u_dq);
        dq_to_abc(e->y + 1, eps, i_abc);
        obs[0] = e->y[0]; obs[1] = torque; obs[2] = i_abc[0]; obs[3] = i_abc[1]; obs[4] = i_abc[2];
        obs[5] = e->y[1]; obs[6] = e->y[2]; obs[7] = u_abc[0]; obs[8] = u_abc[1]; obs[9] = u_abc[2];
        obs[10] = u_dq[0]; obs[11] = u_dq[1]; obs[12] = eps; obs[13] = u_sup;
    } else {
        double eps = e->y[5];
        double eps_fs = atan2(e->y[4], e->y[3]);
        if (eps > M_PI) eps -= 2.0 * M_PI;
        for (int l = 0; l < 3; ++l) u_abc[l] = u_n[l] * u_sup;
        abc_to_dq(u_abc, eps_fs, u_dq);
        q_rot(e->y + 1, -eps_fs, i_dq);
        dq_to_abc(i_dq, eps_fs, i_abc);
        obs[0] = e->y[0]; obs[1] = torque; obs[2] = i_abc[0]; obs[3] = i_abc[1]; obs[4] = i_abc[2];
        obs[5] = i_dq[0]; obs[6] = i_dq[1]; obs[7] = u_abc[0]; obs[8] = u_abc[1]; obs[9] = u_abc[2];
        obs[10] = u_dq[0]; obs[11] = u_dq[1]; obs[12] = eps; obs[13] = u_sup;
    }
    for (int d = 0; d < 8; ++d) /* DeadTimeProcessor.reset (dead_time_processor.py:63-72): the deque is refilled with the reset actions */
        for (int i = 0; i < 6; ++i) e->fifo[d][i] = p->act_delay_reset[i];
    normalise(p, obs);
    for (int i = 0; i < n_out(p); ++i) e->last_state[i] = obs[i] * p->limits[i]; /* dq processor reset(), line 96 */
}

static void system_simulate(const orc_params *p, orc_env *e, const double *action, double *obs);

/* wrapper chain [DqToAbcActionProcessor [DeadTimeProcessor [system]]] */
void orc_step(const orc_params *p, orc_env *e, const double *action, double *obs) {
    double a_abc[6], active[6];
    int nc = conv_nact(p);
    if (p->dq_mode == 2) {
        int eps_idx = p->system == ORC_SYS_EESM ? 14 : 12;
        double adv = 0.5 + p->act_delay; /* set_physical_system, lines 83-86 */
        double angle = e->last_state[eps_idx] + adv * p->tau * e->last_state[0] * p->mp[0]; /* _advance_angle, 98-100 */
        dq_to_abc(action, angle, a_abc);                         /* _transformation, line 15-16 */
        if (p->system == ORC_SYS_EESM) a_abc[3] = action[2];     /* line 170 */
        action = a_abc;
    }
    if (p->act_delay > 0) { /* active = deque.pop(); deque.appendleft(action) */
        int D = p->act_delay;
        for (int i = 0; i < nc; ++i) active[i] = e->fifo[0][i];
        for (int s = 0; s + 1 < D; ++s) memcpy(e->fifo[s], e->fifo[s + 1], sizeof(e->fifo[0]));
        for (int i = 0; i < nc; ++i) e->fifo[D - 1][i] = action[i];
        action = active;
    }
    system_simulate(p, e, action, obs);
    for (int i = 0; i < n_out(p); ++i) e->last_state[i] = obs[i] * p->limits[i];
}

static void system_simulate(const orc_params *p, orc_env *e, const double *action, double *obs) {
    if (ORC_IS_DC(p->system)) simulate_dc(p, e, action, obs);
    else if (p->system == ORC_SYS_PMSM) simulate_pmsm(p, e, action, obs);
    else if (p->system == ORC_SYS_EESM) simulate_eesm(p, e, action, obs);
    else if (p->system == ORC_SYS_DFIM) simulate_dfim(p, e, action, obs);
    else simulate_scim(p, e, action, obs);
}

/* ConstraintMonitor.check_constraints core.py:834-844 (merge 'max'), LimitConstraint constraints.py:55-58,
 * SquaredConstraint 96-98, terminated = violation >= 1.0 core.py:350 */
int orc_done(const orc_params *p, const double *obs) {
    int n = n_out(p), viol = 0;
    double sq = 0.0;
    for (int i = 0; i < n; ++i) {
        if ((p->limit_mask >> i) & 1) viol |= fabs(obs[i]) > 1.0;
        if ((p->squared_mask >> i) & 1) sq += obs[i] * obs[i];
    }
    if (p->squared_mask) viol |= sq > 1.0;
    return viol;
}

/* K steps of one env.  actions: [K][A] doubles (A = 1 | 3; discrete action stored as double).
 * auto_reset: mirror `if terminated: env.reset()` of the reference's usage loop. */
void orc_rollout(const orc_params *p, orc_env *e, const double *actions, int n_act, int K, int auto_reset,
                 double *obs_out, uint8_t *done_out) {
    int no = n_out(p);
    double scratch[ORC_MAX_OUT];
    for (int k = 0; k < K; ++k) {
        double *obs = obs_out + (size_t)k * no;
        orc_step(p, e, actions + (size_t)k * n_act, obs);
        int d = orc_done(p, obs);
        if (done_out) done_out[k] = (uint8_t)d;
        if (d && auto_reset) orc_reset(p, e, scratch);
    }
}

/* orc_rollout with two diagnostics per step for the parity tests' conditioning arguments: diag[k][0] = |psi_r| [Wb] at the START of
 * step k (induction machines: the flux the step's field angle is the arctan2 of; 0 otherwise), diag[k][1] = the smallest |i| [A] a
 * current-sign decision of step k met (1e300: none). */
void orc_rollout_diag(const orc_params *p, orc_env *e, const double *actions, int n_act, int K, int auto_reset,
                      double *obs_out, uint8_t *done_out, double *diag) {
    int no = n_out(p);
    double scratch[ORC_MAX_OUT];
    const int im = p->system == ORC_SYS_SCIM || p->system == ORC_SYS_DFIM;
    for (int k = 0; k < K; ++k) {
        double *obs = obs_out + (size_t)k * no;
        diag[2 * k] = im ? hypot(e->y[3], e->y[4]) : 0.0;
        g_sign_margin = 1e300;
        orc_step(p, e, actions + (size_t)k * n_act, obs);
        diag[2 * k + 1] = g_sign_margin;
        int d = orc_done(p, obs);
        if (done_out) done_out[k] = (uint8_t)d;
        if (d && auto_reset) orc_reset(p, e, scratch);
    }
}

/* Throughput helper for bench.py's cpu_baseline ("port"): n_env independent envs, K steps each, same
 * action stream layout [K][n_env][A]; only the last observation per env is kept. */
void orc_rollout_many(const orc_params *p, int n_env, const double *actions, int n_act, int K, int auto_reset,
                      double *last_obs, int64_t *n_done) {
    int no = n_out(p);
    double scratch[ORC_MAX_OUT];
    int64_t nd = 0;
    for (int j = 0; j < n_env; ++j) {
        orc_env e;
        orc_init(p, &e);
        orc_reset(p, &e, scratch);
        double *obs = last_obs + (size_t)j * no;
        for (int k = 0; k < K; ++k) {
            orc_step(p, &e, actions + ((size_t)k * n_env + j) * n_act, obs);
            if (orc_done(p, obs)) { nd++; if (auto_reset) orc_reset(p, &e, scratch); }
        }
    }
    if (n_done) *n_done = nd;
}

/* Diagnostic (tools/wave_step_statistics.py): `lanes` envs advanced in LOCKSTEP under ORC_SOLVER_DEV_ADAPTIVE, as one wave of the HIP kernel
 * is.  hist_lane[a] = (lane, control step) pairs that took a attempts; hist_wave[a] = control steps whose slowest lane took a (what the
 * wave pays: lanes that are through ride along).  shared = 1: every lane's first try of a control step is the MINIMUM of the lanes' own
 * first tries (the wave-shared proposal of the round-5 verdict, item 5); rejected lanes still cut their own steps.  Histograms of 32 bins
 * (last = that many or more).  actions [K][lanes][A]. */
void orc_wave_attempts(const orc_params *p, int lanes, const double *actions, int n_act, int K, int shared, int64_t *hist_lane, int64_t *hist_wave) {
    if (lanes > 64) lanes = 64;
    orc_env e[64];
    double obs[ORC_MAX_OUT], scratch[ORC_MAX_OUT];
    for (int j = 0; j < lanes; ++j) { orc_init(p, &e[j]); orc_reset(p, &e[j], scratch); }
    for (int k = 0; k < K; ++k) {
        double first = 0.0;
        if (shared) {
            first = p->tau;
            for (int j = 0; j < lanes; ++j) first = fmin(first, dev_first_try(p, p->tau, e[j].dp_h));
        }
        int worst = 0;
        for (int j = 0; j < lanes; ++j) {
            g_dev_first_try = first;
            g_dev_attempts = 0;
            orc_step(p, &e[j], actions + ((size_t)k * lanes + j) * n_act, obs);
            g_dev_first_try = 0.0;
            const int a = g_dev_attempts < 31 ? g_dev_attempts : 31;
            hist_lane[a]++;
            worst = a > worst ? a : worst;
            if (orc_done(p, obs)) orc_reset(p, &e[j], scratch);
        }
        hist_wave[worst]++;
    }
}

/* WeightedSumOfErrors.reward, reward_functions/weighted_sum_of_errors.py:125-129:
 *   (1 - v) * (-sum_i w_i * (|s_i - ref_i| / len_i) ** n_i + bias) + v * violation_reward,  v = violation degree (0 | 1).
 * `reference` is the generator's full-length array (zero at un-referenced states, core.py:346). */
double orc_reward(int n, const double *weights, const double *powers, const double *state_length, double bias,
                  double violation_reward, const double *state, const double *reference, int violated) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) acc += weights[i] * pow(fabs(state[i] - reference[i]) / state_length[i], powers[i]);
    double wse = -acc + bias;
    double v = violated ? 1.0 : 0.0;
    return (1.0 - v) * wse + v * violation_reward;
}

/* Bare pieces exported for known-answer tests */
double orc_kat_poly_load(const orc_params *p, double omega, double torque) { return mechanical_ode(p, omega, torque); }
size_t orc_sizeof_params(void) { return sizeof(orc_params); }
size_t orc_sizeof_env(void) { return sizeof(orc_env); }

/* Converter known-answer hook: set_action(action, t) then convert(i_seg, t_segment_start) for each segment,
 * exactly as *.simulate() drives the converter.  currents/volt: [2][3].  Returns the number of segments. */
int orc_kat_converter_n(const orc_params *p, orc_env *e, const double *action, double t, const double *currents,
                        double *volt, int stride) {
    double seg_end[2];
    int nseg = conv_set_action(p, e, action, t, seg_end);
    double t_seg = t;
    for (int s = 0; s < nseg; ++s) {
        conv_convert(p, e, currents + stride * s, t_seg, volt + stride * s);
        t_seg = seg_end[s];
    }
    return nseg;
}
int orc_kat_converter(const orc_params *p, orc_env *e, const double *action, double t, const double *currents,
                      double *volt) {
    return orc_kat_converter_n(p, e, action, t, currents, volt, 3);
}
void orc_kat_converter_reset(const orc_params *p, orc_env *e, double *u) { conv_reset(p, e, u); }
