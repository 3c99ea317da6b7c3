// gemx_capi.hip -- C ABI of libgemx.so (include/gemx.h): handle management, validation, reset / state access
// kernels, dispatch to the kernel instantiation units (gemx_inst.hip).  No CPU fallback.
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <set>
#include <string>

#include "gemx_common.hpp"

namespace gemx {

// Reset observation of ONE env from its (freshly drawn) initial state: SCMLSystem.reset 256-287 (DC systems),
// SynchronousMotorSystem.reset 527-561, ExternallyExcitedSynchronousMotorSystem.reset 654-691, SquirrelCageInductionMotorSystem.reset
// 816-847, DoublyFedInductionMotorSystem.reset 1031-1113 -- the device-side mirror of host_reset_obs() below (which serves the constant
// initial state).  Converter reset voltages: 0 per 4QC, -0.5 u_sup per B6 leg.  y = [omega, motor states...], eps = electrical angle.
template <class R>
__device__ void reset_obs_row(int sys, const DevParams<R> &P, const double *y, double eps, int nd, int nout, double *o) {
    const double us = (double)P.u_sup, s3 = sqrt(3.0);
    for (int j = 0; j < nout; ++j) o[j] = 0.0;
    const double ua = -0.5 * us, ual = 2.0 / 3.0 * (ua - 0.5 * ua - 0.5 * ua), ube = 2.0 / 3.0 * (0.5 * s3 * ua - 0.5 * s3 * ua);
    if (eps > kPi) eps -= kTwoPi;
    if (sys == GEMX_SYS_SYNC || sys == GEMX_SYS_EESM) {
        const double c = cos(eps), s = sin(eps);
        const double al = c * y[1] - s * y[2], be = s * y[1] + c * y[2];
        o[0] = y[0];
        o[2] = al; o[3] = -0.5 * al + 0.5 * s3 * be; o[4] = -0.5 * al - 0.5 * s3 * be;
        o[5] = y[1]; o[6] = y[2];
        if (sys == GEMX_SYS_SYNC) {
            o[1] = ((double)P.tc0 + (double)P.tc1 * y[1]) * y[2];
            o[7] = ua; o[8] = ua; o[9] = ua; o[10] = c * ual + s * ube; o[11] = -s * ual + c * ube; o[12] = eps; o[13] = us;
        } else {  // (u_a, u_b, u_c, u_e, u_sd, u_sq): the reference's reset layout, lines 679-690
            o[1] = ((double)P.tc0 * y[3] + (double)P.tc1 * y[1]) * y[2];
            o[7] = y[3];
            o[8] = ua; o[9] = ua; o[10] = ua; o[11] = 0.0; o[12] = c * ual + s * ube; o[13] = -s * ual + c * ube; o[14] = eps; o[15] = us;
        }
    } else if (sys == GEMX_SYS_SCIM) {  // y = [omega, i_salpha, i_sbeta, psi_ralpha, psi_rbeta]; dq frame = rotor-flux angle (lines 765-769)
        const double efs = atan2(y[4], y[3]), c = cos(efs), s = sin(efs);
        o[0] = y[0]; o[1] = (double)P.tc0 * (y[3] * y[2] - y[4] * y[1]);
        o[2] = y[1]; o[3] = -0.5 * y[1] + 0.5 * s3 * y[2]; o[4] = -0.5 * y[1] - 0.5 * s3 * y[2];
        o[5] = c * y[1] + s * y[2]; o[6] = -s * y[1] + c * y[2];
        o[7] = ua; o[8] = ua; o[9] = ua; o[10] = c * ual + s * ube; o[11] = -s * ual + c * ube; o[12] = eps; o[13] = us;
    } else if (sys == GEMX_SYS_DFIM) {  // lines 1031-1113: stator in the field frame, rotor quantities rotated by eps_field - eps_el
        double ef = atan2(y[4], y[3]);
        if (ef > kPi) ef -= kTwoPi;
        const double cf = cos(ef), sf = sin(ef), cd = cos(ef - eps), sd = sin(ef - eps);
        const double ira = (double)P.tc2 * y[3] - (double)P.tc3 * y[1], irb = (double)P.tc2 * y[4] - (double)P.tc3 * y[2];
        const double ird = cd * ira + sd * irb, irq = -sd * ira + cd * irb;
        const double ra = cd * ird - sd * irq, rb = sd * ird + cd * irq;
        o[0] = y[0]; o[1] = (double)P.tc0 * (y[3] * y[2] - y[4] * y[1]);
        o[2] = y[1]; o[3] = -0.5 * y[1] + 0.5 * s3 * y[2]; o[4] = -0.5 * y[1] - 0.5 * s3 * y[2];
        o[5] = cf * y[1] + sf * y[2]; o[6] = -sf * y[1] + cf * y[2];
        o[7] = ra; o[8] = -0.5 * ra + 0.5 * s3 * rb; o[9] = -0.5 * ra - 0.5 * s3 * rb;
        o[10] = ird; o[11] = irq;
        for (int l = 0; l < 3; ++l) { o[12 + l] = ua; o[17 + l] = ua; }
        o[15] = cf * ual + sf * ube; o[16] = -sf * ual + cf * ube;
        o[20] = cd * ual + sd * ube; o[21] = -sd * ual + cd * ube;
        o[22] = eps; o[23] = us;
    } else {  // DC systems: [omega, torque, currents..., u (0 per converter), u_sup]
        const int nc = nd - 1, nu = sys == GEMX_SYS_DC_EXTEX ? 2 : 1;
        double torque = (double)P.tc0 * y[1];
        if (sys == GEMX_SYS_DC_SERIES) torque = (double)P.tc0 * y[1] * y[1];
        if (sys == GEMX_SYS_DC_SHUNT || sys == GEMX_SYS_DC_EXTEX) torque = (double)P.tc0 * y[1] * y[2];
        o[0] = y[0]; o[1] = torque;
        for (int i = 0; i < nc; ++i) o[2 + i] = y[1 + i];
        for (int j = 0; j < nu; ++j) o[2 + nc + j] = 0.0;
        o[2 + nc + nu] = us;
    }
}

// reset: masked envs back to the initial ODE state (constant, or drawn per env: electric_motor.py:150-257,
// mechanical_load.py:100-160); optional reset observation
template <class R>
__global__ void reset_kernel(R *state, typename Angle<R>::T *angle, const uint8_t *mask, R *obs, int64_t N, int nd, int nout,
                             int has_angle, int obs_layout, DevParams<R> P, const R *reset_obs, unsigned char *ring, int ring_row_bytes,
                             int sys, const InitDev *rinit, uint32_t *rcnt, int advance) {
    int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    if (mask != nullptr && mask[env] == 0) return;
    double y0[GEMX_MAX_ODE];
    for (int j = 0; j < nd; ++j) y0[j] = (double)P.init[j];
    double eps0 = 0.0;
    if (P.init_kind) {
        // advance = 0 (gemx_reset_again): back to the draw the env last STARTED from -- the counter stays where it is
        const uint32_t count = advance ? rcnt[env] + 1u : (rcnt[env] > 0u ? rcnt[env] : 1u);
        rcnt[env] = count;
        double v[GEMX_MAX_ODE];
        if (sys == GEMX_SYS_SCIM || sys == GEMX_SYS_DFIM) init_draw_all<true>(rinit, env, count, v);
        else init_draw_all<false>(rinit, env, count, v);
        for (int j = 0; j < nd; ++j) y0[j] = v[j];
        if (has_angle) eps0 = v[nd];
    }
    for (int j = 0; j < nd; ++j) state[(int64_t)j * N + env] = (R)y0[j];
    if (P.rc_supply) {  // RCVoltageSupply.reset (voltage_supplies.py:108-114): capacitor loaded, the supply's clock at 0
        state[(int64_t)nd * N + env] = P.u_sup;
        state[(int64_t)(nd + 1) * N + env] = R(0);
    }
    if (P.adaptive) state[(int64_t)(nd + 2) * N + env] = R(0);  // error-controlled solver: no step-size prediction at the start of an episode
    // DeadTimeProcessor.reset (dead_time_processor.py:63-72): the deque is refilled with the reset action (zeros unless the handle
    // carries a custom one: one byte = a discrete index, else ring_row_bytes / sizeof(R) continuous entries)
    for (int d = 0; d < P.delay; ++d) {
        unsigned char *row = ring + ((int64_t)d * N + env) * ring_row_bytes;
        if (ring_row_bytes == 1) row[0] = (unsigned char)P.dreset_d;
        else
            for (int i = 0; i < ring_row_bytes / (int)sizeof(R); ++i) reinterpret_cast<R *>(row)[i] = P.dreset[i];
    }
    if (has_angle) angle[env] = P.init_kind ? Angle<R>::from_rad(eps0) : Angle<R>::from_bits(P.init_angle_rep);
    if (obs != nullptr) {
        double row[GEMX_MAX_OUT];
        if (P.init_kind) {
            reset_obs_row<R>(sys, P, y0, eps0, nd, nout, row);
            for (int j = 0; j < nout; ++j) row[j] *= (double)P.inv_lim[j];
        }
        for (int j = 0; j < nout; ++j) {
            const R v = P.init_kind ? (R)row[j] : reset_obs[j];
            if (obs_layout == GEMX_OBS_AOS) obs[env * nout + j] = v;
            else obs[(int64_t)j * N + env] = v;
        }
    }
}

// gemx_synthetic_actions: the synthetic action stream (gemx_common.hpp: synth_u32) written out, [K][N][A] R or [K][N] uint8
template <class R>
__global__ void synth_actions_kernel(unsigned char *out, int64_t N, int K, int nact, int n_actions, uint64_t seed, uint32_t step0, int64_t env_base) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (k, env)
    if (idx >= (int64_t)K * N) return;
    const int64_t env = env_base + idx % N;  // the stream's env word is the GLOBAL index (gemx_config.env_base)
    const uint32_t t = step0 + (uint32_t)(idx / N);
    if (n_actions > 0) out[idx] = (unsigned char)synth_index(synth_u32(seed, env, t, 0u), (uint32_t)n_actions);
    else
        for (int i = 0; i < nact; ++i) reinterpret_cast<R *>(out)[idx * nact + i] = (R)synth_unit(synth_u32(seed, env, t, (uint32_t)i));
}

template <class R>
__global__ void get_state_kernel(const R *state, const typename Angle<R>::T *angle, R *out, int64_t N, int nd, int has_angle) {
    int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    for (int j = 0; j < nd; ++j) out[(int64_t)j * N + env] = state[(int64_t)j * N + env];
    if (has_angle) out[(int64_t)nd * N + env] = Angle<R>::to_rad(angle[env]);
}
template <class R>
__global__ void set_state_kernel(R *state, typename Angle<R>::T *angle, const R *in, int64_t N, int nd, int has_angle) {
    int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    for (int j = 0; j < nd; ++j) state[(int64_t)j * N + env] = in[(int64_t)j * N + env];
    if (has_angle) angle[env] = Angle<R>::from_rad((double)in[(int64_t)nd * N + env]);
}

}  // namespace gemx

// =================================================================================================
// host side: handle, validation, launch
// =================================================================================================
using namespace gemx;

void gemx_cov_note(const char *key);  // instantiation coverage (below): a process with GEMX_COVERAGE_FILE set lists every distinct kernel it launches
#define GEMX_COV(name) gemx_cov_note(name)
static thread_local char g_err[512] = "";
namespace gemx {
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace gemx
#define HIP_TRY(x) GEMX_HIP_TRY(x)

// which entries of the reference's model-constant matrix each system uses, in the order of DevParams::m
static const int DC_IDX[][2] = {{0, 0}, {0, 1}, {0, 2}};
static const int SERIES_IDX[][2] = {{0, 0}, {0, 1}, {0, 2}};
static const int SHUNT_IDX[][2] = {{0, 0}, {0, 2}, {0, 3}, {1, 1}, {1, 4}};
static const int EESM_IDX[][2] = {{0, 1}, {0, 3}, {0, 4}, {0, 6}, {0, 8}, {1, 2}, {1, 5}, {1, 7}, {1, 9},
                                  {2, 1}, {2, 3}, {2, 4}, {2, 6}, {2, 8}};
static const int SYNC_IDX[][2] = {{0, 1}, {0, 3}, {0, 6}, {1, 0}, {1, 2}, {1, 4}, {1, 5}};
static const int SCIM_IDX[][2] = {{0, 1}, {0, 3}, {0, 6}, {0, 7}, {1, 2}, {1, 4}, {1, 5}, {1, 8},
                                  {2, 1}, {2, 3}, {2, 6}, {3, 2}, {3, 4}, {3, 5}};

static const int DFIM_IDX[][2] = {{0, 1}, {0, 3}, {0, 6}, {0, 7}, {1, 2}, {1, 4}, {1, 5}, {1, 8}, {2, 1}, {2, 3},
                                  {2, 6}, {3, 2}, {3, 4}, {3, 5}, {0, 9}, {1, 10}, {2, 9}, {3, 10}};

static int pack_model(const gemx_config &c, double *m, double *pole) {
    const int(*idx)[2];
    int n, pole_row, rows, cols;
    switch (c.system_kind) {
        case GEMX_SYS_DC_PERMEX: idx = DC_IDX; n = 3; pole_row = -1; rows = 1; cols = 3; break;
        case GEMX_SYS_DC_SERIES: idx = SERIES_IDX; n = 3; pole_row = -1; rows = 1; cols = 3; break;
        case GEMX_SYS_DC_SHUNT: case GEMX_SYS_DC_EXTEX: idx = SHUNT_IDX; n = 5; pole_row = -1; rows = 2; cols = 5; break;
        case GEMX_SYS_EESM: idx = EESM_IDX; n = 14; pole_row = 3; rows = 4; cols = 10; break;
        case GEMX_SYS_SYNC: idx = SYNC_IDX; n = 7; pole_row = 2; rows = 3; cols = 7; break;
        case GEMX_SYS_SCIM: idx = SCIM_IDX; n = 14; pole_row = 4; rows = 5; cols = 9; break;  // u_r columns: zero rotor voltage
        case GEMX_SYS_DFIM: idx = DFIM_IDX; n = 18; pole_row = 4; rows = 5; cols = 11; break;
        default: return fail(GEMX_ERR_ARG, "unknown system_kind %d", c.system_kind);
    }
    bool used[GEMX_MODEL_ROWS][GEMX_MODEL_COLS] = {};
    for (int i = 0; i < n; ++i) {
        m[i] = c.model[idx[i][0] * GEMX_MODEL_COLS + idx[i][1]];
        used[idx[i][0]][idx[i][1]] = true;
    }
    *pole = 0.0;
    if (pole_row >= 0) {
        *pole = c.model[pole_row * GEMX_MODEL_COLS + 0];
        used[pole_row][0] = true;
    }
    (void)rows;
    for (int r = 0; r < GEMX_MODEL_ROWS; ++r)
        for (int k = 0; k < GEMX_MODEL_COLS; ++k)
            if (!used[r][k] && !(k >= cols && c.system_kind == GEMX_SYS_SCIM) && c.model[r * GEMX_MODEL_COLS + k] != 0.0)
                return fail(GEMX_ERR_ARG, "model[%d][%d] = %g is outside the sparsity pattern supported for system_kind %d", r, k,
                            c.model[r * GEMX_MODEL_COLS + k], c.system_kind);
    return GEMX_OK;
}

template <class R> static void fill_params(const gemx_handle &h, const double *m, double pole, DevParams<R> &P) {
    const gemx_config &c = h.cfg;
    memset(&P, 0, sizeof(P));
    for (int i = 0; i < 20; ++i) P.m[i] = (R)m[i];
    P.tc0 = (R)c.torque_coef[0];
    P.tc1 = (R)c.torque_coef[1];
    P.tc2 = (R)c.torque_coef[2];
    P.tc3 = (R)c.torque_coef[3];
    P.pole = (R)pole;
    P.inv_j = (R)(c.j_total > 0 ? 1.0 / c.j_total : 0.0);
    P.la = (R)c.load_a; P.lb = (R)c.load_b; P.lc = (R)c.load_c;
    // PolynomialStaticLoad.set_j_rotor, polynomial_static_load.py:62-66
    P.omega_lim = (R)(c.j_total > 0 ? c.load_a / c.j_total * c.tau_decay : 0.0);
    P.lin_factor = (R)(c.tau_decay > 0 ? c.j_total / c.tau_decay : 0.0);
    P.inv_tau_decay = (R)(c.tau_decay > 0 ? 1.0 / c.tau_decay : 0.0);
    P.u_sup = (R)c.u_nominal;
    P.il_ratio = (R)(c.interlocking_time / c.tau);
    P.tau = (R)c.tau;
    P.t_il = (R)c.interlocking_time;
    P.inv_ns = (R)(1.0 / c.solver_nsteps);
    for (int i = 0; i < GEMX_MAX_OUT; ++i) {
        P.inv_lim[i] = (R)(i < h.nout ? 1.0 / c.limits[i] : 0.0);
    }
    P.cw = (const R *)h.cw_dev;
    for (int i = 0; i < h.nd; ++i) P.init[i] = (R)c.init_state[i];
    P.init_angle_rep = Angle<R>::to_bits(Angle<R>::from_rad(h.has_angle ? c.init_state[h.nd] : 0.0));
    P.nsteps = c.solver_nsteps;
    P.lin = (const R *)h.linmap_dev;
    P.lin_on = 0;  // enabled by the first launch of a linable instantiation (launch_advance_t)
    P.init_kind = c.init_kind;
    P.rc_supply = c.supply_kind == GEMX_SUPPLY_RC;
    P.sup_r = (R)c.supply_r;
    P.sup_inv_rc = (R)(P.rc_supply ? 1.0 / (c.supply_r * c.supply_c) : 0.0);
    P.dq_processor = c.action_frame == GEMX_ACT_DQ_PROCESSOR;
    P.delay = c.action_delay;
    {
        const int ck = c.converter_kind;
        const bool disc = ck == GEMX_CONV_FINITE_B6 || ck == GEMX_CONV_FINITE_4QC || ck == GEMX_CONV_FINITE_2X4QC || ck == GEMX_CONV_FINITE_B6_4QC ||
                          ck == GEMX_CONV_FINITE_2XB6;
        for (int i = 0; i < MAX_ACT; ++i) P.dreset[i] = (R)(c.action_delay > 0 ? c.action_delay_reset[i] : 0.0);
        P.dreset_d = disc && c.action_delay > 0 ? (uint32_t)c.action_delay_reset[0] : 0u;
    }
    P.dq_adv = (R)((0.5 + c.action_delay) * c.tau * pole);  // dq_to_abc_action_processor.py:83-86, 98-100
    P.kink_split = (c.solver_flags & GEMX_SOLVER_SPLIT_KINKS) && c.load_kind == GEMX_LOAD_POLY_STATIC && P.omega_lim > R(0);
    P.adaptive = (c.solver_flags & GEMX_SOLVER_ADAPTIVE) ? 1 : 0;
    P.rtol = (R)(c.solver_rtol > 0 ? c.solver_rtol : 1e-6);
    P.atol = (R)(c.solver_atol > 0 ? c.solver_atol : 1e-9);
    P.atol_w = (R)(c.solver_atol_omega > 0 ? c.solver_atol_omega : (double)P.atol * c.limits[0]);  // (omega is state 0 and observation 0 of every system)
    P.errw = nullptr;  // (set per launch: launch_advance_t)
    // (round 6: the error-controlled solver honours GEMX_SOLVER_SPLIT_KINKS -- every attempt on the smooth model system, dp5_adaptive)
    P.auto_reset = c.auto_reset;
    P.obs_layout = c.obs_layout;
    P.dc_thr[0] = P.dc_thr[1] = (R)INFINITY;
    if (!h.has_angle)
        for (int cidx = 0; cidx < h.nd - 1 && cidx < 2; ++cidx) P.dc_thr[cidx] = viol_threshold<R>(P.inv_lim[2 + cidx]);
    // the env's default constraint gets the 3-instruction fast path (Stepper::default_done)
    const bool is_dc = !h.has_angle;
    const bool two_currents = c.system_kind == GEMX_SYS_DC_SHUNT || c.system_kind == GEMX_SYS_DC_EXTEX;
    uint32_t def_limit = !is_dc ? 0u : (two_currents ? ((1u << 2) | (1u << 3)) : (1u << 2));
    if (c.system_kind == GEMX_SYS_EESM) def_limit = 1u << 7;  // LimitConstraint(('i_e',)), cont_cc_eesm_env.py:108
    const uint32_t def_sq = is_dc ? 0u : ((1u << 5) | (1u << 6));
    if (c.limit_mask == 0 && c.squared_mask == 0) P.constr_kind = 0;
    else if (c.limit_mask == def_limit && c.squared_mask == def_sq) P.constr_kind = 1;
    else P.constr_kind = 2;
}

// reset observation in fp64 on the host (SCMLSystem.reset 256-287, SynchronousMotorSystem.reset 527-561,
// SquirrelCageInductionMotorSystem.reset 816-847) for the constant initial state
static void host_reset_obs(gemx_handle &h, const double *m) {
    (void)m;
    const gemx_config &c = h.cfg;
    double *o = h.reset_obs;
    const double *y = c.init_state;
    const double us = c.u_nominal;
    memset(o, 0, sizeof(double) * GEMX_MAX_OUT);
    auto T32 = [](double al, double be, double *abc) {
        abc[0] = al; abc[1] = -0.5 * al + 0.5 * sqrt(3.0) * be; abc[2] = -0.5 * al - 0.5 * sqrt(3.0) * be;
    };
    auto T23 = [](const double *abc, double *ab) {
        ab[0] = 2.0 / 3.0 * (abc[0] - 0.5 * abc[1] - 0.5 * abc[2]);
        ab[1] = 2.0 / 3.0 * (0.5 * sqrt(3.0) * abc[1] - 0.5 * sqrt(3.0) * abc[2]);
    };
    if (!h.has_angle) {  // DC motors: [omega, torque, currents..., u, u_sup]
        const int nc = h.nd - 1;
        double torque = c.torque_coef[0] * y[1];
        if (c.system_kind == GEMX_SYS_DC_SERIES) torque = c.torque_coef[0] * y[1] * y[1];
        if (c.system_kind == GEMX_SYS_DC_SHUNT || c.system_kind == GEMX_SYS_DC_EXTEX) torque = c.torque_coef[0] * y[1] * y[2];
        const int nu = c.system_kind == GEMX_SYS_DC_EXTEX ? 2 : 1;  // converter.reset() -> 0 V per 4QC
        o[0] = y[0]; o[1] = torque;
        for (int i = 0; i < nc; ++i) o[2 + i] = y[1 + i];
        for (int j = 0; j < nu; ++j) o[2 + nc + j] = 0.0 * us;
        o[2 + nc + nu] = us;
    } else if (c.system_kind == GEMX_SYS_EESM) {  // physical_systems.py:654-691
        double uabc[3] = {-0.5 * us, -0.5 * us, -0.5 * us}, uab[2], iabc[3];
        T23(uabc, uab);
        double eps = y[4];
        if (eps > kPi) eps -= kTwoPi;
        const double cs = cos(eps), sn = sin(eps);
        T32(cs * y[1] - sn * y[2], sn * y[1] + cs * y[2], iabc);
        o[0] = y[0]; o[1] = (c.torque_coef[0] * y[3] + c.torque_coef[1] * y[1]) * y[2];
        o[2] = iabc[0]; o[3] = iabc[1]; o[4] = iabc[2]; o[5] = y[1]; o[6] = y[2]; o[7] = y[3];
        // the reference lays the reset voltages out as (u_a, u_b, u_c, u_e, u_sd, u_sq) (lines 679-690): u_e = 0 lands in
        // the u_sd slot and (u_sd, u_sq) ~ 1e-17 in the u_sq / u_e slots; reproduced as is
        o[8] = uabc[0]; o[9] = uabc[1]; o[10] = uabc[2]; o[11] = 0.0 * us;
        o[12] = cs * uab[0] + sn * uab[1]; o[13] = -sn * uab[0] + cs * uab[1];
        o[14] = eps; o[15] = us;
    } else if (c.system_kind == GEMX_SYS_DFIM) {  // physical_systems.py:1031-1113
        double uabc[3] = {-0.5 * us, -0.5 * us, -0.5 * us}, uab[2], isabc[3], irdef[3];
        T23(uabc, uab);
        double eps_el = y[5], eps_f = atan2(y[4], y[3]);
        if (eps_el > kPi) eps_el -= kTwoPi;
        if (eps_f > kPi) eps_f -= kTwoPi;
        const double cf = cos(eps_f), sf = sin(eps_f), cd = cos(eps_f - eps_el), sd = sin(eps_f - eps_el);
        const double ira = c.torque_coef[2] * y[3] - c.torque_coef[3] * y[1], irb = c.torque_coef[2] * y[4] - c.torque_coef[3] * y[2];
        const double isd = cf * y[1] + sf * y[2], isq = -sf * y[1] + cf * y[2];
        const double ird = cd * ira + sd * irb, irq = -sd * ira + cd * irb;  // reset() rotates by eps_field - eps_el (line 1084)
        T32(y[1], y[2], isabc);
        T32(cd * ird - sd * irq, sd * ird + cd * irq, irdef);
        o[0] = y[0]; o[1] = c.torque_coef[0] * (y[3] * y[2] - y[4] * y[1]);
        for (int l = 0; l < 3; ++l) { o[2 + l] = isabc[l]; o[7 + l] = irdef[l]; o[12 + l] = uabc[l]; o[17 + l] = uabc[l]; }
        o[5] = isd; o[6] = isq; o[10] = ird; o[11] = irq;
        o[15] = cf * uab[0] + sf * uab[1]; o[16] = -sf * uab[0] + cf * uab[1];
        o[20] = cd * uab[0] + sd * uab[1]; o[21] = -sd * uab[0] + cd * uab[1];
        o[22] = eps_el; o[23] = us;
    } else {
        double uabc[3] = {-0.5 * us, -0.5 * us, -0.5 * us}, uab[2], iabc[3], idq[2], udq[2], eps, torque, cs, sn;
        T23(uabc, uab);
        if (c.system_kind == GEMX_SYS_SYNC) {
            eps = y[3];
            cs = cos(eps); sn = sin(eps);
            torque = (c.torque_coef[0] + c.torque_coef[1] * y[1]) * y[2];
            idq[0] = y[1]; idq[1] = y[2];
            T32(cs * y[1] - sn * y[2], sn * y[1] + cs * y[2], iabc);
        } else {
            eps = y[5];
            double efs = atan2(y[4], y[3]);
            cs = cos(efs); sn = sin(efs);
            torque = c.torque_coef[0] * (y[3] * y[2] - y[4] * y[1]);
            idq[0] = cs * y[1] + sn * y[2]; idq[1] = -sn * y[1] + cs * y[2];
            T32(y[1], y[2], iabc);
        }
        udq[0] = cs * uab[0] + sn * uab[1]; udq[1] = -sn * uab[0] + cs * uab[1];
        if (eps > kPi) eps -= kTwoPi;
        o[0] = y[0]; o[1] = torque; o[2] = iabc[0]; o[3] = iabc[1]; o[4] = iabc[2]; o[5] = idq[0]; o[6] = idq[1];
        o[7] = uabc[0]; o[8] = uabc[1]; o[9] = uabc[2]; o[10] = udq[0]; o[11] = udq[1]; o[12] = eps; o[13] = us;
    }
    for (int i = 0; i < h.nout; ++i) o[i] /= c.limits[i];
}

template <class R> static int launch_reset(gemx_handle *h, const uint8_t *mask, void *obs, hipStream_t st, int advance = 1) {
    using AngT = typename Angle<R>::T;
    const DevParams<R> &P = params_of<R>(h);
    int64_t blocks = (h->n + 255) / 256;
    GEMX_COV(sizeof(R) == 4 ? "reset_kernel<float>" : "reset_kernel<double>");
    hipLaunchKernelGGL(reset_kernel<R>, dim3((unsigned)blocks), dim3(256), 0, st, (R *)h->state, (AngT *)h->angle, mask, (R *)obs, h->n,
                       h->nd, h->nout, h->has_angle, h->cfg.obs_layout, P, (const R *)h->reset_obs_dev, (unsigned char *)h->ring,
                       h->cfg.action_delay > 0 ? (int)(h->ring_bytes / ((size_t)h->cfg.action_delay * (size_t)h->n)) : 0, h->cfg.system_kind,
                       (const InitDev *)h->rinit_dev, h->rcnt, advance);
    HIP_TRY(hipGetLastError());
    return GEMX_OK;
}

// ---- checkpoint of everything gemx_get_state / gemx_get_switch_state do not carry (gemx_get_aux_state) ---------------------------
// blob = header (128 bytes) | RC supply rows [2][N] R | DeadTimeProcessor ring | reset counters [N] uint32 | angle words [N] (the fp32
// build's 32-bit fixed-point angle, which gemx_get_state rounds to fp32 radians: the exact words make the restore bit for bit), sections
// padded to 16 bytes
struct AuxHeader {
    uint32_t magic, version;
    int64_t n;
    uint32_t elem_size, delay, nact_conv, supply_rc, init_kind, fifo_phase;
    uint64_t steps_total, rc_bytes, ring_bytes, rcnt_bytes;
    uint32_t system_kind, converter_kind;
    uint64_t angle_bytes;
    int64_t env_base;  // version 2: the shard's global env offset (the counter-based streams continue where they were only on the same shard)
    unsigned char pad[128 - 96];
};
static_assert(sizeof(AuxHeader) == 128, "aux header is 128 bytes");
constexpr uint32_t AUX_MAGIC = 0x55415847u;  // "GXAU"
static size_t pad16(size_t b) { return (b + 15) & ~(size_t)15; }
static void aux_sections(const gemx_handle *h, size_t &rc_b, size_t &ring_b, size_t &rcnt_b, size_t &ang_b) {
    const size_t es = h->cfg.dtype == GEMX_F64 ? 8 : 4;
    rc_b = (size_t)h->extra_rows * (size_t)h->n * es;  // RC supply rows and / or the error-controlled solver's carried step size
    ring_b = h->ring != nullptr ? h->ring_bytes : 0;
    rcnt_b = h->rcnt != nullptr ? sizeof(uint32_t) * (size_t)h->n : 0;
    ang_b = h->has_angle && h->angle != nullptr ? es * (size_t)h->n : 0;
}
__global__ void aux_header_kernel(AuxHeader hd, const uint32_t *fifo_phase, AuxHeader *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        hd.fifo_phase = fifo_phase[0];  // (lives on the device: a launch replayed from a graph advances it without the host)
        *out = hd;
    }
}

// ---- kernel units -----------------------------------------------------------------------------------------------------------------
// Round 6: the stepping kernels of ONE (system, converter unit, dtype) live in their own shared object, libgemx_u<S>_<C>_<F>.so beside this
// library (gemx_inst.hip; gym_electric_motor_amd/build.py), loaded by gemx_create with dlopen -- a handle maps this library (C ABI, reset /
// state access / reference generators: ~2 MB) and the one unit it runs (2-6 MB) instead of 144 MB of all 38.  GEMX_UNIT_DIR names another
// directory.  A unit exports gemx_unit_init (checks that it was compiled against THIS handle layout and receives the error sink) and
// gemx_unit_launch (what gemx::launch_unit_S_C_F used to be).
typedef int (*gemx_unit_launch_fn)(gemx_handle *, const void *, int, void *, uint8_t *, int, hipStream_t);
typedef int (*gemx_unit_init_fn)(unsigned long long handle_bytes, int abi, void (*set_error)(const char *));
struct UnitLib { void *dl = nullptr; gemx_unit_launch_fn launch = nullptr; };
static UnitLib g_units[8][12][2];
static std::mutex g_units_mu;
static void unit_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
static int load_unit(int s, int c, int f, gemx_unit_launch_fn *out) {
    if (s < 0 || s >= 8 || c < 0 || c >= 12) return fail(GEMX_ERR_ARG, "unsupported system/converter combination %d/%d", s, c);
    std::lock_guard<std::mutex> lock(g_units_mu);
    UnitLib &u = g_units[s][c][f];
    if (u.launch == nullptr) {
        char dir[1024] = "";
        const char *ud = getenv("GEMX_UNIT_DIR");
        if (ud != nullptr && ud[0] != 0) snprintf(dir, sizeof(dir), "%s", ud);
        else {
            Dl_info info;
            if (dladdr((const void *)&gemx_abi_version, &info) == 0 || info.dli_fname == nullptr) return fail(GEMX_ERR_DEVICE, "dladdr failed: cannot locate libgemx.so's directory");
            snprintf(dir, sizeof(dir), "%s", info.dli_fname);
            char *slash = strrchr(dir, '/');
            if (slash != nullptr) *slash = 0;
            else snprintf(dir, sizeof(dir), ".");
        }
        char path[1200];
        snprintf(path, sizeof(path), "%s/libgemx_u%d_%d_%d.so", dir, s, c, f);
        void *dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (dl == nullptr) {
            const char *e = dlerror();
            return fail(GEMX_ERR_ARG, "kernel unit %s cannot be loaded (%s): system/converter combination %d/%d is not built, or the package's build is incomplete "
                                      "(python -m gym_electric_motor_amd.build)", path, e ? e : "?", s, c);
        }
        auto init = (gemx_unit_init_fn)dlsym(dl, "gemx_unit_init");
        auto launch = (gemx_unit_launch_fn)dlsym(dl, "gemx_unit_launch");
        if (init == nullptr || launch == nullptr) { dlclose(dl); return fail(GEMX_ERR_ARG, "%s exports no gemx_unit_init / gemx_unit_launch", path); }
        const int rc = init((unsigned long long)sizeof(gemx_handle), GEMX_ABI_VERSION, unit_set_error);
        if (rc != GEMX_OK) { dlclose(dl); return fail(GEMX_ERR_ARG, "%s was built from other sources than libgemx.so (handle layout / ABI differ): rebuild the package", path); }
        u.dl = dl;
        u.launch = launch;
    }
    *out = u.launch;
    return GEMX_OK;
}

// ---- instantiation coverage (tools/instantiation_coverage.py): with GEMX_COVERAGE_FILE set, every DISTINCT kernel instantiation a process
// launches is appended to that file once, as the template arguments of the kernel symbol ("advance_pipe_kernel<1, 1, 0, 1, false, float, 12,
// 6, false, false, false>") -- diffed against the kernel symbols of the built libraries.  Off (one getenv per process) otherwise.
static std::mutex g_cov_mu;
static std::set<std::string> *g_cov_seen = nullptr;
static const char *g_cov_path = nullptr;
static bool g_cov_init = false;
void gemx_cov_note(const char *key) {
    std::lock_guard<std::mutex> lock(g_cov_mu);
    if (!g_cov_init) {
        g_cov_init = true;
        const char *p = getenv("GEMX_COVERAGE_FILE");
        if (p != nullptr && p[0] != 0) { g_cov_path = strdup(p); g_cov_seen = new std::set<std::string>(); }
    }
    if (g_cov_path == nullptr || !g_cov_seen->insert(key).second) return;
    FILE *fh = fopen(g_cov_path, "a");
    if (fh != nullptr) { fprintf(fh, "%s\n", key); fclose(fh); }
}
static void cov_note_launch(const gemx_handle *h, bool linmap_built) {
    const auto &l = h->ll;
    const char *R = l.real_size == 4 ? "float" : "double", *il = l.il ? "true" : "false";
    char key[256];
    if (l.pipe == 3) snprintf(key, sizeof(key), "dc_stream_kernel<%d, %d, %d, float, %d>", l.sys, l.conv, l.solver, l.shape);
    else if (l.pipe == 2) snprintf(key, sizeof(key), "step_kernel<%d, %d, %d, %d, %s, %s>", l.sys, l.conv, l.load, l.solver, il, R);
    else if (l.pipe == 1) {
        static const char *SH[8] = {"12, 3, false, false, false", "4, 2, false, false, false", "2, 2, false, false, false", "12, 6, false, false, false",
                                    "4, 2, true, false, false", "4, 2, false, true, false", "4, 2, true, true, false", "4, 2, true, false, true"};
        snprintf(key, sizeof(key), "advance_pipe_kernel<%d, %d, %d, %d, %s, %s, %s>", l.sys, l.conv, l.load, l.solver, il, R, SH[l.shape & 7]);
    } else snprintf(key, sizeof(key), "advance_kernel<%d, %d, %d, %d, %s, %s>", l.sys, l.conv, l.load, l.solver, il, R);
    gemx_cov_note(key);
    if (linmap_built) { snprintf(key, sizeof(key), "linmap_kernel<%d, %d, %s>", l.sys, l.solver, R); gemx_cov_note(key); }
}

static int launch_advance(gemx_handle *h, const void *actions, int K, void *obs, uint8_t *done, int obs_every, hipStream_t st) {
    if (h->unit_launch == nullptr) return fail(GEMX_ERR_ARG, "internal: the handle's kernel unit is not loaded");
    const int lin_before = h->linmap_state;
    const int rc = ((gemx_unit_launch_fn)h->unit_launch)(h, actions, K, obs, done, obs_every, st);
    if (rc != GEMX_OK) return rc;
    h->steps_total += (unsigned long long)K;  // (only what was launched: a refused launch leaves the count the checkpoint header carries alone)
    cov_note_launch(h, lin_before == 0 && h->linmap_state == 1);
    return GEMX_OK;
}

static int elem_size(const gemx_handle *h) { return h->cfg.dtype == GEMX_F64 ? 8 : 4; }

template <class R> static void build_reward(const gemx_handle *h, const gemx_reward_config *rc, RewardDev<R> &W) {
    memset(&W, 0, sizeof(W));
    int t = 0;
    bool referenced[GEMX_MAX_OUT] = {};
    auto add = [&](int col) {
        W.col[t] = col;
        const double pw = rc->power[col];
        W.kind[t] = pw == 1.0 ? 1 : (pw == 2.0 ? 2 : 3);
        W.coef[t] = (R)rc->weight[col];
        W.inv_len[t] = (R)(1.0 / rc->state_length[col]);
        W.power[t] = (R)pw;
        ++t;
    };
    for (int j = 0; j < rc->n_ref; ++j) { add(rc->ref_index[j]); referenced[rc->ref_index[j]] = true; }
    W.n_ref = rc->n_ref;
    for (int i = 0; i < h->nout; ++i)
        if (!referenced[i] && rc->weight[i] != 0.0) add(i);  // weighted but un-referenced states are compared with 0 (core.py:346)
    W.n_term = t;
    W.bias = (R)rc->bias;
    W.violation_reward = (R)rc->violation_reward;
}

extern "C" {

int gemx_abi_version(void) { return GEMX_ABI_VERSION; }
int gemx_sizeof_config(void) { return (int)sizeof(gemx_config); }
int gemx_debug_read(gemx_handle *h, unsigned long long *out, int n) { gemx::DeviceGuard guard(h->device); return (int)hipMemcpy(out, (char *)h->err + 64, n * 8, hipMemcpyDeviceToHost); }
const char *gemx_last_error(void) { return g_err; }

int gemx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gemx_create(const gemx_config *cfg, int64_t n_envs, int device, gemx_handle **out) {
    if (!cfg || !out) return fail(GEMX_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(gemx_config) || cfg->abi_version != GEMX_ABI_VERSION)
        return fail(GEMX_ERR_ARG, "gemx_config ABI mismatch (struct_size %d vs %d, abi %d vs %d)", cfg->struct_size,
                    (int)sizeof(gemx_config), cfg->abi_version, GEMX_ABI_VERSION);
    if (n_envs <= 0) return fail(GEMX_ERR_ARG, "n_envs must be positive");
    if (cfg->env_base < 0 || cfg->env_base > INT64_MAX - n_envs) return fail(GEMX_ERR_ARG, "env_base must be >= 0 (the global index of this handle's env 0)");
    if (!(cfg->tau > 0)) return fail(GEMX_ERR_ARG, "tau must be positive");
    if (cfg->interlocking_time < 0 || cfg->interlocking_time >= cfg->tau)
        return fail(GEMX_ERR_ARG, "interlocking_time must be in [0, tau)");
    if (cfg->solver_kind < GEMX_SOLVER_EULER || cfg->solver_kind > GEMX_SOLVER_DP5) return fail(GEMX_ERR_ARG, "unknown solver_kind");
    if (cfg->solver_nsteps < 1 || cfg->solver_nsteps > 1024) return fail(GEMX_ERR_ARG, "solver_nsteps must be in [1, 1024]");
    if (cfg->solver_flags & ~(GEMX_SOLVER_SPLIT_KINKS | GEMX_SOLVER_ADAPTIVE)) return fail(GEMX_ERR_ARG, "unknown solver_flags 0x%x", cfg->solver_flags);
    if ((cfg->solver_flags & GEMX_SOLVER_ADAPTIVE) && cfg->solver_kind != GEMX_SOLVER_DP5)
        return fail(GEMX_ERR_ARG, "GEMX_SOLVER_ADAPTIVE needs solver_kind GEMX_SOLVER_DP5 (the embedded error estimate is Dormand-Prince's)");
    if ((cfg->solver_flags & GEMX_SOLVER_ADAPTIVE) && (cfg->solver_rtol < 0 || cfg->solver_atol < 0 || cfg->solver_rtol > 0.1 || cfg->solver_atol_omega < 0))
        return fail(GEMX_ERR_ARG, "solver_rtol must be in [0, 0.1], solver_atol >= 0 and solver_atol_omega >= 0 (0 = default)");
    if (cfg->dtype != GEMX_F32 && cfg->dtype != GEMX_F64) return fail(GEMX_ERR_ARG, "unknown dtype");
    if (cfg->obs_layout != GEMX_OBS_AOS && cfg->obs_layout != GEMX_OBS_SOA) return fail(GEMX_ERR_ARG, "unknown obs_layout");
    if (cfg->load_kind != GEMX_LOAD_CONST_SPEED && cfg->load_kind != GEMX_LOAD_POLY_STATIC) return fail(GEMX_ERR_ARG, "unknown load_kind");
    if (cfg->load_kind == GEMX_LOAD_POLY_STATIC && !(cfg->j_total > 0 && cfg->tau_decay > 0))
        return fail(GEMX_ERR_ARG, "PolynomialStaticLoad needs j_total > 0 and tau_decay > 0");
    const int s = cfg->system_kind, c = cfg->converter_kind;
    const bool dc_sys = s == GEMX_SYS_DC_PERMEX || s == GEMX_SYS_DC_SERIES || s == GEMX_SYS_DC_SHUNT;
    const bool combo = (dc_sys && (c == GEMX_CONV_CONT_4QC || c == GEMX_CONV_FINITE_4QC)) ||
                       ((s == GEMX_SYS_SYNC || s == GEMX_SYS_SCIM) && (c == GEMX_CONV_FINITE_B6 || c == GEMX_CONV_CONT_B6)) ||
                       (s == GEMX_SYS_DC_EXTEX && (c == GEMX_CONV_CONT_2X4QC || c == GEMX_CONV_FINITE_2X4QC)) ||
                       (s == GEMX_SYS_EESM && (c == GEMX_CONV_CONT_B6_4QC || c == GEMX_CONV_FINITE_B6_4QC)) ||
                       (s == GEMX_SYS_DFIM && (c == GEMX_CONV_CONT_2XB6 || c == GEMX_CONV_FINITE_2XB6));
    if (!combo) return fail(GEMX_ERR_ARG, "unsupported system/converter combination %d/%d", s, c);
    if (cfg->action_delay < 0 || cfg->action_delay > GEMX_MAX_DELAY)
        return fail(GEMX_ERR_ARG, "action_delay must be in [0, %d]", GEMX_MAX_DELAY);
    if (cfg->action_frame != GEMX_ACT_ABC) {
        const bool space_ok = cfg->action_frame == GEMX_ACT_DQ_SPACE && (s == GEMX_SYS_SYNC || s == GEMX_SYS_SCIM) && c == GEMX_CONV_CONT_B6;
        const bool proc_ok = cfg->action_frame == GEMX_ACT_DQ_PROCESSOR &&
                             ((s == GEMX_SYS_SYNC && c == GEMX_CONV_CONT_B6) || (s == GEMX_SYS_EESM && c == GEMX_CONV_CONT_B6_4QC));
        if (!space_ok && !proc_ok)
            return fail(GEMX_ERR_ARG, "action_frame %d needs a continuous B6 converter on a synchronous / EESM system (control_space='dq' "
                                      "also SCIM); the SCIM / DFIM dq processors need a flux observer, which is not on the accelerated path",
                        cfg->action_frame);
    }
    if (cfg->init_kind < GEMX_INIT_CONST || cfg->init_kind > GEMX_INIT_GAUSSIAN) return fail(GEMX_ERR_ARG, "unknown init_kind");
    if (cfg->action_delay > 0) {
        const bool disc = c == GEMX_CONV_FINITE_B6 || c == GEMX_CONV_FINITE_4QC || c == GEMX_CONV_FINITE_2X4QC || c == GEMX_CONV_FINITE_B6_4QC ||
                          c == GEMX_CONV_FINITE_2XB6;
        const int nflat = c == GEMX_CONV_FINITE_B6 ? 8 : (c == GEMX_CONV_FINITE_4QC ? 4 : (c == GEMX_CONV_FINITE_2X4QC ? 16 : (c == GEMX_CONV_FINITE_B6_4QC ? 32 : 64)));
        for (int i = 0; i < 6; ++i) {
            const double v = cfg->action_delay_reset[i];
            if (!std::isfinite(v)) return fail(GEMX_ERR_ARG, "action_delay_reset[%d] is not finite", i);
            if (disc && (i > 0 ? v != 0.0 : (v < 0.0 || v >= (double)nflat || v != std::floor(v))))
                return fail(GEMX_ERR_ARG, "action_delay_reset: a discrete converter takes ONE flat action index in [0, %d) in entry 0 (got [%d] = %g)", nflat, i, v);
        }
    }
    if (cfg->init_flux_mode != 0 && !(cfg->init_flux_mode == 1 && (s == GEMX_SYS_SCIM || s == GEMX_SYS_DFIM) && cfg->init_kind != GEMX_INIT_CONST))
        return fail(GEMX_ERR_ARG, "init_flux_mode is 0, or 1 for a SCIM / DFIM system with a random init_kind");
    if (cfg->init_flux_mode == 1 && !(cfg->init_flux[1] > 0 && cfg->init_flux[5] > 0))
        return fail(GEMX_ERR_ARG, "init_flux_mode = 1 needs init_flux[1] (pole pairs) and init_flux[5] (l_m / l_r) positive");
    if (cfg->init_kind != GEMX_INIT_CONST) {
        const int n_ode = (s == GEMX_SYS_DC_PERMEX || s == GEMX_SYS_DC_SERIES) ? 2 : ((s == GEMX_SYS_DC_SHUNT || s == GEMX_SYS_DC_EXTEX) ? 3 :
                          (s == GEMX_SYS_SYNC ? 4 : (s == GEMX_SYS_EESM ? 5 : 6)));
        for (int j = 0; j < n_ode; ++j) {
            if (cfg->init_lo[j] > cfg->init_hi[j]) return fail(GEMX_ERR_ARG, "init_lo[%d] > init_hi[%d]", j, j);
            if (cfg->init_lo[j] < cfg->init_hi[j]) {
                if (cfg->init_kind == GEMX_INIT_GAUSSIAN && !(cfg->init_sigma[j] > 0)) return fail(GEMX_ERR_ARG, "init_sigma[%d] must be positive", j);
                if (j > 0 && (s == GEMX_SYS_SCIM || s == GEMX_SYS_DFIM) && cfg->init_flux_mode != 1)
                    return fail(GEMX_ERR_ARG, "random MOTOR initial states of an induction-motor system need init_flux_mode = 1 (the flux "
                                              "bounds are re-derived at every reset: induction_motor.py:250-285)");
                if (j == 0 && cfg->load_kind == GEMX_LOAD_CONST_SPEED)
                    return fail(GEMX_ERR_ARG, "a ConstantSpeedLoad has no random initial omega (constant_speed_load.py:30-38)");
            }
        }
    }
    if (cfg->supply_kind != GEMX_SUPPLY_IDEAL && cfg->supply_kind != GEMX_SUPPLY_RC) return fail(GEMX_ERR_ARG, "unknown supply_kind");
    if (cfg->supply_kind == GEMX_SUPPLY_RC) {
        if (!(cfg->supply_r > 0 && cfg->supply_c > 0)) return fail(GEMX_ERR_ARG, "RC supply needs supply_r > 0 and supply_c > 0");
        if (c == GEMX_CONV_FINITE_B6_4QC)
            return fail(GEMX_ERR_ARG, "the RC supply is not available for the finite EESM converter (its leg states are not tracked)");
    }
    if (s == GEMX_SYS_EESM && cfg->interlocking_time > 0)
        return fail(GEMX_ERR_ARG, "interlocking_time > 0 is not supported for the EESM system (the reference's dead-time branch, "
                                  "physical_systems.py:628-638, cannot execute either)");

    gemx_handle *h = new (std::nothrow) gemx_handle();
    if (!h) return fail(GEMX_ERR_ALLOC, "out of host memory");
    h->cfg = *cfg;
    h->n = n_envs;
    h->device = device;
    switch (s) {  // ODE rows without the angle / observation length (SysTraits in gemx_common.hpp)
        case GEMX_SYS_DC_PERMEX: case GEMX_SYS_DC_SERIES: h->nd = 2; h->nout = 5; break;
        case GEMX_SYS_DC_SHUNT: h->nd = 3; h->nout = 6; break;
        case GEMX_SYS_DC_EXTEX: h->nd = 3; h->nout = 7; break;
        case GEMX_SYS_SYNC: h->nd = 3; h->nout = 14; break;
        case GEMX_SYS_EESM: h->nd = 4; h->nout = 16; break;
        case GEMX_SYS_DFIM: h->nd = 5; h->nout = 24; break;
        default: h->nd = 5; h->nout = 14; break;
    }
    h->has_angle = !(dc_sys || s == GEMX_SYS_DC_EXTEX);
    h->nact = c == GEMX_CONV_CONT_B6 ? 3 : (c == GEMX_CONV_CONT_2X4QC ? 2 : (c == GEMX_CONV_CONT_B6_4QC ? 4 : (c == GEMX_CONV_CONT_2XB6 ? 6 : 1)));
    h->sw_rows = c == GEMX_CONV_FINITE_2XB6 ? 2 : 1;
    h->nact_conv = h->nact;
    h->conv_unit = c;
    if (cfg->action_frame != GEMX_ACT_ABC) {  // the caller passes (u_d, u_q[, u_e]); the kernel unit is the internal dq kind
        h->nact = h->nact_conv - 1;
        h->conv_unit = c == GEMX_CONV_CONT_B6 ? CONV_CONT_B6_DQ : CONV_CONT_B6_4QC_DQ;
    }
    for (int i = 0; i < h->nout; ++i)
        if (!(cfg->limits[i] > 0) || !std::isfinite(cfg->limits[i])) { delete h; return fail(GEMX_ERR_ARG, "limits[%d] must be positive and finite", i); }
    if ((cfg->limit_mask | cfg->squared_mask) >> h->nout) { delete h; return fail(GEMX_ERR_ARG, "constraint mask has bits beyond S_out=%d", h->nout); }

    double m[20] = {0}, pole = 0;
    int rc = pack_model(*cfg, m, &pole);
    if (rc != GEMX_OK) { delete h; return rc; }

    // all argument validation is done; from here on a device is required
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        delete h;
        return fail(GEMX_ERR_DEVICE, "no HIP device visible: the gemx stepper has no CPU fallback");
    }
    if (device < 0 || device >= ndev) { delete h; return fail(GEMX_ERR_ARG, "device %d out of range (0..%d)", device, ndev - 1); }
    gemx::DeviceGuard guard(device);  // the caller's current device is restored on every return path
    {  // this handle's kernel unit (dlopen on first use in the process)
        gemx_unit_launch_fn fn = nullptr;
        rc = load_unit(cfg->system_kind, h->conv_unit, cfg->dtype == GEMX_F64 ? 1 : 0, &fn);
        if (rc != GEMX_OK) { delete h; return rc; }
        h->unit_launch = (void *)fn;
    }
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device) { delete h; return fail(GEMX_ERR_DEVICE, "hipSetDevice(%d) failed", device); }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
            if (prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
            if (prop.sharedMemPerBlock >= 64 * 1024) h->lds_max = prop.sharedMemPerBlock;
            // The rate limiter's built-in targets are GB/s figures of ONE part in ONE mode: a whole MI355X (gfx950, 256 CUs, ~8 TB/s).  On
            // anything else -- another arch, a partitioned device (CPX / DPX: fewer CUs per logical device) -- they mean nothing: the
            // limiter starts switched off there (advisor finding, round 4; GEMX_PACE_GBPS still turns it on with an explicit target).
            if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 || prop.multiProcessorCount != 256) h->pace_gbps = 0.0;
        }
        // every switch below changes which kernel a PRODUCT call runs: whatever is set is recorded and shows up in gemx_last_launch(), so that
        // a stray variable cannot silently change a benchmark (round 3 verdict)
        for (const char *name : {"GEMX_STEPS_PER_BLOCK", "GEMX_PIPE", "GEMX_PIPE_SHAPE", "GEMX_STEP_KERNEL", "GEMX_DC_STREAM", "GEMX_DCS_EPW", "GEMX_LINMAP", "GEMX_PACE_GBPS", "GEMX_PACE_CAL"}) {
            const char *v = getenv(name);
            if (v == nullptr) continue;
            const size_t used = strlen(h->overrides);
            snprintf(h->overrides + used, sizeof(h->overrides) - used, "%s%s=%.16s", used ? " " : "", name, v);
        }
        const char *ev = getenv("GEMX_STEPS_PER_BLOCK");
        if (ev) h->steps_per_block = atoi(ev);
        ev = getenv("GEMX_PIPE");
        if (ev) h->use_pipe = atoi(ev);
        ev = getenv("GEMX_PIPE_SHAPE");
        if (ev) h->pipe_shape = atoi(ev);
        ev = getenv("GEMX_STEP_KERNEL");
        if (ev) h->use_step_kernel = atoi(ev);
        ev = getenv("GEMX_DC_STREAM");  // 0: never take dc_stream_kernel; 2: at any N (A/B runs and bit-identity tests); 3: also without the
                                        // host-side "omega is at its initial value" knowledge (test of the kernel's own check of that premise)
        if (ev) h->use_dc_stream = atoi(ev);
        ev = getenv("GEMX_DCS_EPW");  // 64: dc_stream_kernel with one env per lane at every size (A/B runs; default: 32 envs per workgroup where they find a CU each)
        if (ev) h->dcs_epw = atoi(ev);
        ev = getenv("GEMX_PACE_GBPS");  // target rate of the large-batch rate limiter in GB/s (A/B runs); 0: off
        if (ev) h->pace_gbps = atof(ev);
        ev = getenv("GEMX_PACE_CAL");  // 0: the rate limiter keeps its built-in target (no closed-loop calibration)
        if (ev) h->pace_cal_on = atoi(ev);
        ev = getenv("GEMX_LINMAP");  // 0: never use the one-step map of the electrical subsystem (A/B runs)
        if (ev && atoi(ev) == 0) h->linmap_state = -1;

    }
    host_reset_obs(*h, m);

    const size_t es = (size_t)elem_size(h);
    auto cleanup = [&](int code) { gemx_destroy(h); return code; };
    // rows: the ODE states | RCVoltageSupply: capacitor voltage, time since its last update (rows nd, nd + 1) | error-controlled solver: the carried
    // step size (row nd + 2, whether or not the two before it are in use)
    h->extra_rows = (cfg->solver_flags & GEMX_SOLVER_ADAPTIVE) ? 3 : (cfg->supply_kind == GEMX_SUPPLY_RC ? 2 : 0);
    if (hipMalloc(&h->state, es * (h->nd + h->extra_rows) * (size_t)h->n) != hipSuccess) return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(state) failed"));
    if (h->extra_rows && hipMemset((char *)h->state + es * (size_t)h->nd * (size_t)h->n, 0, es * (size_t)h->extra_rows * (size_t)h->n) != hipSuccess)
        return cleanup(fail(GEMX_ERR_DEVICE, "hipMemset failed"));
    if (h->has_angle && hipMalloc(&h->angle, (cfg->dtype == GEMX_F64 ? 8 : 4) * (size_t)h->n) != hipSuccess)
        return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(angle) failed"));
    if (hipMalloc((void **)&h->sw, (size_t)h->n * h->sw_rows) != hipSuccess) return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(sw) failed"));
    if (cfg->action_delay > 0) {
        const bool disc = c == GEMX_CONV_FINITE_B6 || c == GEMX_CONV_FINITE_4QC || c == GEMX_CONV_FINITE_2X4QC || c == GEMX_CONV_FINITE_B6_4QC ||
                          c == GEMX_CONV_FINITE_2XB6;
        h->ring_bytes = (size_t)cfg->action_delay * (size_t)h->n * (disc ? 1 : (size_t)h->nact_conv * es);
        if (hipMalloc(&h->ring, h->ring_bytes) != hipSuccess) return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(delay ring) failed"));
    }
    if (cfg->init_kind != GEMX_INIT_CONST) {
        InitDev I;
        memset(&I, 0, sizeof(I));
        I.kind = cfg->init_kind;
        I.n = h->nd + h->has_angle;
        I.seed = cfg->seed;
        I.env_base = cfg->env_base;
        for (int j = 0; j < I.n; ++j) {
            I.lo[j] = cfg->init_lo[j]; I.hi[j] = cfg->init_hi[j]; I.mu[j] = cfg->init_mu[j]; I.sigma[j] = cfg->init_sigma[j];
            I.constant[j] = cfg->init_state[j];
            if (cfg->init_kind == GEMX_INIT_GAUSSIAN && I.lo[j] < I.hi[j]) {
                I.cdf_lo[j] = 0.5 * erfc(-(I.lo[j] - I.mu[j]) / I.sigma[j] / sqrt(2.0));
                I.cdf_hi[j] = 0.5 * erfc(-(I.hi[j] - I.mu[j]) / I.sigma[j] / sqrt(2.0));
            }
        }
        I.flux_mode = cfg->init_flux_mode;
        I.flux_slot = 3;  // [omega, i_s alpha, i_s beta, psi_r alpha, psi_r beta, epsilon]
        for (int j = 0; j < 8; ++j) I.flux[j] = cfg->init_flux[j];
        if (hipMalloc(&h->rinit_dev, sizeof(InitDev)) != hipSuccess || hipMalloc((void **)&h->rcnt, sizeof(uint32_t) * (size_t)h->n) != hipSuccess)
            return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(random initialiser) failed"));
        if (hipMemcpy(h->rinit_dev, &I, sizeof(I), hipMemcpyHostToDevice) != hipSuccess || hipMemset(h->rcnt, 0, sizeof(uint32_t) * (size_t)h->n) != hipSuccess)
            return cleanup(fail(GEMX_ERR_DEVICE, "hipMemcpy failed"));
    }
    // four maps (whole step; the two dead-time segments and the whole step again in the D form: linmap_kernel) of <= 32 coefficients each
    if (hipMalloc(&h->linmap_dev, sizeof(double) * 3 * 64) != hipSuccess) return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(linmap) failed"));
    if (hipMalloc((void **)&h->err, 4096) != hipSuccess) return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(err) failed"));
    if (hipMalloc((void **)&h->fifo_phase, 64) != hipSuccess || hipMemset(h->fifo_phase, 0, 64) != hipSuccess)
        return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(fifo phase) failed"));
    if (hipMalloc(&h->reset_obs_dev, es * GEMX_MAX_OUT) != hipSuccess) return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(reset_obs) failed"));
    if (hipMalloc(&h->cw_dev, es * 2 * GEMX_MAX_OUT) != hipSuccess) return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(cw) failed"));
    {
        double wd[2 * GEMX_MAX_OUT];
        float wf[2 * GEMX_MAX_OUT];
        for (int i = 0; i < GEMX_MAX_OUT; ++i) {
            wd[i] = (double)((cfg->limit_mask >> i) & 1u);
            wd[GEMX_MAX_OUT + i] = (double)((cfg->squared_mask >> i) & 1u);
            wf[i] = (float)wd[i];
            wf[GEMX_MAX_OUT + i] = (float)wd[GEMX_MAX_OUT + i];
        }
        const void *src = cfg->dtype == GEMX_F64 ? (const void *)wd : (const void *)wf;
        if (hipMemcpy(h->cw_dev, src, es * 2 * GEMX_MAX_OUT, hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(fail(GEMX_ERR_DEVICE, "hipMemcpy failed"));
    }
    fill_params<float>(*h, m, pole, h->pf);
    fill_params<double>(*h, m, pole, h->pd);
    if (hipMemset(h->sw, 0, (size_t)h->n * h->sw_rows) != hipSuccess || hipMemset(h->err, 0, 4096) != hipSuccess)
        return cleanup(fail(GEMX_ERR_DEVICE, "hipMemset failed"));
    if (cfg->dtype == GEMX_F64) {
        if (hipMemcpy(h->reset_obs_dev, h->reset_obs, sizeof(double) * GEMX_MAX_OUT, hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(fail(GEMX_ERR_DEVICE, "hipMemcpy failed"));
    } else {
        float tmp[GEMX_MAX_OUT];
        for (int i = 0; i < GEMX_MAX_OUT; ++i) tmp[i] = (float)h->reset_obs[i];
        if (hipMemcpy(h->reset_obs_dev, tmp, sizeof(tmp), hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(fail(GEMX_ERR_DEVICE, "hipMemcpy failed"));
    }
    rc = gemx_reset(h, nullptr, nullptr, nullptr);
    if (rc != GEMX_OK) return cleanup(rc);
    if (hipDeviceSynchronize() != hipSuccess) return cleanup(fail(GEMX_ERR_DEVICE, "hipDeviceSynchronize failed"));
    // random initialisers: the envs now hold draw #1 and their counters say so (1).  A caller that steps with auto_reset straight away
    // gets draw #2 at an env's first in-kernel reset, its first gemx_reset is draw #2 as well.  (Rounds 3-4 put the counters back to 0
    // here so that a binding's construction-time reset -- made only to obtain the reset observation rows -- would not shift the seeded
    // sequence; but then a C caller who never calls gemx_reset started episodes 1 AND 2 from the same state: advisor finding, round 4.
    // A binding now asks for those rows with gemx_reset_again, which re-creates draw #1 without advancing.)
    *out = h;
    return GEMX_OK;
}

int gemx_destroy(gemx_handle *h) {
    if (!h) return GEMX_OK;
    gemx::DeviceGuard guard(h->device);
    if (h->state) (void)hipFree(h->state);
    if (h->angle) (void)hipFree(h->angle);
    if (h->sw) (void)hipFree(h->sw);
    if (h->ring) (void)hipFree(h->ring);
    if (h->rw_dev) (void)hipFree(h->rw_dev);
    if (h->rinit_dev) (void)hipFree(h->rinit_dev);
    if (h->linmap_dev) (void)hipFree(h->linmap_dev);
    if (h->rcnt) (void)hipFree(h->rcnt);
    if (h->err) (void)hipFree(h->err);
    if (h->fifo_phase) (void)hipFree(h->fifo_phase);
    if (h->reset_obs_dev) (void)hipFree(h->reset_obs_dev);
    if (h->cw_dev) (void)hipFree(h->cw_dev);
    if (h->pcal.ev_init)
        for (int i = 0; i < gemx_handle::PaceCal::RING; ++i) {
            if (h->pcal.ev0[i]) (void)hipEventDestroy((hipEvent_t)h->pcal.ev0[i]);
            if (h->pcal.ev1[i]) (void)hipEventDestroy((hipEvent_t)h->pcal.ev1[i]);
        }
    delete h;
    return GEMX_OK;
}

int gemx_n_envs(const gemx_handle *h, int64_t *n) {
    if (!h || !n) return fail(GEMX_ERR_ARG, "null argument");
    *n = h->n;
    return GEMX_OK;
}
int gemx_n_ode(const gemx_handle *h) { return h ? h->nd + h->has_angle : GEMX_ERR_ARG; }
int gemx_n_out(const gemx_handle *h) { return h ? h->nout : GEMX_ERR_ARG; }
int gemx_n_action(const gemx_handle *h) { return h ? h->nact : GEMX_ERR_ARG; }
int gemx_action_itemsize(const gemx_handle *h) {
    if (!h) return GEMX_ERR_ARG;
    const int c = h->cfg.converter_kind;
    const bool discrete = c == GEMX_CONV_FINITE_B6 || c == GEMX_CONV_FINITE_4QC || c == GEMX_CONV_FINITE_2X4QC || c == GEMX_CONV_FINITE_B6_4QC ||
                          c == GEMX_CONV_FINITE_2XB6;
    return discrete ? 1 : elem_size(h);
}
int gemx_n_switch_bytes(const gemx_handle *h) { return h ? h->sw_rows : GEMX_ERR_ARG; }
int gemx_reset_observation(const gemx_handle *h, double *obs_host) {
    if (!h || !obs_host) return fail(GEMX_ERR_ARG, "null argument");
    memcpy(obs_host, h->reset_obs, sizeof(double) * h->nout);
    return GEMX_OK;
}

int gemx_reset(gemx_handle *h, const uint8_t *mask_dev, void *obs_out_dev, void *stream) {
    if (!h) return fail(GEMX_ERR_ARG, "null handle");
    gemx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    const int rc = h->cfg.dtype == GEMX_F64 ? launch_reset<double>(h, mask_dev, obs_out_dev, st) : launch_reset<float>(h, mask_dev, obs_out_dev, st);
    // host-side knowledge "every env's omega is init[0]" (dc_stream_kernel's premise).  A call that is being CAPTURED into a graph runs
    // later, any number of times: from then on the host knows nothing (the kernels that decide on the device serve the handle)
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &capturing);
    if (capturing != hipStreamCaptureStatusNone) { h->omega_unknown = true; h->omega_is_init = false; }
    else if (rc == GEMX_OK && mask_dev == nullptr && h->cfg.init_kind == GEMX_INIT_CONST && !h->omega_unknown) h->omega_is_init = true;
    return rc;
}

int gemx_reset_again(gemx_handle *h, const uint8_t *mask_dev, void *obs_out_dev, void *stream) {
    if (!h) return fail(GEMX_ERR_ARG, "null handle");
    gemx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    const int rc = h->cfg.dtype == GEMX_F64 ? launch_reset<double>(h, mask_dev, obs_out_dev, st, 0) : launch_reset<float>(h, mask_dev, obs_out_dev, st, 0);
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &capturing);
    if (capturing != hipStreamCaptureStatusNone) { h->omega_unknown = true; h->omega_is_init = false; }
    else if (rc == GEMX_OK && mask_dev == nullptr && h->cfg.init_kind == GEMX_INIT_CONST && !h->omega_unknown) h->omega_is_init = true;
    return rc;
}

int64_t gemx_aux_state_bytes(const gemx_handle *h) {
    if (!h) return GEMX_ERR_ARG;
    size_t a, b, c, d;
    aux_sections(h, a, b, c, d);
    return (int64_t)(sizeof(AuxHeader) + pad16(a) + pad16(b) + pad16(c) + pad16(d));
}
int gemx_get_aux_state(gemx_handle *h, void *blob_out_dev, void *stream) {
    if (!h || !blob_out_dev) return fail(GEMX_ERR_ARG, "null argument");
    if (((uintptr_t)blob_out_dev & 15u) != 0) return fail(GEMX_ERR_ARG, "blob_out_dev must be 16-byte aligned");
    gemx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    size_t rc_b, ring_b, rcnt_b, ang_b;
    aux_sections(h, rc_b, ring_b, rcnt_b, ang_b);
    AuxHeader hd;
    memset(&hd, 0, sizeof(hd));
    hd.angle_bytes = ang_b;
    hd.magic = AUX_MAGIC; hd.version = 2; hd.env_base = h->cfg.env_base; hd.n = h->n; hd.elem_size = (uint32_t)elem_size(h); hd.delay = (uint32_t)h->cfg.action_delay;
    hd.nact_conv = (uint32_t)h->nact_conv; hd.supply_rc = h->cfg.supply_kind == GEMX_SUPPLY_RC; hd.init_kind = (uint32_t)h->cfg.init_kind;
    hd.steps_total = h->steps_total; hd.rc_bytes = rc_b; hd.ring_bytes = ring_b; hd.rcnt_bytes = rcnt_b;
    hd.system_kind = (uint32_t)h->cfg.system_kind; hd.converter_kind = (uint32_t)h->cfg.converter_kind;
    // (the 16-byte padding between the sections is part of the blob: zeroed here, so that blobs of equal states are equal bytes for a C
    // caller that hashes or compares them -- advisor finding, round 5; the Python binding used to pre-zero its buffer for that)
    HIP_TRY(hipMemsetAsync(blob_out_dev, 0, (size_t)gemx_aux_state_bytes(h), st));
    GEMX_COV("aux_header_kernel");
    hipLaunchKernelGGL(aux_header_kernel, dim3(1), dim3(64), 0, st, hd, (const uint32_t *)h->fifo_phase, (AuxHeader *)blob_out_dev);
    HIP_TRY(hipGetLastError());
    unsigned char *p = (unsigned char *)blob_out_dev + sizeof(AuxHeader);
    if (rc_b) HIP_TRY(hipMemcpyAsync(p, (const unsigned char *)h->state + (size_t)h->nd * (size_t)h->n * elem_size(h), rc_b, hipMemcpyDeviceToDevice, st));
    p += pad16(rc_b);
    if (ring_b) HIP_TRY(hipMemcpyAsync(p, h->ring, ring_b, hipMemcpyDeviceToDevice, st));
    p += pad16(ring_b);
    if (rcnt_b) HIP_TRY(hipMemcpyAsync(p, h->rcnt, rcnt_b, hipMemcpyDeviceToDevice, st));
    p += pad16(rcnt_b);
    if (ang_b) HIP_TRY(hipMemcpyAsync(p, h->angle, ang_b, hipMemcpyDeviceToDevice, st));
    return GEMX_OK;
}
int gemx_set_aux_state(gemx_handle *h, const void *blob_in_dev, void *stream) {
    if (!h || !blob_in_dev) return fail(GEMX_ERR_ARG, "null argument");
    gemx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &capturing);
    if (capturing != hipStreamCaptureStatusNone) return fail(GEMX_ERR_ARG, "gemx_set_aux_state validates the blob on the host: not inside a stream capture");
    AuxHeader hd;
    HIP_TRY(hipMemcpyAsync(&hd, blob_in_dev, sizeof(hd), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    size_t rc_b, ring_b, rcnt_b, ang_b;
    aux_sections(h, rc_b, ring_b, rcnt_b, ang_b);
    if (hd.magic != AUX_MAGIC || hd.version != 2) return fail(GEMX_ERR_ARG, "not a gemx aux-state blob (magic %08x, version %u)", hd.magic, hd.version);
    if (hd.n != h->n || hd.elem_size != (uint32_t)elem_size(h) || hd.delay != (uint32_t)h->cfg.action_delay || hd.nact_conv != (uint32_t)h->nact_conv ||
        hd.supply_rc != (uint32_t)(h->cfg.supply_kind == GEMX_SUPPLY_RC) || hd.init_kind != (uint32_t)h->cfg.init_kind || hd.rc_bytes != rc_b ||
        hd.ring_bytes != ring_b || hd.rcnt_bytes != rcnt_b || hd.angle_bytes != ang_b || hd.system_kind != (uint32_t)h->cfg.system_kind || hd.converter_kind != (uint32_t)h->cfg.converter_kind || hd.env_base != h->cfg.env_base)
        return fail(GEMX_ERR_ARG, "aux-state blob was taken from a handle of another configuration (n_envs %lld vs %lld, env_base %lld vs %lld, dtype, DeadTimeProcessor steps, supply or initialiser kind differ)",
                    (long long)hd.n, (long long)h->n, (long long)hd.env_base, (long long)h->cfg.env_base);
    if (h->cfg.action_delay > 0 && hd.fifo_phase >= (uint32_t)h->cfg.action_delay) return fail(GEMX_ERR_ARG, "aux-state blob: FIFO phase %u out of range", hd.fifo_phase);
    const unsigned char *p = (const unsigned char *)blob_in_dev + sizeof(AuxHeader);
    if (rc_b) HIP_TRY(hipMemcpyAsync((unsigned char *)h->state + (size_t)h->nd * (size_t)h->n * elem_size(h), p, rc_b, hipMemcpyDeviceToDevice, st));
    p += pad16(rc_b);
    if (ring_b) HIP_TRY(hipMemcpyAsync(h->ring, p, ring_b, hipMemcpyDeviceToDevice, st));
    p += pad16(ring_b);
    if (rcnt_b) HIP_TRY(hipMemcpyAsync(h->rcnt, p, rcnt_b, hipMemcpyDeviceToDevice, st));
    p += pad16(rcnt_b);
    if (ang_b) HIP_TRY(hipMemcpyAsync(h->angle, p, ang_b, hipMemcpyDeviceToDevice, st));
    const uint32_t ph[2] = {hd.fifo_phase, 0u};
    HIP_TRY(hipMemcpyAsync(h->fifo_phase, ph, sizeof(ph), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // (`ph` is a stack buffer)
    h->steps_total = hd.steps_total;
    return GEMX_OK;
}

int gemx_rollout(gemx_handle *h, const void *actions_dev, int32_t K, void *obs_out_dev, uint8_t *done_out_dev, int32_t obs_every,
                 void *stream) {
    if (!h) return fail(GEMX_ERR_ARG, "null handle");
    if (!actions_dev || !obs_out_dev) return fail(GEMX_ERR_ARG, "actions_dev and obs_out_dev must not be null");
    if (K < 1) return fail(GEMX_ERR_ARG, "K must be >= 1");
    if (((uintptr_t)obs_out_dev & 15u) != 0) return fail(GEMX_ERR_ARG, "obs_out_dev must be 16-byte aligned");
    gemx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    return launch_advance(h, actions_dev, K, obs_out_dev, done_out_dev, obs_every ? 1 : 0, st);
}

// number of discrete actions of the handle's converter (0: continuous)
static int n_discrete_actions(const gemx_handle *h) {
    switch (h->cfg.converter_kind) {
        case GEMX_CONV_FINITE_B6: return 8;
        case GEMX_CONV_FINITE_4QC: return 4;
        case GEMX_CONV_FINITE_2X4QC: return 16;
        case GEMX_CONV_FINITE_B6_4QC: return 32;
        case GEMX_CONV_FINITE_2XB6: return 64;
        default: return 0;
    }
}
int gemx_synthetic_actions(gemx_handle *h, uint64_t seed, uint32_t step0, int32_t K, void *actions_out_dev, void *stream) {
    if (!h || !actions_out_dev) return fail(GEMX_ERR_ARG, "null argument");
    if (K < 1) return fail(GEMX_ERR_ARG, "K must be >= 1");
    gemx::DeviceGuard guard(h->device);
    const int64_t total = (int64_t)K * h->n;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    GEMX_COV(h->cfg.dtype == GEMX_F64 ? "synth_actions_kernel<double>" : "synth_actions_kernel<float>");
    if (h->cfg.dtype == GEMX_F64)
        hipLaunchKernelGGL(synth_actions_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char *)actions_out_dev, h->n, K, h->nact, n_discrete_actions(h), seed, step0, h->cfg.env_base);
    else
        hipLaunchKernelGGL(synth_actions_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char *)actions_out_dev, h->n, K, h->nact, n_discrete_actions(h), seed, step0, h->cfg.env_base);
    HIP_TRY(hipGetLastError());
    return GEMX_OK;
}
int gemx_set_rate_limiter(gemx_handle *h, int32_t mode, double target_gbps) {
    if (!h) return fail(GEMX_ERR_ARG, "null handle");
    if (mode < 0 || mode > 2) return fail(GEMX_ERR_ARG, "gemx_set_rate_limiter: mode must be 0 (off), 1 (open loop) or 2 (closed loop)");
    if (!(target_gbps == target_gbps) || target_gbps > 1.0e6) return fail(GEMX_ERR_ARG, "gemx_set_rate_limiter: target_gbps out of range");
    // pace_gbps: < 0 the built-in target, 0 off, > 0 an explicit target; calibration only around the built-in one (launch_advance)
    h->pace_gbps = mode == 0 ? 0.0 : (target_gbps > 0.0 ? target_gbps : -1.0);
    h->pace_cal_on = mode == 2 ? 1 : 0;
    {
        gemx::DeviceGuard guard(h->device);
        for (int i = 0; i < gemx_handle::PaceCal::RING; ++i) {  // (recreated on demand)
            if (h->pcal.ev0[i]) (void)hipEventDestroy((hipEvent_t)h->pcal.ev0[i]);
            if (h->pcal.ev1[i]) (void)hipEventDestroy((hipEvent_t)h->pcal.ev1[i]);
        }
        (void)hipGetLastError();
    }
    h->pcal = gemx_handle::PaceCal();
    h->pace_cal_state = 0;
    h->pace_scale_last = 1.0;
    return GEMX_OK;
}
int gemx_rollout_synthetic(gemx_handle *h, uint64_t seed, uint32_t step0, int32_t K, void *obs_out_dev, uint8_t *done_out_dev, void *stream) {
    if (!h) return fail(GEMX_ERR_ARG, "null handle");
    if (!obs_out_dev) return fail(GEMX_ERR_ARG, "obs_out_dev must not be null");
    if (K < 2) return fail(GEMX_ERR_ARG, "gemx_rollout_synthetic: K must be >= 2 (a single step takes gemx_synthetic_actions + gemx_step)");
    if (((uintptr_t)obs_out_dev & 15u) != 0) return fail(GEMX_ERR_ARG, "obs_out_dev must be 16-byte aligned");
    if (h->cfg.dtype == GEMX_F64) return fail(GEMX_ERR_ARG, "gemx_rollout_synthetic: fp32 handles only (the fp64 build runs gemx_synthetic_actions + gemx_rollout)");
    gemx::DeviceGuard guard(h->device);
    h->cur_synth = true;
    h->cur_seed = seed;
    h->cur_step0 = step0;
    const int rc = launch_advance(h, nullptr, K, obs_out_dev, done_out_dev, 1, (hipStream_t)stream);
    h->cur_synth = false;
    return rc;
}

int gemx_rollout_half(gemx_handle *h, const void *actions_half_dev, int32_t K, void *obs_out_dev, uint8_t *done_out_dev, void *stream) {
    if (!h) return fail(GEMX_ERR_ARG, "null handle");
    if (!actions_half_dev || !obs_out_dev) return fail(GEMX_ERR_ARG, "actions_half_dev and obs_out_dev must not be null");
    if (K < 2) return fail(GEMX_ERR_ARG, "gemx_rollout_half: K must be >= 2");
    if (((uintptr_t)obs_out_dev & 15u) != 0) return fail(GEMX_ERR_ARG, "obs_out_dev must be 16-byte aligned");
    if (((uintptr_t)actions_half_dev & 1u) != 0) return fail(GEMX_ERR_ARG, "actions_half_dev must be 2-byte aligned");
    if (h->cfg.dtype == GEMX_F64) return fail(GEMX_ERR_ARG, "gemx_rollout_half: fp32 handles only");
    if (n_discrete_actions(h) > 0) return fail(GEMX_ERR_ARG, "gemx_rollout_half: continuous converters only (a discrete action is one byte already)");
    gemx::DeviceGuard guard(h->device);
    h->cur_half = true;
    const int rc = launch_advance(h, actions_half_dev, K, obs_out_dev, done_out_dev, 1, (hipStream_t)stream);
    h->cur_half = false;
    return rc;
}

int gemx_set_reward(gemx_handle *h, const gemx_reward_config *rc) {
    if (!h) return fail(GEMX_ERR_ARG, "null handle");
    if (!rc) { h->rw_n_ref = -1; return GEMX_OK; }
    if (rc->struct_size != (int32_t)sizeof(gemx_reward_config)) return fail(GEMX_ERR_ARG, "gemx_reward_config size mismatch");
    if (rc->n_ref < 0 || rc->n_ref > GEMX_MAX_REF) return fail(GEMX_ERR_ARG, "n_ref must be in [0, %d]", GEMX_MAX_REF);
    for (int j = 0; j < rc->n_ref; ++j) {
        if (rc->ref_index[j] < 0 || rc->ref_index[j] >= h->nout) return fail(GEMX_ERR_ARG, "ref_index[%d] out of range", j);
        for (int k = 0; k < j; ++k)
            if (rc->ref_index[k] == rc->ref_index[j]) return fail(GEMX_ERR_ARG, "ref_index has duplicates");
    }
    for (int i = 0; i < h->nout; ++i)
        if (rc->weight[i] != 0.0 && !(rc->state_length[i] > 0)) return fail(GEMX_ERR_ARG, "state_length[%d] must be positive", i);
    gemx::DeviceGuard guard(h->device);
    if (!h->rw_dev && hipMalloc(&h->rw_dev, sizeof(RewardDev<double>)) != hipSuccess) return fail(GEMX_ERR_ALLOC, "hipMalloc(reward) failed");
    if (h->cfg.dtype == GEMX_F64) {
        RewardDev<double> W;
        build_reward(h, rc, W);
        reward_hot_from(W, h->rh_d);
        HIP_TRY(hipMemcpy(h->rw_dev, &W, sizeof(W), hipMemcpyHostToDevice));
    } else {
        RewardDev<float> W;
        build_reward(h, rc, W);
        reward_hot_from(W, h->rh_f);
        HIP_TRY(hipMemcpy(h->rw_dev, &W, sizeof(W), hipMemcpyHostToDevice));
    }
    h->rw_n_ref = rc->n_ref;
    return GEMX_OK;
}

int gemx_rollout_reward(gemx_handle *h, const void *actions_dev, int32_t K, const void *refs_dev, void *obs_out_dev, uint8_t *done_out_dev,
                        void *reward_out_dev, void *stream) {
    if (!h) return fail(GEMX_ERR_ARG, "null handle");
    if (h->rw_n_ref < 0) return fail(GEMX_ERR_ARG, "no reward function installed (gemx_set_reward)");
    if (!reward_out_dev || (h->rw_n_ref > 0 && !refs_dev)) return fail(GEMX_ERR_ARG, "refs_dev and reward_out_dev must not be null");
    h->cur_refs = refs_dev;
    h->cur_reward = reward_out_dev;
    const int rc = gemx_rollout(h, actions_dev, K, obs_out_dev, done_out_dev, 1, stream);
    h->cur_refs = nullptr;
    h->cur_reward = nullptr;
    return rc;
}

int gemx_step(gemx_handle *h, const void *actions_dev, void *obs_out_dev, uint8_t *done_out_dev, void *stream) {
    return gemx_rollout(h, actions_dev, 1, obs_out_dev, done_out_dev, 1, stream);
}

int gemx_get_state(gemx_handle *h, void *soa_out_dev, void *stream) {
    if (!h || !soa_out_dev) return fail(GEMX_ERR_ARG, "null argument");
    gemx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = (h->n + 255) / 256;
    GEMX_COV(h->cfg.dtype == GEMX_F64 ? "get_state_kernel<double>" : "get_state_kernel<float>");
    if (h->cfg.dtype == GEMX_F64)
        hipLaunchKernelGGL(get_state_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, (const double *)h->state, (const double *)h->angle,
                           (double *)soa_out_dev, h->n, h->nd, h->has_angle);
    else
        hipLaunchKernelGGL(get_state_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)h->state, (const int32_t *)h->angle,
                           (float *)soa_out_dev, h->n, h->nd, h->has_angle);
    HIP_TRY(hipGetLastError());
    return GEMX_OK;
}
int gemx_set_state(gemx_handle *h, const void *soa_in_dev, void *stream) {
    if (!h || !soa_in_dev) return fail(GEMX_ERR_ARG, "null argument");
    gemx::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = (h->n + 255) / 256;
    GEMX_COV(h->cfg.dtype == GEMX_F64 ? "set_state_kernel<double>" : "set_state_kernel<float>");
    if (h->cfg.dtype == GEMX_F64)
        hipLaunchKernelGGL(set_state_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, (double *)h->state, (double *)h->angle,
                           (const double *)soa_in_dev, h->n, h->nd, h->has_angle);
    else
        hipLaunchKernelGGL(set_state_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (float *)h->state, (int32_t *)h->angle,
                           (const float *)soa_in_dev, h->n, h->nd, h->has_angle);
    HIP_TRY(hipGetLastError());
    h->omega_is_init = false;  // (dc_stream_kernel assumes omega == init[0]; the next full reset restores that)
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &capturing);
    if (capturing != hipStreamCaptureStatusNone) h->omega_unknown = true;  // replayed at times the host cannot know
    return GEMX_OK;
}
int gemx_get_switch_state(gemx_handle *h, uint8_t *out_dev, void *stream) {
    if (!h || !out_dev) return fail(GEMX_ERR_ARG, "null argument");
    gemx::DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpyAsync(out_dev, h->sw, (size_t)h->n * h->sw_rows, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GEMX_OK;
}
int gemx_set_switch_state(gemx_handle *h, const uint8_t *in_dev, void *stream) {
    if (!h || !in_dev) return fail(GEMX_ERR_ARG, "null argument");
    gemx::DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpyAsync(h->sw, in_dev, (size_t)h->n * h->sw_rows, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GEMX_OK;
}
const char *gemx_last_launch(const gemx_handle *h) {
    if (!h) return "";
    const auto &l = h->ll;
    if (l.pipe == 3)
        snprintf(h->last_launch, sizeof(h->last_launch),
                 "gemx::dc_stream_kernel<sys=%d,conv=%d,solver=%d,f32,D=%d> grid=%lld x %d threads, lds=%zu B, K=%d", l.sys, l.conv, l.solver, l.d,
                 l.blocks, l.threads, l.lds, l.k);
    else if (l.pipe == 2)
        snprintf(h->last_launch, sizeof(h->last_launch), "gemx::step_kernel<sys=%d,conv=%d,load=%d,solver=%d,il=%d,%s> grid=%lld x %d threads, K=1", l.sys, l.conv,
                 l.load, l.solver, l.il, l.real_size == 4 ? "f32" : "f64", l.blocks, l.threads);
    else if (l.pipe) {
        snprintf(h->last_launch, sizeof(h->last_launch),
                 "gemx::advance_pipe_kernel<sys=%d,conv=%d,load=%d,solver=%d,il=%d,%s,D=%d> grid=%lld x %d threads, lds=%zu B, K=%d", l.sys,
                 l.conv, l.load, l.solver, l.il, l.real_size == 4 ? "f32" : "f64", l.d, l.blocks, l.threads, l.lds, l.k);
        if (l.pace != 0) {  // the large-batch rate limiter was set for this launch: interval per hand-off block, and what it was priced for
            const size_t used = strlen(h->last_launch);
            snprintf(h->last_launch + used, sizeof(h->last_launch) - used, ", rate limit %u0 ns per block (%lld resident workgroups%s)", l.pace,
                     l.pace_res < l.blocks ? l.pace_res : l.blocks, l.pace_tail != 0 ? "; shorter in the last round" : "");
        }
        if (h->pace_cal_state != 0) {  // the closed loop: what this launch ran at
            const size_t used = strlen(h->last_launch);
            snprintf(h->last_launch + used, sizeof(h->last_launch) - used, ", limiter %s at %.2f x the built-in target%s", h->pace_cal_state == 2 ? "calibrated" : "calibrating",
                     h->pace_scale_last, h->pace_scale_last == 0.0 ? " (unpaced)" : "");
        }
    } else
        snprintf(h->last_launch, sizeof(h->last_launch),
                 "gemx::advance_kernel<sys=%d,conv=%d,load=%d,solver=%d,il=%d,%s> grid=%lld x %d threads, lds=%zu B, K=%d, S=%d", l.sys, l.conv,
                 l.load, l.solver, l.il, l.real_size == 4 ? "f32" : "f64", l.blocks, l.threads, l.lds, l.k, l.s);
    if (h->overrides[0] != 0) {
        const size_t used = strlen(h->last_launch);
        snprintf(h->last_launch + used, sizeof(h->last_launch) - used, " overrides[%s]", h->overrides);
    }
    return h->last_launch;
}
int gemx_set_steps_per_block(gemx_handle *h, int32_t steps) {
    if (!h || steps < 0) return fail(GEMX_ERR_ARG, "invalid argument");
    h->steps_per_block = steps;
    return GEMX_OK;
}
int gemx_error_flags(gemx_handle *h, uint32_t *flags_host, void *stream) {
    if (!h || !flags_host) return fail(GEMX_ERR_ARG, "null argument");
    gemx::DeviceGuard guard(h->device);
    HIP_TRY(hipMemcpyAsync(flags_host, h->err, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GEMX_OK;
}

}  // extern "C"
