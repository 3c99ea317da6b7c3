// gemx_kernels.hpp -- the MI355X (gfx950 / CDNA4) batched SCML physical-system stepper.
//
// What it replaces (reference = upb-lea/gym-electric-motor 3.0.2, paths relative to src/gym_electric_motor/):
//   SCMLSystem.simulate                          physical_systems/physical_systems.py:171-203
//   SynchronousMotorSystem.simulate              physical_systems/physical_systems.py:487-525
//   SquirrelCageInductionMotorSystem.simulate    physical_systems/physical_systems.py:771-814
//   converters / motors / loads / solvers / constraints on that path (cited at each device function).
//
// Design (MI355X-first, not a translation of the Python):
//   * one lane = one env; one 64-lane wavefront = one workgroup; N envs advance in lockstep.
//   * ODE state lives in HBM as SoA rows [S_ode][N]; in the fused K-step launch it stays in VGPRs between steps,
//     so a step moves only action-in + observation-out + done.
//   * all motor/load/converter/limit parameters are uniform across envs -> kernel arguments -> SGPRs (0 HBM bytes
//     per env, no LDS needed for them).
//   * LDS is spent where lanes exchange data: observation rows are staged in an LDS ring and flushed in bursts of
//     16-byte-per-lane stores; the next block's action tile is prefetched cooperatively (see advance_kernel).
//   * the electrical angle is a 32-bit fixed-point fraction of a turn in the fp32 path (exact wrap, constant
//     resolution however long the rollout is).
//   * compile-time specialisation <SYS, CONV, LOAD, SOLVER, IL, R>: a constant-speed load removes the omega
//     dynamics (per-segment pre-multiplied coefficients, 4 FMAs per PMSM right-hand side); IL = false (no converter
//     dead time, the default of every reference env) removes the current-sign logic and the second segment.
//   * no MFMA: ~100-250 VALU ops and 25-117 bytes per env-step; the path is HBM / issue bound.
#pragma once
#include <cstdio>

#include "gemx_common.hpp"

namespace gemx {

// ------------------------------------------------------------------------------------------------
// load: d(omega)/dt (constant_speed_load.py:40-42; polynomial_static_load.py:62-66, 87-99)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float med3_r(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ double med3_r(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
template <class R> __device__ __forceinline__ R poly_load_ode(const DevParams<R> &P, R omega, R torque) {
    // sign(w) * c * w^2 = c * w * |w|.  The constant term -- sign(w) a beyond |w| = a tau_decay / J, (J / tau_decay) w inside
    // (polynomial_static_load.py:87-92) -- IS the saturation clamp((J / tau_decay) w, -a, a): one multiply and one v_med3_f32.  As a
    // compare + select it was a VALU-written mask in every Runge-Kutta stage, which the next VALU instruction may not read on gfx950
    // (`s_nop 1` + ~8 cycles each, tools/microbench_chain.hip).  At |w| within an ulp of the kink the two forms differ by an ulp of a.
    const R a = med3_r(P.lin_factor * omega, -P.la, P.la);
    const R tl = P.lc * (omega * fabs(omega)) + P.lb * omega + a;
    return (torque - tl) * P.inv_j;
}

// ------------------------------------------------------------------------------------------------
// electrical sub-system.  x = motor states without the angle, u = segment-constant input voltage.
// The reference evaluates matmul(model_constants, feature_vector) (dc_permanently_excited_motor.py:81-84,
// synchronous_motor.py:143-168, induction_motor.py:187-217 + squirrel_cage_induction_motor.py:121-129); here only
// the structurally non-zero entries are used (pack_model() rejects a matrix with any other non-zero entry), and
// everything that depends only on (omega, u) is multiplied out once per call of prep().
// ------------------------------------------------------------------------------------------------
template <int SYS, class R> struct Elec;

template <class R> struct Elec<GEMX_SYS_DC_PERMEX, R> {
    static constexpr int NM = 1;
    struct Pre { R b; };
    static __device__ __forceinline__ Pre prep(const DevParams<R> &P, R w, const R (&u)[MAX_U]) {
        const R bw = P.m[0] * w;  // (wave-uniform and loop invariant behind a ConstantSpeedLoad: the input term is ONE fma per step)
        return Pre{fma(P.m[2], u[0], bw)};
    }
    static __device__ __forceinline__ void f(const DevParams<R> &P, const Pre &p, const R (&x)[1], R (&dx)[1]) { dx[0] = p.b + P.m[1] * x[0]; }
    static __device__ __forceinline__ R torque(const DevParams<R> &P, const R (&x)[1]) { return P.tc0 * x[0]; }  // line 67-69
    static constexpr int NG = 1;  // affine part g of f(x) = A x + g: its first NG rows are non-zero
    static __device__ __forceinline__ void get_b(const Pre &p, R (&g)[NG]) { g[0] = p.b; }
    static __device__ __forceinline__ Pre set_b(Pre p, const R (&g)[NG]) { p.b = g[0]; return p; }
};
template <class R> struct Elec<GEMX_SYS_DC_SERIES, R> {  // dc_series_motor.py:68-83: di = (-(r_a+r_e) i - l_e' omega i + u) / (l_a+l_e)
    static constexpr int NM = 1;
    struct Pre { R a, b; };
    static __device__ __forceinline__ Pre prep(const DevParams<R> &P, R w, const R (&u)[MAX_U]) { return Pre{P.m[0] + P.m[1] * w, P.m[2] * u[0]}; }
    static __device__ __forceinline__ void f(const DevParams<R> &, const Pre &p, const R (&x)[1], R (&dx)[1]) { dx[0] = p.b + p.a * x[0]; }
    static __device__ __forceinline__ R torque(const DevParams<R> &P, const R (&x)[1]) { return P.tc0 * x[0] * x[0]; }  // line 74-76
    static constexpr int NG = 1;
    static __device__ __forceinline__ void get_b(const Pre &p, R (&g)[NG]) { g[0] = p.b; }
    static __device__ __forceinline__ Pre set_b(Pre p, const R (&g)[NG]) { p.b = g[0]; return p; }
};
template <class R> struct Elec<GEMX_SYS_DC_SHUNT, R> {  // dc_motor.py:96-127 with u_a = u_e = u (dc_shunt_motor.py:72-74)
    static constexpr int NM = 2;
    struct Pre { R ba, be, w1; };
    static __device__ __forceinline__ Pre prep(const DevParams<R> &P, R w, const R (&u)[MAX_U]) { return Pre{P.m[2] * u[0], P.m[4] * u[0], P.m[1] * w}; }
    static __device__ __forceinline__ void f(const DevParams<R> &P, const Pre &p, const R (&x)[2], R (&dx)[2]) {
        dx[0] = p.ba + P.m[0] * x[0] + p.w1 * x[1];
        dx[1] = p.be + P.m[3] * x[1];
    }
    static __device__ __forceinline__ R torque(const DevParams<R> &P, const R (&x)[2]) { return P.tc0 * x[0] * x[1]; }  // dc_motor.py:106-108
    static constexpr int NG = 2;
    static __device__ __forceinline__ void get_b(const Pre &p, R (&g)[NG]) { g[0] = p.ba; g[1] = p.be; }
    static __device__ __forceinline__ Pre set_b(Pre p, const R (&g)[NG]) { p.ba = g[0]; p.be = g[1]; return p; }
};
template <class R> struct Elec<GEMX_SYS_DC_EXTEX, R> {  // dc_motor.py:96-127 with separately fed armature / excitation circuits
    static constexpr int NM = 2;
    struct Pre { R ba, be, w1; };
    static __device__ __forceinline__ Pre prep(const DevParams<R> &P, R w, const R (&u)[MAX_U]) { return Pre{P.m[2] * u[0], P.m[4] * u[1], P.m[1] * w}; }
    static __device__ __forceinline__ void f(const DevParams<R> &P, const Pre &p, const R (&x)[2], R (&dx)[2]) {
        dx[0] = p.ba + P.m[0] * x[0] + p.w1 * x[1];
        dx[1] = p.be + P.m[3] * x[1];
    }
    static __device__ __forceinline__ R torque(const DevParams<R> &P, const R (&x)[2]) { return P.tc0 * x[0] * x[1]; }  // dc_motor.py:106-108
    static constexpr int NG = 2;
    static __device__ __forceinline__ void get_b(const Pre &p, R (&g)[NG]) { g[0] = p.ba; g[1] = p.be; }
    static __device__ __forceinline__ Pre set_b(Pre p, const R (&g)[NG]) { p.ba = g[0]; p.be = g[1]; return p; }
};
template <class R> struct Elec<GEMX_SYS_EESM, R> {  // externally_excited_synchronous_motor.py:69-113, 133-136; x = i_sd, i_sq, i_e
    static constexpr int NM = 3;
    struct Pre { R bd, bq, be, w4, w7, w8, w13; };
    static __device__ __forceinline__ Pre prep(const DevParams<R> &P, R w, const R (&u)[MAX_U]) {
        return Pre{P.m[2] * u[0] + P.m[3] * u[2], P.m[6] * u[1], P.m[11] * u[0] + P.m[12] * u[2], P.m[4] * w, P.m[7] * w, P.m[8] * w, P.m[13] * w};
    }
    static __device__ __forceinline__ void f(const DevParams<R> &P, const Pre &p, const R (&x)[3], R (&dx)[3]) {
        dx[0] = p.bd + P.m[0] * x[0] + P.m[1] * x[2] + p.w4 * x[1];
        dx[1] = p.bq + P.m[5] * x[1] + p.w7 * x[0] + p.w8 * x[2];
        dx[2] = p.be + P.m[9] * x[0] + P.m[10] * x[2] + p.w13 * x[1];
    }
    static __device__ __forceinline__ R torque(const DevParams<R> &P, const R (&x)[3]) { return (P.tc0 * x[2] + P.tc1 * x[0]) * x[1]; }
    static constexpr int NG = 3;
    static __device__ __forceinline__ void get_b(const Pre &p, R (&g)[NG]) { g[0] = p.bd; g[1] = p.bq; g[2] = p.be; }
    static __device__ __forceinline__ Pre set_b(Pre p, const R (&g)[NG]) { p.bd = g[0]; p.bq = g[1]; p.be = g[2]; return p; }
};
template <class R> struct Elec<GEMX_SYS_SYNC, R> {  // permanent_magnet_synchronous_motor.py:107-119, 134-139
    static constexpr int NM = 2;
    struct Pre { R bd, bq, wdq, wqd; };
    static __device__ __forceinline__ Pre prep(const DevParams<R> &P, R w, const R (&u)[MAX_U]) {
        // (the back-EMF term as a product of its own: behind a constant-speed load it is loop-invariant and leaves the step -- written
        // m3 w + m5 u_q the compiler fuses it the other way round, a multiply, a fused multiply-add and a register copy of omega per step)
        const R bw = P.m[3] * w;
        return Pre{P.m[1] * u[0], fma(P.m[5], u[1], bw), P.m[2] * w, P.m[6] * w};
    }
    static __device__ __forceinline__ void f(const DevParams<R> &P, const Pre &p, const R (&x)[2], R (&dx)[2]) {
        dx[0] = p.bd + P.m[0] * x[0] + p.wdq * x[1];
        dx[1] = p.bq + P.m[4] * x[1] + p.wqd * x[0];
    }
    static __device__ __forceinline__ R torque(const DevParams<R> &P, const R (&x)[2]) { return (P.tc0 + P.tc1 * x[0]) * x[1]; }
    static constexpr int NG = 2;
    static __device__ __forceinline__ void get_b(const Pre &p, R (&g)[NG]) { g[0] = p.bd; g[1] = p.bq; }
    static __device__ __forceinline__ Pre set_b(Pre p, const R (&g)[NG]) { p.bd = g[0]; p.bq = g[1]; return p; }
};
template <class R> struct Elec<GEMX_SYS_SCIM, R> {  // induction_motor.py:236-248, 287-312
    static constexpr int NM = 4;
    struct Pre { R ba, bb, w2, w6, w10, w13; };
    static __device__ __forceinline__ Pre prep(const DevParams<R> &P, R w, const R (&u)[MAX_U]) {
        return Pre{P.m[3] * u[0], P.m[7] * u[1], P.m[2] * w, P.m[6] * w, P.m[10] * w, P.m[13] * w};
    }
    static __device__ __forceinline__ void f(const DevParams<R> &P, const Pre &p, const R (&x)[4], R (&dx)[4]) {
        dx[0] = p.ba + P.m[0] * x[0] + P.m[1] * x[2] + p.w2 * x[3];
        dx[1] = p.bb + P.m[4] * x[1] + P.m[5] * x[3] + p.w6 * x[2];
        dx[2] = P.m[8] * x[0] + P.m[9] * x[2] + p.w10 * x[3];
        dx[3] = P.m[11] * x[1] + P.m[12] * x[3] + p.w13 * x[2];
    }
    static __device__ __forceinline__ R torque(const DevParams<R> &P, const R (&x)[4]) { return P.tc0 * (x[2] * x[1] - x[3] * x[0]); }
    static constexpr int NG = 2;
    static __device__ __forceinline__ void get_b(const Pre &p, R (&g)[NG]) { g[0] = p.ba; g[1] = p.bb; }
    static __device__ __forceinline__ Pre set_b(Pre p, const R (&g)[NG]) { p.ba = g[0]; p.bb = g[1]; return p; }
};

template <class R> struct Elec<GEMX_SYS_DFIM, R> {  // the SCIM matrix with live rotor-voltage columns; u = u_s alpha/beta, u_r alpha/beta
    static constexpr int NM = 4;
    struct Pre { R ba, bb, bc, bd, w2, w6, w10, w13; };
    static __device__ __forceinline__ Pre prep(const DevParams<R> &P, R w, const R (&u)[MAX_U]) {
        return Pre{P.m[3] * u[0] + P.m[14] * u[2], P.m[7] * u[1] + P.m[15] * u[3], P.m[16] * u[2], P.m[17] * u[3],
                   P.m[2] * w, P.m[6] * w, P.m[10] * w, P.m[13] * w};
    }
    static __device__ __forceinline__ void f(const DevParams<R> &P, const Pre &p, const R (&x)[4], R (&dx)[4]) {
        dx[0] = p.ba + P.m[0] * x[0] + P.m[1] * x[2] + p.w2 * x[3];
        dx[1] = p.bb + P.m[4] * x[1] + P.m[5] * x[3] + p.w6 * x[2];
        dx[2] = p.bc + P.m[8] * x[0] + P.m[9] * x[2] + p.w10 * x[3];
        dx[3] = p.bd + P.m[11] * x[1] + P.m[12] * x[3] + p.w13 * x[2];
    }
    static __device__ __forceinline__ R torque(const DevParams<R> &P, const R (&x)[4]) { return P.tc0 * (x[2] * x[1] - x[3] * x[0]); }
    static constexpr int NG = 4;
    static __device__ __forceinline__ void get_b(const Pre &p, R (&g)[NG]) { g[0] = p.ba; g[1] = p.bb; g[2] = p.bc; g[3] = p.bd; }
    static __device__ __forceinline__ Pre set_b(Pre p, const R (&g)[NG]) { p.ba = g[0]; p.bb = g[1]; p.bc = g[2]; p.bd = g[3]; return p; }
};

// ------------------------------------------------------------------------------------------------
// one explicit Runge-Kutta step of size h for  z' = F(z),  z in R^NZ.  Returns the scheme's quadrature of z[0]
// (used for the angle when omega = z[0] is dynamic).  SOLVER: Euler (solvers.py:124-136), classical RK4,
// Dormand-Prince 5th-order solution without error control.
// ------------------------------------------------------------------------------------------------
// rk_step_k1: the same step with the first stage k1 = F(z) already evaluated by the caller (who may need it to choose h).
// last0 (optional): receives d z[0] / dt of the scheme's LAST stage, which sits at t + h in all three schemes -- the end slope of
// omega's path to the order the kink correction of integrate<> needs, for free.
template <int SOLVER, int NZ, class R, class F>
__device__ __forceinline__ R rk_step_k1(R (&z)[NZ], const R (&k1)[NZ], R h, F &&rhs, R *last0 = nullptr) {
    R zt[NZ];
    if (SOLVER == GEMX_SOLVER_EULER) {
        const R q = z[0];
        if (last0 != nullptr) *last0 = k1[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i) z[i] = z[i] + k1[i] * h;
        return q;
    } else if (SOLVER == GEMX_SOLVER_RK4) {
        R k2[NZ], k3[NZ], k4[NZ];
        R q = z[0];
        const R hh = R(0.5) * h;
#pragma unroll
        for (int i = 0; i < NZ; ++i) zt[i] = z[i] + hh * k1[i];
        rhs(zt, k2);
        q += R(2) * zt[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i) zt[i] = z[i] + hh * k2[i];
        rhs(zt, k3);
        q += R(2) * zt[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i) zt[i] = z[i] + h * k3[i];
        rhs(zt, k4);
        if (last0 != nullptr) *last0 = k4[0];
        q += zt[0];
        const R h6 = h * R(1.0 / 6.0);
#pragma unroll
        for (int i = 0; i < NZ; ++i) z[i] = z[i] + h6 * (k1[i] + R(2) * (k2[i] + k3[i]) + k4[i]);
        return q * R(1.0 / 6.0);
    } else {
        R k2[NZ], k3[NZ], k4[NZ], k5[NZ], k6[NZ];
        R q = R(35.0 / 384.0) * z[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i) zt[i] = z[i] + h * (R(1.0 / 5.0) * k1[i]);
        rhs(zt, k2);
#pragma unroll
        for (int i = 0; i < NZ; ++i) zt[i] = z[i] + h * (R(3.0 / 40.0) * k1[i] + R(9.0 / 40.0) * k2[i]);
        rhs(zt, k3);
        q += R(500.0 / 1113.0) * zt[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i) zt[i] = z[i] + h * (R(44.0 / 45.0) * k1[i] - R(56.0 / 15.0) * k2[i] + R(32.0 / 9.0) * k3[i]);
        rhs(zt, k4);
        q += R(125.0 / 192.0) * zt[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i)
            zt[i] = z[i] + h * (R(19372.0 / 6561.0) * k1[i] - R(25360.0 / 2187.0) * k2[i] + R(64448.0 / 6561.0) * k3[i] -
                                R(212.0 / 729.0) * k4[i]);
        rhs(zt, k5);
        q -= R(2187.0 / 6784.0) * zt[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i)
            zt[i] = z[i] + h * (R(9017.0 / 3168.0) * k1[i] - R(355.0 / 33.0) * k2[i] + R(46732.0 / 5247.0) * k3[i] +
                                R(49.0 / 176.0) * k4[i] - R(5103.0 / 18656.0) * k5[i]);
        rhs(zt, k6);
        if (last0 != nullptr) *last0 = k6[0];
        q += R(11.0 / 84.0) * zt[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i)
            z[i] = z[i] + h * (R(35.0 / 384.0) * k1[i] + R(500.0 / 1113.0) * k3[i] + R(125.0 / 192.0) * k4[i] -
                               R(2187.0 / 6784.0) * k5[i] + R(11.0 / 84.0) * k6[i]);
        return q;
    }
}
template <int SOLVER, int NZ, class R, class F>
__device__ __forceinline__ R rk_step(R (&z)[NZ], R h, F &&rhs) {
    R k1[NZ];
    rhs(z, k1);
    return rk_step_k1<SOLVER, NZ, R>(z, k1, h, rhs);
}

// ------------------------------------------------------------------------------------------------
// GEMX_SOLVER_ADAPTIVE: error-controlled Dormand-Prince 5(4) over one integration segment of length hs -- the semantics of the
// reference's default solver (scipy's dopri5 behind ScipyOdeSolver, solvers.py:139-184: local error estimate from the embedded
// 4th-order solution, norm sqrt(mean((err_i / (atol + rtol max(|y_i|, |y_i new|)))^2)), step accepted iff <= 1, next step
// h * clamp(0.9 err^-1/5, 0.2, 10)), restated for a lock-stepped wave: every lane tries the whole segment first and cuts it where ITS
// estimate demands; the wave iterates until its last lane is through, lanes that are done ride along with h = 0 (z + 0 k = z).  The
// first stage of a sub-step is the last of the one before (FSAL), so an accepted sub-step costs six right-hand sides.  The step size IS
// carried from one control step to the next, per env, as DOPRI5 does (it stores the controller's proposal back into WORK(7);
// solvers.py:139-184 never clears it except through set_initial_value() = reset) -- round 5: a wave used to try every control step whole,
// and since some lane of 64 rejects that almost every time under random actions, every step cost the wave one wasted attempt (three attempts
// where two do).  *hc = the largest step the controller proposed after an accepted sub-step of the previous segment (0: none yet -- a fresh
// episode tries the segment whole); the segment is cut into ceil(hs / *hc) equal first tries.  Differences to scipy's code, none of which
// the tolerance depends on: equal first tries instead of proposal-sized ones with a sliver at the end, no PI term in the step-size rule,
// and a floor of hs / 1024 under which a step is taken as it is and bit GEMX_ERRFLAG_TOLERANCE of the handle's error word is raised.
// Returns the integral of z[0] over the segment (for the angle).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pow_m01(float x) { return __builtin_amdgcn_exp2f(-0.1f * __builtin_amdgcn_logf(x)); }  // x^-0.1, x > 0
__device__ __forceinline__ double pow_m01(double x) { return pow(x, -0.1); }
__device__ __forceinline__ float rcp_r(float x) { return __builtin_amdgcn_rcpf(x); }  // V_RCP_F32, 1 ulp: the error norm is compared with 1
__device__ __forceinline__ double rcp_r(double x) { return 1.0 / x; }
// first try of a segment of length hs with the carried proposal hc: hs / ceil(hs / hc), the whole segment without a proposal
// (Segments shorter than half a control step -- the dead-time segment of a switching leg, t_il << tau -- neither use nor update the carried
// size: the proposal after a 1-us segment is at most 10 us and would cut the 99 us behind it into nine pieces.)
template <class R> __device__ __forceinline__ bool dp5_carries(const DevParams<R> &P, R hs, const R *hc) { return hc != nullptr && !(hs < R(0.5) * P.tau); }
template <class R> __device__ __forceinline__ R dp5_first_try(const DevParams<R> &P, R hs, const R *hc) {
    // (the proposal carries the controller's safety factor 0.9: a sub-step of proposal / 0.9 is the largest the last estimate would have let pass)
    if (!dp5_carries<R>(P, hs, hc) || !(*hc > R(0)) || !(*hc < R(0.9) * hs)) return hs;
    const R n = fmin(ceil(R(0.9) * hs / *hc), R(1024));
    return hs / n;
}
// GEMX_SOLVER_SPLIT_KINKS (integrate<>): the model system's omega path over one step as the cubic Hermite interpolant in th = t / h,
// s(th) = w + V0 th + c2 th^2 + c3 th^3 (V0, V1 = h x the end slopes), and its ramp integrals U(c) = int_0^1 (s - c)_+ dth.
__device__ __forceinline__ float sqrt_r(float x) { return __builtin_amdgcn_sqrtf(x); }  // V_SQRT_F32, 1 ulp
__device__ __forceinline__ double sqrt_r(double x) { return sqrt(x); }
template <class R> struct KinkPath {
    R w, w1, V0, V0sq, q4, c2t, c3q, hV0, M;  // q4 = 4 (w1 - w - V0); c2t = c2 / 3, c3q = c3 / 4, hV0 = V0 / 2; M = int_0^1 s
    bool up;                                   // w1 > w
    __device__ __forceinline__ R ramp(R c) const {
        // crossing instant: the root (towards the end of travel) of the quadratic through both ends with the start slope,
        // th = 2 (c - w) / (V0 +- sqrt(V0^2 + 4 (w1 - w - V0)(c - w))); NaN-safe clamp to [0, 1] (fmax / fmin drop a NaN operand)
        const R cs = c - w;
        const R sq = sqrt_r(fmax(fma(q4, cs, V0sq), R(0)));
        const R th = fmin(fmax((cs + cs) * rcp_r(V0 + (up ? sq : -sq)), R(0)), R(1));
        const R Q = fma(fma(fma(c3q, th, c2t), th, hV0), th, -cs) * th;  // int_0^th (s - c)
        const R Mc = M - c;
        const bool a0 = w >= c, a1 = w1 >= c;
        return (a0 & a1) ? Mc : ((a0 | a1) ? (up ? Mc - Q : Q) : R(0));
    }
};

// KINKS (round 6; GEMX_SOLVER_ADAPTIVE together with GEMX_SOLVER_SPLIT_KINKS, a PolynomialStaticLoad whose torque has kinks; z[0] = omega):
// under random actions the plain controller spends 4.7 attempts per control step and WAVE on BASELINE config 4 -- not because the lanes
// need them (1.1 per lane) but because some lane of 64 crosses the load's kink at |omega| = a tau_decay / J in most steps, where the error
// estimate of a step across the kink demands five to eight cuts (tools/wave_step_statistics.py; without the kink every step of every
// lane is ONE accepted attempt).  So every attempt integrates the SMOOTH model system of the fixed-step kink correction (integrate<>: the
// saturation replaced by the affine piece c0 + c1 omega of the region the mid-step omega is predicted in; `rhs_m(z, dz, c0, c1)`), its
// error estimate never sees the kink, and an accepted sub-step adds the kink's defect to omega in closed form (kink_defect).  The first
// stage of the next sub-step is the last of this one (FSAL) unless the sub-step touched a kink: then the TRUE right-hand side is evaluated
// at the corrected state (behind a wave ballot).  1.8 attempts per control step and wave; 9e-7 / 6e-6 against scipy's dopri5 on i.i.d. /
// held actions where the plain controller has 7e-7 / 3e-6 (fp64: the test suite's CPU restatement of these steps, tools/wave_step_statistics.py).
template <class R>
__device__ __forceinline__ R kink_defect(const DevParams<R> &P, R w, R w1, R V0, R V1, R wmid, bool band, bool needs, R hh) {
    // integrate<>'s closed forms (documented there): the defect of omega over a sub-step of length hh whose model path is the cubic
    // through (w, w1) with the end slopes V0 / hh, V1 / hh
    const R lim = P.omega_lim, phi_lim = copysign(lim, wmid), dl = w1 - w;
    const R c2 = R(3) * dl - R(2) * V0 - V1, c3 = V0 + V1 - R(2) * dl;
    const KinkPath<R> kp{w, w1, V0, V0 * V0, R(4) * (dl - V0), c2 * R(1.0 / 3.0), c3 * R(0.25), R(0.5) * V0,
                         w + R(0.5) * V0 + c2 * R(1.0 / 3.0) + c3 * R(0.25), w1 > w};
    const R lev = (kp.up ? (w < -lim) : !(w > lim)) ? -lim : lim, oth = -lev;
    const R U1 = kp.ramp(lev);
    const bool o0 = w >= oth, o1 = w1 >= oth, full = o0 != o1;
    R U2 = (o0 & o1) ? kp.M - oth : R(0);
    if (__any(full)) {
        const R uf = kp.ramp(oth);
        U2 = full ? uf : U2;
    }
    const R Up = lev > R(0) ? U1 : U2, Um = lev > R(0) ? U2 : U1;
    const R D = -(hh * P.inv_tau_decay) * ((Um - Up - lim) - (band ? kp.M : phi_lim));
    return needs ? D : R(0);
}
struct NoModel {};
template <int NZ, class R, class F, class FM = NoModel>
__device__ __forceinline__ R dp5_adaptive(const DevParams<R> &P, R (&z)[NZ], R hs, F &&rhs, R *hc = nullptr, FM &&rhs_m = FM{}) {
    constexpr bool HAS_M = !std::is_same<typename std::decay<FM>::type, NoModel>::value;
    bool kink = false;  // wave-uniform
    if constexpr (HAS_M) kink = P.kink_split != 0;
    R k1t[NZ];  // the TRUE right-hand side at z
    rhs(z, k1t);
    R t = R(0), h = dp5_first_try<R>(P, hs, hc), integral = R(0), hprop = R(0);
    const R hmin = hs * R(1.0 / 1024.0);
    bool gave_up = false;
#pragma nounroll
    for (int guard = 0; guard < 4096; ++guard) {
        const bool active = t < hs;
        if (!__any(active)) break;
        const bool fin = !(h < hs - t);  // this attempt reaches the end of the segment
        const R hh = active ? (fin ? hs - t : h) : R(0);
        R k1[NZ], k2[NZ], k3[NZ], k4[NZ], k5[NZ], k6[NZ], k7[NZ], zt[NZ], zn[NZ];
#pragma unroll
        for (int i = 0; i < NZ; ++i) k1[i] = k1t[i];
        [[maybe_unused]] R kc0 = R(0), kc1 = R(0), wmid = R(0);
        [[maybe_unused]] bool band = false;
        if constexpr (HAS_M) {
            if (kink) {  // this attempt's model: the region of the Euler-predicted mid-step omega
                wmid = fma(R(0.5) * hh, k1t[0], z[0]);
                band = fabs(wmid) < P.omega_lim;
                kc1 = band ? P.lin_factor : R(0);
                kc0 = band ? R(0) : copysign(P.la, wmid);
                k1[0] = fma(med3_r(P.lin_factor * z[0], -P.la, P.la) - fma(kc1, z[0], kc0), P.inv_j, k1t[0]);  // first stage of the model system
            }
        }
        auto ev = [&](const R (&zz)[NZ], R (&dz)[NZ]) {
            if constexpr (HAS_M) {
                if (kink) { rhs_m(zz, dz, kc0, kc1); return; }
            }
            rhs(zz, dz);
        };
        R q = R(35.0 / 384.0) * z[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i) zt[i] = z[i] + hh * (R(1.0 / 5.0) * k1[i]);
        ev(zt, k2);
#pragma unroll
        for (int i = 0; i < NZ; ++i) zt[i] = z[i] + hh * (R(3.0 / 40.0) * k1[i] + R(9.0 / 40.0) * k2[i]);
        ev(zt, k3);
        q += R(500.0 / 1113.0) * zt[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i) zt[i] = z[i] + hh * (R(44.0 / 45.0) * k1[i] - R(56.0 / 15.0) * k2[i] + R(32.0 / 9.0) * k3[i]);
        ev(zt, k4);
        q += R(125.0 / 192.0) * zt[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i)
            zt[i] = z[i] + hh * (R(19372.0 / 6561.0) * k1[i] - R(25360.0 / 2187.0) * k2[i] + R(64448.0 / 6561.0) * k3[i] -
                                 R(212.0 / 729.0) * k4[i]);
        ev(zt, k5);
        q -= R(2187.0 / 6784.0) * zt[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i)
            zt[i] = z[i] + hh * (R(9017.0 / 3168.0) * k1[i] - R(355.0 / 33.0) * k2[i] + R(46732.0 / 5247.0) * k3[i] +
                                 R(49.0 / 176.0) * k4[i] - R(5103.0 / 18656.0) * k5[i]);
        ev(zt, k6);
        q += R(11.0 / 84.0) * zt[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i)
            zn[i] = z[i] + hh * (R(35.0 / 384.0) * k1[i] + R(500.0 / 1113.0) * k3[i] + R(125.0 / 192.0) * k4[i] -
                                 R(2187.0 / 6784.0) * k5[i] + R(11.0 / 84.0) * k6[i]);
        ev(zn, k7);
        R e2 = R(0);
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const R err = hh * (R(71.0 / 57600.0) * k1[i] - R(71.0 / 16695.0) * k3[i] + R(71.0 / 1920.0) * k4[i] -
                                R(17253.0 / 339200.0) * k5[i] + R(22.0 / 525.0) * k6[i] - R(1.0 / 40.0) * k7[i]);
            const R sk = ((HAS_M && i == 0) ? P.atol_w : P.atol) + P.rtol * fmax(fabs(z[i]), fabs(zn[i]));  // (HAS_M: z[0] is omega)
            const R r = err * rcp_r(sk);
            e2 += r * r;
        }
        const R en2 = e2 * R(1.0 / NZ);  // the norm squared: err <= 1 <=> en2 <= 1, err^-1/5 = en2^-1/10
        const bool floor_hit = !(hh > hmin);
        const bool accept = active && (!(en2 > R(1)) || floor_hit);
        gave_up |= active && floor_hit && en2 > R(1);
        R fac = en2 > R(1e-20) ? R(0.9) * pow_m01(en2) : R(10);
        fac = fmin(fmax(fac, R(0.2)), accept ? R(10) : R(1));
        [[maybe_unused]] const R w0 = z[0];
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            z[i] = accept ? zn[i] : z[i];
            k1t[i] = accept ? k7[i] : k1t[i];  // (FSAL: where the sub-step touched no kink the model's slope at the end IS the true one)
        }
        if constexpr (HAS_M) {
            if (kink) {
                const R lim = P.omega_lim, phi_lim = copysign(lim, wmid), w1 = zn[0];
                const bool needs = accept && ((med3_r(w0, -lim, lim) != (band ? w0 : phi_lim)) | (med3_r(w1, -lim, lim) != (band ? w1 : phi_lim)));
                if (__any(needs)) {  // wave-uniform: the closed forms, and the true slope at the corrected state
                    z[0] = z[0] + kink_defect<R>(P, w0, w1, hh * k1[0], hh * k7[0], wmid, band, needs, hh);
                    R kt[NZ];
                    rhs(z, kt);
#pragma unroll
                    for (int i = 0; i < NZ; ++i) k1t[i] = needs ? kt[i] : k1t[i];
                }
            }
        }
        integral += accept ? hh * q : R(0);
        t = accept ? (fin ? hs : t + hh) : t;
        h = active ? fmax(hh * fac, hmin) : h;
        hprop = accept ? fmax(hprop, h) : hprop;  // (the controller's proposal after an accepted sub-step)
    }
    if (gave_up && P.errw != nullptr) atomicOr(P.errw, (uint32_t)GEMX_ERRFLAG_TOLERANCE);
    if (dp5_carries<R>(P, hs, hc)) *hc = hprop;
    return integral;
}

// ------------------------------------------------------------------------------------------------
// PACKED fp32 (round 5).  gfx950's VALU takes two fp32 operations per lane and issue slot (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), and a
// wave whose instruction stream bounds the launch -- the pipelined kernel's integrator behind a PolynomialStaticLoad, where omega is a state
// and every Runge-Kutta stage evaluates the whole right-hand side: BASELINE config 4, ~300 VALU instructions per step -- runs as fast as it
// has few instructions.  The three-phase machines' states come in PAIRS with the same coefficients pattern, (i_s alpha, i_s beta),
// (psi_r alpha, psi_r beta), (i_sd, i_sq):
//     d i_s   = M0 o i_s + M1 o psi_r + omega (M2 o swap(psi_r)) + b         d psi_r = M8 o i_s + M9 o psi_r + omega (M10 o swap(psi_r)) [+ b_r]
//     d i_sdq = M0 o i + omega (M2 o swap(i) + K3) + b                        (o: element-wise; swap: the other element of the pair)
// Left to the compiler's SLP vectoriser the scalar code came out as 88 packed + 154 scalar fp32 instructions and 38 register moves to
// form the pairs; written on two-element vectors (swap = the packed instructions' op_sel, free) a right-hand side is 8 packed + ~9 scalar
// instructions (the load torque and the motor torque stay scalar).  z = [omega | NP pairs]; the schemes below are rk_step_k1 operation by
// operation, on that type.  fp32 only (the fp64 diagnostic build keeps the array code); used by integrate<> for a dynamic omega.
// ------------------------------------------------------------------------------------------------
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2_t pk_fma(f2_t a, f2_t b, f2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2_t pk_fma(float a, f2_t b, f2_t c) { return __builtin_elementwise_fma((f2_t)(a), b, c); }
template <int NP> struct PkVec {
    float w;
    f2_t p[NP];
};
template <int SYS> constexpr int pk_pairs() { return (SYS == GEMX_SYS_SCIM || SYS == GEMX_SYS_DFIM) ? 2 : (SYS == GEMX_SYS_SYNC ? 1 : 0); }
// the step-constant coefficients of a system's right-hand side as pairs (VGPR pairs, formed once per step from the kernel arguments)
template <int SYS> struct PkElec;
template <> struct PkElec<GEMX_SYS_SYNC> {
    f2_t M0, M2, K3, b;
    __device__ __forceinline__ PkElec(const DevParams<float> &P, const float (&u)[MAX_U])
        : M0{P.m[0], P.m[4]}, M2{P.m[2], P.m[6]}, K3{0.0f, P.m[3]}, b{P.m[1] * u[0], P.m[5] * u[1]} {}
    // permanent_magnet_synchronous_motor.py:107-119: d i_d = m0 i_d + m2 w i_q + m1 u_d, d i_q = m4 i_q + m6 w i_d + m3 w + m5 u_q
    __device__ __forceinline__ float rhs(const DevParams<float> &P, const PkVec<1> &z, PkVec<1> &dz) const {
        const f2_t i = z.p[0];
        dz.p[0] = pk_fma(z.w, pk_fma(M2, i.yx, K3), pk_fma(M0, i, b));
        return (P.tc0 + P.tc1 * i.x) * i.y;  // torque, line 134-139
    }
};
template <> struct PkElec<GEMX_SYS_SCIM> {
    f2_t M0, M1, M2, M8, M9, M10, b;
    __device__ __forceinline__ PkElec(const DevParams<float> &P, const float (&u)[MAX_U])
        : M0{P.m[0], P.m[4]}, M1{P.m[1], P.m[5]}, M2{P.m[2], P.m[6]}, M8{P.m[8], P.m[11]}, M9{P.m[9], P.m[12]}, M10{P.m[10], P.m[13]},
          b{P.m[3] * u[0], P.m[7] * u[1]} {}
    // induction_motor.py:236-248 (the 15 non-zero entries of the 5 x 11 matrix, see Elec<GEMX_SYS_SCIM>)
    __device__ __forceinline__ float rhs(const DevParams<float> &P, const PkVec<2> &z, PkVec<2> &dz) const {
        const f2_t i = z.p[0], psi = z.p[1];
        const f2_t wps = z.w * psi.yx;
        dz.p[0] = pk_fma(M2, wps, pk_fma(M1, psi, pk_fma(M0, i, b)));
        dz.p[1] = pk_fma(M10, wps, pk_fma(M9, psi, M8 * i));
        const f2_t t = psi * i.yx;  // (psi_a i_b, psi_b i_a)
        return P.tc0 * (t.x - t.y);  // torque, line 287-312
    }
};
template <> struct PkElec<GEMX_SYS_DFIM> {
    f2_t M0, M1, M2, M8, M9, M10, b, br;
    __device__ __forceinline__ PkElec(const DevParams<float> &P, const float (&u)[MAX_U])
        : M0{P.m[0], P.m[4]}, M1{P.m[1], P.m[5]}, M2{P.m[2], P.m[6]}, M8{P.m[8], P.m[11]}, M9{P.m[9], P.m[12]}, M10{P.m[10], P.m[13]},
          b{P.m[3] * u[0] + P.m[14] * u[2], P.m[7] * u[1] + P.m[15] * u[3]}, br{P.m[16] * u[2], P.m[17] * u[3]} {}
    __device__ __forceinline__ float rhs(const DevParams<float> &P, const PkVec<2> &z, PkVec<2> &dz) const {
        const f2_t i = z.p[0], psi = z.p[1];
        const f2_t wps = z.w * psi.yx;
        dz.p[0] = pk_fma(M2, wps, pk_fma(M1, psi, pk_fma(M0, i, b)));
        dz.p[1] = pk_fma(M10, wps, pk_fma(M9, psi, pk_fma(M8, i, br)));
        const f2_t t = psi * i.yx;
        return P.tc0 * (t.x - t.y);
    }
};
// z + h (c1 k1 + c2 k2 + ...): the array code's stage expressions on PkVec (same operations per component, in the same order)
template <int NP> __device__ __forceinline__ PkVec<NP> pk_axpy(const PkVec<NP> &z, float h, const PkVec<NP> &k) {
    PkVec<NP> r;
    r.w = z.w + h * k.w;
#pragma unroll
    for (int j = 0; j < NP; ++j) r.p[j] = pk_fma(h, k.p[j], z.p[j]);
    return r;
}
template <int NP> __device__ __forceinline__ PkVec<NP> pk_scale(float c, const PkVec<NP> &k) {
    PkVec<NP> r;
    r.w = c * k.w;
#pragma unroll
    for (int j = 0; j < NP; ++j) r.p[j] = c * k.p[j];
    return r;
}
template <int NP> __device__ __forceinline__ PkVec<NP> pk_acc(const PkVec<NP> &a, float c, const PkVec<NP> &k) {  // a + c k
    PkVec<NP> r;
    r.w = a.w + c * k.w;
#pragma unroll
    for (int j = 0; j < NP; ++j) r.p[j] = pk_fma(c, k.p[j], a.p[j]);
    return r;
}
// rk_step_k1 on PkVec: the first stage k1 = F(z) is the caller's; returns the scheme's quadrature of omega, *last0 = d omega / dt of the last stage
template <int SOLVER, int NP, class F>
__device__ __forceinline__ float rk_step_k1_pk(PkVec<NP> &z, const PkVec<NP> &k1, float h, F &&rhs, float *last0 = nullptr) {
    using V = PkVec<NP>;
    if (SOLVER == GEMX_SOLVER_EULER) {
        const float q = z.w;
        if (last0 != nullptr) *last0 = k1.w;
        z = pk_axpy(z, h, k1);
        return q;
    } else if (SOLVER == GEMX_SOLVER_RK4) {
        V k2, k3, k4, zt;
        float q = z.w;
        const float hh = 0.5f * h;
        zt = pk_axpy(z, hh, k1);
        rhs(zt, k2);
        q += 2.0f * zt.w;
        zt = pk_axpy(z, hh, k2);
        rhs(zt, k3);
        q += 2.0f * zt.w;
        zt = pk_axpy(z, h, k3);
        rhs(zt, k4);
        if (last0 != nullptr) *last0 = k4.w;
        q += zt.w;
        const float h6 = h * (1.0f / 6.0f);
        z.w = z.w + h6 * (k1.w + 2.0f * (k2.w + k3.w) + k4.w);
#pragma unroll
        for (int j = 0; j < NP; ++j) z.p[j] = pk_fma(h6, pk_fma(2.0f, k2.p[j] + k3.p[j], k1.p[j]) + k4.p[j], z.p[j]);
        return q * (1.0f / 6.0f);
    } else {
        V k2, k3, k4, k5, k6, zt;
        float q = (float)(35.0 / 384.0) * z.w;
        zt = pk_axpy(z, h, pk_scale((float)(1.0 / 5.0), k1));
        rhs(zt, k2);
        zt = pk_axpy(z, h, pk_acc(pk_scale((float)(3.0 / 40.0), k1), (float)(9.0 / 40.0), k2));
        rhs(zt, k3);
        q += (float)(500.0 / 1113.0) * zt.w;
        zt = pk_axpy(z, h, pk_acc(pk_acc(pk_scale((float)(44.0 / 45.0), k1), -(float)(56.0 / 15.0), k2), (float)(32.0 / 9.0), k3));
        rhs(zt, k4);
        q += (float)(125.0 / 192.0) * zt.w;
        zt = pk_axpy(z, h, pk_acc(pk_acc(pk_acc(pk_scale((float)(19372.0 / 6561.0), k1), -(float)(25360.0 / 2187.0), k2), (float)(64448.0 / 6561.0), k3),
                                  -(float)(212.0 / 729.0), k4));
        rhs(zt, k5);
        q -= (float)(2187.0 / 6784.0) * zt.w;
        zt = pk_axpy(z, h, pk_acc(pk_acc(pk_acc(pk_acc(pk_scale((float)(9017.0 / 3168.0), k1), -(float)(355.0 / 33.0), k2), (float)(46732.0 / 5247.0), k3),
                                         (float)(49.0 / 176.0), k4), -(float)(5103.0 / 18656.0), k5));
        rhs(zt, k6);
        if (last0 != nullptr) *last0 = k6.w;
        q += (float)(11.0 / 84.0) * zt.w;
        z = pk_axpy(z, h, pk_acc(pk_acc(pk_acc(pk_acc(pk_scale((float)(35.0 / 384.0), k1), (float)(500.0 / 1113.0), k3), (float)(125.0 / 192.0), k4),
                                        -(float)(2187.0 / 6784.0), k5), (float)(11.0 / 84.0), k6));
        return q;
    }
}

// form of a one-step map (linmap_kernel): Phi (x1 = Phi x0 + S g) for the DC machines' whole-step map, D = Phi - I elsewhere
template <int SYS, int SEG> constexpr bool lin_phi_form() { return SEG == 0 && !SysTraits<SYS>::HAS_ANGLE; }
#ifndef GEMX_PACKED_RHS  // 0: the array code for every system (A/B builds)
#define GEMX_PACKED_RHS 1
#endif
// integrate<>'s dynamic-omega branch (PolynomialStaticLoad) for the three-phase machines in fp32, on PkVec: plain sub-steps, or the one-pass
// kink correction (GEMX_SOLVER_SPLIT_KINKS) -- the same steps as the array code below, which documents them.  hs = sub-step, ns = sub-steps.
template <int SYS, int SOLVER, bool NS1>
__device__ __forceinline__ float integrate_pk(const DevParams<float> &P, float (&y)[SysTraits<SYS>::ND], const float (&u)[MAX_U], float hs, int ns) {
    constexpr int NP = pk_pairs<SYS>();
    static_assert(NP > 0 && SysTraits<SYS>::ND == 1 + 2 * NP, "omega + pairs");
    using V = PkVec<NP>;
    const PkElec<SYS> E(P, u);
    V z;
    z.w = y[0];
#pragma unroll
    for (int j = 0; j < NP; ++j) z.p[j] = f2_t{y[1 + 2 * j], y[2 + 2 * j]};
    auto put = [&]() {
        y[0] = z.w;
#pragma unroll
        for (int j = 0; j < NP; ++j) { y[1 + 2 * j] = z.p[j].x; y[2 + 2 * j] = z.p[j].y; }
    };
    auto rhs = [&](const V &zz, V &dz) { dz.w = poly_load_ode<float>(P, zz.w, E.rhs(P, zz, dz)); };
    if (!P.kink_split) {
        float wsum = 0.0f;
        if (NS1 || ns == 1) {
            V k1;
            rhs(z, k1);
            wsum = rk_step_k1_pk<SOLVER, NP>(z, k1, hs, rhs);
        } else {
            for (int s = 0; s < ns; ++s) {
                V k1;
                rhs(z, k1);
                wsum += rk_step_k1_pk<SOLVER, NP>(z, k1, hs, rhs);
            }
        }
        put();
        return P.pole * hs * wsum;
    }
    float deps = 0.0f;
    const float lim = P.omega_lim;
    const float h_td = hs * P.inv_tau_decay;
    for (int s = 0; s < (NS1 ? 1 : ns); ++s) {
        V k1;
        rhs(z, k1);
        const float w = z.w;
        const float wmid = fmaf(0.5f * hs, k1.w, w);
        const bool band = fabsf(wmid) < lim;
        const float c1 = band ? P.lin_factor : 0.0f, c0 = band ? 0.0f : copysignf(P.la, wmid);
        k1.w = fmaf(med3_r(P.lin_factor * w, -P.la, P.la) - fmaf(c1, w, c0), P.inv_j, k1.w);  // first stage of the model system
        auto rhs_m = [&](const V &zz, V &dz) {
            const float om = zz.w;
            dz.w = (E.rhs(P, zz, dz) - (P.lc * (om * fabsf(om)) + P.lb * om + fmaf(c1, om, c0))) * P.inv_j;
        };
        float dw_end;
        deps += (P.pole * hs) * rk_step_k1_pk<SOLVER, NP>(z, k1, hs, rhs_m, &dw_end);
        const float w1 = z.w;
        const float phi_lim = copysignf(lim, wmid);
        const bool needs = (med3_r(w, -lim, lim) != (band ? w : phi_lim)) | (med3_r(w1, -lim, lim) != (band ? w1 : phi_lim));
        if (__any(needs)) {  // wave-uniform
            const float V0 = hs * k1.w, V1 = hs * dw_end, dl = w1 - w;
            const float c2 = 3.0f * dl - 2.0f * V0 - V1, c3 = V0 + V1 - 2.0f * dl;
            const KinkPath<float> kp{w, w1, V0, V0 * V0, 4.0f * (dl - V0), c2 * (1.0f / 3.0f), c3 * 0.25f, 0.5f * V0,
                                     w + 0.5f * V0 + c2 * (1.0f / 3.0f) + c3 * 0.25f, w1 > w};
            const float lev = (kp.up ? (w < -lim) : !(w > lim)) ? -lim : lim, oth = -lev;
            const float U1 = kp.ramp(lev);
            const bool o0 = w >= oth, o1 = w1 >= oth, full = o0 != o1;
            float U2 = (o0 & o1) ? kp.M - oth : 0.0f;
            if (__any(full)) {
                const float uf = kp.ramp(oth);
                U2 = full ? uf : U2;
            }
            const float Up = lev > 0.0f ? U1 : U2, Um = lev > 0.0f ? U2 : U1;
            const float D = -h_td * ((Um - Up - lim) - (band ? kp.M : phi_lim));
            z.w = w1 + (needs ? D : 0.0f);
        }
    }
    put();
    return deps;
}

// dp5_adaptive on PkVec (three-phase machines, fp32, dynamic omega): the same controller, the stages and the error estimate on pairs
template <int SYS>
__device__ __forceinline__ float dp5_adaptive_pk(const DevParams<float> &P, float (&y)[SysTraits<SYS>::ND], const float (&u)[MAX_U], float hs, float *hc) {
    constexpr int NP = pk_pairs<SYS>(), NZ = 1 + 2 * NP;
    using V = PkVec<NP>;
    const PkElec<SYS> E(P, u);
    V z, k1;
    z.w = y[0];
#pragma unroll
    for (int j = 0; j < NP; ++j) z.p[j] = f2_t{y[1 + 2 * j], y[2 + 2 * j]};
    auto rhs = [&](const V &zz, V &dz) { dz.w = poly_load_ode<float>(P, zz.w, E.rhs(P, zz, dz)); };
    const bool kink = P.kink_split != 0;  // wave-uniform: every attempt on the smooth model system, the kink's defect in closed form (dp5_adaptive: KINKS)
    V k1t;  // the TRUE right-hand side at z
    rhs(z, k1t);
    float t = 0.0f, h = dp5_first_try<float>(P, hs, hc), integral = 0.0f, hprop = 0.0f;
    const float hmin = hs * (1.0f / 1024.0f);
    bool gave_up = false;
#pragma nounroll
    for (int guard = 0; guard < 4096; ++guard) {
        const bool active = t < hs;
        if (!__any(active)) break;
        const bool fin = !(h < hs - t);
        const float hh = active ? (fin ? hs - t : h) : 0.0f;
        V k2, k3, k4, k5, k6, k7, zt, zn;
        k1 = k1t;
        float kc0 = 0.0f, kc1 = 0.0f, wmid = 0.0f;
        bool band = false;
        if (kink) {  // this attempt's model: the region of the Euler-predicted mid-step omega
            wmid = fmaf(0.5f * hh, k1t.w, z.w);
            band = fabsf(wmid) < P.omega_lim;
            kc1 = band ? P.lin_factor : 0.0f;
            kc0 = band ? 0.0f : copysignf(P.la, wmid);
            k1.w = fmaf(med3_r(P.lin_factor * z.w, -P.la, P.la) - fmaf(kc1, z.w, kc0), P.inv_j, k1t.w);  // first stage of the model system
        }
        auto ev = [&](const V &zz, V &dz) {
            const float tq = E.rhs(P, zz, dz);
            const float om = zz.w;
            if (kink) dz.w = (tq - (P.lc * (om * fabsf(om)) + P.lb * om + fmaf(kc1, om, kc0))) * P.inv_j;  // (a wave-uniform branch: not both sides)
            else dz.w = poly_load_ode<float>(P, om, tq);
        };
        float q = (float)(35.0 / 384.0) * z.w;
        zt = pk_axpy(z, hh, pk_scale((float)(1.0 / 5.0), k1));
        ev(zt, k2);
        zt = pk_axpy(z, hh, pk_acc(pk_scale((float)(3.0 / 40.0), k1), (float)(9.0 / 40.0), k2));
        ev(zt, k3);
        q += (float)(500.0 / 1113.0) * zt.w;
        zt = pk_axpy(z, hh, pk_acc(pk_acc(pk_scale((float)(44.0 / 45.0), k1), -(float)(56.0 / 15.0), k2), (float)(32.0 / 9.0), k3));
        ev(zt, k4);
        q += (float)(125.0 / 192.0) * zt.w;
        zt = pk_axpy(z, hh, pk_acc(pk_acc(pk_acc(pk_scale((float)(19372.0 / 6561.0), k1), -(float)(25360.0 / 2187.0), k2), (float)(64448.0 / 6561.0), k3),
                                   -(float)(212.0 / 729.0), k4));
        ev(zt, k5);
        q -= (float)(2187.0 / 6784.0) * zt.w;
        zt = pk_axpy(z, hh, pk_acc(pk_acc(pk_acc(pk_acc(pk_scale((float)(9017.0 / 3168.0), k1), -(float)(355.0 / 33.0), k2), (float)(46732.0 / 5247.0), k3),
                                          (float)(49.0 / 176.0), k4), -(float)(5103.0 / 18656.0), k5));
        ev(zt, k6);
        q += (float)(11.0 / 84.0) * zt.w;
        zn = pk_axpy(z, hh, pk_acc(pk_acc(pk_acc(pk_acc(pk_scale((float)(35.0 / 384.0), k1), (float)(500.0 / 1113.0), k3), (float)(125.0 / 192.0), k4),
                                          -(float)(2187.0 / 6784.0), k5), (float)(11.0 / 84.0), k6));
        ev(zn, k7);
        // error estimate hh (e1 k1 + e3 k3 + e4 k4 + e5 k5 + e6 k6 + e7 k7), scaled per component by atol + rtol max(|z|, |z new|)
        const V er = pk_scale(hh, pk_acc(pk_acc(pk_acc(pk_acc(pk_acc(pk_scale((float)(71.0 / 57600.0), k1), -(float)(71.0 / 16695.0), k3), (float)(71.0 / 1920.0), k4),
                                                        -(float)(17253.0 / 339200.0), k5), (float)(22.0 / 525.0), k6), -(float)(1.0 / 40.0), k7));
        auto sq = [&](float e, float a0, float a1, float at) { const float r = e * rcp_r(at + P.rtol * fmaxf(fabsf(a0), fabsf(a1))); return r * r; };
        float e2 = sq(er.w, z.w, zn.w, P.atol_w);  // (omega: its own absolute tolerance, gemx_config.solver_atol_omega)
#pragma unroll
        for (int j = 0; j < NP; ++j) e2 += sq(er.p[j].x, z.p[j].x, zn.p[j].x, P.atol) + sq(er.p[j].y, z.p[j].y, zn.p[j].y, P.atol);
        const float en2 = e2 * (1.0f / NZ);
        const bool floor_hit = !(hh > hmin);
        const bool accept = active && (!(en2 > 1.0f) || floor_hit);
        gave_up |= active && floor_hit && en2 > 1.0f;
        float fac = en2 > 1e-20f ? 0.9f * pow_m01(en2) : 10.0f;
        fac = fminf(fmaxf(fac, 0.2f), accept ? 10.0f : 1.0f);
        const float w0 = z.w;
        z.w = accept ? zn.w : z.w;
        k1t.w = accept ? k7.w : k1t.w;  // (FSAL: where the sub-step touched no kink the model's slope at the end IS the true one)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            z.p[j] = accept ? zn.p[j] : z.p[j];
            k1t.p[j] = accept ? k7.p[j] : k1t.p[j];
        }
        if (kink) {
            const float lim = P.omega_lim, phi_lim = copysignf(lim, wmid), w1 = zn.w;
            const bool needs = accept && ((med3_r(w0, -lim, lim) != (band ? w0 : phi_lim)) | (med3_r(w1, -lim, lim) != (band ? w1 : phi_lim)));
            if (__any(needs)) {  // wave-uniform: the closed forms, and the true slope at the corrected state
                z.w = z.w + kink_defect<float>(P, w0, w1, hh * k1.w, hh * k7.w, wmid, band, needs, hh);
                V kt;
                rhs(z, kt);
                k1t.w = needs ? kt.w : k1t.w;
#pragma unroll
                for (int j = 0; j < NP; ++j) k1t.p[j] = needs ? kt.p[j] : k1t.p[j];
            }
        }
        integral += accept ? hh * q : 0.0f;
        t = accept ? (fin ? hs : t + hh) : t;
        h = active ? fmaxf(hh * fac, hmin) : h;
        hprop = accept ? fmaxf(hprop, h) : hprop;
    }
    if (gave_up && P.errw != nullptr) atomicOr(P.errw, (uint32_t)GEMX_ERRFLAG_TOLERANCE);
    if (dp5_carries<float>(P, hs, hc)) *hc = hprop;
    y[0] = z.w;
#pragma unroll
    for (int j = 0; j < NP; ++j) { y[1 + 2 * j] = z.p[j].x; y[2 + 2 * j] = z.p[j].y; }
    return integral;
}

// ------------------------------------------------------------------------------------------------
// integrate one segment of length h with `nsteps` sub-steps (EulerSolver(nsteps), solvers.py:103-122).
// y = [omega, motor states].  Returns the angle increment  pole * int(omega dt)  of the scheme.
// ------------------------------------------------------------------------------------------------
// LIN (constant-speed load, one segment of length tau, omega == init[0]): with omega fixed the electrical subsystem is LINEAR with
// constant coefficients, x' = A x + g (g constant over the step), and one step of ANY explicit Runge-Kutta scheme is the affine map
// x1 = Phi x0 + S g with Phi = R(hA), S = h (R(hA) - I)(hA)^-1 (R = the scheme's stability polynomial).  P.lin holds Phi and S as
// linmap_kernel obtained them by pushing unit vectors through rk_step itself; a step is then NM*(NM+NG) FMAs instead of 4 (RK4) or
// 6 (DP5) right-hand sides plus stage combinations.  Same polynomial, so same result up to rounding.
// SEG (LIN only): which of the handle's maps steps this segment (linmap_kernel builds four) -- 0: the whole control step tau; with
// converter dead time a step may be cut at the switching instant (converters.py:302-310): 1 = FIRST segment, of length t_il in the lanes
// with a switching leg (`two`) and tau in the others (per-lane select of the coefficients), 2 = the rest, tau - t_il.
template <int SYS, int LOAD, int SOLVER, class R, bool NS1 = false, bool LIN = false, int SEG = 0>
__device__ __forceinline__ R integrate(const DevParams<R> &P, R (&y)[SysTraits<SYS>::ND], const R (&u)[MAX_U], R h, const R *linr = nullptr, bool two = false,
                                       R *hc = nullptr) {  // hc: the error-controlled solver's step size carried between control steps (dp5_adaptive)
    using E = Elec<SYS, R>;
    constexpr int NM = E::NM;
    const int ns = NS1 ? 1 : P.nsteps;     // NS1: the caller guarantees solver_nsteps == 1 (branch-free code)
    const R hs = NS1 ? h : h * P.inv_ns;
    if (LIN) {
        constexpr int NG = E::NG;
        R g[NG], x[NM];
        // the map's coefficients: from the caller's registers when it preloaded them (lin_preload), else through the device pointer --
        // which, inside a rolled step loop, is TWO vector loads and a full trip to memory on every step (they cannot be hoisted out of the
        // `lin_ok` branch): the run-time-checked copies of the step spent 680 of their 1190 cycles per step there (s_memtime probe)
        constexpr int NC = NM * (NM + NG);
        // (SEG != 0: the caller's registers / the array start at map 1 -- [t_il | tau - t_il | tau], see linmap_kernel)
        const R *L0 = linr != nullptr ? linr : P.lin + (SEG != 0 ? NC : 0);
        R L[NC];
        [[maybe_unused]] uint32_t two_mask = (SEG == 1 && two) ? 0xFFFFFFFFu : 0u;
        if constexpr (SEG == 1 && sizeof(R) == 4) asm volatile("" : "+v"(two_mask));
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if constexpr (SEG == 1 && sizeof(R) == 4) {
                // a per-lane select between two REGISTERS as a bit-field insert under an all-ones / all-zeros lane mask (one v_bfi_b32 per
                // coefficient, VGPR operands only).  Written as `two ? a[i] : b[i]` the optimiser makes ONE load at a selected address of
                // it -- which demotes the preloaded coefficients to scratch memory and puts two scratch loads on every step -- and an
                // opaque copy of both operands in front of the select (round 3a) cost two v_mov per coefficient and step; the mask is
                // opaque instead, so that the and / or form is not folded back into that select
                uint32_t ca, cb;
                memcpy(&ca, &L0[2 * NC + i], 4);  // tau
                memcpy(&cb, &L0[i], 4);           // t_il
                const uint32_t cs = (two_mask & cb) | (~two_mask & ca);
                memcpy(&L[i], &cs, 4);
            } else if constexpr (SEG == 1) {
                L[i] = two ? L0[i] : L0[2 * NC + i];
            } else {
                L[i] = L0[SEG == 2 ? NC + i : i];
            }
        }
        const R om = P.init[0];  // == y[0] in every lane (lin_usable); wave-uniform, so everything derived from it is loop-invariant
        E::get_b(E::prep(P, om, u), g);
#pragma unroll
        for (int i = 0; i < NM; ++i) x[i] = y[1 + i];
        for (int s = 0; s < ns; ++s) {
            R xn[NM];
#pragma unroll
            for (int r = 0; r < NM; ++r) {
                // Phi form: S g + Phi x.  D form, whole step (the three-phase machines, the headline): x + S g + D x with x as the FIRST
                // addend -- the same instruction count as the Phi form (the leading multiply becomes a fused multiply-add), every
                // partial sum rounded at |x|'s scale, unbiased.  D form, dead-time segments: the small terms first, then x.
                // (x first for the synchronous and squirrel-cage machines: 3.2e-6 / 4.4e-6 against their recorded runs; the EESM and the
                // DFIM, whose five- and four-state maps have more partial sums to round and whose steps are 150+ instructions anyway, add x
                // LAST like the dead-time segments: one rounding at |x|'s scale)
                constexpr bool X_FIRST = SEG == 0 && (SYS == GEMX_SYS_SYNC || SYS == GEMX_SYS_SCIM);
                constexpr bool X_LAST = !lin_phi_form<SYS, SEG>() && !X_FIRST;
                R acc = X_FIRST ? fma(L[NM * NM + r * NG], g[0], x[r]) : L[NM * NM + r * NG] * g[0];
#pragma unroll
                for (int i = 1; i < NG; ++i) acc += L[NM * NM + r * NG + i] * g[i];
#pragma unroll
                for (int c = 0; c < NM; ++c) acc += L[r * NM + c] * x[c];
                xn[r] = X_LAST ? x[r] + acc : acc;
            }
#pragma unroll
            for (int r = 0; r < NM; ++r) x[r] = xn[r];
        }
#pragma unroll
        for (int i = 0; i < NM; ++i) y[1 + i] = x[i];
        return P.pole * om * h;
    }
    if (LOAD == GEMX_LOAD_CONST_SPEED) {
        // omega is constant (constant_speed_load.py:40-42): integrate the electrical states only; every scheme's
        // quadrature weights sum to one, so the angle increment is exactly pole * omega * h.
        const typename E::Pre pre = E::prep(P, y[0], u);
        R x[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) x[i] = y[1 + i];
        auto rhs = [&](const R (&xx)[NM], R (&dx)[NM]) { E::f(P, pre, xx, dx); };
        if (SOLVER == GEMX_SOLVER_DP5 && P.adaptive) {
            dp5_adaptive<NM, R>(P, x, h, rhs, hc);
        } else if (NS1 || ns == 1) {
            rk_step<SOLVER, NM, R>(x, hs, rhs);
        } else {
            for (int s = 0; s < ns; ++s) rk_step<SOLVER, NM, R>(x, hs, rhs);
        }
#pragma unroll
        for (int i = 0; i < NM; ++i) y[1 + i] = x[i];
        return P.pole * y[0] * h;
    } else {
        if constexpr (sizeof(R) == 4 && pk_pairs<SYS>() > 0 && GEMX_PACKED_RHS) {
            // fp32, three-phase machine: the packed form of the same schemes (see PkElec); error control keeps the array code
            if (!(SOLVER == GEMX_SOLVER_DP5 && P.adaptive)) return integrate_pk<SYS, SOLVER, NS1>(P, y, u, hs, ns);
            if constexpr (SOLVER == GEMX_SOLVER_DP5) return P.pole * dp5_adaptive_pk<SYS>(P, y, u, h, hc);
        }
        // SCMLSystem._system_equation (physical_systems.py:205-236): [load derivative, motor derivative]
        auto rhs = [&](const R (&z)[NM + 1], R (&dz)[NM + 1]) {
            R x[NM], dx[NM];
#pragma unroll
            for (int i = 0; i < NM; ++i) x[i] = z[1 + i];
            const typename E::Pre pre = E::prep(P, z[0], u);
            E::f(P, pre, x, dx);
            dz[0] = poly_load_ode(P, z[0], E::torque(P, x));
#pragma unroll
            for (int i = 0; i < NM; ++i) dz[1 + i] = dx[i];
        };
        if (SOLVER == GEMX_SOLVER_DP5 && P.adaptive) {
            // (with GEMX_SOLVER_SPLIT_KINKS: every attempt on the smooth model system c0 + c1 omega of the load's saturation, see dp5_adaptive)
            auto rhs_mc = [&](const R (&z)[NM + 1], R (&dz)[NM + 1], R c0, R c1) {
                R x[NM], dx[NM];
#pragma unroll
                for (int i = 0; i < NM; ++i) x[i] = z[1 + i];
                const typename E::Pre pre = E::prep(P, z[0], u);
                E::f(P, pre, x, dx);
                dz[0] = (E::torque(P, x) - (P.lc * (z[0] * fabs(z[0])) + P.lb * z[0] + fma(c1, z[0], c0))) * P.inv_j;
#pragma unroll
                for (int i = 0; i < NM; ++i) dz[1 + i] = dx[i];
            };
            return P.pole * dp5_adaptive<NM + 1, R>(P, y, h, rhs, hc, rhs_mc);
        }
        if (!P.kink_split) {
            R wsum = R(0);
            if (NS1 || ns == 1) {
                wsum = rk_step<SOLVER, NM + 1, R>(y, hs, rhs);
            } else {
                for (int s = 0; s < ns; ++s) wsum += rk_step<SOLVER, NM + 1, R>(y, hs, rhs);
            }
            return P.pole * hs * wsum;
        }
        // GEMX_SOLVER_SPLIT_KINKS (PolynomialStaticLoad only).  The load torque's constant term is the saturation
        // sigma(omega) = clamp(J / tau_decay * omega, -a, a) (polynomial_static_load.py:87-92): the right-hand side has kinks at
        // |omega| = omega_lim, where a fixed step loses its order -- the reference's adaptive default solver rejects and splits exactly
        // those steps.  Rounds 2-3 cut the step at the predicted crossings, up to three passes of the scheme for the whole wave whenever ANY
        // lane crossed (BASELINE config 4: 0.70 -> 0.37 of the roofline).  Now ONE pass, for every lane:
        //   * the pass integrates a SMOOTH system: sigma replaced by the affine piece sigma_m(omega) = c0 + c1 omega of the region the
        //     Euler-predicted mid-step omega lies in (-a | J / tau_decay omega | +a), so the scheme keeps its order whatever the lanes do;
        //   * the defect D = omega_true - omega_model obeys D' = -(1 / tau_decay) [clamp(omega, -lim, lim) - phi_m(omega)] (phi_m: region m's
        //     piece of the clamp, extended), integrated to first order along the model's own omega path -- the cubic through both ends
        //     of the step with the slopes of the scheme's first and last stage (the last stage sits at t + h) -- in closed form:
        //     int clamp = -lim + U(-lim) - U(lim) with the ramp integrals U(c) = int (omega - c)_+ dt; U needs the instant the path
        //     crosses c, but is stationary in it (the integrand vanishes there), so the root of the quadratic through both ends with the
        //     start slope is accurate enough; D is added to omega at the end of the step.
        // Lanes whose path stays inside region m have D = 0 exactly; the closed forms run behind a wave ballot.  Against the recorded runs
        // of the reference's default solver this is MORE accurate than the cuts were (fp64, worst over every PolynomialStaticLoad fixture:
        // 6.3e-6 against 2.0e-5; tools/oracle_solver_scan.py), because the crossing is located a posteriori from a third-order path instead of
        // predicted from the first stage.  (The test suite's fp64 CPU restatement follows the same steps: 1e-9 agreement with the fp64 build.)
        R deps = R(0);
        const R lim = P.omega_lim;
        const R h_td = hs * P.inv_tau_decay;
        for (int s = 0; s < ns; ++s) {
            R k1[NM + 1];
            rhs(y, k1);
            const R w = y[0];
            const R wmid = fma(R(0.5) * hs, k1[0], w);
            const bool band = fabs(wmid) < lim;
            const R c1 = band ? P.lin_factor : R(0), c0 = band ? R(0) : copysign(P.la, wmid);
            k1[0] = fma(med3_r(P.lin_factor * w, -P.la, P.la) - fma(c1, w, c0), P.inv_j, k1[0]);  // first stage of the model system
            auto load_m = [&](R om, R torque) { return (torque - (P.lc * (om * fabs(om)) + P.lb * om + fma(c1, om, c0))) * P.inv_j; };
            auto rhs_m = [&](const R (&z)[NM + 1], R (&dz)[NM + 1]) {
                R x[NM], dx[NM];
#pragma unroll
                for (int i = 0; i < NM; ++i) x[i] = z[1 + i];
                const typename E::Pre pre = E::prep(P, z[0], u);
                E::f(P, pre, x, dx);
                dz[0] = load_m(z[0], E::torque(P, x));
#pragma unroll
                for (int i = 0; i < NM; ++i) dz[1 + i] = dx[i];
            };
            R dw_end;  // d omega / dt of the last stage (at t + hs)
            deps += (P.pole * hs) * rk_step_k1<SOLVER, NM + 1, R>(y, k1, hs, rhs_m, &dw_end);
            const R w1 = y[0];
            const R phi_lim = copysign(lim, wmid);
            const bool needs = (med3_r(w, -lim, lim) != (band ? w : phi_lim)) | (med3_r(w1, -lim, lim) != (band ? w1 : phi_lim));
#ifdef GEMX_KINK_NO_DEFECT  // timing-only A/B build: what the closed forms behind the ballot cost (results are WRONG in steps that cross a kink)
            if (false) {
#else
            if (__any(needs)) {  // wave-uniform
#endif
                const R V0 = hs * k1[0], V1 = hs * dw_end, dl = w1 - w;
                const R c2 = R(3) * dl - R(2) * V0 - V1, c3 = V0 + V1 - R(2) * dl;
                const KinkPath<R> kp{w, w1, V0, V0 * V0, R(4) * (dl - V0), c2 * R(1.0 / 3.0), c3 * R(0.25), R(0.5) * V0,
                                     w + R(0.5) * V0 + c2 * R(1.0 / 3.0) + c3 * R(0.25), w1 > w};
                // the first kink ahead of the start in the direction of travel; the other one is crossed in the same step only by a lane
                // that passes the whole band (rare: second ballot)
                const R lev = (kp.up ? (w < -lim) : !(w > lim)) ? -lim : lim, oth = -lev;
                const R U1 = kp.ramp(lev);
                const bool o0 = w >= oth, o1 = w1 >= oth, full = o0 != o1;
                R U2 = (o0 & o1) ? kp.M - oth : R(0);
                if (__any(full)) {
                    const R uf = kp.ramp(oth);
                    U2 = full ? uf : U2;
                }
                const R Up = lev > R(0) ? U1 : U2, Um = lev > R(0) ? U2 : U1;
                const R D = -h_td * ((Um - Up - lim) - (band ? kp.M : phi_lim));
                y[0] = w1 + (needs ? D : R(0));
            }
        }
        return deps;
    }
}

// The constant-speed electrical step (one sub-step of length tau) split at its state-independent part, for dc_stream_kernel: elec_input() is
// everything that depends on (omega, u) only, elec_apply() advances x with it.  Together they are integrate<SYS, CONST_SPEED, SOLVER, R,
// NS1 = true, LIN>() operation by operation (same expressions in the same order: bit-identical results, asserted by the tests).
template <int SYS, class R, bool LIN>
__device__ __forceinline__ void elec_input(const DevParams<R> &P, R om, const R (&u)[MAX_U], const R *L, R (&in)[Elec<SYS, R>::NM]) {
    using E = Elec<SYS, R>;
    constexpr int NM = E::NM, NG = E::NG;
    static_assert(NG == NM, "the DC machines' affine part has one entry per state");
    R g[NG];
    E::get_b(E::prep(P, om, u), g);
#pragma unroll
    for (int r = 0; r < NM; ++r) {
        if (LIN) {
            R acc = L[NM * NM + r * NG] * g[0];
#pragma unroll
            for (int i = 1; i < NG; ++i) acc += L[NM * NM + r * NG + i] * g[i];
            in[r] = acc;
        } else {
            in[r] = g[r];
        }
    }
}
template <int SYS, int SOLVER, class R, bool LIN>
__device__ __forceinline__ void elec_apply(const DevParams<R> &P, R om, R (&x)[Elec<SYS, R>::NM], const R (&in)[Elec<SYS, R>::NM], const R *L) {
    using E = Elec<SYS, R>;
    constexpr int NM = E::NM;
    if (LIN) {
        R xn[NM];
#pragma unroll
        for (int r = 0; r < NM; ++r) {
            R acc = in[r];
#pragma unroll
            for (int c = 0; c < NM; ++c) acc += L[r * NM + c] * x[c];
            xn[r] = acc;
        }
#pragma unroll
        for (int r = 0; r < NM; ++r) x[r] = xn[r];
    } else {
        const R zu[MAX_U] = {R(0), R(0), R(0), R(0)};
        const typename E::Pre pre = E::set_b(E::prep(P, om, zu), in);  // (the omega-only entries; the affine part is the pre-wave's)
        auto rhs = [&](const R (&xx)[NM], R (&dx)[NM]) { E::f(P, pre, xx, dx); };
        rk_step<SOLVER, NM, R>(x, P.tau, rhs);
    }
}

// ------------------------------------------------------------------------------------------------
// converters: phase voltages for one segment
// ------------------------------------------------------------------------------------------------
// ContTwoQuadrantConverter via ContDynamicallyAveragedConverter (converters.py:144-158, 177-184, 425-427):
// clip(duty - sign(i) * t_il / tau, 0, 1).  IL = false: t_il == 0, the current is irrelevant.
template <bool IL, class R> __device__ __forceinline__ R cont_leg(const DevParams<R> &P, R duty, R i) {
    if (!IL) return duty;  // duty is already clipped to [0, 1]
    return clip01(duty - sgn(i) * P.il_ratio);
}
// FiniteTwoQuadrantConverter.convert (converters.py:277-285): leg state 1 -> upper rail, 2 -> lower rail, 0 (dead)
// -> freewheeling diode (upper rail while i < 0).  Branch-free; returns +-0.5 * u_sup.
template <class R> __device__ __forceinline__ R fin_leg_u(uint32_t st, R i, R half_us) {
    const bool upper = (st == 1u) | ((st == 0u) & (i < R(0)));
    return upper ? half_us : -half_us;
}
// Finite-B6C action -> per-leg sub-action (1 = upper, 2 = lower), converters.py:788-797, packed 2 bits per leg.
// b6_subactions(): test-and-select per leg -- transparent to the optimiser, which folds a later `leg state == 1` back into the action
// bit (the RC supply's i_sup of a dead-time-free bridge, Stepper::legs_of: the opaque look-up below cost that path 12 %).
// b6_subactions_packed(): a look-up in a 48-bit constant (8 actions x 6 bits), three instructions: for the dead-time code, which works
// on the packed value (b6_interlock).
constexpr uint32_t b6_subactions_of(uint32_t a) {
    return ((a & 4u) ? 1u : 2u) | (((a & 2u) ? 1u : 2u) << 2) | (((a & 1u) ? 1u : 2u) << 4);
}
__device__ __forceinline__ uint32_t b6_subactions(uint32_t a) { return b6_subactions_of(a); }
constexpr uint64_t b6_subaction_table() {
    uint64_t t = 0;
    for (uint32_t a = 0; a < 8; ++a) t |= (uint64_t)b6_subactions_of(a) << (6 * a);
    return t;
}
__device__ __forceinline__ uint32_t b6_subactions_packed(uint32_t a) {
    constexpr uint64_t T = b6_subaction_table();
    return (uint32_t)(T >> (6u * (a & 7u))) & 63u;
}
// Interlocking (FiniteTwoQuadrantConverter._set_switching_pattern 300-310 + convert 270-276 as driven by
// *.simulate(), which passes the segment START time): a leg that changes between upper and lower goes to the
// dead state 0 for the WHOLE step (two segments [t, t+t_il], [t+t_il, t+tau]) and takes the new state on the
// next step.  Returns the leg states used during this step; `two` = this env integrates two segments.
// All legs at once: a leg's states are 1 (upper), 2 (lower), 0 (dead) and its sub-action is 1 or 2, so "changes between upper and lower"
// is exactly prev ^ want == 3 in that leg's two bits -- a dead leg (0 ^ want = want) never does, an unchanged one gives 0.
template <int NLEG = 3> __device__ __forceinline__ uint32_t b6_interlock(uint32_t prev, uint32_t want, bool &two) {
    constexpr uint32_t LOW = ((1u << (2 * NLEG)) - 1u) / 3u;  // bit 0 of every leg: 0b0101...
    const uint32_t x = prev ^ want;
    const uint32_t t = x & (x >> 1) & LOW;  // bit 0 of every switching leg
    two = t != 0u;
    return want & ~(t * 3u);                // switching legs are dead for this step
}

// B6 bridges: phase voltages u_a, u_b, u_c [V].  `legs` only matters for Finite-B6C with IL; A0 = index of this bridge's
// first entry in the (concatenated) continuous action.
template <int CONV, bool IL, class R, int A0 = 0>
__device__ __forceinline__ void b6_voltages(const DevParams<R> &P, const R (&act)[MAX_ACT], uint32_t dact, uint32_t legs, R ia, R ib, R ic,
                                            R &ua, R &ub, R &uc) {
    if (CONV == GEMX_CONV_CONT_B6) {  // converters.py:888-903
        ua = (cont_leg<IL, R>(P, duty_pos(act[A0]), ia) - R(0.5)) * P.u_sup;
        ub = (cont_leg<IL, R>(P, duty_pos(act[A0 + 1]), ib) - R(0.5)) * P.u_sup;
        uc = (cont_leg<IL, R>(P, duty_pos(act[A0 + 2]), ic) - R(0.5)) * P.u_sup;
    } else {  // converters.py:816-823
        const R hu = R(0.5) * P.u_sup;
        if (!IL) {  // every leg follows its sub-action immediately: upper rail iff the action bit is set
            ua = (dact & 4u) ? hu : -hu;
            ub = (dact & 2u) ? hu : -hu;
            uc = (dact & 1u) ? hu : -hu;
        } else {
            ua = fin_leg_u<R>(legs & 3u, ia, hu);
            ub = fin_leg_u<R>((legs >> 2) & 3u, ib, hu);
            uc = fin_leg_u<R>((legs >> 4) & 3u, ic, hu);
        }
    }
}

// cos/sin of the rotor-flux angle eps_fs = atan2(psi_b, psi_a) (calculate_field_angle, physical_systems.py:765-769) without atan2
__device__ __forceinline__ float rsqrt_r(float x) { return __frsqrt_rn(x); }  // V_RSQ_F32, 1 ulp (an IEEE division is ~12 instructions)
__device__ __forceinline__ double rsqrt_r(double x) { return 1.0 / sqrt(x); }
template <class R> __device__ __forceinline__ void flux_angle(R pa, R pb, R &s, R &c) {
    // Branch- and compare-free: both components scaled by 2^40 (exact; the squares of any flux from 1e-31 to 1e6 Wb stay normal
    // numbers) and 2^-50 added to the alpha component -- invisible next to a flux above 1e-20 Wb, and a flux of exactly zero (every
    // episode starts there) gives c = 1, s = 0, i.e. atan2(0, 0) = 0 as the reference's np.arctan2 does.  (Round 2: three compares and
    // four selects per call, each compare a VALU-written mask with its `s_nop`.)
    const R qa = fma(pa, R(1099511627776.0), R(8.8817841970012523e-16)), qb = pb * R(1099511627776.0);
    const R rn = rsqrt_r(qa * qa + qb * qb);
    c = qa * rn;
    s = qb * rn;
}

// ------------------------------------------------------------------------------------------------
// action stage in front of the converter (gemx_config.action_frame): (u_d, u_q[, u_e]) -> (u_a, u_b, u_c[, u_e]).
//   control_space='dq' (physical_systems.py:491-492 / 777-778): Park angle = the step-start electrical angle (synchronous
//     motors) or rotor-flux angle (SCIM);
//   DqToAbcActionProcessor (dq_to_abc_action_processor.py:98-114, EESM 158-175): angle of the last returned state advanced
//     by (0.5 + dead-time steps) * tau * omega * p.
// ------------------------------------------------------------------------------------------------
template <int SYS, int CONV, class R>
__device__ __forceinline__ void dq_action_stage(const DevParams<R> &P, const R (&y)[SysTraits<SYS>::ND], typename Angle<R>::T ang,
                                                R (&act)[MAX_ACT]) {
    if (!conv_dq<CONV>()) return;
    R s, c;
    if (P.dq_processor) Angle<R>::sincos(Angle<R>::advance(ang, P.dq_adv * y[0]), s, c);
    else if (SYS == GEMX_SYS_SCIM) flux_angle<R>(y[SysTraits<SYS>::ND - 2], y[SysTraits<SYS>::ND - 1], s, c);
    else Angle<R>::sincos(ang, s, c);
    const R ud = act[0], uq = act[1], ue = act[2];
    t32(c * ud - s * uq, s * ud + c * uq, act[0], act[1], act[2]);
    if (CONV == CONV_CONT_B6_4QC_DQ) act[3] = ue;
}

// ------------------------------------------------------------------------------------------------
// one control step of one env, in two halves so that they can run in different waves (advance_pipe_kernel):
//   advance(): converter -> voltages -> ODE integration -> new (y, ang, sw); leaves in ho[] what observe() needs
//   observe(): normalised observation row from (y, ang, ho)
//   state_violation(): the value whose excess over 1 violates the env's DEFAULT constraint, from (y, ho) with the same arithmetic as
//     default_done() applies to the row
// step() = advance() + observe() (single-wave kernel).
// ------------------------------------------------------------------------------------------------
// the angle after a segment whose integrate<> returned `deps`.  First dead-time segment of the one-step-map instantiations (LIN, SEG == 1):
// omega is the wave-uniform init[0] there and the segment is t_il or tau long, so BOTH possible increments are launch constants (hoisted
// out of the step loop) and the lane picks one -- a select and an add instead of the float -> fixed-point conversion of a per-lane product
// (8 instructions).  Same expressions as integrate<>'s `pole * omega * h`, so the same bits.
template <class R, bool LIN, int SEG>
__device__ __forceinline__ typename Angle<R>::T advance_angle(const DevParams<R> &P, typename Angle<R>::T ang, R deps, bool two) {
    if constexpr (LIN && SEG == 1) {
        const R w = P.pole * P.init[0];
        const typename Angle<R>::Inc i_il = Angle<R>::increment(w * P.t_il), i_tau = Angle<R>::increment(w * P.tau);
        return Angle<R>::add(ang, two ? i_il : i_tau);
    } else {
        return Angle<R>::advance(ang, deps);
    }
}

template <int SYS, int CONV, int LOAD, int SOLVER, bool IL, class R> struct Stepper;

// ---- DcMotorSystem (physical_systems.py:171-203, 290-318): permanently excited, series, shunt motors behind ONE
// Cont-4QC (converters.py:438-495) / Finite-4QC (313-368); externally excited motor behind a MultiConverter of TWO
// (converters.py:498-740: armature, excitation) ----------------------------------------------------------------------
template <int SYS, int CONV, int LOAD, int SOLVER, bool IL, class R> struct DcStepper {
    using AngT = typename Angle<R>::T;
    static constexpr int ND = SysTraits<SYS>::ND, NOUT = SysTraits<SYS>::NOUT, NC = ND - 1;
    static constexpr int NU = SYS == GEMX_SYS_DC_EXTEX ? 2 : 1;  // converter output voltages
    static constexpr int NH = NU;                                // ho: u [V]
    static constexpr int NVT = 0;                                // no per-action voltage table (see the synchronous machines)
    static constexpr int row_slot(int j) { return j; }           // hand-off row of the pipelined kernel: logical index -> LDS slot
    static constexpr bool CONT = CONV == GEMX_CONV_CONT_4QC || CONV == GEMX_CONV_CONT_2X4QC;
    // leg states a finite converter WITHOUT dead time is left in by the action `dact` (what the IL code stores in `sw`): needed by the
    // RC supply's i_sup of the next step when the dead-time-free instantiation serves the handle
    static __device__ __forceinline__ uint32_t legs_of(uint32_t dact) {
        uint32_t legs = 0;
        if (!CONT) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const uint32_t aj = (dact >> (2 * j)) & 3u;
                legs |= (((aj & 2u) ? 2u : 1u) | (((aj & 1u) ? 2u : 1u) << 2)) << (4 * j);
            }
        }
        return legs;
    }
    static_assert((CONV == GEMX_CONV_CONT_2X4QC || CONV == GEMX_CONV_FINITE_2X4QC) == (NU == 2), "system / converter width mismatch");
    // i_in = motor.i_in(currents): the current (dc_permanently_excited_motor.py:77-79, dc_series_motor.py:85-87),
    // i_a + i_e for the shunt motor (dc_shunt_motor.py:68-70), [i_a, i_e] for the externally excited motor
    // (dc_motor.py:110-112; MultiConverter.convert slices it per sub-converter, converters.py:550-558)
    static __device__ __forceinline__ R i_in(const R (&y)[ND], int j) {
        if (SYS == GEMX_SYS_DC_EXTEX) return y[1 + j];
        return SYS == GEMX_SYS_DC_SHUNT ? y[1] + y[ND - 1] : y[1];
    }
    // converter output WITHOUT dead time: a function of the action alone (dc_stream_kernel evaluates it off the integrator's wave)
    static __device__ __forceinline__ void input_voltages(const DevParams<R> &P, const R (&act)[MAX_ACT], uint32_t dact, R (&u)[MAX_U]) {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            if (CONT) {
                u[j] = (duty_pos(act[j]) - duty_neg(act[j])) * P.u_sup;
            } else {  // Finite-4QC: action -> (leg0, leg1) sub-actions [1,1,2,2] / [1,2,1,2] (converters.py:360-361); 1 = upper rail
                const uint32_t aj = (dact >> (2 * j)) & 3u;
                u[j] = (((aj & 2u) ? R(0) : R(1)) - ((aj & 1u) ? R(0) : R(1))) * P.u_sup;
            }
        }
    }
    template <bool NS1 = false, bool LIN = false, bool TAB = false>
    static __device__ __forceinline__ void advance(const DevParams<R> &P, R (&y)[ND], AngT &, uint32_t &sw, const R (&act)[MAX_ACT], uint32_t dact,
                                                   R (&ho)[NH], const R * = nullptr, const R *linr = nullptr, R *hc = nullptr) {
        R u[MAX_U] = {R(0), R(0), R(0), R(0)};
        if (CONT) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const R d0 = duty_pos(act[j]);
                const R d1 = duty_neg(act[j]);
                const R i = i_in(y, j);
                u[j] = (cont_leg<IL, R>(P, d0, i) - cont_leg<IL, R>(P, d1, i)) * P.u_sup;  // both sub-converters see the same i (line 483)
            }
            // (one segment; the dead-time instantiations' registers hold maps 1..3, so they name the whole step as `first segment, no
            // switching leg`: integrate<..., SEG = 1>(two = false) takes the tau map among them)
            integrate<SYS, LOAD, SOLVER, R, NS1, LIN, IL ? 1 : 0>(P, y, u, P.tau, linr, false, hc);
        } else {  // Finite-4QC: action -> (leg0, leg1) sub-actions [1,1,2,2] / [1,2,1,2] (converters.py:360-361)
            uint32_t legs = 0;
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const uint32_t aj = (dact >> (2 * j)) & 3u;
                legs |= (((aj & 2u) ? 2u : 1u) | (((aj & 1u) ? 2u : 1u) << 2)) << (4 * j);
            }
            bool two = false;
            if (IL) {
                if (P.t_il > R(0)) legs = b6_interlock<2 * NU>(sw, legs, two);
                sw = legs;
            }
            auto segment = [&](R h, auto seg_tag) {
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    const uint32_t s0 = (legs >> (4 * j)) & 3u, s1 = (legs >> (4 * j + 2)) & 3u;
                    R v0, v1;
                    if (!IL) {
                        v0 = s0 == 1u ? R(1) : R(0);
                        v1 = s1 == 1u ? R(1) : R(0);
                    } else {  // converters.py:350-352: leg 1 sees -i_out; dead leg -> freewheeling diode (277-285)
                        const R i = i_in(y, j);
                        v0 = ((s0 == 1u) | ((s0 == 0u) & (i < R(0)))) ? R(1) : R(0);
                        v1 = ((s1 == 1u) | ((s1 == 0u) & (-i < R(0)))) ? R(1) : R(0);
                    }
                    u[j] = (v0 - v1) * P.u_sup;
                }
                integrate<SYS, LOAD, SOLVER, R, NS1, LIN, decltype(seg_tag)::value>(P, y, u, h, linr, two, hc);
            };
            if (IL) {
                segment(two ? P.t_il : P.tau, std::integral_constant<int, 1>{});
                if (two) segment(P.tau - P.t_il, std::integral_constant<int, 2>{});
            } else {
                segment(P.tau, std::integral_constant<int, 0>{});
            }
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) ho[j] = u[j];
    }
    static __device__ __forceinline__ void observe(const DevParams<R> &P, const R (&y)[ND], AngT, const R (&ho)[NH], R (&obs)[NOUT]) {
        R x[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) x[c] = y[1 + c];
        obs[0] = y[0] * P.inv_lim[0];
        obs[1] = Elec<SYS, R>::torque(P, x) * P.inv_lim[1];
#pragma unroll
        for (int c = 0; c < NC; ++c) obs[2 + c] = y[1 + c] * P.inv_lim[2 + c];
#pragma unroll
        for (int j = 0; j < NU; ++j) obs[2 + NC + j] = ho[j] * P.inv_lim[2 + NC + j];
        obs[2 + NC + NU] = P.u_sup * P.inv_lim[2 + NC + NU];
    }
    // default constraints of the DC envs: LimitConstraint on the current(s): ('i',) (cont_cc_permex_dc_env.py:104,
    // cont_cc_series_dc_env.py:102) / ('i_a', 'i_e') (cont_cc_shunt_dc_env.py:103, cont_cc_extex_dc_env.py)
    static __device__ __forceinline__ bool default_done(const R (&obs)[NOUT]) {
        bool d = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) d |= fabs(obs[2 + c]) > R(1);
        return d;
    }
    static __device__ __forceinline__ R state_violation(const DevParams<R> &P, const R (&y)[ND], const R (&)[NH]) {
        R v = fabs(y[1] * P.inv_lim[2]);
#pragma unroll
        for (int c = 1; c < NC; ++c) v = fmax(v, fabs(y[1 + c] * P.inv_lim[2 + c]));
        return v;
    }
};
template <int CONV, int LOAD, int SOLVER, bool IL, class R>
struct Stepper<GEMX_SYS_DC_PERMEX, CONV, LOAD, SOLVER, IL, R> : DcStepper<GEMX_SYS_DC_PERMEX, CONV, LOAD, SOLVER, IL, R> {};
template <int CONV, int LOAD, int SOLVER, bool IL, class R>
struct Stepper<GEMX_SYS_DC_SERIES, CONV, LOAD, SOLVER, IL, R> : DcStepper<GEMX_SYS_DC_SERIES, CONV, LOAD, SOLVER, IL, R> {};
template <int CONV, int LOAD, int SOLVER, bool IL, class R>
struct Stepper<GEMX_SYS_DC_SHUNT, CONV, LOAD, SOLVER, IL, R> : DcStepper<GEMX_SYS_DC_SHUNT, CONV, LOAD, SOLVER, IL, R> {};
template <int CONV, int LOAD, int SOLVER, bool IL, class R>
struct Stepper<GEMX_SYS_DC_EXTEX, CONV, LOAD, SOLVER, IL, R> : DcStepper<GEMX_SYS_DC_EXTEX, CONV, LOAD, SOLVER, IL, R> {};

// ---- SynchronousMotorSystem (physical_systems.py:487-525), control_space 'abc' ---------------------------------
template <int CONV, int LOAD, int SOLVER, bool IL, class R> struct Stepper<GEMX_SYS_SYNC, CONV, LOAD, SOLVER, IL, R> {
    using AngT = typename Angle<R>::T;
    static constexpr int NH = 7;  // ho: sin, cos of the last segment-start angle, u_a, u_b, u_c, u_sd, u_sq
    static __device__ __forceinline__ uint32_t legs_of(uint32_t dact) { return CONV == GEMX_CONV_FINITE_B6 ? b6_subactions(dact) : 0u; }
    // Finite-B6C without dead time and with an ideal supply: the bridge's output is a function of the action index alone, so the
    // pipelined kernel keeps it in an 8-entry LDS table (u_a, u_b, u_c, u_alpha, u_beta per switching state), computed ONCE per launch
    // with the very code below, instead of decoding the action and Clarke-transforming it in every step.
    static constexpr int NVT = (CONV == GEMX_CONV_FINITE_B6 && !IL) ? 5 : 0;
    // Hand-off row of the pipelined kernel, logical order [omega, i_sd, i_sq, eps, sin, cos, u_a, u_b, u_c, u_sd, u_sq, done] -> LDS
    // slots grouped the way the integrator's registers come out of its instructions, so that the three ds_write_b128 need (almost) no
    // v_mov packing: [i_sd, i_sq (a v_pk_fma result pair), omega, eps | sin, cos (V_SIN / V_COS), u_sd, u_sq (pair) | u_a, u_b, u_c
    // (the table entry's quad), done]
    static constexpr int row_slot(int j) {
        constexpr int slot[12] = {2, 0, 1, 3, 4, 5, 8, 9, 10, 6, 7, 11};
        return j < 12 ? slot[j] : j;  // (entry 12: the per-lane supply voltage of the FULL variant)
    }
    static __device__ __forceinline__ void action_entry(const DevParams<R> &P, uint32_t dact, R (&e)[8]) {
        const R zero[MAX_ACT] = {R(0), R(0), R(0), R(0), R(0), R(0)};
        // entry = [u_alpha, u_beta | u_a, u_b, u_c]: the pair the integrator needs first, 8-byte aligned (ONE LDS read per step where the
        // hand-off row is compact and u_abc is looked up by the output waves: advance_pipe_kernel, COMPACT_K)
        b6_voltages<CONV, false, R>(P, zero, dact, 0u, R(0), R(0), R(0), e[2], e[3], e[4]);
        t23(e[2], e[3], e[4], e[0], e[1]);
        e[5] = e[6] = e[7] = R(0);
    }
    template <bool NS1 = false, bool LIN = false, bool TAB = false>
    static __device__ __forceinline__ void advance(const DevParams<R> &P, R (&y)[3], AngT &ang, uint32_t &sw, const R (&act)[MAX_ACT],
                                                   uint32_t dact, R (&ho)[NH], const R *tab = nullptr, const R *linr = nullptr, R *hc = nullptr) {
        R s, c;
        Angle<R>::sincos(ang, s, c);
        uint32_t legs = 0;
        bool two = false;
        if (IL && CONV == GEMX_CONV_FINITE_B6) {
            legs = b6_subactions_packed(dact);
            if (P.t_il > R(0)) legs = b6_interlock(sw, legs, two);  // converters.py:302: no dead time -> pattern [action]
            sw = legs;
        }
        R ua, ub, uc, u[MAX_U] = {R(0), R(0), R(0), R(0)};
        auto segment = [&](R h, auto seg_tag) {
            R ia = R(0), ib = R(0), ic = R(0);
            if (IL) {  // i_in = T32(Q(i_dq, eps)) (line 493/505); only its sign matters (dead legs / cont. interlocking)
                const R ial = c * y[1] - s * y[2], ibe = s * y[1] + c * y[2];
                t32(ial, ibe, ia, ib, ic);
            }
            R ual, ube;
            if (TAB) {  // this action's table entry (action_entry)
                ual = tab[0]; ube = tab[1]; ua = tab[2]; ub = tab[3]; uc = tab[4];
            } else {
                b6_voltages<CONV, IL, R>(P, act, dact, legs, ia, ib, ic, ua, ub, uc);
                t23(ua, ub, uc, ual, ube);
            }
            u[0] = c * ual + s * ube;  // Q^-1(., eps): u_dq frozen at the segment-start angle (line 501/511)
            u[1] = -s * ual + c * ube;
            const R deps = integrate<GEMX_SYS_SYNC, LOAD, SOLVER, R, NS1, LIN, decltype(seg_tag)::value>(P, y, u, h, linr, two, hc);
            ang = advance_angle<R, LIN, decltype(seg_tag)::value>(P, ang, deps, two);
        };
        if (IL) {
            segment(two ? P.t_il : P.tau, std::integral_constant<int, 1>{});
            if (two) {  // exec-masked; the whole wave skips it when no lane switches (s_cbranch_execz)
                Angle<R>::sincos(ang, s, c);  // eps / i_in refreshed at the switching instant (lines 504-505)
                segment(P.tau - P.t_il, std::integral_constant<int, 2>{});
            }
        } else {
            segment(P.tau, std::integral_constant<int, 0>{});
        }
        ho[0] = s; ho[1] = c; ho[2] = ua; ho[3] = ub; ho[4] = uc; ho[5] = u[0]; ho[6] = u[1];
    }
    static __device__ __forceinline__ void observe(const DevParams<R> &P, const R (&y)[3], AngT ang, const R (&ho)[NH], R (&obs)[14]) {
        // i_abc from the NEW i_dq with the angle of the last segment start (line 519, reference quirk)
        const R s = ho[0], c = ho[1];
        const R ial = c * y[1] - s * y[2], ibe = s * y[1] + c * y[2];
        R ia, ib, ic;
        t32(ial, ibe, ia, ib, ic);
        const R x[2] = {y[1], y[2]};
        obs[0] = y[0] * P.inv_lim[0];
        obs[1] = Elec<GEMX_SYS_SYNC, R>::torque(P, x) * P.inv_lim[1];
        obs[2] = ia * P.inv_lim[2];
        obs[3] = ib * P.inv_lim[3];
        obs[4] = ic * P.inv_lim[4];
        obs[5] = y[1] * P.inv_lim[5];
        obs[6] = y[2] * P.inv_lim[6];
        obs[7] = ho[2] * P.inv_lim[7];
        obs[8] = ho[3] * P.inv_lim[8];
        obs[9] = ho[4] * P.inv_lim[9];
        obs[10] = ho[5] * P.inv_lim[10];
        obs[11] = ho[6] * P.inv_lim[11];
        obs[12] = Angle<R>::wrapped(ang) * P.inv_lim[12];
        obs[13] = P.u_sup * P.inv_lim[13];
    }
    // default constraint: SquaredConstraint(('i_sq','i_sd')) (finite_cc_pmsm_env.py:106)
    static __device__ __forceinline__ bool default_done(const R (&obs)[14]) { return obs[5] * obs[5] + obs[6] * obs[6] > R(1); }
    static __device__ __forceinline__ R state_violation(const DevParams<R> &P, const R (&y)[3], const R (&)[NH]) {
        const R o5 = y[1] * P.inv_lim[5], o6 = y[2] * P.inv_lim[6];
        return o5 * o5 + o6 * o6;
    }
};

// ---- ExternallyExcitedSynchronousMotorSystem (physical_systems.py:619-652): SynchronousMotorSystem plus the
// excitation circuit (i_e, u_e) fed by the 4QC of the MultiConverter.  Single segment only: gemx_create refuses a
// dead time for this system (the reference's interlocking loop, lines 628-638, cannot execute). -------------------
template <int CONV, int LOAD, int SOLVER, bool IL, class R> struct Stepper<GEMX_SYS_EESM, CONV, LOAD, SOLVER, IL, R> {
    using AngT = typename Angle<R>::T;
    // ho: sin, cos of the step-start angle, u_a, u_b, u_c, u_e.  (u_sd, u_sq = Q^-1(T23(u_abc), eps) are NOT handed over: observe() computes them
    // again from the same values, and the pipelined kernel's hand-off row is 12 values at a 16-byte aligned stride: see Stepper<SCIM>.)
    static constexpr int NH = 6;
    static __device__ __forceinline__ uint32_t legs_of(uint32_t) { return 0u; }  // (no RC supply behind the finite EESM converter: gemx_create)
    static constexpr int B6 = CONV == GEMX_CONV_CONT_B6_4QC ? GEMX_CONV_CONT_B6 : GEMX_CONV_FINITE_B6;
    // converter output of a flat action index: u_a, u_b, u_c, u_e (the multi-converter has no dead time here)
    static __device__ __forceinline__ void voltages(const DevParams<R> &P, const R (&act)[MAX_ACT], uint32_t dact, R &ua, R &ub, R &uc, R &ue) {
        b6_voltages<B6, false, R>(P, act, dact & 7u, 0u, R(0), R(0), R(0), ua, ub, uc);
        if (CONV == GEMX_CONV_CONT_B6_4QC) {  // converters.py:481-491 with t_il = 0
            ue = (duty_pos(act[3]) - duty_neg(act[3])) * P.u_sup;
        } else {  // Finite-4QC sub-actions [1,1,2,2] / [1,2,1,2] (converters.py:360-361), flat action = a_b6 + 8 * a_4qc
            const uint32_t a1 = (dact >> 3) & 3u;
            ue = (((a1 & 2u) ? R(0) : R(1)) - ((a1 & 1u) ? R(0) : R(1))) * P.u_sup;
        }
    }
    // per-action voltage table of the pipelined kernel (32 entries: u_a, u_b, u_c, u_alpha, u_beta, u_e), as for Stepper<GEMX_SYS_SYNC>
    static constexpr int NVT = CONV == GEMX_CONV_FINITE_B6_4QC ? 6 : 0;
    static constexpr int row_slot(int j) { return j; }
    static __device__ __forceinline__ void action_entry(const DevParams<R> &P, uint32_t dact, R (&e)[8]) {
        const R zero[MAX_ACT] = {R(0), R(0), R(0), R(0), R(0), R(0)};
        voltages(P, zero, dact, e[0], e[1], e[2], e[5]);
        t23(e[0], e[1], e[2], e[3], e[4]);
        e[6] = e[7] = R(0);
    }
    template <bool NS1 = false, bool LIN = false, bool TAB = false>
    static __device__ __forceinline__ void advance(const DevParams<R> &P, R (&y)[4], AngT &ang, uint32_t &, const R (&act)[MAX_ACT],
                                                   uint32_t dact, R (&ho)[NH], const R *tab = nullptr, const R *linr = nullptr, R *hc = nullptr) {
        R s, c;
        Angle<R>::sincos(ang, s, c);
        R ua, ub, uc, ue, u[MAX_U] = {R(0), R(0), R(0), R(0)};
        R ual, ube;
        if (TAB) {
            ua = tab[0]; ub = tab[1]; uc = tab[2]; ual = tab[3]; ube = tab[4]; ue = tab[5];
        } else {
            voltages(P, act, dact, ua, ub, uc, ue);
            t23(ua, ub, uc, ual, ube);
        }
        u[0] = c * ual + s * ube;  // Q^-1(., eps) at the step-start angle (line 643)
        u[1] = -s * ual + c * ube;
        u[2] = ue;
        const R deps = integrate<GEMX_SYS_EESM, LOAD, SOLVER, R, NS1, LIN, IL ? 1 : 0>(P, y, u, P.tau, linr, false, hc);  // (see DcStepper: one segment, the tau map)
        ang = Angle<R>::advance(ang, deps);
        ho[0] = s; ho[1] = c; ho[2] = ua; ho[3] = ub; ho[4] = uc; ho[5] = ue;
    }
    static __device__ __forceinline__ void observe(const DevParams<R> &P, const R (&y)[4], AngT ang, const R (&ho)[NH], R (&obs)[16]) {
        const R s = ho[0], c = ho[1];  // i_abc from the NEW i_dq with the step-start angle (line 646)
        const R ial = c * y[1] - s * y[2], ibe = s * y[1] + c * y[2];
        R ia, ib, ic;
        t32(ial, ibe, ia, ib, ic);
        const R x[3] = {y[1], y[2], y[3]};
        obs[0] = y[0] * P.inv_lim[0];
        obs[1] = Elec<GEMX_SYS_EESM, R>::torque(P, x) * P.inv_lim[1];
        obs[2] = ia * P.inv_lim[2];
        obs[3] = ib * P.inv_lim[3];
        obs[4] = ic * P.inv_lim[4];
        obs[5] = y[1] * P.inv_lim[5];
        obs[6] = y[2] * P.inv_lim[6];
        obs[7] = y[3] * P.inv_lim[7];
        obs[8] = ho[2] * P.inv_lim[8];
        obs[9] = ho[3] * P.inv_lim[9];
        obs[10] = ho[4] * P.inv_lim[10];
        R ual, ube;
        t23(ho[2], ho[3], ho[4], ual, ube);  // (what advance() integrated with: the same expressions on the same values)
        obs[11] = (c * ual + s * ube) * P.inv_lim[11];
        obs[12] = (-s * ual + c * ube) * P.inv_lim[12];
        obs[13] = ho[5] * P.inv_lim[13];
        obs[14] = Angle<R>::wrapped(ang) * P.inv_lim[14];
        obs[15] = P.u_sup * P.inv_lim[15];
    }
    // default constraints: SquaredConstraint(('i_sq','i_sd')) + LimitConstraint(('i_e',)) (cont_cc_eesm_env.py:108)
    static __device__ __forceinline__ bool default_done(const R (&obs)[16]) {
        return (obs[5] * obs[5] + obs[6] * obs[6] > R(1)) | (fabs(obs[7]) > R(1));
    }
    static __device__ __forceinline__ R state_violation(const DevParams<R> &P, const R (&y)[4], const R (&)[NH]) {
        const R o5 = y[1] * P.inv_lim[5], o6 = y[2] * P.inv_lim[6], o7 = y[3] * P.inv_lim[7];
        return fmax(o5 * o5 + o6 * o6, fabs(o7));
    }
};

// ---- SquirrelCageInductionMotorSystem (physical_systems.py:771-814), control_space 'abc' -----------------------
template <int CONV, int LOAD, int SOLVER, bool IL, class R> struct Stepper<GEMX_SYS_SCIM, CONV, LOAD, SOLVER, IL, R> {
    using AngT = typename Angle<R>::T;
    // ho: sin, cos of the last segment-start field angle, u_a, u_b, u_c.  (u_alpha, u_beta = T23(u_abc) are NOT handed over: observe() computes
    // them again -- the same function of the same three values -- and the pipelined kernel's hand-off row is 12 values at a 48-byte stride,
    // 16-byte aligned like the synchronous machines', instead of 14 at 56 bytes that left as eight LDS instructions per step.)
    static constexpr int NH = 5;
    static __device__ __forceinline__ uint32_t legs_of(uint32_t dact) { return CONV == GEMX_CONV_FINITE_B6 ? b6_subactions(dact) : 0u; }
    // per-action voltage table of the pipelined kernel, as for the synchronous machines (Stepper<GEMX_SYS_SYNC>::action_entry)
    static constexpr int NVT = (CONV == GEMX_CONV_FINITE_B6 && !IL) ? 5 : 0;
    static constexpr int row_slot(int j) { return j; }
    static __device__ __forceinline__ void action_entry(const DevParams<R> &P, uint32_t dact, R (&e)[8]) {
        const R zero[MAX_ACT] = {R(0), R(0), R(0), R(0), R(0), R(0)};
        b6_voltages<CONV, false, R>(P, zero, dact, 0u, R(0), R(0), R(0), e[0], e[1], e[2]);
        t23(e[0], e[1], e[2], e[3], e[4]);
        e[5] = e[6] = e[7] = R(0);
    }
    static __device__ __forceinline__ void field_angle(R pa, R pb, R &s, R &c) { flux_angle<R>(pa, pb, s, c); }
    template <bool NS1 = false, bool LIN = false, bool TAB = false>
    static __device__ __forceinline__ void advance(const DevParams<R> &P, R (&y)[5], AngT &ang, uint32_t &sw, const R (&act)[MAX_ACT],
                                                   uint32_t dact, R (&ho)[NH], const R *tab = nullptr, const R *linr = nullptr, R *hc = nullptr) {
        R s, c;
        field_angle(y[3], y[4], s, c);
        uint32_t legs = 0;
        bool two = false;
        if (IL && CONV == GEMX_CONV_FINITE_B6) {
            legs = b6_subactions_packed(dact);
            if (P.t_il > R(0)) legs = b6_interlock(sw, legs, two);
            sw = legs;
        }
        R ua, ub, uc, u[MAX_U] = {R(0), R(0), R(0), R(0)};
        auto segment = [&](R h, auto seg_tag) {
            R ia = R(0), ib = R(0), ic = R(0);
            if (IL) t32(y[1], y[2], ia, ib, ic);  // i_in = T32(i_alphabeta) (line 780/792)
            if (TAB) {  // this action's table entry
                ua = tab[0]; ub = tab[1]; uc = tab[2]; u[0] = tab[3]; u[1] = tab[4];
            } else {
                b6_voltages<CONV, IL, R>(P, act, dact, legs, ia, ib, ic, ua, ub, uc);
                t23(ua, ub, uc, u[0], u[1]);  // u_alphabeta constant over the segment (line 788/799)
            }
            const R deps = integrate<GEMX_SYS_SCIM, LOAD, SOLVER, R, NS1, LIN, decltype(seg_tag)::value>(P, y, u, h, linr, two, hc);
            ang = advance_angle<R, LIN, decltype(seg_tag)::value>(P, ang, deps, two);
        };
        if (IL) {
            segment(two ? P.t_il : P.tau, std::integral_constant<int, 1>{});
            if (two) {
                field_angle(y[3], y[4], s, c);  // line 791
                segment(P.tau - P.t_il, std::integral_constant<int, 2>{});
            }
        } else {
            segment(P.tau, std::integral_constant<int, 0>{});
        }
        ho[0] = s; ho[1] = c; ho[2] = ua; ho[3] = ub; ho[4] = uc;
    }
    static __device__ __forceinline__ void observe(const DevParams<R> &P, const R (&y)[5], AngT ang, const R (&ho)[NH], R (&obs)[14]) {
        // i_dq = Q^-1(i_alphabeta_new, eps_fs of the last segment start) (line 806, reference quirk);
        // i_abc = T32(Q(i_dq, eps_fs)) == T32(i_alphabeta_new) (line 807); u_dq = Q^-1(u_alphabeta, eps_fs) (798)
        const R s = ho[0], c = ho[1];
        R ia, ib, ic;
        t32(y[1], y[2], ia, ib, ic);
        const R x[4] = {y[1], y[2], y[3], y[4]};
        obs[0] = y[0] * P.inv_lim[0];
        obs[1] = Elec<GEMX_SYS_SCIM, R>::torque(P, x) * P.inv_lim[1];
        obs[2] = ia * P.inv_lim[2];
        obs[3] = ib * P.inv_lim[3];
        obs[4] = ic * P.inv_lim[4];
        obs[5] = (c * y[1] + s * y[2]) * P.inv_lim[5];
        obs[6] = (-s * y[1] + c * y[2]) * P.inv_lim[6];
        obs[7] = ho[2] * P.inv_lim[7];
        obs[8] = ho[3] * P.inv_lim[8];
        obs[9] = ho[4] * P.inv_lim[9];
        R ual, ube;
        t23(ho[2], ho[3], ho[4], ual, ube);  // (what advance() integrated with: T23 of the same u_abc)
        obs[10] = (c * ual + s * ube) * P.inv_lim[10];
        obs[11] = (-s * ual + c * ube) * P.inv_lim[11];
        obs[12] = Angle<R>::wrapped(ang) * P.inv_lim[12];
        obs[13] = P.u_sup * P.inv_lim[13];
    }
    // default constraint: SquaredConstraint(('i_sq','i_sd')) (cont_sc_scim_env.py:111)
    static __device__ __forceinline__ bool default_done(const R (&obs)[14]) { return obs[5] * obs[5] + obs[6] * obs[6] > R(1); }
    static __device__ __forceinline__ R state_violation(const DevParams<R> &P, const R (&y)[5], const R (&ho)[NH]) {
        const R s = ho[0], c = ho[1];
        const R o5 = (c * y[1] + s * y[2]) * P.inv_lim[5], o6 = (-s * y[1] + c * y[2]) * P.inv_lim[6];
        return o5 * o5 + o6 * o6;
    }
};

// ---- DoublyFedInductionMotorSystem (physical_systems.py:948-1029): stator AND rotor behind a B6 bridge each ---------------
template <int CONV, int LOAD, int SOLVER, bool IL, class R> struct Stepper<GEMX_SYS_DFIM, CONV, LOAD, SOLVER, IL, R> {
    using AngT = typename Angle<R>::T;
    using SC = Stepper<GEMX_SYS_SCIM, GEMX_CONV_CONT_B6, LOAD, SOLVER, IL, R>;  // field_angle()
    // ho: sin, cos of the last segment-start field angle; sin, cos of the last segment-start electrical angle;
    //     u_sa, u_sb, u_sc; u_rd, u_re, u_rf (rotor-fixed three-phase frame)
    static constexpr int NH = 10;
    static __device__ __forceinline__ uint32_t legs_of(uint32_t dact) {
        return CONV == GEMX_CONV_FINITE_2XB6 ? (b6_subactions(dact & 7u) | (b6_subactions((dact >> 3) & 7u) << 6)) : 0u;
    }
    static constexpr int NVT = 0;
    static constexpr int row_slot(int j) { return j; }
    static constexpr int B6 = CONV == GEMX_CONV_CONT_2XB6 ? GEMX_CONV_CONT_B6 : GEMX_CONV_FINITE_B6;
    template <bool NS1 = false, bool LIN = false, bool TAB = false>
    static __device__ __forceinline__ void advance(const DevParams<R> &P, R (&y)[5], AngT &ang, uint32_t &sw, const R (&act)[MAX_ACT],
                                                   uint32_t dact, R (&ho)[NH], const R * = nullptr, const R *linr = nullptr, R *hc = nullptr) {
        R sf, cf, se, ce;
        SC::field_angle(y[3], y[4], sf, cf);
        Angle<R>::sincos_precise(ang, se, ce);
        uint32_t legs = 0;
        bool two = false;
        if (IL && CONV == GEMX_CONV_FINITE_2XB6) {  // flat action = a_stator + 8 * a_rotor; 6 half-bridges, 2 bits each
            legs = b6_subactions_packed(dact & 7u) | (b6_subactions_packed((dact >> 3) & 7u) << 6);
            if (P.t_il > R(0)) legs = b6_interlock<6>(sw, legs, two);
            sw = legs;
        }
        R usa, usb, usc, urd, ure, urf, u[MAX_U];
        auto segment = [&](R h, auto seg_tag) {
            R isa = R(0), isb = R(0), isc = R(0), ird = R(0), ire = R(0), irf = R(0);
            if (IL) {  // i_sabc = T32(i_s alphabeta); i_rdef = T32(calculate_rotor_current(state)) (lines 960-963, 980-981)
                t32(y[1], y[2], isa, isb, isc);
                t32(P.tc2 * y[3] - P.tc3 * y[1], P.tc2 * y[4] - P.tc3 * y[2], ird, ire, irf);
            }
            b6_voltages<B6, IL, R, 0>(P, act, dact & 7u, legs & 63u, isa, isb, isc, usa, usb, usc);
            b6_voltages<B6, IL, R, 3>(P, act, (dact >> 3) & 7u, (legs >> 6) & 63u, ird, ire, irf, urd, ure, urf);
            t23(usa, usb, usc, u[0], u[1]);
            // u_r alphabeta = Q(eps_field) Q^-1(T23 u_rdef, eps_field - eps_el) (lines 972-974) == Q(T23 u_rdef, eps_el)
            R urg, urh;
            t23(urd, ure, urf, urg, urh);
            u[2] = ce * urg - se * urh;
            u[3] = se * urg + ce * urh;
            const R deps = integrate<GEMX_SYS_DFIM, LOAD, SOLVER, R, NS1, LIN, decltype(seg_tag)::value>(P, y, u, h, linr, two, hc);
            ang = advance_angle<R, LIN, decltype(seg_tag)::value>(P, ang, deps, two);
        };
        if (IL) {
            segment(two ? P.t_il : P.tau, std::integral_constant<int, 1>{});
            if (two) {
                SC::field_angle(y[3], y[4], sf, cf);  // lines 978-979
                Angle<R>::sincos_precise(ang, se, ce);
                segment(P.tau - P.t_il, std::integral_constant<int, 2>{});
            }
        } else {
            segment(P.tau, std::integral_constant<int, 0>{});
        }
        ho[0] = sf; ho[1] = cf; ho[2] = se; ho[3] = ce; ho[4] = usa; ho[5] = usb; ho[6] = usc; ho[7] = urd; ho[8] = ure; ho[9] = urf;
    }
    static __device__ __forceinline__ void observe(const DevParams<R> &P, const R (&y)[5], AngT ang, const R (&ho)[NH], R (&obs)[24]) {
        const R sf = ho[0], cf = ho[1], se = ho[2], ce = ho[3];
        const R sd = sf * ce - cf * se, cd = cf * ce + sf * se;  // sin / cos of (eps_field - eps_el)
        // stator: i_sdq = Q^-1(i_s alphabeta NEW, eps_field of the last segment start) (line 1003); i_sabc = T32(i_s alphabeta) (1004)
        R isa, isb, isc;
        t32(y[1], y[2], isa, isb, isc);
        // rotor: i_rdq = Q^-1(i_r alphabeta NEW, eps_field) (1005); i_rdef = T32(Q(i_rdq, eps_field - eps_el)) (1006)
        const R ira = P.tc2 * y[3] - P.tc3 * y[1], irb = P.tc2 * y[4] - P.tc3 * y[2];
        const R ird_ = cf * ira + sf * irb, irq_ = -sf * ira + cf * irb;
        R ird, ire, irf;
        t32(cd * ird_ - sd * irq_, sd * ird_ + cd * irq_, ird, ire, irf);
        R usal, usbe, urg, urh;
        t23(ho[4], ho[5], ho[6], usal, usbe);
        t23(ho[7], ho[8], ho[9], urg, urh);
        const R x[4] = {y[1], y[2], y[3], y[4]};
        obs[0] = y[0] * P.inv_lim[0];
        obs[1] = Elec<GEMX_SYS_DFIM, R>::torque(P, x) * P.inv_lim[1];
        obs[2] = isa * P.inv_lim[2];
        obs[3] = isb * P.inv_lim[3];
        obs[4] = isc * P.inv_lim[4];
        obs[5] = (cf * y[1] + sf * y[2]) * P.inv_lim[5];
        obs[6] = (-sf * y[1] + cf * y[2]) * P.inv_lim[6];
        obs[7] = ird * P.inv_lim[7];
        obs[8] = ire * P.inv_lim[8];
        obs[9] = irf * P.inv_lim[9];
        obs[10] = ird_ * P.inv_lim[10];
        obs[11] = irq_ * P.inv_lim[11];
        obs[12] = ho[4] * P.inv_lim[12];
        obs[13] = ho[5] * P.inv_lim[13];
        obs[14] = ho[6] * P.inv_lim[14];
        obs[15] = (cf * usal + sf * usbe) * P.inv_lim[15];   // u_sdq = Q^-1(T23 u_sabc, eps_field) (line 990)
        obs[16] = (-sf * usal + cf * usbe) * P.inv_lim[16];
        obs[17] = ho[7] * P.inv_lim[17];
        obs[18] = ho[8] * P.inv_lim[18];
        obs[19] = ho[9] * P.inv_lim[19];
        obs[20] = (cd * urg + sd * urh) * P.inv_lim[20];     // u_rdq = Q^-1(T23 u_rdef, eps_field - eps_el) (line 991)
        obs[21] = (-sd * urg + cd * urh) * P.inv_lim[21];
        obs[22] = Angle<R>::wrapped(ang) * P.inv_lim[22];
        obs[23] = P.u_sup * P.inv_lim[23];
    }
    // default constraint: SquaredConstraint(('i_sq','i_sd')) (cont_cc_dfim_env.py:115)
    static __device__ __forceinline__ bool default_done(const R (&obs)[24]) { return obs[5] * obs[5] + obs[6] * obs[6] > R(1); }
    static __device__ __forceinline__ R state_violation(const DevParams<R> &P, const R (&y)[5], const R (&ho)[NH]) {
        const R s = ho[0], c = ho[1];
        const R o5 = (c * y[1] + s * y[2]) * P.inv_lim[5], o6 = (-s * y[1] + c * y[2]) * P.inv_lim[6];
        return o5 * o5 + o6 * o6;
    }
};

// ------------------------------------------------------------------------------------------------
// RCVoltageSupply: i_sup = converter.i_sup(i_in) as *.simulate() evaluates it at the start of a step -- after set_action(), before
// convert() -- i.e. with the NEW duty cycles of a continuous converter (Cont-2QC: converters.py:429-435) but the switching state the
// PREVIOUS convert() left behind in a finite one (Finite-2QC: 289-298); 4QC = leg(i) + leg(-i) (366-368, 493-495), B6 = sum over the
// legs (837-839, 909-911), MultiConverter = sum over the sub-converters (572-580).  `act` is the converter-side action, `sw` the stored
// leg states (2 bits per half-bridge).
// ------------------------------------------------------------------------------------------------
template <class R> __device__ __forceinline__ R cont_leg_i_sup(const DevParams<R> &P, R duty, R i) {
    const R ic = i < R(0) ? R(1) : R(0);
    return (duty + P.il_ratio * (ic - duty)) * i;
}
template <class R> __device__ __forceinline__ R fin_leg_i_sup(uint32_t st, R i) {
    return st == 1u ? i : ((st == 0u && i < R(0)) ? i : R(0));
}
template <bool FIN, class R> __device__ __forceinline__ R qc4_i_sup(const DevParams<R> &P, R a, uint32_t legs, R i) {
    if (FIN) return fin_leg_i_sup<R>(legs & 3u, i) + fin_leg_i_sup<R>((legs >> 2) & 3u, -i);
    return cont_leg_i_sup<R>(P, duty_pos(a), i) + cont_leg_i_sup<R>(P, duty_neg(a), -i);
}
template <bool FIN, class R> __device__ __forceinline__ R b6_i_sup(const DevParams<R> &P, R a0, R a1, R a2, uint32_t legs, R ia, R ib, R ic) {
    if (FIN) return fin_leg_i_sup<R>(legs & 3u, ia) + fin_leg_i_sup<R>((legs >> 2) & 3u, ib) + fin_leg_i_sup<R>((legs >> 4) & 3u, ic);
    return cont_leg_i_sup<R>(P, duty_pos(a0), ia) + cont_leg_i_sup<R>(P, duty_pos(a1), ib) +
           cont_leg_i_sup<R>(P, duty_pos(a2), ic);
}
template <int SYS, int CONV, class R>
__device__ __forceinline__ R supply_current(const DevParams<R> &P, const R (&y)[SysTraits<SYS>::ND], typename Angle<R>::T ang, uint32_t sw,
                                            const R (&act)[MAX_ACT]) {
    constexpr bool FIN = ConvTraits<CONV>::DISCRETE != 0;
    if (SYS == GEMX_SYS_DC_PERMEX || SYS == GEMX_SYS_DC_SERIES) return qc4_i_sup<FIN, R>(P, act[0], sw, y[1]);
    if (SYS == GEMX_SYS_DC_SHUNT) return qc4_i_sup<FIN, R>(P, act[0], sw, y[1] + y[SysTraits<SYS>::ND - 1]);
    if (SYS == GEMX_SYS_DC_EXTEX) return qc4_i_sup<FIN, R>(P, act[0], sw, y[1]) + qc4_i_sup<FIN, R>(P, act[1], sw >> 4, y[SysTraits<SYS>::ND - 1]);
    R ia, ib, ic;
    if (SYS == GEMX_SYS_SYNC || SYS == GEMX_SYS_EESM) {
        R s, c;
        Angle<R>::sincos(ang, s, c);
        t32(c * y[1] - s * y[2], s * y[1] + c * y[2], ia, ib, ic);
    } else {
        t32(y[1], y[2], ia, ib, ic);
    }
    R tot = b6_i_sup<FIN, R>(P, act[0], act[1], act[2], sw, ia, ib, ic);
    if (SYS == GEMX_SYS_EESM) tot += qc4_i_sup<false, R>(P, act[3], 0u, y[SysTraits<SYS>::ND - 1]);  // continuous converter only (gemx_create)
    if (SYS == GEMX_SYS_DFIM) {  // rotor bridge: i_rdef = T32(calculate_rotor_current(state))
        R ra, rb, rc;
        t32(P.tc2 * y[3] - P.tc3 * y[1], P.tc2 * y[SysTraits<SYS>::ND - 1] - P.tc3 * y[2], ra, rb, rc);
        tot += b6_i_sup<FIN, R>(P, act[3], act[4], act[5], sw >> 6, ra, rb, rc);
    }
    return tot;
}

#ifndef GEMX_DRAW_NS  // 0: the number of states read from the description at run time, as rounds 4-5 did (A/B builds)
#define GEMX_DRAW_NS 1
#endif
// in-kernel auto-reset with random initialisers: a fresh initial state for this env, its reset counter advanced (rare path).
// (Round 4 tried this as a real call, `noinline`: advance_kernel's spills went from 41-55 to 2-27 registers -- not to zero, the 256-VGPR
// cap is the blocked I/O's prefetch and flush registers, not this path -- while step_kernel, which calls it too, went from no scratch
// at all to 247 VGPRs and 112 bytes of stack for the callee.  Kept inline.)
template <int SYS, class R>
__device__ __forceinline__ void draw_initial_state(const KArgs<R> &a, int64_t env, R (&y)[SysTraits<SYS>::ND], typename Angle<R>::T &ang) {
    constexpr int ND = SysTraits<SYS>::ND;
    const uint32_t count = a.rcnt[env] + 1u;
    a.rcnt[env] = count;
    double v[GEMX_MAX_ODE];
    init_draw_all<SYS == GEMX_SYS_SCIM || SYS == GEMX_SYS_DFIM, GEMX_DRAW_NS * (SysTraits<SYS>::ND + (SysTraits<SYS>::HAS_ANGLE ? 1 : 0))>(a.rinit, env, count, v);
#pragma unroll
    for (int j = 0; j < ND; ++j) y[j] = (R)v[j];
    if (SysTraits<SYS>::HAS_ANGLE) ang = Angle<R>::from_rad(v[ND]);
}

// the same draw with the env's reset counter held in a register (pipelined kernel: no global memory traffic besides the description).
// A REAL CALL since round 6 (the pipelined kernel's integrator only; the single-wave kernels keep the inline form, see above): inlined into
// every unrolled copy of the step, the draw -- generator blocks, the induction machines' flux bounds, the truncated normal's erfc and inverse
// CDF in fp64 -- was 13 600 of the SCIM kernel's 38 000 instructions, and a lane that outran its queue of prepared draws (2-7 % of the SCIM's
// resets) sent its wave through ~1500 instructions that are never in the instruction cache: ~12 000 cycles per event, 7000 of the
// integrator's 10 000 cycles per block (profiles/r06_rinit_probe.md).  One copy per kernel, the result in registers.
template <int SYS, class R> struct DrawnState {
    R y[SysTraits<SYS>::ND];
    typename Angle<R>::T ang;
};
#ifndef GEMX_DRAW_CALL  // 0: the inline form of rounds 4-5 (A/B builds)
#define GEMX_DRAW_CALL 1
#endif
#if GEMX_DRAW_CALL
#define GEMX_DRAW_CALL_ATTR __attribute__((noinline))
#else
#define GEMX_DRAW_CALL_ATTR __forceinline__
#endif
template <int SYS, class R>
__device__ GEMX_DRAW_CALL_ATTR DrawnState<SYS, R> draw_initial_state_call(const InitDev *rinit, int64_t env, uint32_t count) {
    constexpr int ND = SysTraits<SYS>::ND;
    DrawnState<SYS, R> o;
    double v[GEMX_MAX_ODE];
    init_draw_all<SYS == GEMX_SYS_SCIM || SYS == GEMX_SYS_DFIM, GEMX_DRAW_NS * (SysTraits<SYS>::ND + (SysTraits<SYS>::HAS_ANGLE ? 1 : 0))>(rinit, env, count, v);
#pragma unroll
    for (int j = 0; j < ND; ++j) o.y[j] = (R)v[j];
    o.ang = typename Angle<R>::T(0);
    if (SysTraits<SYS>::HAS_ANGLE) o.ang = Angle<R>::from_rad(v[ND]);
    return o;
}
template <int SYS, class R>
__device__ __forceinline__ void draw_initial_state_cnt(const InitDev *rinit, int64_t env, uint32_t &count, R (&y)[SysTraits<SYS>::ND],
                                                       typename Angle<R>::T &ang) {
    constexpr int ND = SysTraits<SYS>::ND;
    count += 1u;
    const DrawnState<SYS, R> o = draw_initial_state_call<SYS, R>(rinit, env, count);
#pragma unroll
    for (int j = 0; j < ND; ++j) y[j] = o.y[j];
    if (SysTraits<SYS>::HAS_ANGLE) ang = o.ang;
}

// step() for the single-wave kernel
template <class ST, int ND, int NOUT, class R, bool LIN = false>
__device__ __forceinline__ void full_step(const DevParams<R> &P, R (&y)[ND], typename Angle<R>::T &ang, uint32_t &sw, const R (&act)[MAX_ACT],
                                          uint32_t dact, R (&obs)[NOUT], const R *linr = nullptr, R *hc = nullptr) {
    R ho[ST::NH];
    ST::template advance<false, LIN>(P, y, ang, sw, act, dact, ho, nullptr, linr, hc);
    ST::observe(P, y, ang, ho, obs);
}
// instantiations whose electrical subsystem can be stepped by the precomputed one-step map (see integrate<..., LIN>).  Euler: only the
// ONE-state machines -- there the map is one multiply (the input term, off the recurrence) and one FMA against Euler's two dependent
// FMAs; with two states it is 8 FMAs against 6
template <int SYS, int LOAD, int SOLVER, bool IL, class R> constexpr bool linable() {
    return LOAD == GEMX_LOAD_CONST_SPEED && (SOLVER != GEMX_SOLVER_EULER || Elec<SYS, R>::NM == 1) && sizeof(R) == 4;
}
// the map is valid for a wave if every lane's omega equals init[0] (then it stays so: a ConstantSpeedLoad never changes omega, and a
// reset puts init[0] back); omega set to something else through gemx_set_state falls back to the stage-by-stage solver
template <int SYS, int LOAD, int SOLVER, bool IL, class R> __device__ __forceinline__ bool lin_usable(const DevParams<R> &P, R omega) {
    if (!linable<SYS, LOAD, SOLVER, IL, R>()) return false;
    return P.lin_on && __all(omega == P.init[0]);
}
// the map's NM * (NM + NG) coefficients into registers, once per kernel (see integrate<..., LIN>)
template <int SYS, class R> constexpr int lin_count() { return Elec<SYS, R>::NM * (Elec<SYS, R>::NM + Elec<SYS, R>::NG); }
// registers of a kernel instantiation's preloaded coefficients: one map (tau), or the three maps of the dead-time (IL) instantiations
// (t_il, tau - t_il, tau: maps 1..3 of linmap_kernel)
template <int SYS, class R, bool IL> constexpr int lin_regs() { return lin_count<SYS, R>() * (IL ? 3 : 1); }
template <int SYS, class R, int NR> __device__ __forceinline__ void lin_preload(const DevParams<R> &P, bool lin_ok, R (&c)[NR]) {
    constexpr int OFF = NR > lin_count<SYS, R>() ? lin_count<SYS, R>() : 0;  // the dead-time instantiations take maps 1..3 (linmap_kernel)
#pragma unroll
    for (int i = 0; i < NR; ++i) c[i] = lin_ok ? P.lin[OFF + i] : R(0);
}
// builds the map for one handle: Phi's columns are rk_step(e_j) with g = 0, S's columns rk_step(0) with g = e_i
template <int SYS, int SOLVER, class R> __global__ void linmap_kernel(DevParams<double> P, R *out) {
    // Evaluated in DOUBLE whatever R is: the maps are Phi = I + D with D = O(h A), and what a step needs of them is D.  Formed in fp32,
    // 1 + D carries D to an absolute 6e-8 only -- a relative 6e-8 / (h A) error of every decay rate, SYSTEMATIC (the same coefficient
    // every step): 1e-3 for the 1 us dead-time segment of a PMSM, which turned currents near zero to the wrong sign and with them a
    // freewheeling leg's voltage (r03m: u_a off by 2.0 normalised with nsteps = 8).  So: rk_step in double, and
    //   every map holds D = Phi - I and S, and the step is x + D x + S g -- except map 0 (whole step tau, no dead time) of the DC machines,
    //   which keeps Phi: their h A is 0.01 ... 0.1 (relative error of the decay <= 6e-6) and dc_stream_kernel's split of the step into a
    //   state-independent input term and ONE fused multiply-add on the recurrence needs that form.  The three-phase machines' whole-step
    //   map in the Phi form cost a factor 5-10 of accuracy against the stage-by-stage solver on every recorded run (DFIM 9.8e-5 against
    //   9.6e-6, EESM 6.4e-5 / 5.3e-6, PMSM at tau = 1e-5 8.1e-6 / 1.1e-6: tools/probe_map_bias.py, profiles/r03p_map_bias.md).
    //   maps: 0 tau | 1 t_il | 2 tau - t_il | 3 tau (1..3: the dead-time instantiations' registers)
    using E = Elec<SYS, double>;
    constexpr int NM = E::NM, NG = E::NG, NC = NM * (NM + NG);
    constexpr bool PHI_FORM0 = lin_phi_form<SYS, 0>();
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double u0[MAX_U] = {0.0, 0.0, 0.0, 0.0};
    double zg[NG];
    for (int i = 0; i < NG; ++i) zg[i] = 0.0;
    const typename E::Pre pre0 = E::set_b(E::prep(P, P.init[0], u0), zg);
    const double hseg[4] = {P.tau, P.t_il, P.tau - P.t_il, P.tau};
    for (int k = 0; k < 4; ++k) {
        const double hs = hseg[k] * P.inv_ns;
        R *o = out + k * NC;
        for (int j = 0; j < NM; ++j) {
            double x[NM];
            for (int i = 0; i < NM; ++i) x[i] = i == j ? 1.0 : 0.0;
            auto rhs = [&](const double (&xx)[NM], double (&dx)[NM]) { E::f(P, pre0, xx, dx); };
            rk_step<SOLVER, NM, double>(x, hs, rhs);
            for (int r = 0; r < NM; ++r) o[r * NM + j] = (R)((k == 0 && PHI_FORM0) ? x[r] : x[r] - (r == j ? 1.0 : 0.0));
        }
        for (int i = 0; i < NG; ++i) {
            double gi[NG], x[NM];
            for (int q = 0; q < NG; ++q) gi[q] = q == i ? 1.0 : 0.0;
            for (int q = 0; q < NM; ++q) x[q] = 0.0;
            const typename E::Pre prei = E::set_b(pre0, gi);
            auto rhs = [&](const double (&xx)[NM], double (&dx)[NM]) { E::f(P, prei, xx, dx); };
            rk_step<SOLVER, NM, double>(x, hs, rhs);
            for (int r = 0; r < NM; ++r) o[NM * NM + r * NG + i] = (R)x[r];
        }
    }
}

// ConstraintMonitor with merge 'max' over LimitConstraint / SquaredConstraint; terminated = violation >= 1
// (core.py:350, 834-844; constraints.py:55-58, 96-98).  constr_kind is wave-uniform: 0 none, 1 the env's default
// constraint (3 VALU ops), 2 arbitrary masks as 0/1 weights (branch-free).
template <class ST, int NOUT, class R> __device__ __forceinline__ bool constraint_done(const DevParams<R> &P, const R (&obs)[NOUT]) {
    if (P.constr_kind == 0) return false;
    if (P.constr_kind == 1) return ST::default_done(obs);
    R lim = R(0), sq = R(0);
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
        lim = fmax(lim, P.cw[i] * fabs(obs[i]));
        sq += (P.cw[GEMX_MAX_OUT + i] * obs[i]) * obs[i];
    }
    return (lim > R(1)) | (sq > R(1));
}

extern __shared__ __attribute__((aligned(16))) unsigned char gemx_smem[];

// Flush `sb` observation rows / done rows (control steps k0 .. k0+sb-1) of one EB-env workgroup from the LDS rings to
// the caller's tensors: 16-byte-per-lane stores over each row's contiguous span (AoS) or coalesced dword stores (SoA).
// EB = envs per workgroup (ring row length): BLOCK for the single-wave kernel, 16 / 32 / 64 for the pipelined one; `tid` is the lane.
template <int NOUT, class R, int EB = BLOCK>
__device__ __forceinline__ void flush_rings(const KArgs<R> &a, const R *ring, const unsigned char *donebuf, int k0, int sb, int tid,
                                            int64_t blk0, int rows, bool full, bool valid, int64_t env) {
    constexpr int VEC = 16 / sizeof(R);
    constexpr int ROWV = EB * NOUT / VEC;
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    using V = typename std::conditional<sizeof(R) == 4, v4f_t, v2d_t>::type;
    const int64_t N = a.N;
    const bool stream_out = a.K > 1;  // rollouts: non-temporal stores (see flush_rows_pipe); a single step's row is read next by the policy
    const int el = tid < EB ? tid : EB - 1;  // this lane's env slot (lanes beyond EB duplicate the last env of the workgroup)
    if (a.P.obs_layout == GEMX_OBS_AOS) {
        if (a.obs_vec) {
            const int nvec = rows * NOUT / VEC;  // == ROWV for full blocks
#pragma unroll 2
            for (int s = 0; s < sb; ++s) {
                V *gv = reinterpret_cast<V *>(a.obs + ((int64_t)(k0 + s) * N + blk0) * NOUT);
                const V *lv = reinterpret_cast<const V *>(ring + (size_t)s * EB * NOUT);
#pragma unroll
                for (int i = 0; i < (ROWV + BLOCK - 1) / BLOCK; ++i) {
                    const int idx = tid + i * BLOCK;
                    if (idx < nvec) {
                        const V v = lv[idx];
                        if (stream_out) __builtin_nontemporal_store(v, &gv[idx]);
                        else gv[idx] = v;
                    }
                }
                for (int idx = nvec * VEC + tid; idx < rows * NOUT; idx += BLOCK)
                    a.obs[((int64_t)(k0 + s) * N + blk0) * NOUT + idx] = ring[(size_t)s * EB * NOUT + idx];
            }
        } else {
            for (int s = 0; s < sb; ++s)
                for (int idx = tid; idx < rows * NOUT; idx += BLOCK)
                    a.obs[((int64_t)(k0 + s) * N + blk0) * NOUT + idx] = ring[(size_t)s * EB * NOUT + idx];
        }
    } else if (valid) {
        for (int s = 0; s < sb; ++s) {
#pragma unroll
            for (int j = 0; j < NOUT; ++j) a.obs[((int64_t)(k0 + s) * NOUT + j) * N + env] = ring[(s * NOUT + j) * EB + el];
        }
    }
    if (a.done != nullptr) {
        if (a.coop && full) {  // done rows are EB contiguous bytes: EB / 16 chunks of 16 bytes per row
            constexpr int CPD = EB / 16;
            const int nchunk = sb * CPD;
            for (int idx = tid; idx < nchunk; idx += BLOCK) {
                const int row = idx / CPD, col = idx - row * CPD;
                *reinterpret_cast<uint4 *>(a.done + (int64_t)(k0 + row) * N + blk0 + col * 16) =
                    *reinterpret_cast<const uint4 *>(donebuf + row * EB + col * 16);
            }
        } else if (valid) {
            for (int s = 0; s < sb; ++s) a.done[(int64_t)(k0 + s) * N + env] = donebuf[s * EB + el];
        }
    }
}

// The pipelined kernel's output waves, full blocks: the same RPW rows (AoS, 16-byte aligned, full 64-env workgroup) with ALL their LDS
// reads issued before the first store and wave-uniform row addresses (k0 must be uniform).  flush_rings' per-row chain -- 3 reads,
// wait, 3 stores, a predicated 4th read, wait, store -- exposes the LDS latency twice per row: ~380 cycles per row on an otherwise idle
// chip, where the stores themselves issue at 23 cycles each (tools/microbench_store.hip).  Lanes past the end of a row's last,
// partial chunk repeat the row's final chunk (same address, same data) instead of branching around the store.
template <int NOUT, int RPW, class R, int EB = BLOCK>
__device__ __forceinline__ void flush_rows_pipe(const KArgs<R> &a, const R *ring, const unsigned char *donebuf, int k0, int tid, int64_t blk0) {
    constexpr int VEC = 16 / sizeof(R);
    constexpr int ROWV = EB * NOUT / VEC;               // 16-byte chunks per row (EB envs)
    constexpr int NV = (ROWV + BLOCK - 1) / BLOCK;      // chunks per lane
    static_assert((EB * NOUT) % VEC == 0, "rows are whole 16-byte chunks");
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    using V = typename std::conditional<sizeof(R) == 4, v4f_t, v2d_t>::type;
    // NON-TEMPORAL stores: the observation stream is written once and is far larger than L2 / MALL (0.9 GB per 1000-step launch of the
    // headline); without the allocation in the caches the same kernels run 3-7 % faster at every size (same-box A/B: 79.0 -> 83.0 G
    // env-steps/s at 16384 envs, 88.6 -> 94.4 G at 131072, 104.4 -> 107.5 G at 1M).  The single-step path (K = 1, advance_kernel) keeps
    // ordinary stores: its one row per env is what a policy kernel reads next.
#define GEMX_ROW_STORE(ptr, val) __builtin_nontemporal_store(val, ptr)
    const int64_t N = a.N;
    // (named variables, not arrays: the optimiser puts a [RPW][NV] array of 16-byte values into scratch memory, and the scratch
    // loads' vmcnt(0) then waits for every observation store in flight)
    static_assert(RPW <= 4 && NV <= 6, "flush_rows_pipe: named buffers");
    auto chunk = [&](int i) { const int c = tid + i * BLOCK; return c > ROWV - 1 ? ROWV - 1 : c; };
    const int c0 = chunk(0), c1 = chunk(1), c2 = chunk(2), c3 = chunk(3), c4 = chunk(4), c5 = chunk(5);
    const V *lv = reinterpret_cast<const V *>(ring);
    constexpr int RS = EB * NOUT / VEC;  // chunks per ring row
    V b00, b01, b02, b03, b04, b05;
    V b10, b11, b12, b13, b14, b15;
    V b20, b21, b22, b23, b24, b25;
    V b30, b31, b32, b33, b34, b35;
    if constexpr (0 < RPW && 0 < NV) b00 = lv[0 * RS + c0];
    if constexpr (0 < RPW && 1 < NV) b01 = lv[0 * RS + c1];
    if constexpr (0 < RPW && 2 < NV) b02 = lv[0 * RS + c2];
    if constexpr (0 < RPW && 3 < NV) b03 = lv[0 * RS + c3];
    if constexpr (0 < RPW && 4 < NV) b04 = lv[0 * RS + c4];
    if constexpr (0 < RPW && 5 < NV) b05 = lv[0 * RS + c5];
    if constexpr (1 < RPW && 0 < NV) b10 = lv[1 * RS + c0];
    if constexpr (1 < RPW && 1 < NV) b11 = lv[1 * RS + c1];
    if constexpr (1 < RPW && 2 < NV) b12 = lv[1 * RS + c2];
    if constexpr (1 < RPW && 3 < NV) b13 = lv[1 * RS + c3];
    if constexpr (1 < RPW && 4 < NV) b14 = lv[1 * RS + c4];
    if constexpr (1 < RPW && 5 < NV) b15 = lv[1 * RS + c5];
    if constexpr (2 < RPW && 0 < NV) b20 = lv[2 * RS + c0];
    if constexpr (2 < RPW && 1 < NV) b21 = lv[2 * RS + c1];
    if constexpr (2 < RPW && 2 < NV) b22 = lv[2 * RS + c2];
    if constexpr (2 < RPW && 3 < NV) b23 = lv[2 * RS + c3];
    if constexpr (2 < RPW && 4 < NV) b24 = lv[2 * RS + c4];
    if constexpr (2 < RPW && 5 < NV) b25 = lv[2 * RS + c5];
    if constexpr (3 < RPW && 0 < NV) b30 = lv[3 * RS + c0];
    if constexpr (3 < RPW && 1 < NV) b31 = lv[3 * RS + c1];
    if constexpr (3 < RPW && 2 < NV) b32 = lv[3 * RS + c2];
    if constexpr (3 < RPW && 3 < NV) b33 = lv[3 * RS + c3];
    if constexpr (3 < RPW && 4 < NV) b34 = lv[3 * RS + c4];
    if constexpr (3 < RPW && 5 < NV) b35 = lv[3 * RS + c5];
    if constexpr (0 < RPW) {
        V *gv = reinterpret_cast<V *>(a.obs + ((int64_t)(k0 + 0) * N + blk0) * NOUT);
        if constexpr (0 < NV) GEMX_ROW_STORE(&gv[c0], b00);
        if constexpr (1 < NV) GEMX_ROW_STORE(&gv[c1], b01);
        if constexpr (2 < NV) GEMX_ROW_STORE(&gv[c2], b02);
        if constexpr (3 < NV) GEMX_ROW_STORE(&gv[c3], b03);
        if constexpr (4 < NV) GEMX_ROW_STORE(&gv[c4], b04);
        if constexpr (5 < NV) GEMX_ROW_STORE(&gv[c5], b05);
    }
    if constexpr (1 < RPW) {
        V *gv = reinterpret_cast<V *>(a.obs + ((int64_t)(k0 + 1) * N + blk0) * NOUT);
        if constexpr (0 < NV) GEMX_ROW_STORE(&gv[c0], b10);
        if constexpr (1 < NV) GEMX_ROW_STORE(&gv[c1], b11);
        if constexpr (2 < NV) GEMX_ROW_STORE(&gv[c2], b12);
        if constexpr (3 < NV) GEMX_ROW_STORE(&gv[c3], b13);
        if constexpr (4 < NV) GEMX_ROW_STORE(&gv[c4], b14);
        if constexpr (5 < NV) GEMX_ROW_STORE(&gv[c5], b15);
    }
    if constexpr (2 < RPW) {
        V *gv = reinterpret_cast<V *>(a.obs + ((int64_t)(k0 + 2) * N + blk0) * NOUT);
        if constexpr (0 < NV) GEMX_ROW_STORE(&gv[c0], b20);
        if constexpr (1 < NV) GEMX_ROW_STORE(&gv[c1], b21);
        if constexpr (2 < NV) GEMX_ROW_STORE(&gv[c2], b22);
        if constexpr (3 < NV) GEMX_ROW_STORE(&gv[c3], b23);
        if constexpr (4 < NV) GEMX_ROW_STORE(&gv[c4], b24);
        if constexpr (5 < NV) GEMX_ROW_STORE(&gv[c5], b25);
    }
    if constexpr (3 < RPW) {
        V *gv = reinterpret_cast<V *>(a.obs + ((int64_t)(k0 + 3) * N + blk0) * NOUT);
        if constexpr (0 < NV) GEMX_ROW_STORE(&gv[c0], b30);
        if constexpr (1 < NV) GEMX_ROW_STORE(&gv[c1], b31);
        if constexpr (2 < NV) GEMX_ROW_STORE(&gv[c2], b32);
        if constexpr (3 < NV) GEMX_ROW_STORE(&gv[c3], b33);
        if constexpr (4 < NV) GEMX_ROW_STORE(&gv[c4], b34);
        if constexpr (5 < NV) GEMX_ROW_STORE(&gv[c5], b35);
    }
    if (a.done != nullptr) {  // done rows are EB contiguous bytes: EB / 16 chunks of 16 bytes per row
        constexpr int CPD = EB / 16;
        static_assert(RPW * CPD <= BLOCK, "one pass");
        typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
        if (tid < RPW * CPD) {
            const int row = tid / CPD, col = tid - row * CPD;
            __builtin_nontemporal_store(*reinterpret_cast<const v4u_t *>(donebuf + row * EB + col * 16),
                                        reinterpret_cast<v4u_t *>(a.done + (int64_t)(k0 + row) * N + blk0 + col * 16));
        }
    }
#undef GEMX_ROW_STORE
}

// Fused reward (WeightedSumOfErrors.reward, weighted_sum_of_errors.py:125-129) of staged observation rows of one 64-env
// workgroup, from the LDS ring and the caller's reference tensor.  One lane = one env.  Two halves, so that the reference
// loads can be issued long before they are needed (their latency is never on the critical path):
//   reward_fetch: this lane's references of rows k0 .. k0+nr-1 (nr <= RB) -> registers;
//   reward_apply: reward of those rows -> reward tensor.
template <int RB, class R>
__device__ __forceinline__ void reward_fetch(const KArgs<R> &a, int k0, int nr, int64_t e, R (&rv)[RB][GEMX_MAX_REF]) {
    const int n_ref = a.rh.n_ref;
    const int64_t N = a.N;
#pragma unroll
    for (int s = 0; s < RB; ++s) {
#pragma unroll
        for (int j = 0; j < GEMX_MAX_REF; ++j) {
            rv[s][j] = R(0);
            if (s < nr && j < n_ref) rv[s][j] = a.refs[((int64_t)(k0 + s) * N + e) * n_ref + j];
        }
    }
}
// The reward description as wave-uniform VALUES (SGPRs): read once per kernel, so that no scalar load sits in the row loops
// (stores to the reward tensor could alias the description as far as the compiler knows).  The first HOT terms cover every
// reference env (<= 3 referenced states); further weighted states take the slow path through memory.
template <class R> struct RewardRegs {
    static constexpr int HOT = GEMX_REWARD_HOT;
    int32_t n_term, general, col[HOT], kind[HOT];
    R coef[HOT], inv_len[HOT], power[HOT], bias, violation_reward;
    // from the kernel arguments (scalar loads, see RewardHot); unused hot terms carry weight 0, so the hot path has no per-term branch
    __device__ __forceinline__ void load(const RewardHot<R> &w) {
        n_term = w.n_term;
        general = w.general;
#pragma unroll
        for (int t = 0; t < HOT; ++t) {
            col[t] = w.col[t]; kind[t] = w.kind[t]; coef[t] = w.coef[t]; inv_len[t] = w.inv_len[t]; power[t] = w.power[t];
        }
        bias = w.bias;
        violation_reward = w.violation_reward;
    }
};
// GENERAL: reward_power other than 1 or 2 allowed (pow(): a ~300-instruction expansion, so it must not be unrolled per term)
template <bool GENERAL, class R> __device__ __forceinline__ R reward_term(R o, R ref, R inv_len, int kind, R power, R coef) {
    const R dlt = fabs(o - ref) * inv_len;
    R p = dlt * (kind == 2 ? dlt : R(1));  // (a select, not a branch: dlt * 1 == dlt exactly)
    if (GENERAL && kind == 3) p = pow(dlt, power);
    return coef * p;
}
template <int NOUT, int RB, class R, int EB = BLOCK>
__device__ __forceinline__ void reward_apply(const KArgs<R> &a, const RewardRegs<R> &W, const R *ring, const unsigned char *donebuf, int k0,
                                             int row0, int nr, int tid, int64_t env, bool valid, const R (&rv)[RB][GEMX_MAX_REF]) {
    constexpr int HOT = RewardRegs<R>::HOT;
    static_assert(HOT >= GEMX_MAX_REF, "referenced states must be hot terms");
    const int64_t N = a.N;
    const bool aos = a.P.obs_layout == GEMX_OBS_AOS;
    const int el = tid < EB ? tid : EB - 1;
    auto obs_at = [&](int row, int c) { return aos ? ring[((size_t)row * EB + el) * NOUT + c] : ring[((size_t)row * NOUT + c) * EB + el]; };
    // the hot terms' observations and the done bytes of ALL rows first: RB * (HOT + 1) LDS reads in flight at once instead of one
    // exposed LDS latency per term (the output waves spent ~1250 cycles per row here, s_memtime probe)
    R oh[RB][HOT];
    unsigned char dn[RB];
#pragma unroll
    for (int s = 0; s < RB; ++s) {
        const int row = row0 + (s < nr ? s : 0);
#pragma unroll
        for (int t = 0; t < HOT; ++t) oh[s][t] = obs_at(row, W.col[t]);
        dn[s] = donebuf[row * EB + el];
    }
#pragma unroll
    for (int s = 0; s < RB; ++s) {
        if (s < nr) {
            const int row = row0 + s;  // row of the ring / done ring
            R acc = R(0);
            if (!W.general) {
#pragma unroll
                for (int t = 0; t < HOT; ++t)  // terms < n_ref are the referenced states (reference column t), the others compare with 0
                    acc += reward_term<false, R>(oh[s][t], t < GEMX_MAX_REF ? rv[s][t] : R(0), W.inv_len[t], W.kind[t], W.power[t], W.coef[t]);
            }
            // slow path through memory: terms beyond the hot ones, and EVERY term when some reward_power is not 1 or 2 (one pow() site)
#pragma nounroll
            for (int t = W.general ? 0 : HOT; t < W.n_term; ++t) {
                R ref = R(0);
#pragma unroll
                for (int j = 0; j < GEMX_MAX_REF; ++j) ref = (t == j) ? rv[s][j] : ref;
                acc += reward_term<true, R>(obs_at(row, a.rw->col[t]), ref, a.rw->inv_len[t], a.rw->kind[t], a.rw->power[t], a.rw->coef[t]);
            }
            const R wse = W.bias - acc;
            const R r = dn[s] ? W.violation_reward : wse;  // (1 - v) * wse + v * violation_reward, v in {0, 1}
            if (valid) a.reward[(int64_t)(k0 + s) * N + env] = r;
        }
    }
}
// all `nr` rows of an I/O block (single-wave kernel): groups of RB rows, group g+1 fetched while group g is evaluated; the
// first group (`ra`) was fetched by the caller BEFORE the block's compute phase
constexpr int REWARD_RB = 4;
template <int NOUT, class R>
__device__ __forceinline__ void reward_rows(const KArgs<R> &a, const R *ring, const unsigned char *donebuf, int k0, int nr, int tid,
                                            int64_t env, bool valid, R (&ra)[REWARD_RB][GEMX_MAX_REF]) {
    constexpr int RB = REWARD_RB;
    const int64_t e = valid ? env : a.N - 1;
    RewardRegs<R> W;
    W.load(a.rh);
    R rb[RB][GEMX_MAX_REF];
    auto cnt = [&](int s0) { return nr - s0 < RB ? (nr - s0 < 0 ? 0 : nr - s0) : RB; };
    for (int s0 = 0; s0 < nr; s0 += 2 * RB) {
        reward_fetch<RB, R>(a, k0 + s0 + RB, cnt(s0 + RB), e, rb);
        reward_apply<NOUT, RB, R>(a, W, ring, donebuf, k0 + s0, s0, cnt(s0), tid, env, valid, ra);
        reward_fetch<RB, R>(a, k0 + s0 + 2 * RB, cnt(s0 + 2 * RB), e, ra);
        reward_apply<NOUT, RB, R>(a, W, ring, donebuf, k0 + s0 + RB, s0 + RB, cnt(s0 + RB), tid, env, valid, rb);
    }
}

// S control steps of one I/O block.  COOP: actions come from the LDS tile (no global memory access at all in
// this loop); otherwise straight from global memory (K == 1, tail workgroup, unaligned tensors).  The two variants
// are separate instantiations on purpose: a pointer that may be LDS or global would compile to FLAT loads, whose
// s_waitcnt covers vmcnt as well and would wait for every outstanding observation store.
template <bool COOP, int SYS, int CONV, int LOAD, int SOLVER, bool IL, class R>
__device__ __forceinline__ void compute_block(const KArgs<R> &a, R (&y)[SysTraits<SYS>::ND], typename Angle<R>::T &ang, uint32_t &sw,
                                              R (&obs)[SysTraits<SYS>::NOUT], uint32_t &done_or, uint32_t &bad_action, R *ring,
                                              const unsigned char *atile, unsigned char *donebuf, int k0, int sb, int tid,
                                              int64_t e, typename Angle<R>::T init_ang, R *fifo, int &slot, R (&sup)[2], bool lin_ok,
                                              const R (&linc)[lin_regs<SYS, R, IL>()], R &hcar) {
    constexpr int ND = SysTraits<SYS>::ND, NOUT = SysTraits<SYS>::NOUT, NACT = ConvTraits<CONV>::NACT;
    constexpr bool DISCRETE = ConvTraits<CONV>::DISCRETE;
    constexpr int ABYTES = DISCRETE ? 1 : NACT * (int)sizeof(R);
    constexpr int ROWB = BLOCK * ABYTES;
    using ST = Stepper<SYS, conv_base<CONV>(), LOAD, SOLVER, IL, R>;
    constexpr int NACTC = conv_nact_c<CONV>();  // converter-side action width (FIFO entries)
    const DevParams<R> &P = a.P;
    // Software pipeline (COOP): the action of step s+1 is read from LDS while step s computes, and the observation
    // row of step s is written to the ring at the top of iteration s+1, so the only LDS wait of an iteration (for the
    // action it is about to use) finds a read that was issued a whole step earlier.
    auto read_action = [&](int s, R (&dst)[MAX_ACT], uint32_t &ddst) {
        if (COOP) {
            if (DISCRETE) ddst = atile[s * ROWB + tid];
            else {
#pragma unroll
                for (int i = 0; i < NACT; ++i) dst[i] = reinterpret_cast<const R *>(atile + s * ROWB)[tid * NACT + i];
            }
        } else {
            const unsigned char *g = a.actions + ((int64_t)(k0 + s) * a.N + e) * ABYTES;
            if (DISCRETE) ddst = *g;
            else {
#pragma unroll
                for (int i = 0; i < NACT; ++i) dst[i] = reinterpret_cast<const R *>(g)[i];
            }
        }
    };
    auto write_ring = [&](int s, bool dn) {
        if (P.obs_layout == GEMX_OBS_AOS) {
#pragma unroll
            for (int j = 0; j < NOUT; ++j) ring[(s * BLOCK + tid) * NOUT + j] = obs[j];
        } else {
#pragma unroll
            for (int j = 0; j < NOUT; ++j) ring[(s * NOUT + j) * BLOCK + tid] = obs[j];
        }
        donebuf[s * BLOCK + tid] = dn ? 1 : 0;
    };
    R nact[MAX_ACT];
#pragma unroll
    for (int i = 0; i < MAX_ACT; ++i) nact[i] = R(0);
    uint32_t ndact = 0;
    read_action(0, nact, ndact);
    bool pdone = false;
    for (int s = 0; s < sb; ++s) {
        R act[MAX_ACT];
#pragma unroll
        for (int i = 0; i < MAX_ACT; ++i) act[i] = nact[i];
        uint32_t dact = ndact;
        if (a.obs_every && s > 0) write_ring(s - 1, pdone);       // row of the previous step (obs still holds it)
        if (COOP && s + 1 < sb) read_action(s + 1, nact, ndact);   // prefetch (LDS only; global loads would add vmcnt waits)
        if (DISCRETE) { bad_action |= dact >= (uint32_t)ConvTraits<CONV>::NACTIONS; dact &= (uint32_t)(ConvTraits<CONV>::NACTIONS - 1); }
        // wrapper order of the reference: [DqToAbcActionProcessor [DeadTimeProcessor [system(control_space)]]] -- the processor
        // transforms BEFORE the delay queue, a system built with control_space='dq' transforms the delayed (u_d, u_q) AFTER it
        if (conv_dq<CONV>() && P.dq_processor) dq_action_stage<SYS, CONV, R>(P, y, ang, act);
        if (P.delay > 0) {  // DeadTimeProcessor.simulate (dead_time_processor.py:74-85): swap with the FIFO slot of this step
            R *f = fifo + ((size_t)slot * BLOCK + tid) * NACTC;
            if (DISCRETE) {
                const uint32_t old = (uint32_t)f[0];  // small integers are exact in R
                f[0] = (R)dact;
                dact = old;
            } else {
#pragma unroll
                for (int i = 0; i < NACTC; ++i) { const R old = f[i]; f[i] = act[i]; act[i] = old; }
            }
            slot = slot + 1 == P.delay ? 0 : slot + 1;
        }
        if (conv_dq<CONV>() && !P.dq_processor) dq_action_stage<SYS, CONV, R>(P, y, ang, act);
        // RCVoltageSupply.get_voltage(self._t, i_sup) (voltage_supplies.py:116-123): one explicit Euler step of the supply's own
        // state over the time since the previous control step (0 right after a reset), before the converter is evaluated
        DevParams<R> PL = P;  // per-lane view of the parameters: only u_sup differs between lanes
        if (P.rc_supply) {
            const R isup = supply_current<SYS, conv_base<CONV>(), R>(P, y, ang, sw, act);
            sup[0] = sup[0] + (P.u_sup - sup[0] - P.sup_r * isup) * P.sup_inv_rc * sup[1];
            sup[1] = P.tau;
            PL.u_sup = sup[0];
        }
        if (linable<SYS, LOAD, SOLVER, IL, R>() && lin_ok) full_step<ST, ND, NOUT, R, linable<SYS, LOAD, SOLVER, IL, R>()>(PL, y, ang, sw, act, dact, obs, linc);
        else full_step<ST, ND, NOUT, R, false>(PL, y, ang, sw, act, dact, obs, nullptr, &hcar);
        if (!IL && conv_has_legs<CONV>() && P.rc_supply) sw = ST::legs_of(dact);  // (the IL code keeps `sw` itself)
        const bool done = constraint_done<ST, NOUT, R>(P, obs);
        done_or |= done ? 1u : 0u;
        pdone = done;
        if (done && P.auto_reset) {  // `if terminated: env.reset()`; switching state survives (converters.py:45-54)
#pragma unroll
            for (int j = 0; j < ND; ++j) y[j] = P.init[j];
            ang = init_ang;
            hcar = R(0);  // (the error-controlled solver starts an episode without a step-size prediction, like set_initial_value())
            if (P.init_kind && (int64_t)blockIdx.x * BLOCK + tid < a.N) draw_initial_state<SYS, R>(a, e, y, ang);  // (not the clamped tail lanes)
            sup[0] = P.u_sup;  // RCVoltageSupply.reset: the capacitor is loaded again, the supply's clock restarts
            sup[1] = R(0);
            for (int d = 0; d < P.delay; ++d) {  // DeadTimeProcessor.reset: the deque is refilled with the reset action
#pragma unroll
                for (int i = 0; i < NACTC; ++i) fifo[((size_t)d * BLOCK + tid) * NACTC + i] = P.dreset[i];
            }
        }
        if (!COOP && s + 1 < sb) read_action(s + 1, nact, ndact);
    }
    if (a.obs_every) write_ring(sb - 1, pdone);
}

// ------------------------------------------------------------------------------------------------
// THE kernel: K control steps of N envs (K = 1 is the single-step path of gemx_step()).
//
// I/O is blocked in groups of S control steps, because on gfx950 loads AND stores retire through the same
// in-order vmcnt counter: waiting for the next action right after issuing this step's observation stores would
// expose the HBM store latency (~2 us) on every step.  Per block of S steps:
//   1. issue the global loads of the NEXT block's action tile (S rows x 64 envs, contiguous per row) as 16-byte
//      per-lane loads into registers (no wait);
//   2. S control steps, reading this block's actions from LDS and writing observation rows / done bytes to an
//      LDS ring (no global memory traffic at all inside the compute loop);
//   3. park the prefetched action tile in the other half of the LDS action buffer (its loads had S steps of
//      arithmetic to land);
//   4. flush the ring: each 64-env row is a contiguous 64*S_out*sizeof(R) span of the [K, N, S_out] output, written
//      with 16-byte-per-lane stores; the stores drain while the next block computes.
// ------------------------------------------------------------------------------------------------
template <int SYS, int CONV, int LOAD, int SOLVER, bool IL, class R>
// (waves_per_eu: at least two waves per SIMD, i.e. at most 256 VGPRs -- the general kernel of the synchronous machines sits right at
// that edge (255 .. 258 depending on what else is inlined), and 258 would halve its residency; the DFIM's needs ~280 either way)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(SYS == GEMX_SYS_DFIM ? 1 : 2))) void advance_kernel(const KArgs<R> a) {
    constexpr int ND = SysTraits<SYS>::ND, NOUT = SysTraits<SYS>::NOUT, NACT = ConvTraits<CONV>::NACT;
    constexpr bool DISCRETE = ConvTraits<CONV>::DISCRETE;
    constexpr int ABYTES = DISCRETE ? 1 : NACT * (int)sizeof(R);  // action bytes per env and step
    constexpr int ROWB = BLOCK * ABYTES;                           // action bytes per 64-env row
    constexpr int CPR = ROWB / 16;                                 // 16-byte chunks per action row
    constexpr int VEC = 16 / sizeof(R);
    constexpr int ROWV = BLOCK * NOUT / VEC;                       // 16-byte chunks per observation row
    constexpr int NCH = act_chunks(CPR);                           // 16-byte action chunks per lane and I/O block
    using AngT = typename Angle<R>::T;
    using V = typename std::conditional<sizeof(R) == 4, float4, double2>::type;

#ifdef GEMX_TIMING
    const unsigned long long pT0 = clock64(), pW0 = wall_clock64();
    unsigned long long pT1 = 0, pT2 = 0, pT3 = 0;
#endif
    const DevParams<R> &P = a.P;
    const int tid = threadIdx.x;
    const int64_t blk0 = (int64_t)blockIdx.x * BLOCK;
    const int64_t env = blk0 + tid;
    const int64_t N = a.N;
    const bool valid = env < N;
    const int64_t e = valid ? env : N - 1;  // clamp loads of the tail lanes; their stores are masked
    const int rows = (int)((N - blk0) < BLOCK ? (N - blk0) : BLOCK);
    const bool full = rows == BLOCK;
    const int S = a.S;
    const int K = a.K;

    // LDS carve-up: observation ring [S][64*NOUT] R | action tiles [2][S*ROWB] bytes | done ring [S][64] bytes
    R *ring = reinterpret_cast<R *>(gemx_smem);
    unsigned char *actbuf = gemx_smem + (size_t)S * BLOCK * NOUT * sizeof(R);
    unsigned char *donebuf = actbuf + 2 * (size_t)S * ROWB;
    constexpr int NACTC = conv_nact_c<CONV>();
    // DeadTimeProcessor FIFO [delay][64][NACTC] R behind the (16-byte padded) done ring
    R *fifo = reinterpret_cast<R *>(donebuf + (((size_t)S * BLOCK + 15) & ~(size_t)15));
    const int ring_phase = fifo_phase_read(a);
    int slot = ring_phase;
    const bool coop = a.coop && full && K > 1;   // cooperative action staging for this workgroup (uniform)

    R y[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) y[j] = a.state[(int64_t)j * N + e];
    AngT ang = AngT(0);
    if (SysTraits<SYS>::HAS_ANGLE) ang = a.angle[e];
    uint32_t sw = 0;
    const bool USE_SW = conv_has_legs<CONV>() && (IL || P.rc_supply);  // leg states: dead time, or the RC supply's i_sup
    if (USE_SW) {
        sw = a.sw[e];
        if (conv_sw_bytes<CONV>() == 2) sw |= (uint32_t)a.sw[N + e] << 8;
    }
    const AngT init_ang = Angle<R>::from_bits(P.init_angle_rep);
    const bool lin_ok = lin_usable<SYS, LOAD, SOLVER, IL, R>(P, y[0]);  // wave-uniform
    R linc[lin_regs<SYS, R, IL>()];  // the one-step map's coefficients, in registers for the whole launch (drained with the prologue loads)
    lin_preload<SYS, R>(P, linable<SYS, LOAD, SOLVER, IL, R>() && lin_ok, linc);
    R sup[2] = {P.u_sup, R(0)};  // RCVoltageSupply: capacitor voltage, time since the supply's last update
    if (P.rc_supply) {
        sup[0] = a.state[(int64_t)ND * N + e];
        sup[1] = a.state[(int64_t)(ND + 1) * N + e];
    }
    // error-controlled solver: the step size its controller proposed last (state row ND + 2; 0 = none yet), see dp5_adaptive
    R hcar = R(0);
    if (SOLVER == GEMX_SOLVER_DP5 && P.adaptive) hcar = a.state[(int64_t)(ND + 2) * N + e];
    for (int d = 0; d < P.delay; ++d) {  // this lane's FIFO entries (only this lane ever touches them)
#pragma unroll
        for (int i = 0; i < NACTC; ++i) {
            const int64_t gi = ((int64_t)d * N + e) * NACTC + i;
            fifo[((size_t)d * BLOCK + tid) * NACTC + i] = DISCRETE ? (R)a.ring[gi] : reinterpret_cast<const R *>(a.ring)[gi];
        }
    }

    // per-lane 16-byte loads of one action tile (steps [k0, k0+sb)) into registers: tile_load / tile_park below
    V tile[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) tile[c] = V{};
    const unsigned char *act_blk = a.actions + blk0 * ABYTES;  // this workgroup's column of the action tensor
    const int64_t act_row_stride = N * ABYTES;
#define GEMX_TILE_LOAD(k0_, sb_)                                                                                      \
    do {                                                                                                              \
        const int nchunk_ = (sb_) * CPR;                                                                              \
        _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                                             \
            const int idx = c * BLOCK + tid;                                                                          \
            if (idx < nchunk_) {                                                                                      \
                const int row = idx / CPR, col = idx - row * CPR;                                                     \
                tile[c] = *reinterpret_cast<const V *>(act_blk + (int64_t)((k0_) + row) * act_row_stride + col * 16); \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
#define GEMX_TILE_PARK(half_, sb_)                                                                                    \
    do {                                                                                                              \
        const int nchunk_ = (sb_) * CPR;                                                                              \
        V *dst_ = reinterpret_cast<V *>(actbuf + (size_t)(half_) * S * ROWB);                                         \
        _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                                             \
            const int idx = c * BLOCK + tid;                                                                          \
            if (idx < nchunk_) dst_[idx] = tile[c];                                                                   \
        }                                                                                                             \
    } while (0)

    if (coop) {
        GEMX_TILE_LOAD(0, S < K ? S : K);
        GEMX_TILE_PARK(0, S < K ? S : K);
    }
    // Drain the prologue loads (state, angle, first action tile) HERE, once.  Otherwise the compiler parks a
    // conservative `s_waitcnt vmcnt(0)` at the first use inside the step loop, and since stores retire through the
    // same counter every I/O block would wait for the previous block's whole flush burst instead of overlapping it.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // gfx9 encoding: vmcnt(0), expcnt/lgkmcnt untouched
    __syncthreads();
#ifdef GEMX_TIMING
    pT1 = clock64();
#endif

    uint32_t done_or = 0, bad_action = 0;
    R obs[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) obs[j] = R(0);
    int half = 0;
    for (int k0 = 0; k0 < K; k0 += S) {
        const int sb = (K - k0) < S ? (K - k0) : S;
        const int k1 = k0 + sb;
        const int sb_next = (K - k1) < S ? (K - k1) : S;
        if (coop && sb_next > 0) GEMX_TILE_LOAD(k1, sb_next);  // 1. prefetch (no wait)
        R rfirst[REWARD_RB][GEMX_MAX_REF];  // fused reward: references of this block's first rows, in flight during the compute phase
        if (a.obs_every && a.rw != nullptr) reward_fetch<REWARD_RB, R>(a, k0, sb < REWARD_RB ? sb : REWARD_RB, e, rfirst);

        // 2. compute: no global memory traffic in here when coop
        const unsigned char *atile = actbuf + (size_t)half * S * ROWB;
        if (coop) compute_block<true, SYS, CONV, LOAD, SOLVER, IL, R>(a, y, ang, sw, obs, done_or, bad_action, ring, atile, donebuf, k0, sb, tid, e, init_ang, fifo, slot, sup, lin_ok, linc, hcar);
        else compute_block<false, SYS, CONV, LOAD, SOLVER, IL, R>(a, y, ang, sw, obs, done_or, bad_action, ring, atile, donebuf, k0, sb, tid, e, init_ang, fifo, slot, sup, lin_ok, linc, hcar);
        __syncthreads();
#ifdef GEMX_TIMING
        if (k0 == 0) pT2 = clock64();
#endif

        // 3. park the prefetched tile
        if (coop && sb_next > 0) GEMX_TILE_PARK(half ^ 1, sb_next);

        // 4. flush the rings (+ the fused reward of the staged rows)
        if (a.obs_every && a.rw != nullptr) reward_rows<NOUT, R>(a, ring, donebuf, k0, sb, tid, env, valid, rfirst);
        if (a.obs_every) flush_rings<NOUT, R>(a, ring, donebuf, k0, sb, tid, blk0, rows, full, valid, env);
        __syncthreads();
        half ^= 1;
    }

    if (!a.obs_every) {  // last-step-only mode: one row through ring slot 0
        if (P.obs_layout == GEMX_OBS_AOS) {
#pragma unroll
            for (int j = 0; j < NOUT; ++j) ring[tid * NOUT + j] = obs[j];
            __syncthreads();
            for (int idx = tid; idx < rows * NOUT; idx += BLOCK) a.obs[blk0 * NOUT + idx] = ring[idx];
        } else if (valid) {
#pragma unroll
            for (int j = 0; j < NOUT; ++j) a.obs[(int64_t)j * N + env] = obs[j];
        }
        if (a.done != nullptr && valid) a.done[env] = (uint8_t)done_or;
    }

    if (valid) {
#pragma unroll
        for (int j = 0; j < ND; ++j) a.state[(int64_t)j * N + env] = y[j];
        if (SysTraits<SYS>::HAS_ANGLE) a.angle[env] = ang;
        if (USE_SW) {
            a.sw[env] = (uint8_t)sw;
            if (conv_sw_bytes<CONV>() == 2) a.sw[N + env] = (uint8_t)(sw >> 8);
        }
        if (P.rc_supply) {
            a.state[(int64_t)ND * N + env] = sup[0];
            a.state[(int64_t)(ND + 1) * N + env] = sup[1];
        }
        if (SOLVER == GEMX_SOLVER_DP5 && P.adaptive) a.state[(int64_t)(ND + 2) * N + env] = hcar;
        for (int d = 0; d < P.delay; ++d) {
#pragma unroll
            for (int i = 0; i < NACTC; ++i) {
                const int64_t gi = ((int64_t)d * N + env) * NACTC + i;
                const R v = fifo[((size_t)d * BLOCK + tid) * NACTC + i];
                if (DISCRETE) a.ring[gi] = (unsigned char)(uint32_t)v;
                else reinterpret_cast<R *>(a.ring)[gi] = v;
            }
        }
    }
    if (bad_action && valid) atomicOr(a.err, 1u);
    fifo_phase_advance(a, ring_phase, tid == 0);
#ifdef GEMX_TIMING
    pT3 = clock64();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the stores have left the CU
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64) + 32 + (blockIdx.x ? 8 : 0);
        dbg[0] = pT1 - pT0; dbg[1] = pT2 - pT1; dbg[2] = pT3 - pT2; dbg[3] = clock64() - pT3; dbg[4] = pW0; dbg[5] = wall_clock64();
    }
#endif
#undef GEMX_TILE_LOAD
#undef GEMX_TILE_PARK
}

// ------------------------------------------------------------------------------------------------
// K = 1 (gemx_step, the closed-loop path: a policy between every two control steps).  Such a launch runs every instruction ONCE, so
// its time is latency, not throughput.  Through advance_kernel a step took 4.4 us inside the kernel (s_memtime probe, PMSM, 16384
// envs): 3200 cycles until the state had arrived, 4600 for the one control step (the action was loaded only then: a second trip to
// HBM; a step of the fused rollout takes ~350 cycles), 2200 for LDS ring -> barrier -> 16-byte stores -> barrier -> state stores.
// step_kernel does the same step with ONE batch of loads (state, angle, leg states, supply state, action, FIFO slot -- all issued
// before the first wait), no LDS, no barrier, and the observation row stored straight from the lane's registers (a row is
// NOUT * sizeof(R) contiguous bytes, so consecutive lanes still cover whole cache lines).  Same device functions as the other two
// kernels: bit-identical results (tests).
// ------------------------------------------------------------------------------------------------
template <int SYS, int CONV, int LOAD, int SOLVER, bool IL, class R>
__global__ __launch_bounds__(BLOCK) void step_kernel(const KArgs<R> a) {
    constexpr int ND = SysTraits<SYS>::ND, NOUT = SysTraits<SYS>::NOUT, NACT = ConvTraits<CONV>::NACT;
    constexpr bool DISCRETE = ConvTraits<CONV>::DISCRETE;
    constexpr int NACTC = conv_nact_c<CONV>();
    using AngT = typename Angle<R>::T;
    using ST = Stepper<SYS, conv_base<CONV>(), LOAD, SOLVER, IL, R>;
    const DevParams<R> &P = a.P;
    const bool USE_SW = conv_has_legs<CONV>() && (IL || P.rc_supply);
    const int64_t N = a.N;
    const int64_t env = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = env < N;
    const int64_t e = valid ? env : N - 1;  // tail lanes recompute the last env; their stores are masked

    // ---- one batch of loads
    R y[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) y[j] = a.state[(int64_t)j * N + e];
    AngT ang = AngT(0);
    if (SysTraits<SYS>::HAS_ANGLE) ang = a.angle[e];
    uint32_t sw = 0;
    if (USE_SW) {
        sw = a.sw[e];
        if (conv_sw_bytes<CONV>() == 2) sw |= (uint32_t)a.sw[N + e] << 8;
    }
    R sup[2] = {P.u_sup, R(0)};
    if (P.rc_supply) {
        sup[0] = a.state[(int64_t)ND * N + e];
        sup[1] = a.state[(int64_t)(ND + 1) * N + e];
    }
    R hcar = R(0);  // error-controlled solver: carried step size (state row ND + 2)
    if (SOLVER == GEMX_SOLVER_DP5 && P.adaptive) hcar = a.state[(int64_t)(ND + 2) * N + e];
    R act[MAX_ACT];
#pragma unroll
    for (int i = 0; i < MAX_ACT; ++i) act[i] = R(0);
    uint32_t dact = 0;
    if (DISCRETE) dact = a.actions[e];
    else {
#pragma unroll
        for (int i = 0; i < NACT; ++i) act[i] = reinterpret_cast<const R *>(a.actions)[e * NACT + i];
    }
    // DeadTimeProcessor (dead_time_processor.py:74-85): this step's slot of the env's FIFO (global step count mod delay)
    R popped[NACTC];
#pragma unroll
    for (int i = 0; i < NACTC; ++i) popped[i] = R(0);
    const int ring_phase = fifo_phase_read(a);
    const int64_t slot0 = ((int64_t)ring_phase * N + e) * NACTC;
    if (P.delay > 0) {
#pragma unroll
        for (int i = 0; i < NACTC; ++i) popped[i] = DISCRETE ? (R)a.ring[slot0 + i] : reinterpret_cast<const R *>(a.ring)[slot0 + i];
    }
    const AngT init_ang = Angle<R>::from_bits(P.init_angle_rep);
    const bool lin_ok = lin_usable<SYS, LOAD, SOLVER, IL, R>(P, y[0]);  // wave-uniform

    // ---- the control step, exactly as compute_block() does it
    uint32_t bad_action = 0;
    if (DISCRETE) { bad_action = dact >= (uint32_t)ConvTraits<CONV>::NACTIONS; dact &= (uint32_t)(ConvTraits<CONV>::NACTIONS - 1); }
    if (conv_dq<CONV>() && P.dq_processor) dq_action_stage<SYS, CONV, R>(P, y, ang, act);
    if (P.delay > 0) {  // swap with the FIFO slot
        if (DISCRETE) {
            if (valid) a.ring[slot0] = (unsigned char)dact;
            dact = (uint32_t)popped[0];
        } else {
#pragma unroll
            for (int i = 0; i < NACTC; ++i) {
                if (valid) reinterpret_cast<R *>(a.ring)[slot0 + i] = act[i];
                act[i] = popped[i];
            }
        }
    }
    if (conv_dq<CONV>() && !P.dq_processor) dq_action_stage<SYS, CONV, R>(P, y, ang, act);
    DevParams<R> PL = P;  // per-lane view: only u_sup differs between lanes (RCVoltageSupply)
    if (P.rc_supply) {
        const R isup = supply_current<SYS, conv_base<CONV>(), R>(P, y, ang, sw, act);
        sup[0] = sup[0] + (P.u_sup - sup[0] - P.sup_r * isup) * P.sup_inv_rc * sup[1];
        sup[1] = P.tau;
        PL.u_sup = sup[0];
    }
    R obs[NOUT];
    if (linable<SYS, LOAD, SOLVER, IL, R>() && lin_ok) full_step<ST, ND, NOUT, R, linable<SYS, LOAD, SOLVER, IL, R>()>(PL, y, ang, sw, act, dact, obs);
    else full_step<ST, ND, NOUT, R, false>(PL, y, ang, sw, act, dact, obs, nullptr, &hcar);
    if (!IL && conv_has_legs<CONV>() && P.rc_supply) sw = ST::legs_of(dact);
    const bool done = constraint_done<ST, NOUT, R>(P, obs);
    if (done && P.auto_reset) {  // `if terminated: env.reset()`; switching state survives (converters.py:45-54)
#pragma unroll
        for (int j = 0; j < ND; ++j) y[j] = P.init[j];
        ang = init_ang;
        hcar = R(0);
        if (P.init_kind && valid) draw_initial_state<SYS, R>(a, e, y, ang);
        sup[0] = P.u_sup;
        sup[1] = R(0);
        if (valid) {
            for (int d = 0; d < P.delay; ++d) {  // DeadTimeProcessor.reset: the deque is refilled with the reset action
#pragma unroll
                for (int i = 0; i < NACTC; ++i) {
                    const int64_t gi = ((int64_t)d * N + e) * NACTC + i;
                    if (DISCRETE) a.ring[gi] = (unsigned char)P.dreset_d;
                    else reinterpret_cast<R *>(a.ring)[gi] = P.dreset[i];
                }
            }
        }
    }

    // ---- stores
    if (valid) {
        if (P.obs_layout == GEMX_OBS_AOS) {
            R *row = a.obs + env * NOUT;
            constexpr int W = (NOUT * sizeof(R)) % 16 == 0 ? 16 / sizeof(R) : ((NOUT * sizeof(R)) % 8 == 0 ? 8 / sizeof(R) : 1);  // row alignment
            typedef R vec_t __attribute__((ext_vector_type(W)));
            if constexpr (W > 1) {
#pragma unroll
                for (int j = 0; j < NOUT; j += W) {
                    vec_t v;
#pragma unroll
                    for (int q = 0; q < W; ++q) v[q] = obs[j + q];
                    *reinterpret_cast<vec_t *>(row + j) = v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NOUT; ++j) row[j] = obs[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NOUT; ++j) a.obs[(int64_t)j * N + env] = obs[j];
        }
        if (a.done != nullptr) a.done[env] = done ? 1 : 0;
#pragma unroll
        for (int j = 0; j < ND; ++j) a.state[(int64_t)j * N + env] = y[j];
        if (SysTraits<SYS>::HAS_ANGLE) a.angle[env] = ang;
        if (USE_SW) {
            a.sw[env] = (uint8_t)sw;
            if (conv_sw_bytes<CONV>() == 2) a.sw[N + env] = (uint8_t)(sw >> 8);
        }
        if (P.rc_supply) {
            a.state[(int64_t)ND * N + env] = sup[0];
            a.state[(int64_t)(ND + 1) * N + env] = sup[1];
        }
        if (SOLVER == GEMX_SOLVER_DP5 && P.adaptive) a.state[(int64_t)(ND + 2) * N + env] = hcar;
        if (bad_action) atomicOr(a.err, 1u);
    }
    fifo_phase_advance(a, ring_phase, threadIdx.x == 0);
}

// ------------------------------------------------------------------------------------------------
// Pipelined variant: a single wave per SIMD is bound by its own instruction issue rate (one VALU instruction per >= 4.5 cycles,
// tools/microbench_issue.hip), so the control step is split across the 1 + OW + 1 waves of a workgroup that serve the SAME 64 envs:
//   wave 0 (integrator):   action (LDS staging buffer / per-action voltage table, read one and two steps ahead) -> converter -> ODE
//                          integration -> default constraint -> auto-reset -> hand-off row to LDS.  No global memory instruction
//                          between its prologue and its epilogue.
//   waves 1..OW (output):  hand-off row -> observation row -> LDS ring -> fused reward -> 16-byte non-temporal stores to HBM.  They
//                          issue no global loads, so they never wait on vmcnt: their stores are fire-and-forget.
//   last wave (loader):    the NEXT block's actions / reward references global -> LDS (global_load_lds), awaited here: issued by the
//                          integrator they queued behind the output waves' stores and stalled it at their issue.
// They meet at ONE s_barrier per block of D control steps; the hand-off buffer is double-buffered so the output waves work on block
// b-1 while the integrator runs block b and the loader fetches block b+1.  Full blocks are unrolled into branch-free basic blocks of
// four steps.  Shapes <D, OW>: <12, 3>, <4, 2>, <2, 2> (launch_advance_t picks by N and LDS footprint).
// Preconditions (checked by the launcher): full 64-env workgroups, 16-byte aligned tensors (coop), obs_every, K >= 2,
// constraint kind none/default, one solver sub-step; RC supply / random initialisers: the FULL instantiation only.
// ------------------------------------------------------------------------------------------------
// FULL = true: the variant that also serves an RCVoltageSupply (per-lane supply voltage: two more registers of state in the integrator
// and one more hand-off value, from which the output waves take the u_sup column) and random initialisers (the rare auto-reset path
// draws the new state in the integrator wave, reset counter in a register).  One extra instantiation (shape <4, 2>) instead of
// burdening the common ones with the registers of that code (the fp64 Philox / inverse-CDF draw alone costs ~50 VGPRs).
// The shallow shapes (<4, 2>, <2, 2>: four waves per workgroup) exist to put FOUR workgroups on a CU -- four waves per SIMD, i.e. at most 128
// VGPRs -- and the launcher's residency arithmetic (LDS, wave slots) takes that for granted.  Several instantiations land a handful of
// registers above the line on their own (SCIM RK4 <4, 2>: 129; SCIM error-controlled <2, 2>: 134, which ran BASELINE config 4 under
// ScipyOdeSolver() in two rounds instead of one: tools/vgpr_report.py, profiles/r04j_vgpr_report.md), and which side of it a kernel falls
// on moves with every unrelated edit.  So the line is REQUESTED where the natural count is close to it (everything but the DFIM's rows and
// the dead-time variants of the fixed DP5 step, which need 160-170: forcing those would spill into the step loop).
// target of the large-batch rate limiter: chip-wide algorithmic GB/s (tools/microbench_rowpitch.hip, profiles/r04k_*; GEMX_PACE_GBPS overrides)
#ifndef GEMX_PACE_DEFAULT_ON
#define GEMX_PACE_DEFAULT_ON 1
#endif
// prepared draws per lane (FULL pipelined kernel, random initialisers): four; eight for the induction machines, whose episodes under random
// initial states are short (a third of them a few steps) and whose draw takes the loader three to four passes -- with four entries 7.7 % of
// the SCIM's resets found the queue empty and drew inline (round 5 A/B: +10 % for the SCIM, -5 to -15 % for the PMSM, hence per system)
template <int SYS> constexpr int prep_q() { return (SYS == GEMX_SYS_SCIM || SYS == GEMX_SYS_DFIM) ? 8 : 4; }
#ifndef GEMX_PREP_DRAWS
#define GEMX_PREP_DRAWS 1  // (0: A/B builds -- every reset draws inline)
#endif
// RINIT: waves per SIMD the instantiation is compiled for.  Left alone the induction machines' kernels took 286 VGPRs (one workgroup
// per CU): at most 256, two per CU, SCIM 13.4 -> 22.6 G env-steps/s.  The others: at most 168, three per CU, with the inline draw marked
// as the cold side of its branch so that the spills land THERE (PMSM at 131072 envs 59 -> 68 G, level elsewhere; without the hint 57 ->
// 50, and 39 at four waves; the induction machines lose 20-35 % at three or four).  profiles/r05t_ab_rinit_waves.txt, r05u_ab_rinit_expect.txt
#define GEMX_EXPECT_PREPARED(x) __builtin_expect((x), 1)
// Phases of the loader's prepared-draw state machine (scan + generator block | generator block | ... | finish) per hand-off block
// (KArgs::prep_phases).  Round 5: one -- a pass cost 1450-6600 cycles (Philox, and every field of the description re-read from global memory)
// and more per pass made the loader the slowest wave of a block.  Round 6 (Threefry, the description in LDS: ~560 cycles per generator block,
// 870-2500 for the finish): TWO for the induction machines, whose draw takes three phases and whose lanes drain their queues fastest
// (resets that find the queue empty 7.4 % -> 2.4 %; same box, product builds, SCIM cont at 131072 envs 0.373 -> 0.430 of the roofline, 16384
// envs 0.216 -> 0.250; three phases: 0.360 / 0.196), ONE for the others (PMSM finite at 16384 envs 0.411 -> 0.336 with two, speed control
// level).  GEMX_PREP_PHASES overrides (A/B builds).  profiles/r06_rinit_probe.md.
#ifndef GEMX_RINIT_WAVES
#define GEMX_RINIT_WAVES 0  // (A/B builds: one figure for every system)
#endif
template <int SYS, int SOLVER, bool IL, int D, bool FULL, bool SLOW = false, bool RINIT = false> constexpr int pipe_waves_per_eu() {
    if (RINIT) return GEMX_RINIT_WAVES != 0 ? GEMX_RINIT_WAVES : ((SYS == GEMX_SYS_SCIM || SYS == GEMX_SYS_DFIM) ? 2 : 3);
    return (D <= 4 && !FULL && !SLOW && SYS != GEMX_SYS_DFIM && !(SOLVER == GEMX_SOLVER_DP5 && IL)) ? 4 : 1;
}
// SLOW (round 5; <4, 2> only): the instantiation for solver sub-steps (nsteps > 1) and CUSTOM constraint sets (constr_kind 2).  Every
// block of such a launch takes the rolled, run-time-checked copy of the step, which here also loops over the sub-steps and evaluates the
// weighted Limit / Squared constraint on an observation row it computes for that purpose.  A separate instantiation because that code
// inside the common kernels cost them registers (config 4's <4, 2>: 119 VGPRs -> 128 with 10 spills); with SLOW = false the kernel
// is what it was.
// RINIT (round 5; with FULL only): random initial states.  The FULL kernel without it serves the RC supply alone and carries no draw code
// at all (the prepared-draw code inside one shared kernel cost the RC-supply launches 15 % of their rate, same box).
template <int SYS, int CONV, int LOAD, int SOLVER, bool IL, class R, int D, int OW, bool FULL = false, bool SLOW = false, bool RINIT = false>
__global__ __launch_bounds__((1 + OW + pipe_loader_waves(D)) * BLOCK) __attribute__((amdgpu_waves_per_eu(pipe_waves_per_eu<SYS, SOLVER, IL, D, FULL, SLOW, RINIT>())))
void advance_pipe_kernel(const KArgs<R> a) {
    constexpr int LW = pipe_loader_waves(D);  // 1: a loader wave stages actions / references instead of the integrator
    constexpr int ND = SysTraits<SYS>::ND, NOUT = SysTraits<SYS>::NOUT, NACT = ConvTraits<CONV>::NACT;
    constexpr bool DISCRETE = ConvTraits<CONV>::DISCRETE;
    constexpr bool HAS_ANGLE = SysTraits<SYS>::HAS_ANGLE;
    using AngT = typename Angle<R>::T;
    using ST = Stepper<SYS, conv_base<CONV>(), LOAD, SOLVER, IL, R>;
    constexpr int NH = ST::NH;
    constexpr int NDONE = ND + (HAS_ANGLE ? 1 : 0) + NH;    // hand-off row: y, angle bits, ho, done[, supply voltage]
    constexpr int NHT = NDONE + 1 + (FULL ? 1 : 0);

    const DevParams<R> &P = a.P;
    // readfirstlane: the wave index is wave-uniform, but the compiler cannot know that of a value derived from threadIdx -- without
    // it every row address of the output waves (k0 = pb * D + ow * RPW) is 64-bit per-lane VALU arithmetic (v_mad_u64_u32 chains)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tid = threadIdx.x & (BLOCK - 1);
    // Workgroups take the 64-env groups in REVERSE order: the group that may be partial (the last one) goes to workgroup 0, which is
    // dispatched first.  A partial workgroup is 15-20 % slower than a full one (general I/O paths); as the last workgroup of the last round it
    // ran alone at the end of the launch (100000 envs: 0.67 of the roofline against 0.75 at 100032), in the first round it hides behind the rest.
    const int64_t blk0 = (int64_t)(gridDim.x - 1u - blockIdx.x) * BLOCK;
    const int64_t env = blk0 + tid;
    const int64_t N = a.N;
    const int K = a.K;
    // The LAST workgroup of a batch that is not a multiple of 64 envs is PARTIAL (round 5; such a batch used to fall back to the single-wave
    // kernel, 6-8 x slower): its lanes beyond the batch integrate a copy of env N - 1 (loads clamped, actions read as zeros) and store
    // nothing; its loader wave stages actions / references lane by lane instead of in 16-byte units that would reach past the row (and,
    // in the last row, past the tensor); its output waves flush through flush_rings (16-byte units over the rows' valid span + a scalar
    // tail).  All three tests are wave-uniform and sit outside the step loops; every full workgroup runs the code it ran before.
    const bool full_wg = blk0 + BLOCK <= N;
    const int rows_n = full_wg ? BLOCK : (int)(N - blk0);  // envs of this workgroup
    // A batch whose rows are NOT 16-byte aligned (N % 16 != 0 with one-byte actions / done bytes, n_envs * n_out % 4 != 0, a caller's
    // unaligned tensor: KArgs::coop / obs_vec are 0) keeps the 16-byte units in its full workgroups all the same: gfx950 under ROCm's
    // default alignment mode executes global_store_dwordx4, 16-byte stores at byte offsets and global_load_lds_dwordx4 from byte-misaligned
    // addresses correctly (tools/microbench_unaligned.hip, profiles/r05j_unaligned_access.txt) -- such a batch used to run the single-wave
    // fallback kernel at a sixth of the rate.  fast_io: this workgroup moves its rows in 16-byte units.
    const bool fast_io = full_wg;
    const bool valid = env < N;
    const int64_t envc = valid ? env : N - 1;  // clamped env index for loads
    constexpr int S = D;              // the observation ring holds exactly one hand-off block
    const int nb = (K + D - 1) / D;   // hand-off blocks

    // LDS: observation ring [S][64*NOUT] R | done ring [S][64] | hand-off [2][D][64][NHT] R
    R *ring = reinterpret_cast<R *>(gemx_smem);
    unsigned char *donebuf = gemx_smem + (size_t)S * BLOCK * NOUT * sizeof(R);
    R *hand = reinterpret_cast<R *>(donebuf + (size_t)S * BLOCK);
    constexpr int NACTC = conv_nact_c<CONV>();
    R *fifo = hand + 2 * (size_t)D * BLOCK * NHT;  // DeadTimeProcessor FIFO [delay][64][NACTC], touched by the integrator wave only
    // action staging buffer [2][D][64 * action bytes], filled by global -> LDS direct loads of the integrator wave
    unsigned char *actb = reinterpret_cast<unsigned char *>(fifo + (size_t)pipe_queue_rows(D, P.delay, conv_dq<CONV>() && P.dq_processor, FULL) * BLOCK * NACTC);
    // fused reward: reference rows [3][D][64 * n_ref] R, staged global -> LDS by the integrator wave one block ahead, read by the output
    // waves one block behind (hence three buffers).  The output waves thus issue NO global loads: a load would make them wait, through
    // the in-order vmcnt, for all their older observation stores once per block.
    constexpr int AHEAD = pipe_act_ahead(D), NBUF = pipe_act_bufs(D), NRBUF = pipe_ref_bufs(D);
    constexpr int ACTB_BYTES = NBUF * ((D + 3) / 4 * 4) * BLOCK * (DISCRETE ? 1 : NACT * (int)sizeof(R));
    R *refb = reinterpret_cast<R *>(actb + ACTB_BYTES);
    const int n_ref = a.rw != nullptr ? a.rh.n_ref : 0;
    // per-action voltage table [NACTIONS][8] R of steppers that have one (ST::NVT > 0): written by the integrator wave before its first block, read by it and
    // (COMPACT rows, below) by the output waves
    constexpr bool USE_TAB = ST::NVT > 0 && DISCRETE && !FULL;  // (FULL: the supply voltage may differ per lane; the table is built from the uniform one)
    R *vtab = refb + NRBUF * (size_t)D * BLOCK * n_ref;
    // PREPARED DRAWS (FULL, random initialisers; round 5).  Under random actions some lane of a wave terminates in most control steps (PMSM,
    // 2.3 % of the env-steps: 78 % of a wave's steps), and the fp64 draw of its fresh initial state (Philox blocks + the transforms: 1100-
    // 3500 cycles) sat, exec-masked, on the integrator's instruction stream nearly every step.  The draw is a pure function of (env,
    // reset count), so the LOADER wave computes every lane's next PREP_Q draws ahead of time into a per-lane queue: entry of count c in
    // slot c % PREP_Q = eight dwords [states.., angle bits, .., tag = c], the tag written last.  The integrator keeps the entry of
    // rcount + 1 in registers (read at every block start and after every consumption -- tag half first; LDS operations of a wave complete
    // in order and the loader rewrites a slot only after its count was consumed, so no torn entry is ever taken), takes it at a reset and
    // publishes the new count (`prep_cnt`); a lane that outruns its queue draws inline as before.  Both routes give the same bits.
    // [PREP_Q][64][8] dwords | cnt[64]
    // (pointers in the LDS address space, explicitly: address-space inference does not rewrite VOLATILE accesses, and as flat_load /
    // flat_store -- which count on vmcnt AND lgkmcnt -- these cost the RC-supply launches of the same kernel 15 % of their rate)
    typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
    typedef volatile __attribute__((address_space(3))) uint32_t lds_u32_t;
    typedef volatile __attribute__((address_space(3))) u4_t lds_u4_t;
    constexpr int PREP_Q = prep_q<SYS>();
    lds_u32_t *prep = (lds_u32_t *)reinterpret_cast<uint32_t *>(vtab + ((ST::NVT > 0 && DISCRETE) ? ConvTraits<CONV>::NACTIONS * 8 : 0));
    lds_u32_t *prep_cnt = prep + (size_t)PREP_Q * BLOCK * 8;
    static_assert(ND + 1 <= 7, "a prepared draw is eight dwords: states, angle, tag");
    // The initialiser's description in LDS (round 6; GEMX_PREP_DESC_COPY = 2).  Read through the global pointer, every field was loaded
    // again in every pass of the loader (nothing may stay in registers across the block's barrier and the volatile LDS traffic): a pass
    // that computes one generator block -- ~340 cycles of arithmetic -- took 1450, the finishing pass 3000-6600.  Copied into the loader's
    // REGISTERS (= 1) the passes are fast where a workgroup has its CU to itself, but the ~140 dwords live across the whole kernel cost
    // the heavier integrators their registers at four workgroups per CU (same box, 131072 envs: PMSM speed control 0.40 -> 0.30 of the
    // roofline, SCIM 0.30 -> 0.26; PMSM finite 0.69 -> 0.75; profiles/r06_rinit_probe.md).  In LDS it costs neither: the waves that draw
    // (integrator: inline draws; loader: prepared draws) each write the same 552 bytes before their first use -- no barrier needed, LDS
    // operations of one wave complete in order and both write identical values.
#ifndef GEMX_PREP_DESC_COPY  // 0: through the pointer (round 5), 1: the loader's registers, 2: LDS
#define GEMX_PREP_DESC_COPY 2
#endif
    InitDev *desc_lds = reinterpret_cast<InitDev *>(gemx_smem);  // (unused unless FULL && RINIT)
    if constexpr (FULL && RINIT) {
        desc_lds = reinterpret_cast<InitDev *>(reinterpret_cast<uint32_t *>(vtab + ((ST::NVT > 0 && DISCRETE) ? ConvTraits<CONV>::NACTIONS * 8 : 0)) +
                                              (size_t)(PREP_Q * 8 + 1) * BLOCK);
    }
    auto copy_desc_to_lds = [&]() {
        if constexpr (FULL && RINIT && GEMX_PREP_DESC_COPY == 2) {
            static_assert(sizeof(InitDev) % 4 == 0, "dword copy");
            const uint32_t *src = reinterpret_cast<const uint32_t *>(a.rinit);
            uint32_t *dst = reinterpret_cast<uint32_t *>(desc_lds);
            for (int i = tid; i < (int)(sizeof(InitDev) / 4); i += BLOCK) dst[i] = src[i];
        }
    };
    // COMPACT hand-off rows (synchronous machines behind a finite converter and a constant-speed load: the headline): the integrator's
    // time per step is dominated by its LDS instructions (~25 cycles of issue apiece against ~5 for a VALU instruction: six of them were
    // 150 of the step's 320 cycles), so the blocks that run on the voltage table and the one-step map hand over EIGHT values instead of
    // twelve -- [i_sd, i_sq, done << 8 | action, eps | sin, cos, u_sd, u_sq], two 16-byte writes instead of three -- and read only
    // u_alpha, u_beta of the action's table entry: omega is the launch constant init[0] in those blocks, and u_a, u_b, u_c are the
    // table entry of the action, which the output waves look up themselves.  The format of block b is published in the padding of table
    // entry 0 (vtab[6 + (b & 1)]) before the block's barrier; every other copy of the step keeps the full row.
    constexpr bool COMPACT_K = USE_TAB && SYS == GEMX_SYS_SYNC && linable<SYS, LOAD, SOLVER, IL, R>() && ST::NVT <= 5;
    auto steps_of = [&](int b) { return (K - b * D) < D ? (K - b * D) : D; };

    // Action staging: global memory -> LDS DIRECTLY (`global_load_lds_dword`: each lane's dword lands at M0 + 4 * lane, no VGPR
    // destination), one block ahead, double-buffered.  Staging through registers instead put up to D*NACT pending-load VGPRs
    // into the unrolled steps, and whenever the register allocator placed one of them next to an operand of a packed
    // instruction (v_pk_* read register PAIRS) the compiler had to insert `s_waitcnt vmcnt(0)` in the middle of the block
    // (measured 150 -> 164 us per 500-step launch).  The integrator reads its action of a step from LDS one step ahead.
    constexpr int ABYTES = DISCRETE ? 1 : NACT * (int)sizeof(R);
    constexpr int ROWB = BLOCK * ABYTES;  // bytes of one 64-env action row (contiguous in the [K][N][A] tensor)
    constexpr int DP = ROWB == 64 ? (D + 3) / 4 * 4 : D;  // rows per buffer half (uint8: padded to whole 256-byte groups of four rows)
    // one instruction moves 64 sixteen-byte units (`global_load_lds_dwordx4`, gfx950): the block's D rows of ROWB bytes are contiguous in LDS,
    // so unit q = 64 j + lane of instruction j belongs to row q / U at byte 16 (q % U).  (Dword-wide staging took ROWB / 256 instructions per
    // row -- 36 per block for three-phase duty cycles, each a 256-byte request with a row stride of N * ABYTES between them.)
    constexpr int U16 = ROWB / 16, NU16 = D * U16, NSTAGE = (NU16 + 63) / 64;
    auto stage_actions = [&](int b) {
        const int sb = steps_of(b);
        unsigned char *dst = actb + (size_t)(b % NBUF) * DP * ROWB;
        if (a.act_synth) {  // synthetic actions: this lane's own values of every row, generated (gemx_common.hpp: synth_u32); no memory read
            for (int s = 0; s < D; ++s) {
                const uint32_t t = a.act_step0 + (uint32_t)(b * D + (s < sb ? s : sb - 1));
                if constexpr (DISCRETE) {
                    dst[(size_t)s * ROWB + tid] = (unsigned char)synth_index(synth_u32(a.act_seed, a.act_env_base + envc, t, 0u), (uint32_t)ConvTraits<CONV>::NACTIONS);
                } else {
#pragma unroll
                    for (int i = 0; i < NACT; ++i) reinterpret_cast<R *>(dst + (size_t)s * ROWB)[tid * NACT + i] = (R)synth_unit(synth_u32(a.act_seed, a.act_env_base + envc, t, (uint32_t)i));
                }
            }
            return;
        }
        if constexpr (!DISCRETE && sizeof(R) == 4) {
            if (a.act_half) {  // NARROW action tensor (gemx_rollout_half, round 6): [K][N][A] halves, 2 A bytes per env-step from the HBM instead
                               // of 4 A.  Each lane loads its own env's values and widens them into the SAME fp32 rows the direct
                               // global -> LDS loads write for an fp32 tensor: everything downstream (integrator, delayed reads of the
                               // DeadTimeProcessor, dq stage) is untouched, and a half tensor gives the bits of its values fed as fp32.
                const _Float16 *srch = reinterpret_cast<const _Float16 *>(a.actions) + (int64_t)b * D * N * NACT;
                for (int s = 0; s < D; ++s) {
                    const int row = s < sb ? s : sb - 1;
                    const _Float16 *g = srch + ((int64_t)row * N + envc) * NACT;
#pragma unroll
                    for (int i = 0; i < NACT; ++i) {
                        const float v = (float)g[i];
                        reinterpret_cast<float *>(dst + (size_t)s * ROWB)[tid * NACT + i] = valid ? v : 0.0f;
                    }
                }
                return;
            }
        }
        const unsigned char *src = a.actions + ((int64_t)b * D * N + blk0) * ABYTES;
        if (!fast_io) {  // partial workgroup / unaligned rows: this lane's own action of every row through a register; lanes beyond the batch stage zeros
            for (int s = 0; s < D; ++s) {
                const int row = s < sb ? s : sb - 1;
                const unsigned char *g = src + ((int64_t)row * N + (envc - blk0)) * ABYTES;
                if constexpr (DISCRETE) {
                    const unsigned char v = *g;
                    dst[(size_t)s * ROWB + tid] = valid ? v : (unsigned char)0;
                } else {
#pragma unroll
                    for (int i = 0; i < NACT; ++i) {
                        const R v = reinterpret_cast<const R *>(g)[i];
                        reinterpret_cast<R *>(dst + (size_t)s * ROWB)[tid * NACT + i] = valid ? v : R(0);
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NSTAGE; ++j) {
            const int q = j * 64 + tid;
            int row = q / U16;
            const int col = q - row * U16;
            row = row < sb ? row : sb - 1;  // tail block: re-read the last valid row
            if (NU16 % 64 == 0 || q < NU16)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + (int64_t)row * N * ABYTES + col * 16),
                                                 (void __attribute__((address_space(3))) *)(dst + (size_t)j * 1024), 16, 0, 0);
        }
    };
    auto stage_refs = [&](int b) {  // rows of 64 * n_ref references, 64 consecutive dwords per instruction
        const int sb = steps_of(b);
        const int dwords = n_ref * (int)(sizeof(R) / 4);  // per env
        R *dst = refb + (size_t)(b % NRBUF) * D * BLOCK * n_ref;
        if (!full_wg) {  // partial workgroup: lane by lane (see stage_actions)
            for (int s = 0; s < D; ++s) {
                const int row = s < sb ? s : sb - 1;
                const R *g = a.refs + (((int64_t)b * D + row) * N + envc) * n_ref;
                for (int j = 0; j < n_ref; ++j) dst[((size_t)s * BLOCK + tid) * n_ref + j] = g[j];
            }
            return;
        }
        for (int s = 0; s < D; ++s) {
            const int row = s < sb ? s : sb - 1;
            const unsigned char *src = reinterpret_cast<const unsigned char *>(a.refs + (((int64_t)b * D + row) * N + blk0) * n_ref);
            for (int i = 0; i < dwords; ++i)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + i * 256 + tid * 4),
                                                 (void __attribute__((address_space(3))) *)(reinterpret_cast<unsigned char *>(dst + (size_t)s * BLOCK * n_ref) + i * 256),
                                                 4, 0, 0);
        }
    };
    if (wave == 0) {
        // ------------------------------------------------------------------ integrator
        // (the wave whose instruction stream is the launch time issues first wherever it shares a SIMD -- with one of the six output waves
        // of the deep shape, with other workgroups' waves in the shallow ones: headline 148.3 -> 144.3 us, SCIM 65536 envs 893 -> 878 us, same box)
        __builtin_amdgcn_s_setprio(3);
        R y[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j) y[j] = a.state[(int64_t)j * N + envc];
        AngT ang = AngT(0);
        if (HAS_ANGLE) ang = a.angle[envc];
        uint32_t sw = 0;
        const bool USE_SW = conv_has_legs<CONV>() && (IL || (FULL && P.rc_supply));
        if (USE_SW) {
            sw = a.sw[envc];
            if (conv_sw_bytes<CONV>() == 2) sw |= (uint32_t)a.sw[N + envc] << 8;
        }
        const AngT init_ang = Angle<R>::from_bits(P.init_angle_rep);
        // the reset values in VGPRs of their own, opaque to the optimiser: as kernel arguments (SGPRs) they cannot be the VGPR operand
        // v_cndmask wants, and the compiler re-creates them with a v_mov per select and step (three of the headline step's 44 instructions)
        R init_v[ND];
#pragma unroll
        for (int j = 0; j < ND; ++j) { init_v[j] = P.init[j]; asm volatile("" : "+v"(init_v[j])); }
        AngT init_ang_v = init_ang;
        asm volatile("" : "+v"(init_ang_v));
        // (The DeadTimeProcessor's reset action, P.dreset / P.dreset_d, is NOT given VGPRs of its own like the reset values above: four
        // registers that are live through every instantiation pushed the <4, 2> SCIM kernel from 125 to 129 VGPRs and the error-controlled
        // <2, 2> one from 130 to 134 -- across the 128-register line, three resident workgroups per CU instead of four, BASELINE config 4
        // with ScipyOdeSolver() at half its rate (profiles/r04i_bench.json against r04g).  The selects that read it sit in the
        // DeadTimeProcessor copies of the step only and pay a v_mov each there.)
        R sup[2] = {P.u_sup, R(0)};  // RCVoltageSupply: capacitor voltage, time since the supply's last update (FULL)
        uint32_t rcount = 0;         // random initialisers: resets of this env so far (FULL)
        if constexpr (FULL) {
            if (P.rc_supply) {
                sup[0] = a.state[(int64_t)ND * N + envc];
                sup[1] = a.state[(int64_t)(ND + 1) * N + envc];
            }
            if constexpr (RINIT) {
                copy_desc_to_lds();
                rcount = a.rcnt[envc];
                prep_cnt[tid] = rcount;
#pragma unroll
                for (int q = 0; q < PREP_Q; ++q) prep[((size_t)q * BLOCK + tid) * 8 + 7] = 0u;  // (counts start at 1: tag 0 = no entry)
            }
        }
        R hcar = R(0);  // error-controlled solver: the step size its controller proposed last (state row ND + 2), see dp5_adaptive
        if (SOLVER == GEMX_SOLVER_DP5 && P.adaptive) hcar = a.state[(int64_t)(ND + 2) * N + envc];
        constexpr bool LINABLE = linable<SYS, LOAD, SOLVER, IL, R>();
        const bool lin_ok = lin_usable<SYS, LOAD, SOLVER, IL, R>(P, y[0]);  // wave-uniform
        // DeadTimeProcessor, two representations of the same queue (both leave the same [delay][N] ring in HBM):
        //   FIFO:    an LDS ring per lane, swapped every step (the rolled, run-time-checked copy of the step);
        //   DELAYED: no queue at all -- the converter sees the action row staged `delay` steps EARLIER (rows before the block: `carry`,
        //            the last `delay` rows of the previous block / of the HBM ring), zeroed while fewer than `delay` steps have passed
        //            since the env's last reset (`since`).  It keeps the unrolled, table-driven blocks; deep shape only, and not
        //            behind a DqToAbcActionProcessor, whose transform (a function of the state at SUBMISSION time) precedes the queue.
        R linc[lin_regs<SYS, R, IL>()];
        lin_preload<SYS, R>(P, LINABLE && lin_ok, linc);
        constexpr bool CAN_DELAY = D == PIPE_D && !FULL;
        const uint32_t delay_u = (uint32_t)P.delay;
        // SLOW: solver sub-steps and custom constraint sets -- every block takes mode 1 (see the kernel's head)
        const bool ns1 = !SLOW || P.nsteps == 1;
        const bool generic_constr = SLOW && P.constr_kind == 2;
        constexpr bool slow_path = SLOW;  // (the launcher takes a SLOW instantiation exactly when nsteps != 1 or constr_kind == 2: its unrolled
                                          // copies of the step are never run and, behind this constant, never compiled)
        const bool delayed_any = CAN_DELAY && P.delay > 0 && (!LINABLE || lin_ok) && !slow_path;            // wave-uniform
        const bool delayed_t = delayed_any && conv_dq<CONV>() && P.dq_processor;  // queue of TRANSFORMED actions (row buffer, see one_step)
        const bool delayed = delayed_any && !delayed_t;                           // delayed read of the staged raw rows
        uint32_t since = delay_u;  // DELAYED: control steps since this env's last reset, saturating at `delay` (the HBM ring holds zeros already)
        const int ring_phase = fifo_phase_read(a);
        int slot = ring_phase;
        for (int d = 0; d < P.delay; ++d) {
            // FIFO: slot d of the ring; DELAYED: carry row d = the entry popped d steps from now = ring slot (phase + d) mod delay
            int src = d;
            if (delayed_any) { src = ring_phase + d; src = src >= P.delay ? src - P.delay : src; }
#pragma unroll
            for (int i = 0; i < NACTC; ++i) {
                const int64_t gi = ((int64_t)src * N + envc) * NACTC + i;
                fifo[((size_t)d * BLOCK + tid) * NACTC + i] = DISCRETE ? (R)a.ring[gi] : reinterpret_cast<const R *>(a.ring)[gi];
            }
        }
        R pop[NACTC];  // FIFO: the queue entry the next step pops, read one step ahead
#pragma unroll
        for (int i = 0; i < NACTC; ++i) pop[i] = (P.delay > 0 && !delayed_any) ? fifo[((size_t)slot * BLOCK + tid) * NACTC + i] : R(0);
        // DELAYED_T (mode 3): row r of the queue buffer holds the converter-side action of step r of the current block; the processor's
        // output of step s goes to row s + delay.  qpre = row s, read one step ahead; tprev = the previous step's output (queue one deep:
        // that IS row s, and its LDS round trip would sit on the chain)
        R qpre[NACTC], tprev[NACTC];
#pragma unroll
        for (int i = 0; i < NACTC; ++i) { qpre[i] = delayed_t ? fifo[(size_t)tid * NACTC + i] : R(0); tprev[i] = qpre[i]; }
        const bool check_default = SLOW ? P.constr_kind >= 1 : P.constr_kind == 1;  // (SLOW: the default constraint or a custom set -- the same thresholds)
        const bool auto_reset = P.auto_reset != 0;
        const R thr_done = check_default ? R(1) : R(INFINITY), thr_reset = (check_default && auto_reset) ? R(1) : R(INFINITY);
        uint32_t bad_action = 0;
        // prepared draws (FULL, random initialisers): the queue entry of rcount + 1, tag in pre_hi.w (see `prep`)
        u4_t pre_lo = {0u, 0u, 0u, 0u}, pre_hi = {0u, 0u, 0u, 0u};
#ifdef GEMX_TIMING
        uint32_t n_prepared = 0u, n_inline = 0u;  // resets served from the queue / drawn inline (debug words 500, 501: tools/probe_prepared_draws.py)
#endif
        auto prefetch_draw = [&]() {
            if constexpr (FULL && RINIT && GEMX_PREP_DRAWS != 0) {
                lds_u4_t *e = (lds_u4_t *)(prep + ((size_t)((rcount + 1u) % PREP_Q) * BLOCK + tid) * 8);
                pre_hi = e[1];  // (the tag's half first: a valid tag implies the other half was written before it)
                pre_lo = e[0];
            }
        };
        if constexpr (USE_TAB) {  // (written and read by this wave only: LDS operations of one wave complete in order)
            if (tid < ConvTraits<CONV>::NACTIONS) {
                R e[8];
                ST::action_entry(P, (uint32_t)tid, e);
#pragma unroll
                for (int j = 0; j < 8; ++j) vtab[tid * 8 + j] = e[j];
            }
        }

        auto read_action = [&](int b, int s, R (&dst)[NACT], uint32_t &ddst) {
            const unsigned char *row = actb + ((size_t)(b % NBUF) * DP + s) * ROWB;
            if (DISCRETE) ddst = row[tid];
            else {
#pragma unroll
                for (int i = 0; i < NACT; ++i) dst[i] = reinterpret_cast<const R *>(row)[tid * NACT + i];
            }
        };
        // `fifo_possible` (a std::bool_constant): false_type compiles the DeadTimeProcessor queue out, so that the fully
        // unrolled blocks of the common case stay ONE branch-free basic block; true_type keeps the wave-uniform run-time test
        // `mode` (a std::integral_constant): 0 compiles the DeadTimeProcessor queue out, so that the fully unrolled blocks of the common
        // case stay ONE branch-free basic block; 1 = the FIFO representation with its wave-uniform run-time tests; 2 = DELAYED (the
        // caller hands in the delayed, already masked action: only the `since` count is kept here)
        auto one_step = [&](auto mode, const R (&act_in)[NACT], uint32_t dact, R *row, const R *tab, R *qw = nullptr, const R *qn = nullptr) {
            constexpr int MODE = decltype(mode)::value;
            constexpr bool FIFO = MODE == 1;
            constexpr bool TAB = USE_TAB && !FIFO;  // the unrolled blocks take the action's table entry
            R act[MAX_ACT];
#pragma unroll
            for (int i = 0; i < MAX_ACT; ++i) act[i] = i < NACT ? act_in[i] : R(0);
            if (DISCRETE) {  // (with a loader wave the range check is ITS job: three instructions less in every step of this wave)
                if (LW == 0) bad_action |= dact >= (uint32_t)ConvTraits<CONV>::NACTIONS;
                dact &= (uint32_t)(ConvTraits<CONV>::NACTIONS - 1);
            }
            // action stage, exactly as in compute_block(): [DqToAbcActionProcessor [DeadTimeProcessor [system(control_space)]]]
            R qnext[NACTC];
            if constexpr (MODE == 3) {
#pragma unroll
                for (int i = 0; i < NACTC; ++i) qnext[i] = qn[i];  // row s + 1 (written at least one step ago unless the queue is one deep)
            }
            if (conv_dq<CONV>() && P.dq_processor) dq_action_stage<SYS, CONV, R>(P, y, ang, act);
            if constexpr (MODE == 3) {
                const bool one_deep = P.delay == 1, queued = since >= delay_u;
#pragma unroll
                for (int i = 0; i < NACTC; ++i) {
                    const R t = act[i];
                    qw[i] = t;                                   // row s + delay
                    const R q = one_deep ? tprev[i] : qpre[i];   // row s
                    tprev[i] = t;
                    act[i] = queued ? q : P.dreset[i];           // (the refilled reset action right after a reset)
                    qpre[i] = qnext[i];
                }
            }
            if (FIFO && P.delay > 0) {
                // the value popped in THIS step was read a step ago (`pop`); push the new one, then read the next step's pop -- the
                // slot after this one, or what was just pushed when the queue is one deep -- so that its LDS latency hides behind this step
                R *f = fifo + ((size_t)slot * BLOCK + tid) * NACTC;
                slot = slot + 1 == P.delay ? 0 : slot + 1;
                const R *fn = fifo + ((size_t)slot * BLOCK + tid) * NACTC;
                const bool one_deep = P.delay == 1;
                if (DISCRETE) {
                    const R pushed = (R)dact;
                    f[0] = pushed;
                    dact = (uint32_t)pop[0];
                    pop[0] = one_deep ? pushed : fn[0];
                } else {
#pragma unroll
                    for (int i = 0; i < NACTC; ++i) {
                        const R pushed = act[i];
                        f[i] = pushed;
                        act[i] = pop[i];
                        pop[i] = one_deep ? pushed : fn[i];
                    }
                }
            }
            if (conv_dq<CONV>() && !P.dq_processor) dq_action_stage<SYS, CONV, R>(P, y, ang, act);
            R ho[NH];
            // launcher guarantees solver_nsteps == 1; LINABLE instantiations take the one-step map whenever it is valid for this wave
            auto run_advance = [&](const DevParams<R> &Q) {
                if constexpr (TAB) {
                    ST::template advance<true, LINABLE, true>(Q, y, ang, sw, act, dact, ho, tab, linc, &hcar);
                } else {
                    if constexpr (FIFO && SLOW) {
                        if (!ns1) { ST::template advance<false, false>(Q, y, ang, sw, act, dact, ho, nullptr, nullptr, &hcar); return; }  // sub-steps
                    }
                    if (LINABLE && (!FIFO || lin_ok)) ST::template advance<true, LINABLE>(Q, y, ang, sw, act, dact, ho, nullptr, linc);
                    else ST::template advance<true, false>(Q, y, ang, sw, act, dact, ho, nullptr, nullptr, &hcar);
                }
            };
            // custom constraint set (mode 1): constraint_done()'s expressions on the row observe() gives for THIS lane's parameters
            R gviol = R(0);
            auto generic_violation = [&](const DevParams<R> &Q) {
                if constexpr (FIFO && SLOW) {
                    if (generic_constr) {
                        R ob[NOUT];
                        ST::observe(Q, y, ang, ho, ob);
                        R lim = R(0), sq = R(0);
#pragma unroll
                        for (int i = 0; i < NOUT; ++i) {
                            lim = fmax(lim, P.cw[i] * fabs(ob[i]));  // (wave-uniform addresses: scalar loads)
                            sq += (P.cw[GEMX_MAX_OUT + i] * ob[i]) * ob[i];
                        }
                        gviol = fmax(lim, sq);
                    }
                }
            };
            R usup_lane = P.u_sup;
            if constexpr (FULL) {
                // RCVoltageSupply, exactly as in compute_block(): one explicit Euler step of the supply's own state before the converter
                DevParams<R> PL = P;  // per-lane view of the parameters: only u_sup differs between lanes
                if (P.rc_supply) {
                    const R isup = supply_current<SYS, conv_base<CONV>(), R>(P, y, ang, sw, act);
                    sup[0] = sup[0] + (P.u_sup - sup[0] - P.sup_r * isup) * P.sup_inv_rc * sup[1];
                    sup[1] = P.tau;
                    PL.u_sup = sup[0];
                }
                usup_lane = PL.u_sup;
                run_advance(PL);
                generic_violation(PL);
                if (!IL && conv_has_legs<CONV>() && P.rc_supply) sw = ST::legs_of(dact);
            } else {
                run_advance(P);
                generic_violation(P);
            }
            // (the wave-uniform switches "default constraint on" / "auto-reset on" live in the two thresholds, not in scalar ANDs of the
            // compare mask: VALU compare -> SALU and -> VALU select sat twice on every step's dependency chain; probe: -3 % integrator cycles)
            R viol = ST::state_violation(P, y, ho);
            if constexpr (FIFO && SLOW) viol = generic_constr ? gviol : viol;
            const bool done = viol > thr_done;
            constexpr bool COMPACT_ROW = COMPACT_K && TAB;  // (the table-driven copies: modes 0 and 2)
            if constexpr (COMPACT_ROW) {
                // Stepper<SYNC>::row_slot: i_sd -> 0, i_sq -> 1, (omega -> 2), eps -> 3 | sin, cos -> 4, 5, u_sd, u_sq -> 6, 7
                const uint32_t pk = (done ? 0x100u : 0u) | dact;
                R pkf, bits;
                memcpy(&pkf, &pk, sizeof(R));
                memcpy(&bits, &ang, sizeof(R));
                row[0] = y[1]; row[1] = y[2]; row[2] = pkf; row[3] = bits;
                row[4] = ho[0]; row[5] = ho[1]; row[6] = ho[5]; row[7] = ho[6];
            } else {
#pragma unroll
                for (int j = 0; j < ND; ++j) row[ST::row_slot(j)] = y[j];
                if (HAS_ANGLE) {
                    R bits;
                    memcpy(&bits, &ang, sizeof(R));
                    row[ST::row_slot(ND)] = bits;
                }
#pragma unroll
                for (int j = 0; j < NH; ++j) row[ST::row_slot(ND + (HAS_ANGLE ? 1 : 0) + j)] = ho[j];
                row[ST::row_slot(NDONE)] = done ? R(1) : R(0);
                if constexpr (FULL) row[ST::row_slot(NDONE + 1)] = usup_lane;
            }
            const bool rs = viol > thr_reset;  // `if terminated: env.reset()`; switching state survives
            // every copy of the step but the FIFO one runs only while the one-step map is valid for this wave (`lin_ok`), i.e. while every
            // lane's omega IS init[0] and stays so: putting it back is a no-op there
            constexpr bool OMEGA_FIXED = LINABLE && MODE != 1;
#pragma unroll
            for (int j = OMEGA_FIXED ? 1 : 0; j < ND; ++j) y[j] = rs ? init_v[j] : y[j];
            ang = rs ? init_ang_v : ang;
            if constexpr (SOLVER == GEMX_SOLVER_DP5) hcar = rs ? R(0) : hcar;
            if constexpr (FULL) {
#if defined(GEMX_AB_NO_RINIT_BLOCK)  // timing-only A/B build: a reset restores the constant state (results are WRONG)
                if constexpr (false) if (rs) {
#else
                if constexpr (RINIT) if (rs) {  // (exec-masked, skipped wave-wide)
#endif
#if defined(GEMX_AB_NO_INLINE)  // timing-only A/B build: every reset takes whatever the registers hold (results are WRONG where the queue was empty)
                    if (true) {
#else
                    if (GEMX_PREP_DRAWS != 0 && GEMX_EXPECT_PREPARED(pre_hi.w == rcount + 1u)) {  // the loader wave's prepared draw of this count, in registers since the block's start
#endif
                        const uint32_t w8[8] = {pre_lo.x, pre_lo.y, pre_lo.z, pre_lo.w, pre_hi.x, pre_hi.y, pre_hi.z, pre_hi.w};
#pragma unroll
                        for (int j = 0; j < ND; ++j) memcpy(&y[j], &w8[j], sizeof(R));
                        if (HAS_ANGLE) memcpy(&ang, &w8[ND], sizeof(R));
                        rcount += 1u;
#ifdef GEMX_TIMING
                        n_prepared += 1u;
#endif
                    } else {
                        draw_initial_state_cnt<SYS, R>(GEMX_PREP_DESC_COPY == 2 ? desc_lds : a.rinit, envc, rcount, y, ang);
#ifdef GEMX_TIMING
                        n_inline += 1u;
#endif
                    }
                    if (GEMX_PREP_DRAWS != 0) prep_cnt[tid] = rcount;
                    prefetch_draw();  // the next count's entry (its latency ends long before the next step's reset test)
                }
                sup[0] = rs ? P.u_sup : sup[0];  // RCVoltageSupply.reset: the capacitor is loaded again, the supply's clock restarts
                sup[1] = rs ? R(0) : sup[1];
            }
            if constexpr (MODE == 2 || MODE == 3) since = rs ? 0u : (since < delay_u ? since + 1u : delay_u);
            if (FIFO && P.delay > 0) {  // DeadTimeProcessor.reset: the deque is refilled with the reset action
                if (rs) {  // (rare, exec-masked: the reset action is an SGPR operand of plain moves here -- as the VGPR operand of a select in
                           // every step it held NACTC registers live through the loop and took the shallow kernels across the 128-VGPR line)
                    for (int d = 0; d < P.delay; ++d) {
#pragma unroll
                        for (int i = 0; i < NACTC; ++i) fifo[((size_t)d * BLOCK + tid) * NACTC + i] = P.dreset[i];
                    }
#pragma unroll
                    for (int i = 0; i < NACTC; ++i) pop[i] = P.dreset[i];
                }
            }
        };
        stage_actions(0);
        if (n_ref > 0) stage_refs(0);
        if (LW != 0 && DISCRETE) {  // block 0 has landed and is visible to the loader wave, which validates the action indices
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
        }
#ifdef GEMX_TIMING
        unsigned long long tv = 0, tc = 0, tw = 0, tlong = 0, T0 = clock64(), W0 = wall_clock64(), t0, t0b, t1, t2;
#endif
        // Rate limit (round 4; a.pace_block_ticks, set by the launcher for large batches only): the write path of this chip delivers MORE when
        // it is offered slightly less than it can take -- the rollout's store pattern as a pure store kernel (tools/microbench_rowpitch.hip)
        // moves 5.2-5.4 TB/s at every batch size when the stores are issued as fast as they go, and 6.0-6.9 TB/s when every workgroup is
        // held to one row per interval (16384 envs 0.675 -> 0.869 of the 8 TB/s; 65536 envs, four workgroups per CU, 0.655 -> 0.795).  At 16384
        // envs the integrator's own instruction stream is that pacing (one workgroup per CU, a row per ~330 cycles); in the large batches
        // nothing paces the output waves, they run into the congested path and the whole launch sits at 0.60-0.70.  So the wave that drives
        // the block loop waits for the block's slot on the constant 100 MHz clock; one s_memrealtime per block.
        // Not in the deep shape with six output waves (PIPE_PACED; the launcher uses it for one workgroup per CU only while the limiter is on):
        // there the mere presence of the clock read and the wait loop in the block loop cost the UNPACED headline launch 5 % (same box, 16384
        // envs: 145 -> 153 us per 1000 steps, profiles/r04p_ab_old.txt).  <12, 3> keeps it: the DC machines run it at two workgroups per CU
        // (32768 envs, ShuntDc / ExtExDc finite 0.62 unpaced -> 0.83 paced; <4, 2> paced 0.73-0.76: profiles/r04p_shape_32768.txt).
        constexpr bool PIPE_PACED = !(D == PIPE_D && OW == PIPE_OUT_WAVES_RW);
        uint32_t pace = 0u;
        unsigned long long pace_t0 = 0ull;
        if constexpr (PIPE_PACED) {
            pace = blockIdx.x >= a.pace_tail_from ? a.pace_tail_ticks : a.pace_block_ticks;
            pace_t0 = pace != 0u ? wall_clock64() : 0ull;
        }
        for (int b = 0; b < nb; ++b) {
            if constexpr (PIPE_PACED) {
                if (pace != 0u) {
                    const unsigned long long due = pace_t0 + (unsigned long long)b * pace;
                    while (wall_clock64() < due) __builtin_amdgcn_s_sleep(4);
                }
            }
#ifdef GEMX_TIMING
            t0 = clock64();
#endif
            if constexpr (FULL) {
                if constexpr (RINIT) prefetch_draw();  // (what the loader wave has prepared since)
            }
            const int sb = steps_of(b);
            R *hb = hand + (size_t)(b & 1) * D * BLOCK * NHT + (size_t)tid * NHT;
            // block b's actions (and, the first time, the state) have landed: staged a whole block ago.  Only THEN issue the next
            // block's staging loads -- they go to the other half of the buffer, which nobody reads during this block.
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#ifdef GEMX_TIMING
            t0b = clock64();
#endif
            if (LW == 0 && b + 1 < nb) {
                stage_actions(b + 1);
                if (n_ref > 0) stage_refs(b + 1);
            }
            R an[NACT], ac[NACT] = {};
            uint32_t dn = 0, dc = 0;
#pragma unroll
            for (int i = 0; i < NACT; ++i) an[i] = R(0);
            // DELAYED: the row the converter sees at step s of this block was staged `delay` steps earlier
            auto read_delayed = [&](int s, R (&dst)[NACT], uint32_t &ddst) {
                if (s >= P.delay) {
                    read_action(b, s - P.delay, dst, ddst);
                } else {
                    const R *c = fifo + ((size_t)s * BLOCK + tid) * NACTC;
                    if (DISCRETE) ddst = (uint32_t)c[0];
                    else {
#pragma unroll
                        for (int i = 0; i < NACT; ++i) dst[i] = c[i];
                    }
                }
            };
            [[maybe_unused]] constexpr uint32_t AMASK = DISCRETE ? (uint32_t)(ConvTraits<CONV>::NACTIONS - 1) : 0u;
            [[maybe_unused]] auto fetch_entry = [&](uint32_t d, R (&e)[8]) {
                const R *src = vtab + (size_t)(d & AMASK) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = j < ST::NVT ? src[j] : R(0);
            };
            // full block, unrolled into branch-free basic blocks of FOUR steps (unrolling all twelve makes basic blocks of up to ~8000
            // instructions for the heavier systems, on which the instruction scheduler's compile time explodes; the run time is the same)
            // (`ns_tag`: the steps of the block -- D, or, deep shape, a tail block of 4 or 8 steps: 1000 steps = 83 blocks of 12 + 4, and through the
            // rolled run-time-checked step below those four cost 2.9 us of the headline's 143.6 where four steps of a whole block cost 0.56:
            // profiles/r04t_tail_pipe.txt)
            auto run_block = [&](auto delayed_tag, auto ns_tag) {
                constexpr bool DEL = decltype(delayed_tag)::value;
                constexpr int NS = decltype(ns_tag)::value;
                using Mode = std::integral_constant<int, DEL ? 2 : 0>;
                auto rd = [&](int s, R (&dst)[NACT], uint32_t &ddst) {
                    if constexpr (DEL) read_delayed(s, dst, ddst);
                    else read_action(b, s, dst, ddst);
                };
                rd(0, an, dn);
                if constexpr (USE_TAB) {
                    // two-deep software pipeline: step s runs on table entry ec; the entry of step s+1 and the action of step s+2 are
                    // in flight (each LDS read has a whole step to land)
                    R en[8], ec[8], e0[8];
                    uint32_t dnn = 0;
                    if constexpr (DEL) fetch_entry(P.dreset_d, e0);  // the reset action's entry, for the steps right after a reset
                    fetch_entry(dn, en);
                    rd(1, an, dnn);
#pragma unroll 4
                    for (int s = 0; s < NS; ++s) {
                        dc = dn;
#pragma unroll
                        for (int j = 0; j < 8; ++j) ec[j] = en[j];
                        dn = dnn;
                        fetch_entry(dn, en);
                        rd(s + 2 < NS ? s + 2 : NS - 1, an, dnn);
                        if constexpr (DEL) {
                            const bool queued = since >= delay_u;  // else: the refilled reset action
#pragma unroll
                            for (int j = 0; j < ST::NVT; ++j) ec[j] = queued ? ec[j] : e0[j];
                            if constexpr (COMPACT_K) dc = queued ? dc : P.dreset_d;  // (compact rows carry the action the converter saw)
                        }
                        one_step(Mode{}, ac, dc, hb + (size_t)s * BLOCK * NHT, ec);
                    }
                } else {
#pragma unroll 4
                    for (int s = 0; s < NS; ++s) {
                        dc = dn;
#pragma unroll
                        for (int i = 0; i < NACT; ++i) ac[i] = an[i];
                        rd(s + 1 < NS ? s + 1 : s, an, dn);  // one step ahead: its LDS latency hides behind this step
                        if constexpr (DEL) {
                            const bool queued = since >= delay_u;
                            dc = queued ? dc : P.dreset_d;
#pragma unroll
                            for (int i = 0; i < NACT; ++i) ac[i] = queued ? ac[i] : (i < NACTC ? P.dreset[i < NACTC ? i : 0] : R(0));
                        }
                        one_step(Mode{}, ac, dc, hb + (size_t)s * BLOCK * NHT, nullptr);
                    }
                }
            };
            bool compact_blk = false;  // this block's rows are COMPACT (see COMPACT_K)
            using WholeBlock = std::integral_constant<int, D>;
            // tail blocks of 4 / 8 steps through the unrolled code: deep shape, synchronous machines (the headline's family, where the rolled
            // tail was 1.6 % of a 1000-step launch; every system would gain its 1-1.5 %, at +24 % compile time for the library -- not taken)
            constexpr bool TAIL48 = D == PIPE_D && PIPE_D == 12 && SYS == GEMX_SYS_SYNC;
            if ((sb == D || (TAIL48 && (sb == 4 || sb == 8))) && P.delay == 0 && (!LINABLE || lin_ok) && !slow_path) {
                if (sb == D) run_block(std::false_type{}, WholeBlock{});
                else if constexpr (TAIL48) {
                    if (sb == 8) run_block(std::false_type{}, std::integral_constant<int, 8>{});
                    else run_block(std::false_type{}, std::integral_constant<int, 4>{});
                }
                compact_blk = COMPACT_K;
            } else if (CAN_DELAY && delayed) {
                compact_blk = COMPACT_K;
                if constexpr (CAN_DELAY) {
                    if (sb == D) {
                        run_block(std::true_type{}, WholeBlock{});
                    } else {  // tail block: the same steps, rolled
                        R ect[8] = {};
#pragma nounroll
                        for (int s = 0; s < sb; ++s) {
                            read_delayed(s, ac, dc);
                            const bool queued = since >= delay_u;
                            dc = queued ? dc : P.dreset_d;
#pragma unroll
                            for (int i = 0; i < NACT; ++i) ac[i] = queued ? ac[i] : (i < NACTC ? P.dreset[i < NACTC ? i : 0] : R(0));
                            if constexpr (USE_TAB) fetch_entry(dc, ect);
                            one_step(std::integral_constant<int, 2>{}, ac, dc, hb + (size_t)s * BLOCK * NHT, ect);
                        }
                    }
                    // carry <- the last `delay` rows submitted so far (ascending: entry j + sb is read before it is overwritten)
                    for (int j = 0; j < P.delay; ++j) {
                        const int idx = j + sb - P.delay;
                        R *c = fifo + ((size_t)j * BLOCK + tid) * NACTC;
                        R v[NACTC];
#pragma unroll
                        for (int i = 0; i < NACTC; ++i) v[i] = R(0);
                        if (idx >= 0) {
                            R t[NACT];
                            uint32_t td = 0;
#pragma unroll
                            for (int i = 0; i < NACT; ++i) t[i] = R(0);
                            read_action(b, idx, t, td);
                            if (DISCRETE) v[0] = (R)(td & AMASK);
                            else {
#pragma unroll
                                for (int i = 0; i < NACT; ++i) v[i] = t[i];
                            }
                        } else {
                            const R *o = fifo + ((size_t)(j + sb) * BLOCK + tid) * NACTC;
#pragma unroll
                            for (int i = 0; i < NACTC; ++i) v[i] = o[i];
                        }
#pragma unroll
                        for (int i = 0; i < NACTC; ++i) c[i] = v[i];
                    }
                }
            } else if (CAN_DELAY && delayed_t) {
                if constexpr (CAN_DELAY && conv_dq<CONV>()) {
                    auto qrow = [&](int r) { return fifo + ((size_t)r * BLOCK + tid) * NACTC; };
                    using Mode3 = std::integral_constant<int, 3>;
                    read_action(b, 0, an, dn);
                    if (sb == D) {
#pragma unroll 4
                        for (int s = 0; s < D; ++s) {
#pragma unroll
                            for (int i = 0; i < NACT; ++i) ac[i] = an[i];
                            read_action(b, s + 1 < D ? s + 1 : s, an, dn);
                            one_step(Mode3{}, ac, 0u, hb + (size_t)s * BLOCK * NHT, nullptr, qrow(s + P.delay), qrow(s + 1));
                        }
                    } else {
#pragma nounroll
                        for (int s = 0; s < sb; ++s) {
#pragma unroll
                            for (int i = 0; i < NACT; ++i) ac[i] = an[i];
                            read_action(b, s + 1 < sb ? s + 1 : s, an, dn);
                            one_step(Mode3{}, ac, 0u, hb + (size_t)s * BLOCK * NHT, nullptr, qrow(s + P.delay), qrow(s + 1));
                        }
                    }
                    // rows sb .. sb + delay - 1 (the pending entries) move to the front; qpre / tprev already hold row sb
                    for (int j = 0; j < P.delay; ++j) {
                        R v[NACTC];
#pragma unroll
                        for (int i = 0; i < NACTC; ++i) v[i] = qrow(sb + j)[i];
#pragma unroll
                        for (int i = 0; i < NACTC; ++i) qrow(j)[i] = v[i];
                    }
                }
            } else {  // tail block or DeadTimeProcessor FIFO: ONE rolled copy of the run-time-checked step
                // (the action of step s+1 is read from LDS before step s runs, as in the unrolled blocks)
                read_action(b, 0, an, dn);
#pragma nounroll
                for (int s = 0; s < sb; ++s) {
                    dc = dn;
#pragma unroll
                    for (int i = 0; i < NACT; ++i) ac[i] = an[i];
                    read_action(b, s + 1 < sb ? s + 1 : s, an, dn);
                    one_step(std::integral_constant<int, 1>{}, ac, dc, hb + (size_t)s * BLOCK * NHT, nullptr);
                }
            }
            if constexpr (COMPACT_K) {
                if (tid == 0) vtab[6 + (b & 1)] = compact_blk ? R(1) : R(0);
            }
#ifdef GEMX_TIMING
            t1 = clock64();
#endif
            __syncthreads();  // publishes hand-off block b; wave 1 is done reading block b-1 (other half)
#ifdef GEMX_TIMING
            t2 = clock64();
            tv += t0b - t0; tc += t1 - t0b; tw += t2 - t1;
            tlong += (t2 - t1) > 1000 ? 1 : 0;
            if (blockIdx.x == 0 && tid == 0 && b >= 100 && b < 148) {  // barrier trace of blocks 100..147 (arrive, release) per wave
                unsigned long long *tr = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64) + 64 + 0 * 96 + 2 * (b - 100);
                tr[0] = t1; tr[1] = t2;
            }
#endif
        }
#ifdef GEMX_TIMING
        if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 37)) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64) + (blockIdx.x ? 16 : 0);
            dbg[0] = tv; dbg[1] = tc; dbg[2] = tw; dbg[3] = clock64() - T0; dbg[4] = wall_clock64() - W0; dbg[5] = (unsigned long long)nb | (tlong << 32);
        }
        if (tid == 0 && blockIdx.x < 1024) {  // where the integrator wave of every workgroup ran: HW_ID[15:0] (wave, SIMD, pipe, CU, SH, SE) | XCC_ID << 16
            uint32_t hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(a.err) + 64 + 1024)[blockIdx.x] =
                (uint16_t)(((hw >> 4) & 3u) | (((hw >> 8) & 15u) << 2) | (((hw >> 12) & 1u) << 6) | (((hw >> 13) & 7u) << 7) | ((xcc & 15u) << 10));
        }
#endif
        if (valid) {  // (lanes of a partial workgroup beyond the batch store nothing)
#pragma unroll
            for (int j = 0; j < ND; ++j) a.state[(int64_t)j * N + env] = y[j];
            if (HAS_ANGLE) a.angle[env] = ang;
            if (USE_SW) {
                a.sw[env] = (uint8_t)sw;
                if (conv_sw_bytes<CONV>() == 2) a.sw[N + env] = (uint8_t)(sw >> 8);
            }
            if constexpr (FULL) {
                if (P.rc_supply) {
                    a.state[(int64_t)ND * N + env] = sup[0];
                    a.state[(int64_t)(ND + 1) * N + env] = sup[1];
                }
                if constexpr (RINIT) a.rcnt[env] = rcount;
#ifdef GEMX_TIMING
                if constexpr (RINIT) {
                    unsigned long long *dbgc = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64);
                    atomicAdd(&dbgc[500], (unsigned long long)n_prepared);
                    atomicAdd(&dbgc[501], (unsigned long long)n_inline);
                }
#endif
            }
            if (SOLVER == GEMX_SOLVER_DP5 && P.adaptive) a.state[(int64_t)(ND + 2) * N + env] = hcar;
            int phase_end = 0;
            if (P.delay > 0) phase_end = (ring_phase + K) % P.delay;
            for (int d = 0; d < P.delay; ++d) {
                // FIFO: slot d as it stands.  DELAYED: carry row d is the entry popped d steps after this launch -> ring slot (phase_end + d)
                // mod delay; entries submitted before the env's last reset are the refilled reset action
                int dst = d;
                bool keep = true;
                if (delayed_any) {
                    dst = phase_end + d;
                    dst = dst >= P.delay ? dst - P.delay : dst;
                    keep = (uint32_t)d + since >= delay_u;
                }
#pragma unroll
                for (int i = 0; i < NACTC; ++i) {
                    const int64_t gi = ((int64_t)dst * N + env) * NACTC + i;
                    const R v = keep ? fifo[((size_t)d * BLOCK + tid) * NACTC + i] : P.dreset[i];
                    if (DISCRETE) a.ring[gi] = (unsigned char)(uint32_t)v;
                    else reinterpret_cast<R *>(a.ring)[gi] = v;
                }
            }
        }
        if (bad_action) atomicOr(a.err, 1u);
        fifo_phase_advance(a, ring_phase, tid == 0);
    } else if (LW != 0 && wave == 1 + OW) {
        // ------------------------------------------------------------------ loader
        // The staging loads of block b+1 are issued and awaited HERE while the integrator works on block b.  Issued by the
        // integrator they queue, in the CU's vector-memory path, behind the output waves' observation stores, and the integrator
        // -- the one wave whose instruction stream sets the launch time at small N -- stalled at their issue for ~1300 of its
        // ~7600 cycles per 12-step block (s_memtime probe, 16384 envs; profiles/r01f_pipe_probe.md).
        // It also validates discrete actions (converters.py:204-206: an index outside the action space is an error): block b's rows
        // are in LDS -- block 0 behind the initial barrier, the others staged and awaited here one iteration earlier.
#ifdef GEMX_TIMING
        unsigned long long tl = 0, tb = 0, tph[6] = {0, 0, 0, 0, 0, 0}, nph[6] = {0, 0, 0, 0, 0, 0};
#endif
        uint32_t bad = 0;
        // prepared draws (FULL, random initialisers): the Philox blocks of the draw in progress, its count, the lanes it is for
        uint32_t pq0[4] = {0u, 0u, 0u, 0u}, pq1[4] = {0u, 0u, 0u, 0u}, pp0[4] = {0u, 0u, 0u, 0u}, pp1[4] = {0u, 0u, 0u, 0u}, prep_c = 0u;
        uint32_t lq0[4] = {0u, 0u, 0u, 0u}, prep_lastw = 0u;  // Philox block 0 of the draw this lane got last, and its count
        bool prep_act = false;
        int prep_phase = 0;  // wave-uniform
        // (the initialiser's description: see GEMX_PREP_DESC_COPY at the kernel's LDS layout)
        InitDev prep_desc;
        if constexpr (FULL && RINIT && GEMX_PREP_DRAWS != 0 && GEMX_PREP_DESC_COPY == 1) prep_desc = *a.rinit;
        copy_desc_to_lds();
        if (DISCRETE) __syncthreads();
        // (Staging TWO blocks ahead through a third buffer was tried in round 2 -- the s_memtime probe shows this wave's loads taking longer
        // than the integrator's block in the shallow shapes -- and changed nothing, same box, over all motor families:
        // profiles/r02h_loader_depth.md; and again in round 4 for action tensors too large for the Infinity Cache: PIPE_ACT_BUFS.)
        // ... which round 5 did for the shallow shapes after all (AHEAD = 2, see pipe_act_ahead): block b + 2 is issued, block b + 1 awaited
        if (AHEAD == 2 && 1 < nb) {
            stage_actions(1);
            if (n_ref > 0) stage_refs(1);
        }
        for (int b = 0; b < nb; ++b) {
#ifdef GEMX_TIMING
            const unsigned long long l0 = clock64();
#endif
            const bool issued = b + AHEAD < nb;
            if (issued) {
                stage_actions(b + AHEAD);
                if (n_ref > 0) stage_refs(b + AHEAD);
            }
            if constexpr (FULL && RINIT && GEMX_PREP_DRAWS != 0) {
#pragma unroll 1
                for (int prep_rep = 0; prep_rep < a.prep_phases; ++prep_rep) {  // (phases of the state machine below per pass: see launch_advance_t)
#ifdef GEMX_TIMING
                    const unsigned long long pt0 = clock64();
                    const int pt_phase = prep_phase;
#endif
                    // prepared draws (see `prep`): the next initial state of every lane whose entry was consumed -- ONE Philox block per pass
                    // (~900 cycles: forty quarter-rate multiplies), so that this wave still reaches the block's barrier before the
                    // integrator does; the whole draw in one pass (~2000-3500 cycles) made it the slowest wave of a block wherever the
                    // integrator runs on its one-step map (PMSM, 32768 envs: 41 G env-steps/s against 89 G without random initial states).
                    constexpr bool FLUX = SYS == GEMX_SYS_SCIM || SYS == GEMX_SYS_DFIM;
                    const InitDev *I = GEMX_PREP_DESC_COPY == 2 ? desc_lds : (GEMX_PREP_DESC_COPY == 1 ? &prep_desc : a.rinit);  // (see GEMX_PREP_DESC_COPY)
                    const bool blk1 = init_needs_block1(I), fprev = FLUX && I->flux_mode != 0;
                    if (prep_phase == 0) {  // scan: the first of the next PREP_Q counts whose slot does not hold it; lanes with a full queue sit out
                        const uint32_t c = prep_cnt[tid];
                        prep_act = false;
                        prep_c = c;
#pragma unroll
                        for (int q = PREP_Q; q >= 1; --q) {
                            const uint32_t w = c + (uint32_t)q;
                            if (prep[((size_t)(w % PREP_Q) * BLOCK + tid) * 8 + 7] != w) { prep_act = true; prep_c = w - 1u; }
                        }
                        if (__any(prep_act)) prep_phase = 1;
                    }
                    // the induction machines' previous draw (its stator currents bound this one's flux): filling a queue count by count, that
                    // is the draw this lane got last -- its block 0 was kept, and the draw takes three passes instead of four
                    auto after_own_blocks = [&]() {
                        if (!fprev) return 5;
                        if (I->flux_slot <= 4 && !__any(prep_act && prep_lastw != prep_c)) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) pp0[i] = lq0[i];
                            return 5;
                        }
                        return 3;
                    };
                    const uint64_t prep_genv = (uint64_t)(I->env_base + envc);  // the streams' env word is the GLOBAL index
                    if (prep_phase == 1) {
                        InitRng::block(I->seed, prep_genv, prep_c + 1u, 0u, pq0);
                        prep_phase = blk1 ? 2 : after_own_blocks();
                    } else if (prep_phase == 2) {
                        InitRng::block(I->seed, prep_genv, prep_c + 1u, 1u, pq1);
                        prep_phase = after_own_blocks();
                    } else if (prep_phase == 3) {
                        InitRng::block(I->seed, prep_genv, prep_c, 0u, pp0);
                        prep_phase = I->flux_slot > 4 ? 4 : 5;  // (the currents' uniforms, slots flux_slot - 2 and - 1, sit in block 0 for every machine built)
                    } else if (prep_phase == 4) {
                        InitRng::block(I->seed, prep_genv, prep_c, 1u, pp1);
                        prep_phase = 5;
                    } else if (prep_phase == 5) {
                        if (prep_act) {
                            double u[GEMX_MAX_ODE], v[GEMX_MAX_ODE];
                            init_uniforms_from(pq0, pq1, u);
                            init_draw_from<FLUX, GEMX_DRAW_NS * (ND + (HAS_ANGLE ? 1 : 0))>(I, prep_c + 1u, u, [&](double (&up)[GEMX_MAX_ODE]) { init_uniforms_from(pp0, pp1, up); }, v);
                            uint32_t w8[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
                            for (int j = 0; j < ND; ++j) { const R x = (R)v[j]; memcpy(&w8[j], &x, sizeof(R)); }
                            if (HAS_ANGLE) { const AngT ad = Angle<R>::from_rad(v[ND]); memcpy(&w8[ND], &ad, sizeof(R)); }
                            w8[7] = prep_c + 1u;
                            lds_u4_t *e = (lds_u4_t *)(prep + ((size_t)((prep_c + 1u) % PREP_Q) * BLOCK + tid) * 8);
                            e[0] = u4_t{w8[0], w8[1], w8[2], w8[3]};
                            e[1] = u4_t{w8[4], w8[5], w8[6], w8[7]};  // (the tag's half last)
                            prep_lastw = prep_c + 1u;
#pragma unroll
                            for (int i = 0; i < 4; ++i) lq0[i] = pq0[i];
                        }
                        prep_phase = 0;
                    }
#ifdef GEMX_TIMING
                    tph[pt_phase] += clock64() - pt0; nph[pt_phase] += 1ull;
#endif
                }
            }
            if (DISCRETE) {
                const unsigned char *rows = actb + (size_t)(b % NBUF) * DP * ROWB;
                const int sb = steps_of(b);
                for (int s = 0; s < sb; ++s) bad |= (uint32_t)rows[(size_t)s * ROWB + tid] >= (uint32_t)ConvTraits<CONV>::NACTIONS;  // (lanes beyond the batch: staged as 0)
            }
            // block b + 1 has landed in LDS before the barrier publishes it.  vmcnt retires in order: with block b + 2's NSTAGE staging
            // instructions just issued, "at most NSTAGE outstanding" is "block b + 1 complete" (fused reward: the reference rows' instruction
            // count is a run-time value, so those launches wait for everything)
            if (b + 1 < nb) {
                constexpr int VM_KEEP = 0x0F70 | (NSTAGE & 0xF) | ((NSTAGE >> 4) << 14);
                static_assert(NSTAGE < 64, "vmcnt immediate");
                if (AHEAD == 2 && issued && n_ref == 0 && fast_io && !a.act_half) __builtin_amdgcn_s_waitcnt(VM_KEEP);
                else __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            }
#ifdef GEMX_TIMING
            const unsigned long long l1 = clock64();
#endif
            __syncthreads();
#ifdef GEMX_TIMING
            tl += l1 - l0; tb += clock64() - l1;
            if (blockIdx.x == 0 && tid == 0 && b >= 100 && b < 148) {  // barrier trace of blocks 100..147 (arrive, release) per wave
                unsigned long long *tr = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64) + 64 + 3 * 96 + 2 * (b - 100);
                tr[0] = l1; tr[1] = clock64();
            }
#endif
        }
        if (bad) atomicOr(a.err, 1u);
#ifdef GEMX_TIMING
        if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 37)) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64) + (blockIdx.x ? 16 : 0);
            dbg[12] = tl; dbg[13] = tb;
            if (blockIdx.x == 0) for (int i = 0; i < 6; ++i) { dbg[484 + i] = tph[i]; dbg[490 + i] = nph[i]; }  // loader: cycles / passes per prepared-draw phase
        }
#endif
    } else {
        // ------------------------------------------------------------------ output + stores
        const bool aos = P.obs_layout == GEMX_OBS_AOS;
        auto one_row = [&](auto compact_tag, const R (&row)[NHT], int rs) {
            constexpr bool COMPACT = decltype(compact_tag)::value;
            R y[ND], ho[NH], obs[NOUT];
            AngT ang = AngT(0);
            R dn;
            if constexpr (COMPACT) {  // (see COMPACT_K: eight values; omega is the launch constant, u_abc the action's table entry)
                uint32_t pk;
                memcpy(&pk, &row[2], sizeof(R));
                memcpy(&ang, &row[3], sizeof(R));
                const R *e = vtab + (size_t)(pk & 0xFFu) * 8;
                y[0] = P.init[0]; y[1] = row[0]; y[2] = row[1];
                ho[0] = row[4]; ho[1] = row[5]; ho[2] = e[2]; ho[3] = e[3]; ho[4] = e[4]; ho[5] = row[6]; ho[6] = row[7];  // (entry: Stepper<SYNC>::action_entry)
                dn = (pk & 0x100u) ? R(1) : R(0);
            } else {
#pragma unroll
                for (int j = 0; j < ND; ++j) y[j] = row[ST::row_slot(j)];
                if (HAS_ANGLE) {
                    const R bits = row[ST::row_slot(ND)];
                    memcpy(&ang, &bits, sizeof(R));
                }
#pragma unroll
                for (int j = 0; j < NH; ++j) ho[j] = row[ST::row_slot(ND + (HAS_ANGLE ? 1 : 0) + j)];
                dn = row[ST::row_slot(NDONE)];
            }
            ST::observe(P, y, ang, ho, obs);
            if constexpr (FULL) obs[NOUT - 1] = row[ST::row_slot(NDONE + 1)] * P.inv_lim[NOUT - 1];  // u_sup column: this lane's supply voltage
            if (aos) {
#pragma unroll
                for (int j = 0; j < NOUT; ++j) ring[(rs * BLOCK + tid) * NOUT + j] = obs[j];
            } else {
#pragma unroll
                for (int j = 0; j < NOUT; ++j) ring[(rs * NOUT + j) * BLOCK + tid] = obs[j];
            }
            donebuf[rs * BLOCK + tid] = dn != R(0) ? 1 : 0;
        };
        auto load_row = [&](auto compact_tag, const R *src, R (&row)[NHT]) {
#pragma unroll
            for (int j = 0; j < (decltype(compact_tag)::value ? 8 : NHT); ++j) row[j] = src[j];
        };
        // output wave `ow` of OW owns rows [ow*RPW, (ow+1)*RPW) of every hand-off block: it turns them into
        // observation rows in ITS part of the ring and flushes them itself -- the output waves never synchronise with
        // each other, only with the integrator at the block barrier.
        static_assert(D % OW == 0, "rows per output wave");
        constexpr int RPW = D / OW;
        const int ow = wave - 1;
        const int r0 = ow * RPW;
        auto rows_of = [&](int pb) {  // rows of this wave in block pb (may be <= 0 in the tail block)
            const int sb = steps_of(pb);
            return sb - r0 < RPW ? (sb - r0 < 0 ? 0 : sb - r0) : RPW;
        };

#ifdef GEMX_TIMING
        unsigned long long tflush = 0;
#endif
        auto process = [&](int pb) {
            const int nr = rows_of(pb);
            if (nr <= 0) return;
            const R *hb = hand + (size_t)(pb & 1) * D * BLOCK * NHT + (size_t)r0 * BLOCK * NHT + (size_t)tid * NHT;
            R rows[2][NHT];
            auto convert_rows = [&](auto compact_tag) {
                load_row(compact_tag, hb, rows[0]);
                if (nr == RPW) {
#pragma unroll
                    for (int s = 0; s < RPW; ++s) {  // the next row's LDS reads are issued BEFORE this row's ring writes
                        if (s + 1 < RPW) load_row(compact_tag, hb + (size_t)(s + 1) * BLOCK * NHT, rows[(s + 1) & 1]);
                        one_row(compact_tag, rows[s & 1], r0 + s);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < RPW; ++s) {
                        if (s < nr) {
                            if (s + 1 < nr) load_row(compact_tag, hb + (size_t)(s + 1) * BLOCK * NHT, rows[(s + 1) & 1]);
                            one_row(compact_tag, rows[s & 1], r0 + s);
                        }
                    }
                }
            };
            if constexpr (COMPACT_K) {
                // (the block's format, published by the integrator before the barrier that handed the block over)
                if (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, vtab[6 + (pb & 1)])) != 0) convert_rows(std::true_type{});
                else convert_rows(std::false_type{});
            } else {
                convert_rows(std::false_type{});
            }
#ifdef GEMX_TIMING
            const unsigned long long f0 = clock64();
#endif
            if (a.rw != nullptr) {
                // this block's references are in LDS (staged by the integrator wave); the description is (re)read from the scalar
                // cache once per block, so that its ~25 SGPRs are live only here
                RewardRegs<R> WR;
                WR.load(a.rh);
                R rv[RPW][GEMX_MAX_REF];
                const R *rb = refb + ((size_t)(pb % NRBUF) * D + r0) * BLOCK * n_ref + (size_t)tid * n_ref;
                // unconditional loads from clamped addresses + selects: a conditional load compiles to one scalar branch per element
                auto fetch_refs = [&](int nrr) {
#pragma unroll
                    for (int s = 0; s < RPW; ++s) {
#pragma unroll
                        for (int j = 0; j < GEMX_MAX_REF; ++j) {
                            const R v = rb[(size_t)(s < nrr ? s : 0) * BLOCK * n_ref + (j < n_ref ? j : 0)];
                            rv[s][j] = (s < nrr && j < n_ref) ? v : R(0);
                        }
                    }
                };
                if (nr == RPW) {  // full block: the row count is a compile-time constant in this copy
                    fetch_refs(RPW);
                    reward_apply<NOUT, RPW, R>(a, WR, ring, donebuf, pb * D + r0, r0, RPW, tid, env, valid, rv);
                } else {
                    fetch_refs(nr);
                    reward_apply<NOUT, RPW, R>(a, WR, ring, donebuf, pb * D + r0, r0, nr, tid, env, valid, rv);
                }
            }
            if (aos && nr == RPW && fast_io) flush_rows_pipe<NOUT, RPW, R>(a, ring + (size_t)r0 * BLOCK * NOUT, donebuf + (size_t)r0 * BLOCK, pb * D + r0, tid, blk0);
            else flush_rings<NOUT, R>(a, ring + (size_t)r0 * BLOCK * NOUT, donebuf + (size_t)r0 * BLOCK, pb * D + r0, nr, tid, blk0, rows_n, full_wg,
                                      valid, env);
#ifdef GEMX_TIMING
            tflush += clock64() - f0;
#endif
        };
#ifdef GEMX_TIMING
        unsigned long long tp = 0, tq = 0, pmax = 0, pg2 = 0, pg3 = 0;
#endif
        if (LW != 0 && DISCRETE) __syncthreads();  // (the integrator's and the loader's initial barrier)
        for (int b = 0; b < nb; ++b) {
#ifdef GEMX_TIMING
            const unsigned long long q0 = clock64();
#endif
            if (b >= 1) process(b - 1);
#ifdef GEMX_TIMING
            const unsigned long long q1 = clock64();
#endif
            __syncthreads();
#ifdef GEMX_TIMING
            tp += q1 - q0; tq += clock64() - q1;
            pmax = (q1 - q0) > pmax ? (q1 - q0) : pmax; pg2 += (q1 - q0) > 2000; pg3 += (q1 - q0) > 3000;
            if (blockIdx.x == 0 && tid == 0 && b >= 100 && b < 148) {  // barrier trace of blocks 100..147 (arrive, release) per wave
                unsigned long long *tr = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64) + 64 + (wave < 3 ? wave : 2) * 96 + 2 * (b - 100);
                tr[0] = q1; tr[1] = clock64();
            }
#endif
        }
        process(nb - 1);
#ifdef GEMX_TIMING
        if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 37)) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64) + (blockIdx.x ? 16 : 0);
            dbg[6 + 2 * (wave - 1)] = tp; dbg[7 + 2 * (wave - 1)] = tq;
            if (wave == 1) { dbg[14] = tflush; dbg[15] = (pmax << 40) | (pg3 << 20) | pg2; }
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------------
// dc_stream_kernel: small batches of the DC machines behind a constant-speed load (BASELINE config 2: Cont-CC-PermExDc-v0, 4096 envs).
//
// A launch takes at least K times what ONE wave needs per control step, and in advance_pipe_kernel that wave carries the
// converter stage, the solver, the constraint, the reset and a 16-byte hand-off write: 150 cycles per step for a one-state motor.  Here the
// step's recurrence is all the integrator wave keeps.  With omega constant and no dead time the electrical right-hand side is
// f(x) = A x + g(u_k), and g(u_k) does not depend on the state, so
//   * PRE waves (alternating groups of four steps) load the action rows DCS_PREFETCH blocks ahead straight into registers -- through a
//     buffer descriptor, one instruction per row --, run the converter stage and Elec::prep, and leave the step's input term
//     (elec_input(): g, or S g of the one-step map) and the voltages in LDS;
//   * the INTEGRATOR wave (wave 0, raised priority, alone on its SIMD) reads the input terms of four steps with one 16-byte LDS read two
//     groups ahead, applies the solver (elec_apply(): a 1-state Euler step is one FMA), applies the reset, and hands over NOTHING but the
//     new motor states, four steps per 16-byte write.  The reset is a multiplication when the initial state is zero (one_step(), ZERO):
//     a VALU-written mask may not be read by the next VALU instruction on gfx950, so compare -> select costs 15.6 cycles where
//     FMA -> clamp -> multiply costs 7.8 (tools/microbench_chain.hip, profiles/r03d_microbench_chain.txt).  A lone wave issues one
//     instruction per ~5 cycles whatever its kind, so the step costs what its instruction COUNT is: 3 VALU + 1/4 LDS read + 1/4 LDS write;
//   * OUTPUT waves one block behind rebuild the rest -- observation row, done flag: the same device functions on the same values as
//     everywhere else.  The rows of a group of four steps go through a per-wave LDS staging buffer and leave as FULL, 16-byte ALIGNED
//     16-byte stores (4 rows x 64 envs x NOUT dwords = 64 NOUT chunks = NOUT instructions, through a buffer descriptor whose base is
//     advanced per block): stored straight from the lanes' registers a row is 64 pieces of 20-28 bytes that straddle 16-byte boundaries,
//     ~50 cycles per row in the CU's store path against 20 for aligned quads.  The constant columns of a DC machine's row (omega, u_sup)
//     are put into the staging rows once per launch.  The done bytes of four steps go through a byte staging area and leave in ONE store.
// LDS rows are [group of 4 steps][lane][step in group][value]: 16 bytes per lane and value.  One s_barrier per block of D = 32 steps
// (64 with 32 envs per workgroup -- the form the launcher takes up to 4096 envs, see the note at the kernel: the bound named next is
// what that form removes).
// What bounds the 64-env form (profiles/r03h_dcs_probe.txt): the output waves (1200-1600 cycles of a 1700-cycle block period) -- a block
// moves ~130 KB through the LDS (~1000 cycles at 128 B/clk), and, decisively, the CU issues one 1-KiB wave store per ~33 cycles: 48 of them
// per 32 steps.  With 32 envs per workgroup every CU has half of them and the integrator (34 cycles per step) is the bound again.
// Every value is produced by the code the other kernels run (prep / rk_step / observe / state_violation), so the results are
// bit-identical to theirs; the tests assert it.  Preconditions beyond the pipelined kernel's (checked by the launcher): DC machine,
// ConstantSpeedLoad, no dead time of either kind, ideal supply, constant initial state, no fused reward, AoS observations, and
// omega == init[0] in every env (gemx_set_state clears that until the next full reset; the launcher never takes this kernel while its
// stream is being captured into a graph, and the integrator wave CHECKS the omega row: a moved omega raises GEMX_ERRFLAG_OMEGA_MOVED).
// ------------------------------------------------------------------------------------------------
// Waves of a workgroup go to the CU's four SIMDs round robin, so the waves whose index is a multiple of four would share the integrator's
// SIMD and its issue slots: waves 4, 8, 12 stay resident but do nothing except meet the others at every barrier (a parked wave costs its
// SIMD's other wave ~2 %; round 2's wave 4 ENDED at once, relying on "an ended wave no longer counts at the barrier" -- true on this
// hardware, but not something the programming model promises; GEMX_DCS_WAVE4_EXITS keeps that as an A/B).
// Roles per system: the one-state machines (PermEx, Series) run <D = 32, 4 pre waves, 8 output waves> = 16 waves, the two-state ones
// (Shunt, ExtEx: twice the LDS per step) <32, 2, 4> = 8 waves.  Same-box A/B at 4096 envs, PermExDc, Euler, us per 1000 steps
// (profiles/r03e_dcs_ab.md, r03h_dcs_ab.md): <64, 2, 4> 32.4, <64, 4, 4> 30.2, <32, 2, 8> 30.2, <32, 4, 8> 29.3 -> 28.45 with the running
// descriptor bases, <64, 4, 8> 27.7 (but 33.5 against 31.2 at 8192 envs: twice the unrolled code, and two CUs share an instruction
// cache once more than half of them are busy); round 2's kernel: 34.4.
// GEMX_DCS_D1 / GEMX_DCS_PRE / GEMX_DCS_OUT override the one-state choice (A/B builds, tools/dev_build.py).
#ifndef GEMX_DCS_D1
#define GEMX_DCS_D1 32
#endif
#ifndef GEMX_DCS_PRE
#define GEMX_DCS_PRE 4
#endif
#ifndef GEMX_DCS_OUT
#define GEMX_DCS_OUT 8
#endif
constexpr int DCS_PREFETCH = 3;
template <int SYS> constexpr int dcs_depth() { return SysTraits<SYS>::ND == 2 ? GEMX_DCS_D1 : 32; }  // ND == 2: one motor state, one voltage
template <int SYS> constexpr int dcs_pre() { return SysTraits<SYS>::ND == 2 ? GEMX_DCS_PRE : 2; }
template <int SYS> constexpr int dcs_out() { return SysTraits<SYS>::ND == 2 ? GEMX_DCS_OUT : 4; }
// worker waves = every wave whose index is not a multiple of four: wave w is worker w - 1 - w / 4 (waves 1, 2, 3, 5, 6, 7, 9, ...); the
// first dcs_pre() workers are pre waves, the next dcs_out() output waves
constexpr int dcs_worker_index(int wave) { return wave - 1 - (wave >> 2); }
template <int SYS> constexpr int dcs_waves() {  // smallest workgroup whose worker waves cover the roles
    int w = 2;
    while (dcs_worker_index(w - 1) + 1 < dcs_pre<SYS>() + dcs_out<SYS>() || ((w - 1) & 3) == 0) ++w;
    return w;
}
template <int SYS, int CONV> constexpr size_t dcs_smem_bytes() {
    constexpr int NM = SysTraits<SYS>::ND - 1, NU = SYS == GEMX_SYS_DC_EXTEX ? 2 : 1;
    return (size_t)dcs_depth<SYS>() * BLOCK * sizeof(float) * (2 * NM + 3 * NU + 2 * NM) +
           (size_t)dcs_out<SYS>() * 4 * BLOCK * SysTraits<SYS>::NOUT * sizeof(float) + (size_t)dcs_out<SYS>() * 4 * BLOCK;
}
// EPW = envs per workgroup.  64: a lane is an env.  32 (what the launcher takes: twice the workgroups, i.e. twice the CUs, for the same
// batch): the integrator's lanes 32..63 mirror lanes 0..31, while the pre and output waves -- whose work is not a recurrence -- give their
// upper half lanes the NEXT group of four steps of the same envs.  What bounds the 64-env form at 4096 envs is a CU's store path: ~33 cycles
// per 1-KiB wave store, 48 of them per 32 steps = the 1600 of its 1700-cycle block period (r03h probe); with 32 envs per workgroup a
// store instruction is still full (8 rows x 32 envs x NOUT dwords) but every CU has half of them to issue.  LDS rows stay [group][lane]: lane
// l of double group G holds steps 4 (2 G + l / 32) .. + 3 of env l % 32, so every index below is the 64-env one with "group" read as
// "double group" -- except the integrator's, which walks the groups in step order.
template <int SYS, int CONV, int SOLVER, class R, int EPW = BLOCK>
__global__ __launch_bounds__(dcs_waves<SYS>() * BLOCK) void dc_stream_kernel(const KArgs<R> a) {
    static_assert(EPW == BLOCK || EPW == BLOCK / 2, "envs per workgroup");
    constexpr int HALVES = BLOCK / EPW;
    constexpr int DCS_PRE = dcs_pre<SYS>(), DCS_OUT = dcs_out<SYS>(), DCS_WAVES = dcs_waves<SYS>();
    static_assert(DCS_WAVES <= 16 && dcs_worker_index(DCS_WAVES - 1) == DCS_PRE + DCS_OUT - 1, "wave roles");
    constexpr int ND = SysTraits<SYS>::ND, NOUT = SysTraits<SYS>::NOUT, NM = ND - 1, NACT = ConvTraits<CONV>::NACT;
    constexpr bool DISCRETE = ConvTraits<CONV>::DISCRETE;
    constexpr int D = dcs_depth<SYS>() * HALVES, NGR = D / 4, NG2 = NGR / HALVES;  // steps, groups of four steps, (double) groups = LDS rows per block
    using ST = Stepper<SYS, CONV, GEMX_LOAD_CONST_SPEED, SOLVER, false, R>;
    using AngT = typename Angle<R>::T;
    constexpr int NU = ST::NU;
    constexpr bool LINABLE = linable<SYS, GEMX_LOAD_CONST_SPEED, SOLVER, false, R>();
    static_assert(sizeof(R) == 4 && !SysTraits<SYS>::HAS_ANGLE && D % 4 == 0 && NM <= 2, "fp32 DC machines");

    const DevParams<R> &P = a.P;
#ifdef GEMX_TIMING
    const unsigned long long WE = wall_clock64();  // kernel entry of this wave, 100 MHz ticks: where the launch's FIXED part goes (dbg[70..77])
#endif
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tid = threadIdx.x & (BLOCK - 1);
    // (32 envs per workgroup on twice as many CUs was tried for the output waves' sake: their time per row does not go down with the
    // active lanes of a store -- 1650 cycles per 32-step block either way -- and the exec masking around the stores cost 20 %.)
    const int le = tid & (EPW - 1), hf = HALVES == 2 ? tid >> 5 : 0;  // env lane, half (which group of a double group)
    const int64_t N = a.N, blk0 = (int64_t)blockIdx.x * EPW, env = blk0 + le;
    const int K = a.K, nb = (K + D - 1) / D;
    auto steps_of = [&](int b) __attribute__((always_inline)) { return (K - b * D) < D ? (K - b * D) : D; };
    // LDS: input terms [2][NGR][64][4][NM] | voltages [3][NGR][64][4][NU] | new motor states [2][NGR][64][4][NM] | row staging [DCS_OUT][4][64][NOUT]
    // (in iteration b the integrator reads the input terms of block b while the pre waves write block b + 1's: two buffers; the voltages of
    // block b - 1 are still being read by the output waves then: three)
    R *gin = reinterpret_cast<R *>(gemx_smem);
    R *uu = gin + 2 * (size_t)D * EPW * NM;
    R *hand = uu + 3 * (size_t)D * EPW * NU;
    R *stage = hand + 2 * (size_t)D * EPW * NM;
    const R om = P.init[0];  // == omega of every env (launcher); a ConstantSpeedLoad never changes it, a reset puts it back
    const bool lin_ok = LINABLE && P.lin_on != 0;  // wave-uniform
    R linc[lin_regs<SYS, R, false>()];
    lin_preload<SYS, R>(P, lin_ok, linc);
    const bool check_default = P.constr_kind == 1;
    const R thr_done = check_default ? R(1) : R(INFINITY);

    if (wave == 0) {
        // ------------------------------------------------------------------ integrator: the recurrence and nothing else
        __builtin_amdgcn_s_setprio(3);
        R y[NM];  // the state the last step returned (BEFORE its reset select)
#pragma unroll
        for (int i = 0; i < NM; ++i) y[i] = a.state[(int64_t)(1 + i) * N + env];
        const R om_lane = a.state[env];
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) here, once: otherwise the compiler parks the wait for x inside the step loop
        // the kernel's premise, checked where it is used: a launch replayed from a graph captured before a gemx_set_state, or enqueued on
        // another stream than the state change, would otherwise integrate the wrong machine in silence
        if (!__all(om_lane == om) && tid == 0) atomicOr(a.err, (uint32_t)GEMX_ERRFLAG_OMEGA_MOVED);
        const bool resets = check_default && P.auto_reset != 0;
        R thr[NM];  // |i_c| >= thr[c]  <=>  this step's state violates the default constraint (DevParams::dc_thr)
#pragma unroll
        for (int i = 0; i < NM; ++i) thr[i] = resets ? P.dc_thr[i] : R(INFINITY);
        // The reset, `x <- init where some |y_c| >= thr_c`, WITHOUT a compare: on gfx950 a VALU-written mask (VCC or any SGPR pair) may
        // not be read by the next VALU instruction -- the compiler pads v_cmp -> v_cndmask with `s_nop 1`, and the pair costs 15.6 cycles
        // on top of the 6.3 of the step's FMA (tools/microbench_chain.hip: 21.9 cycles per step; 13.9 without the nop, which the hazard
        // rules do not allow).  With a ZERO initial state -- every DC env's default -- the select is a multiplication:
        //     keep_c = sat((thr_c - |y_c|) 2^100)  in {0, 1} exactly (the FMA's single rounding keeps the sign of thr_c - |y_c|,
        //              zero only for equality; 2^100 lifts the smallest non-zero difference far above 1),
        //     x <- (prod_c keep_c) y          = y, or (+-)0 = init
        // three dependent VALU instructions for the one-state machines, 14.1 cycles per step.  Other initial states keep the select.
        R tbig[NM];
        constexpr R BIG = R(1.2676506002282294e30);  // 2^100
#pragma unroll
        for (int i = 0; i < NM; ++i) tbig[i] = thr[i] * BIG;
        bool zero_init = true;
#pragma unroll
        for (int i = 0; i < NM; ++i) zero_init &= P.init[1 + i] == R(0);
        R x[NM];  // the state the next step starts from (the last step's result with its reset applied)
#pragma unroll
        for (int i = 0; i < NM; ++i) x[i] = y[i];
        auto one_step = [&](auto lin_tag, auto zero_tag, const R *in_, R *out_) __attribute__((always_inline)) {  // (tags: std::bool_constant -- no branch in the step)
            constexpr bool LIN = decltype(lin_tag)::value, ZERO = decltype(zero_tag)::value;
            R in[NM], xa[NM];
#pragma unroll
            for (int i = 0; i < NM; ++i) { in[i] = in_[i]; xa[i] = x[i]; }
            elec_apply<SYS, SOLVER, R, LIN>(P, om, xa, in, LIN ? linc : nullptr);
#pragma unroll
            for (int i = 0; i < NM; ++i) out_[i] = xa[i];
            if constexpr (ZERO) {
                R keep = clip01(fma(-fabs(xa[0]), BIG, tbig[0]));
#pragma unroll
                for (int i = 1; i < NM; ++i) keep *= clip01(fma(-fabs(xa[i]), BIG, tbig[i]));
#pragma unroll
                for (int i = 0; i < NM; ++i) x[i] = keep * xa[i];
            } else {
                bool rs = fabs(xa[0]) >= thr[0];
#pragma unroll
                for (int i = 1; i < NM; ++i) rs |= fabs(xa[i]) >= thr[i];
#pragma unroll
                for (int i = 0; i < NM; ++i) x[i] = rs ? P.init[1 + i] : xa[i];
            }
        };
        auto run_block = [&](auto lin_tag, auto zero_tag, int b) __attribute__((always_inline)) {
            const int sb = steps_of(b);
            // (group g of the block = half g % HALVES of double group g / HALVES)
            const R *gb = gin + ((size_t)(b & 1) * NG2 * BLOCK + le) * 4 * NM;
            R *hb = hand + ((size_t)(b & 1) * NG2 * BLOCK + le) * 4 * NM;
            auto goff = [](int g) __attribute__((always_inline)) { return ((size_t)(g / HALVES) * BLOCK + (size_t)(g % HALVES) * EPW) * 4 * NM; };
            // groups of four steps: a group's input terms are read two groups ahead (a step is far shorter than an LDS round trip).
            // run_groups<NG>(g0): NG consecutive groups from group g0 (even, so that with 32 envs per workgroup g0 is a whole number of
            // double groups and the offsets inside stay compile-time constants), unrolled, no branch.
            auto run_groups = [&](auto ng_tag, int g0) __attribute__((always_inline)) {
                constexpr int NG = decltype(ng_tag)::value;
                const size_t base = goff(g0);
                R in4[3][4 * NM], out4[4 * NM];
                auto fetch = [&](int g, R (&dst)[4 * NM]) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < 4 * NM; ++i) dst[i] = gb[base + goff(g) + i];
                };
                fetch(0, in4[0]);
                if (NG > 1) fetch(1, in4[1]);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    // LDS operations retire IN ORDER, so the wait for a group's input terms also waits for every LDS operation issued
                    // before that read: read two groups ahead, then the four steps, then this group's write -- and keep the compiler from
                    // clustering several groups' writes and reads in front of one wait (r03d: 37 cycles per step for 3 VALU instructions)
                    if (g + 2 < NG) fetch(g + 2, in4[(g + 2) % 3]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) one_step(lin_tag, zero_tag, &in4[g % 3][j * NM], &out4[j * NM]);
#pragma unroll
                    for (int i = 0; i < 4 * NM; ++i) hb[base + goff(g) + i] = out4[i];
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (sb == D) {  // whole block
                run_groups(std::integral_constant<int, NGR>{}, 0);
            } else {
                // The LAST block of a launch whose length is not a multiple of D: chunks of four groups (16 steps) through the same unrolled
                // code, then pairs of groups, then single steps.  It used to take the per-step loop alone, one LDS round trip per step:
                // 150 cycles a step against 27 -- the 40-step tail of BASELINE config 2's 1000-step launch (15 blocks of 64 + 40) was 6100
                // cycles, 2.6 us of 19.8 (profiles/r04t_dcs_fixed.txt: K = 250, tail of 58 steps: 8916 cycles against 1750 for a whole block).
                const int ng = sb >> 2;  // whole groups
                int g0 = 0;
#pragma nounroll
                for (; g0 + 4 <= ng; g0 += 4) run_groups(std::integral_constant<int, 4>{}, g0);
                if (g0 + 2 <= ng) { run_groups(std::integral_constant<int, 2>{}, g0); g0 += 2; }
#pragma nounroll
                for (int s = 4 * g0; s < sb; ++s) {
                    const size_t o = goff(s >> 2) + (size_t)(s & 3) * NM;
                    R in[NM], out[NM];
#pragma unroll
                    for (int i = 0; i < NM; ++i) in[i] = gb[o + i];
                    one_step(lin_tag, zero_tag, in, out);
#pragma unroll
                    for (int i = 0; i < NM; ++i) hb[o + i] = out[i];
                }
            }
        };
#ifdef GEMX_TIMING
        unsigned long long tc = 0, tw = 0, T0 = clock64(), W0 = wall_clock64();
#endif
        __syncthreads();  // block 0's input terms are in LDS
        for (int b = 0; b < nb; ++b) {
#ifdef GEMX_TIMING
            const unsigned long long t0 = clock64();
#endif
            if (LINABLE && lin_ok) {
                if (zero_init) run_block(std::bool_constant<LINABLE>{}, std::true_type{}, b);
                else run_block(std::bool_constant<LINABLE>{}, std::false_type{}, b);
            } else {
                if (zero_init) run_block(std::false_type{}, std::true_type{}, b);
                else run_block(std::false_type{}, std::false_type{}, b);
            }
#ifdef GEMX_TIMING
            const unsigned long long t1 = clock64();
#endif
            __syncthreads();  // publishes the states of block b; the pre waves have block b + 1 ready
#ifdef GEMX_TIMING
            tc += t1 - t0; tw += clock64() - t1;
            if (tid == 0 && blockIdx.x == 0 && b < 8) {  // the first blocks: [80 + b] = barrier b released (ticks since entry), [88 + b] = this wave's cycles in block b
                unsigned long long *dbg = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64);
                dbg[80 + b] = wall_clock64() - WE; dbg[88 + b] = t1 - t0;
            }
#endif
        }
#ifdef GEMX_TIMING
        if (tid == 0 && blockIdx.x == 0) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64);
            dbg[0] = 0; dbg[1] = tc; dbg[2] = tw; dbg[3] = clock64() - T0; dbg[4] = wall_clock64() - W0; dbg[5] = nb;
            dbg[32] = tc; dbg[33] = tw;  // every wave of workgroup 0: [32 + 2 wave] = cycles at work, [33 + 2 wave] = cycles at the barriers
            dbg[70] = W0 - WE;                // entry -> the integrator is at its first barrier (state loaded)
            dbg[71] = wall_clock64() - WE;    // entry -> the integrator's last barrier passed
        }
#endif
#pragma unroll
        for (int i = 0; i < NM; ++i) a.state[(int64_t)(1 + i) * N + env] = x[i];
    } else if ((wave & 3) == 0) {
        // ------------------------------------------------------------------ idle: keep SIMD 0 to the integrator, meet every barrier
#ifdef GEMX_DCS_WAVE4_EXITS  // A/B: round 2's behaviour (an ended wave no longer counts at the barrier on gfx950)
        return;
#endif
        __builtin_amdgcn_s_setprio(0);
        for (int b = 0; b <= nb; ++b) __syncthreads();
    } else if (dcs_worker_index(wave) < DCS_PRE) {
        // ------------------------------------------------------------------ pre: actions -> converter -> input term, DCS_PREFETCH blocks ahead
        constexpr int GP = NG2 / DCS_PRE, RP = 4 * GP;  // (double) groups / rows per lane, pre wave and block: (double) groups j * DCS_PRE + pw
        static_assert(NG2 % DCS_PRE == 0, "groups per pre wave");
        const int pw = dcs_worker_index(wave);
        uint32_t bad = 0;
        struct Rows { R f[RP][NACT]; uint32_t d[RP]; };
        constexpr int ABYTES = DISCRETE ? 1 : NACT * (int)sizeof(R);
        const int64_t rowb = N * ABYTES, bstride = (int64_t)D * rowb;
        // Loads through a buffer descriptor (wave-uniform base, rebased per block) with the row AND lane offsets in ONE precomputed 32-bit
        // VGPR per row: a load is exactly one instruction -- a per-lane 64-bit pointer costs a v_lshl_add_u64 per row, and this wave, like
        // every wave here, is bound by the number of instructions it issues (one per ~5 cycles, whatever their kind).  The descriptor's
        // size is what is left of the tensor, so rows past the end of the rollout (the last blocks' prefetch) read as 0 instead of being
        // clamped in a second copy of the loads: two copies merge in PHIs, and the copies of their registers wait for the loads just
        // issued (r03a: 48 v_mov behind an s_waitcnt vmcnt per block).
        const int64_t abase_off = ((int64_t)(4 * HALVES * pw) * N + blk0) * ABYTES;  // this workgroup's span of this wave's first row
        const int64_t atotal = (int64_t)K * N * ABYTES;
        uint32_t voff[RP];
#pragma unroll
        for (int j = 0; j < RP; ++j) {
            const int r = ((j >> 2) * DCS_PRE * HALVES + hf) * 4 + (j & 3);  // this lane's row within the block, relative to this wave's first row
            voff[j] = (uint32_t)((int64_t)r * rowb) + (uint32_t)le * (uint32_t)ABYTES;
        }
        auto load = [&](int bb, Rows &v) __attribute__((always_inline)) {
            const int64_t off = abase_off + (int64_t)bb * bstride;
            int64_t rem = atotal - off;
            rem = rem < 0 ? 0 : (rem > 0xFFFFFFFFll ? 0xFFFFFFFFll : rem);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.actions + (rem > 0 ? off : 0)), 0, (int)(uint32_t)rem, 0x00020000);
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                if (DISCRETE) v.d[j] = __builtin_amdgcn_raw_buffer_load_b8(rs, (int)voff[j], 0, 0);
                else {
#pragma unroll
                    for (int i = 0; i < NACT; ++i) v.f[j][i] = __builtin_bit_cast(R, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff[j] + 4 * i, 0, 0));
                }
            }
        };
        auto convert_t = [&](auto lin_tag, int bb, const Rows &v) __attribute__((always_inline)) {
            constexpr bool LIN = decltype(lin_tag)::value;
            R *gb = gin + ((size_t)(bb & 1) * NG2 * BLOCK + tid) * 4 * NM;
            R *ub = uu + ((size_t)(bb % 3) * NG2 * BLOCK + tid) * 4 * NU;
#pragma unroll
            for (int jg = 0; jg < GP; ++jg) {
                const int g = jg * DCS_PRE + pw;
                R in4[4 * NM], u4[4 * NU];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int j = jg * 4 + s4;
                    R act[MAX_ACT] = {R(0), R(0), R(0), R(0), R(0), R(0)};
                    uint32_t dact = 0;
                    if (DISCRETE) {
                        dact = v.d[j];
                        bad |= ((int64_t)bb * D + 4 * (g * HALVES + hf) + s4 < K) & (dact >= (uint32_t)ConvTraits<CONV>::NACTIONS);
                        dact &= (uint32_t)(ConvTraits<CONV>::NACTIONS - 1);
                    } else {
#pragma unroll
                        for (int i = 0; i < NACT; ++i) act[i] = v.f[j][i];
                    }
                    R u[MAX_U] = {R(0), R(0), R(0), R(0)};
                    ST::input_voltages(P, act, dact, u);
                    R in[NM];
                    elec_input<SYS, R, LIN>(P, om, u, LIN ? linc : nullptr, in);
#pragma unroll
                    for (int i = 0; i < NM; ++i) in4[s4 * NM + i] = in[i];
#pragma unroll
                    for (int i = 0; i < NU; ++i) u4[s4 * NU + i] = u[i];
                }
#pragma unroll
                for (int i = 0; i < 4 * NM; ++i) gb[(size_t)g * BLOCK * 4 * NM + i] = in4[i];
#pragma unroll
                for (int i = 0; i < 4 * NU; ++i) ub[(size_t)g * BLOCK * 4 * NU + i] = u4[i];
            }
        };
        auto convert = [&](int bb, const Rows &v) __attribute__((always_inline)) {
            if (LINABLE && lin_ok) convert_t(std::bool_constant<LINABLE>{}, bb, v);
            else convert_t(std::false_type{}, bb, v);
        };
        // NPF register sets: block b + NPF is requested while block b + 1 is converted.  Loads and conversions are issued UNCONDITIONALLY --
        // blocks past the end re-read the last row and convert into a buffer nobody reads any more: inside a conditional the compiler's
        // vmcnt bookkeeping falls back to waiting for the loads it has just issued.
        constexpr int NPF = DCS_PREFETCH;
#ifdef GEMX_TIMING
        unsigned long long ptl = 0, ptb = 0;
#endif
        Rows S_[NPF];
#pragma unroll
        for (int q = 0; q < NPF; ++q) load(q, S_[q]);
#ifdef GEMX_TIMING
        const unsigned long long WP0 = wall_clock64();
#endif
        convert(0, S_[0]);
#ifdef GEMX_TIMING
        const unsigned long long WP1 = wall_clock64();
#endif
        __syncthreads();
        auto iteration = [&](int b, Rows &free_set, const Rows &next_set, bool fetch) __attribute__((always_inline)) {
#ifdef GEMX_TIMING
            const unsigned long long l0 = clock64();
#endif
            if (fetch) load(b + NPF, free_set);  // (compile-time constant at both call sites)
            convert(b + 1, next_set);
#ifdef GEMX_TIMING
            const unsigned long long l1 = clock64();
#endif
            __syncthreads();
#ifdef GEMX_TIMING
            ptl += l1 - l0; ptb += clock64() - l1;
#endif
        };
        int b0 = 0;
        for (; b0 + NPF <= nb; b0 += NPF) {  // iteration b: set b % NPF is free (block b was converted an iteration ago)
#pragma unroll
            for (int q = 0; q < NPF; ++q) iteration(b0 + q, S_[q], S_[(q + 1) % NPF], true);
        }
#pragma unroll
        for (int q = 0; q < NPF - 1; ++q)  // the last nb % NPF iterations: nothing left to request
            if (b0 + q < nb) iteration(b0 + q, S_[q], S_[(q + 1) % NPF], false);
#ifdef GEMX_TIMING
        if (tid == 0 && blockIdx.x == 0) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64);
            if (pw == 0) { dbg[12] = ptl; dbg[13] = ptb; dbg[75] = WP0 - WE; dbg[76] = WP1 - WE; dbg[77] = wall_clock64() - WE; }  // loads issued | block 0 converted | done
            dbg[32 + 2 * wave] = ptl; dbg[33 + 2 * wave] = ptb;
        }
#endif
        if (bad) atomicOr(a.err, 1u);
    } else {
        // ------------------------------------------------------------------ output: observation row + done flag, one block behind
        constexpr int GPW = NG2 / DCS_OUT;  // (double) groups per output wave and block: (double) groups j * DCS_OUT + ow
        static_assert(NG2 % DCS_OUT == 0, "groups per output wave");
        constexpr int RG = 4 * HALVES;     // rows of a (double) group
        const int ow = dcs_worker_index(wave) - DCS_PRE;
        struct __attribute__((packed, aligned(4))) Row { R v[NOUT]; };
        typedef float v4f_t __attribute__((ext_vector_type(4)));
        typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
        const bool has_done = a.done != nullptr;
        auto observe_row = [&](const R *xr, const R *ur, R (&obs)[NOUT]) -> bool {
            R y[ND], ho[ST::NH];
            y[0] = om;
#pragma unroll
            for (int i = 0; i < NM; ++i) y[1 + i] = xr[i];
#pragma unroll
            for (int i = 0; i < NU; ++i) ho[i] = ur[i];
            ST::observe(P, y, AngT(0), ho, obs);
            return ST::state_violation(P, y, ho) > thr_done;
        };
        auto emit = [&](const R *xr, const R *ur, R *orow) __attribute__((always_inline)) -> bool {  // (tail blocks: the lane's row straight from registers)
            R obs[NOUT];
            const bool done = observe_row(xr, ur, obs);
            Row row;
#pragma unroll
            for (int i = 0; i < NOUT; ++i) row.v[i] = obs[i];
            *reinterpret_cast<Row *>(orow) = row;
            return done;
        };
        // whole blocks: the four rows of a group through this wave's staging buffer [4][64][NOUT] (lane stride NOUT dwords: conflict-free
        // for NOUT = 5 / 7, two-way -- free for a ds_write_b32 -- for 6) and out as 16-byte chunks: chunk q = 64 i + lane of the group
        // is bytes 16 (q % CPRW) .. + 15 of row q / CPRW (CPRW = 16 NOUT chunks per 64-env row; LDS operations of one wave complete in
        // order, so the reads need no barrier behind the writes)
        R *stg = stage + (size_t)ow * 4 * BLOCK * NOUT;
        constexpr int CPRW = EPW * NOUT / 4;
        uint32_t coff[NOUT];  // byte offset of this lane's chunk i from the (wave-uniform) address of the group's first row
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int q = i * BLOCK + tid, row = q / CPRW, col = q - row * CPRW;
            coff[i] = (uint32_t)((int64_t)row * N * NOUT * (int64_t)sizeof(R) + (int64_t)col * 16);  // (the launcher checks 4 rows < 4 GiB)
        }
        // columns 0 (omega: constant behind a ConstantSpeedLoad) and NOUT - 1 (u_sup: ideal supply) of a DC machine's row are the same in
        // every row of the launch (DcStepper::observe): the staging rows get them ONCE, every row then writes columns 1 .. NOUT - 2 only
        {
            R y0[ND], ho0[ST::NH], obs0[NOUT];
            y0[0] = om;
#pragma unroll
            for (int i = 0; i < NM; ++i) y0[1 + i] = R(0);
#pragma unroll
            for (int i = 0; i < ST::NH; ++i) ho0[i] = R(0);
            ST::observe(P, y0, AngT(0), ho0, obs0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                stg[((size_t)(hf * 4 + s4) * EPW + le) * NOUT] = obs0[0];
                stg[((size_t)(hf * 4 + s4) * EPW + le) * NOUT + NOUT - 1] = obs0[NOUT - 1];
            }
            asm volatile("" ::: "memory");
        }
        const int64_t ostride = N * NOUT;
        // stores through buffer descriptors rebased per block (wave-uniform), the lane's part of the address in ONE constant 32-bit VGPR per
        // chunk, the group's in an SGPR: a store is one instruction (see the pre waves' loads)
        const int64_t obase_off = (((int64_t)(RG * ow) * N + blk0) * NOUT) * (int64_t)sizeof(R);  // this workgroup's span of this wave's first row
        R *obase = a.obs + ((int64_t)(RG * ow) * N + env) * NOUT;  // this lane's row of this wave's first step (tail blocks)
        // done bytes of a group of four steps: every lane leaves its env's byte of each step in a [4][64]-byte staging area; read back as one
        // dword per lane that is bytes 4 (l % 16) .. + 3 of step l / 16 -- ONE store per group and no bit fiddling
        unsigned char *dstg = reinterpret_cast<unsigned char *>(stage + (size_t)DCS_OUT * 4 * BLOCK * NOUT) + (size_t)ow * 4 * BLOCK;
        const int64_t dbase_off = (int64_t)(RG * ow) * N + blk0;
        const uint32_t doff = (uint32_t)((int64_t)((4 * tid) / EPW) * N + (4 * tid) % EPW);  // (dword `tid` of the [RG][EPW] byte staging area)
        unsigned char *optr = reinterpret_cast<unsigned char *>(a.obs) + obase_off;  // block pb's rows of this wave / workgroup (wave-uniform, advanced per block)
        unsigned char *dptr = a.done + (has_done ? dbase_off : 0);
        const int64_t ostep = (int64_t)D * ostride * (int64_t)sizeof(R), dstep = has_done ? (int64_t)D * N : 0;
#ifdef GEMX_TIMING
        unsigned long long to_w = 0, to_r = 0, to_s = 0;  // per group: compute + staging writes | read back + wait | stores
#endif
        auto process_t = [&](auto done_tag, int pb) __attribute__((always_inline)) {
            constexpr bool HAS_DONE = decltype(done_tag)::value;  // (compile-time: a wave-uniform run-time test costs two issue slots per group)
            const int sb = steps_of(pb);
            const R *hb = hand + ((size_t)(pb & 1) * NG2 * BLOCK + tid) * 4 * NM;
            const R *ub = uu + ((size_t)(pb % 3) * NG2 * BLOCK + tid) * 4 * NU;
            if (sb == D) {  // whole block: all LDS reads of this wave's groups first, then group after group without a branch
                R xs[GPW][4 * NM], us[GPW][4 * NU];
#pragma unroll
                for (int jg = 0; jg < GPW; ++jg) {
                    const int g = jg * DCS_OUT + ow;
#pragma unroll
                    for (int i = 0; i < 4 * NM; ++i) xs[jg][i] = hb[(size_t)g * BLOCK * 4 * NM + i];
#pragma unroll
                    for (int i = 0; i < 4 * NU; ++i) us[jg][i] = ub[(size_t)g * BLOCK * 4 * NU + i];
                }
                // (whole blocks lie inside the tensors: no clipping needed, and the block's base is a running 64-bit value -- rebuilding
                // offset and remaining size from pb cost ~40 scalar instructions per block, every one an issue slot of this wave)
                const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)optr, 0, -1, 0x00020000);
                const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)dptr, 0, HAS_DONE ? -1 : 0, 0x00020000);
#pragma unroll
                for (int jg = 0; jg < GPW; ++jg) {
                    const int r0 = RG * jg * DCS_OUT;  // first row of the (double) group, relative to this wave's first row
#ifdef GEMX_TIMING
                    const unsigned long long o0 = clock64();
#endif
#ifdef GEMX_DCS_DIRECT_ROWS
                    static_assert(EPW == BLOCK, "A/B path of the 64-env form");
                    unsigned long long m[4];  // A/B: round 2's rows, stored straight from the lanes' registers
                    R *ob = obase + (int64_t)pb * D * ostride;  // (A/B only)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
                        m[s4] = __ballot(emit(&xs[jg][s4 * NM], &us[jg][s4 * NU], ob + (int64_t)(r0 + s4) * ostride));
#else
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        R obs[NOUT];
                        const bool dn = observe_row(&xs[jg][s4 * NM], &us[jg][s4 * NU], obs);
#pragma unroll
                        for (int i = 1; i < NOUT - 1; ++i) stg[((size_t)(hf * 4 + s4) * EPW + le) * NOUT + i] = obs[i];
                        if (HAS_DONE) dstg[(hf * 4 + s4) * EPW + le] = dn ? 1 : 0;
                    }
                    // (compiler-level fences: the rows are written as floats / bytes and read back as float4 / dwords -- distinct types to
                    // the alias analysis, which would otherwise keep the PREVIOUS group's chunks in registers or move the next group's
                    // writes up; LDS operations of one wave complete in order, so the hardware needs nothing)
                    asm volatile("" ::: "memory");
#ifdef GEMX_TIMING
                    const unsigned long long o1 = clock64();
#endif
                    v4f_t c[NOUT];
#pragma unroll
                    for (int i = 0; i < NOUT; ++i) c[i] = reinterpret_cast<const v4f_t *>(stg)[i * BLOCK + tid];
                    const uint32_t dbytes = reinterpret_cast<const uint32_t *>(dstg)[tid];
                    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) ONCE (else the compiler puts a separate wait in front of every store)
                    asm volatile("" ::: "memory");
#ifdef GEMX_TIMING
                    const unsigned long long o2 = clock64();
#endif
                    const int goff = (int)((int64_t)r0 * ostride * (int64_t)sizeof(R));  // (a group's offset within the block: wave-uniform, < 4 GiB by the launcher's check)
#pragma unroll
                    for (int i = 0; i < NOUT; ++i) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, c[i]), ro, (int)coff[i], goff, /*nt*/ 2);
                    if (HAS_DONE) __builtin_amdgcn_raw_buffer_store_b32(dbytes, rd, (int)doff, (int)((int64_t)r0 * N), 0);
#ifdef GEMX_TIMING
                    to_w += o1 - o0; to_r += o2 - o1; to_s += clock64() - o2;
#endif
#endif
#ifdef GEMX_DCS_DIRECT_ROWS
                    if (has_done) {
                        const int dq = tid >> 4, dc = tid & 15;
                        const unsigned long long mq = dq == 0 ? m[0] : (dq == 1 ? m[1] : (dq == 2 ? m[2] : m[3]));
                        const uint32_t nib = (uint32_t)(mq >> (4 * dc)) & 15u;
                        const uint32_t bytes = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
                        *reinterpret_cast<uint32_t *>(dptr + (int64_t)r0 * N + doff) = bytes;
                    }
#endif
                }
            } else {
#pragma nounroll
                for (int jg = 0; jg < GPW; ++jg) {
                    const int g = jg * DCS_OUT + ow;
#pragma nounroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int r = 4 * (g * HALVES + hf) + s4;  // this lane's row within the block
                        if (r < sb) {
                            R xr[NM], ur[NU];
#pragma unroll
                            for (int i = 0; i < NM; ++i) xr[i] = hb[(size_t)g * BLOCK * 4 * NM + s4 * NM + i];
#pragma unroll
                            for (int i = 0; i < NU; ++i) ur[i] = ub[(size_t)g * BLOCK * 4 * NU + s4 * NU + i];
                            const int64_t k = (int64_t)pb * D + r;
                            const bool done = emit(xr, ur, a.obs + (k * N + env) * NOUT);
                            if (has_done) a.done[k * N + env] = done ? 1 : 0;
                        }
                    }
                }
            }
        };
        (void)obase;
        auto process = [&](int pb) __attribute__((always_inline)) {
            if (has_done) process_t(std::true_type{}, pb);
            else process_t(std::false_type{}, pb);
        };
        __syncthreads();
#ifdef GEMX_TIMING
        unsigned long long tp = 0, tq = 0;
        const unsigned long long WO1 = wall_clock64();  // the first barrier released: block 0's input terms are in LDS
#endif
        for (int b = 0; b <= nb; ++b) {  // (ONE call site of process(): a second one doubles the kernel's code and tempts the inliner to refuse)
#ifdef GEMX_TIMING
            const unsigned long long q0 = clock64();
#endif
            if (b >= 1) { process(b - 1); optr += ostep; dptr += dstep; }
#ifdef GEMX_TIMING
            const unsigned long long q1 = clock64();
#endif
            if (b < nb) __syncthreads();
#ifdef GEMX_TIMING
            tp += q1 - q0; tq += clock64() - q1;
#endif
        }
#ifdef GEMX_TIMING
        if (tid == 0 && blockIdx.x == 0) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.err) + 64);
            if (ow < 3) { dbg[6 + 2 * ow] = tp; dbg[7 + 2 * ow] = tq; }
            dbg[32 + 2 * wave] = tp; dbg[33 + 2 * wave] = tq;
            if (ow == 0) { dbg[64] = to_w; dbg[65] = to_r; dbg[66] = to_s; dbg[73] = WO1 - WE; dbg[74] = wall_clock64() - WE; }
        }
#endif
    }
}

// Which shapes of the pipelined kernel exist.  Round 6, from the instantiation-coverage record (tools/instantiation_coverage.py,
// profiles/r06_instantiation_coverage.md: of 2455 compiled kernels the whole GPU suite, bench.py and the throughput matrix launched 473):
//   * <4, 2> and its FULL / FULL + SLOW / FULL + RINIT forms serve EVERY (load, solver, dead time) combination at every batch size;
//   * the two DEEP shapes (<12, 3>, <12, 6>: one workgroup per CU, small batches) exist for the default fixed-step solver (RK4) without
//     converter dead time only -- the configurations every bench leg, matrix row and `make(env_id)` default runs at 16384 envs.  Euler, the
//     Dormand-Prince kernels (integrator-bound at 0.1-0.3 of the roofline whatever the hand-off depth) and the dead-time code (IL) take
//     <4, 2> there; their twelve-step unrolled blocks were the most expensive code of the library to compile;
//   * <2, 2> (one resident round where <4, 2> would run a round plus a short tail) exists without dead time for RK4 and Dormand-Prince
//     (BASELINE config 4 runs it under both);
//   * SLOW without FULL is gone: solver sub-steps and custom constraint sets take FULL + SLOW (a per-lane supply row they do not need:
//     those launches are integrator-bound).
// Every shape that IS built is launched by tests/test_gpu_instantiations.py and compared bit for bit with the single-wave kernel.
// (the DFIM's 24-value rows leave no room for twelve-step blocks in 160 KB of LDS: its deep shapes could never be launched;
// the finite EESM converter is not available behind an RC supply, the one user of the plain FULL form)
template <int SYS, int SOLVER, bool IL> constexpr bool pipe_deep_built() { return SOLVER == GEMX_SOLVER_RK4 && !IL && SYS != GEMX_SYS_DFIM; }
template <int SYS, int CONV> constexpr bool pipe_full_built() { return !(SYS == GEMX_SYS_EESM && CONV == GEMX_CONV_FINITE_B6_4QC); }
template <int SOLVER, bool IL> constexpr bool pipe_d3_built() { return SOLVER != GEMX_SOLVER_EULER && !IL; }
// shape index -> kernel (0: <12, 3>, 1: <4, 2>, 2: <2, 2>, 3: <12, 6>, 4: <4, 2> FULL, 6: <4, 2> FULL SLOW, 7: <4, 2> FULL RINIT; 5 (SLOW without
// FULL) is no longer built); nullptr for a shape that is not built
template <int SYS, int CONV, int LOAD, int SOLVER, bool IL, class R> inline void (*pipe_kernel_of(int shape))(const KArgs<R>) {
    if constexpr (pipe_deep_built<SYS, SOLVER, IL>()) {
        if (shape == 0) return advance_pipe_kernel<SYS, CONV, LOAD, SOLVER, IL, R, PIPE_D, PIPE_OUT_WAVES>;
        if (shape == 3) return advance_pipe_kernel<SYS, CONV, LOAD, SOLVER, IL, R, PIPE_D, PIPE_OUT_WAVES_RW>;
    }
    if constexpr (pipe_d3_built<SOLVER, IL>()) {
        if (shape == 2) return advance_pipe_kernel<SYS, CONV, LOAD, SOLVER, IL, R, PIPE_D3, PIPE_OUT_WAVES3>;
    }
    if (shape == 1) return advance_pipe_kernel<SYS, CONV, LOAD, SOLVER, IL, R, PIPE_D2, PIPE_OUT_WAVES2>;
    if constexpr (pipe_full_built<SYS, CONV>()) {
        if (shape == 4) return advance_pipe_kernel<SYS, CONV, LOAD, SOLVER, IL, R, PIPE_D2, PIPE_OUT_WAVES2, true>;
    }
    if (shape == 6) return advance_pipe_kernel<SYS, CONV, LOAD, SOLVER, IL, R, PIPE_D2, PIPE_OUT_WAVES2, true, true>;
    if (shape == 7) return advance_pipe_kernel<SYS, CONV, LOAD, SOLVER, IL, R, PIPE_D2, PIPE_OUT_WAVES2, true, false, true>;
    return nullptr;
}

// ------------------------------------------------------------------------------------------------
// host launcher
// ------------------------------------------------------------------------------------------------
// I/O block depth S (control steps staged in LDS between global-memory bursts): as deep as the LDS allows for the
// number of workgroups that should be co-resident per CU, capped by the per-lane action-prefetch registers.
inline int choose_steps_per_block(const gemx_handle *h, int K, int es, int abytes, int max_wg_per_cu = 16) {
    if (K <= 1) return 1;
    const size_t lds_max = h->lds_max - (size_t)h->cfg.action_delay * BLOCK * h->nact_conv * es - 64;  // minus the DeadTimeProcessor FIFO
    const size_t per_step = (size_t)BLOCK * h->nout * es + 2 * (size_t)BLOCK * abytes + BLOCK;
    int S = h->steps_per_block;
    if (S <= 0) {
        const int64_t nblocks = (h->n + BLOCK - 1) / BLOCK;
        int64_t per_cu = (nblocks + h->n_cu - 1) / h->n_cu;
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 16) per_cu = 16;
        if (per_cu > max_wg_per_cu) per_cu = max_wg_per_cu;  // register-limited residency: deeper I/O blocks instead of idle LDS
        S = (int)((lds_max - 1024) / per_cu / per_step);
        if (S > MAX_STEPS_PER_BLOCK) S = MAX_STEPS_PER_BLOCK;
    }
    const int cpr = BLOCK * abytes / 16;
    const int s_regs = act_chunks(cpr) * BLOCK / cpr;  // steps whose action rows fit the per-lane prefetch registers
    if (S > s_regs) S = s_regs;
    const int s_lds = (int)((lds_max - 256) / per_step);
    if (S > s_lds) S = s_lds;
    if (S > K) S = K;
    if (S < 1) S = 1;
    return S;
}

template <int SYS, int CONV, int LOAD, int SOLVER, bool IL, class R>
int launch_advance_t(gemx_handle *h, const void *actions, int K, void *obs, uint8_t *done, int obs_every, hipStream_t st) {
    constexpr int ABYTES = ConvTraits<CONV>::DISCRETE ? 1 : ConvTraits<CONV>::NACT * (int)sizeof(R);
    KArgs<R> a;
    a.P = params_of<R>(h);
    a.state = (R *)h->state;
    a.angle = (typename Angle<R>::T *)h->angle;
    a.sw = h->sw;
    a.actions = (const unsigned char *)actions;
    a.act_synth = h->cur_synth ? 1 : 0;
    a.act_seed = h->cur_seed;
    a.act_env_base = h->cfg.env_base;
    a.act_half = h->cur_half ? 1 : 0;
#ifdef GEMX_PREP_PHASES
    a.prep_phases = GEMX_PREP_PHASES;
#else
    a.prep_phases = ((SYS == GEMX_SYS_SCIM || SYS == GEMX_SYS_DFIM) && (h->n + BLOCK - 1) / BLOCK <= (int64_t)h->n_cu) ? 2 : 1;
#endif
    a.act_step0 = h->cur_step0;
    a.obs = (R *)obs;
    a.done = done;
    a.ring = (unsigned char *)h->ring;
    const int delay = h->cfg.action_delay;
    a.fifo_phase = h->fifo_phase;
    a.err = h->err;
    if (h->linmap_state == 0) {  // once per handle: the electrical subsystem's one-step map (constant-speed loads)
        // (random initialisers that draw OMEGA per episode keep the stage-by-stage solver.  Round 5: only those -- a random MOTOR initialiser
        // beside a constant-speed load leaves omega at init[0], the map stays valid, and the kernels check every lane's omega at launch
        // anyway (lin_usable); until then every random-initialiser handle ran the full Runge-Kutta stages, 2.2 x the integrator time)
        if constexpr (linable<SYS, LOAD, SOLVER, IL, R>()) {
            hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(st, &capturing);
            const bool omega_drawn = h->cfg.init_kind != GEMX_INIT_CONST && h->cfg.init_lo[0] < h->cfg.init_hi[0];
            if (omega_drawn || (h->cfg.solver_flags & GEMX_SOLVER_ADAPTIVE) || h->cfg.solver_nsteps != 1) {  // (error control, sub-steps: stage by stage)
                h->linmap_state = -1;
            } else if (capturing == hipStreamCaptureStatusNone) {
                // built and COMPLETED here, so that later launches on any stream (or from a captured graph) find it; a first launch
                // that is itself being captured into a graph goes without the map and leaves the attempt to the next eager launch
                hipLaunchKernelGGL((linmap_kernel<SYS, SOLVER, R>), dim3(1), dim3(64), 0, st, params_of<double>(h), (R *)h->linmap_dev);
                GEMX_HIP_TRY(hipGetLastError());
                GEMX_HIP_TRY(hipStreamSynchronize(st));
                h->linmap_state = 1;
                h->pf.lin_on = 1;
            }
        } else {
            h->linmap_state = -1;
        }
    }
    a.P = params_of<R>(h);
    a.P.errw = h->err;
    a.rinit = (const InitDev *)h->rinit_dev;
    a.rcnt = h->rcnt;
    a.rw = h->cur_reward != nullptr ? (const RewardDev<R> *)h->rw_dev : nullptr;
    if constexpr (sizeof(R) == 4) a.rh = h->rh_f;
    else a.rh = h->rh_d;
    a.refs = (const R *)h->cur_refs;
    a.reward = (R *)h->cur_reward;
    a.N = h->n;
    a.K = K;
    a.obs_every = obs_every;
    auto kern = advance_kernel<SYS, CONV, LOAD, SOLVER, IL, R>;
    if (h->wg_per_cu == 0) {  // single-wave workgroups a CU can hold, from the kernel's VGPR count
        hipFuncAttributes fa;
        h->wg_per_cu = 16;
        if (hipFuncGetAttributes(&fa, (const void *)kern) == hipSuccess && fa.numRegs > 0) {
            int waves = 512 / ((fa.numRegs + 7) & ~7);  // gfx950: 512 VGPRs per SIMD lane, allocation granule 8
            if (waves < 1) waves = 1;
            if (waves > 8) waves = 8;
            h->wg_per_cu = 4 * waves;
        }
    }
    a.S = choose_steps_per_block(h, K, (int)sizeof(R), ABYTES, h->wg_per_cu);
    a.D = 1;
    // 16-byte alignment of every full block's rows: row starts are (k*N + blk0) * bytes_per_env with blk0 % 64 == 0
    a.coop = (((uintptr_t)actions & 15u) == 0 && ((size_t)h->n * ABYTES) % 16 == 0 &&
              (done == nullptr || (((uintptr_t)done & 15u) == 0 && (size_t)h->n % 16 == 0))) ? 1 : 0;
    a.obs_vec = (((size_t)h->n * h->nout * sizeof(R)) % 16 == 0) ? 1 : 0;
    size_t smem = (size_t)a.S * BLOCK * h->nout * sizeof(R) + 2 * (size_t)a.S * BLOCK * ABYTES + (size_t)a.S * BLOCK;
    smem = (smem + 15) & ~(size_t)15;
    smem += (size_t)delay * BLOCK * conv_nact_c<CONV>() * sizeof(R);  // DeadTimeProcessor FIFO
    const int64_t blocks = (h->n + BLOCK - 1) / BLOCK;
    // pipelined kernel (integrator / output / loader waves) whenever the launch qualifies
    // (round 5: ANY batch size -- a last workgroup of fewer than 64 envs is handled inside the pipelined kernel (advance_pipe_kernel:
    // full_wg / fast_io), rows that are not 16-byte aligned are moved by the same instructions (unaligned 16-byte accesses are legal);
    // dc_stream_kernel still wants whole workgroups and aligned rows)
    const bool pipe_ok = h->use_pipe != 0 && K >= 2 && obs_every;
    const bool fast_step = params_of<R>(h).constr_kind <= 1 && h->cfg.solver_nsteps == 1;  // (else: the pipelined kernel's rolled copy of the step)
    // RC supply / random initialisers: the FULL instantiation (shape <4, 2> only; see advance_pipe_kernel)
    // (round 6: also solver sub-steps / custom constraint sets -- the SLOW copy of the step exists in its FULL form only, pipe_kernel_of)
    const bool need_full = h->cfg.supply_kind != GEMX_SUPPLY_IDEAL || h->cfg.init_kind != GEMX_INIT_CONST || !fast_step;
    // (fp32 only: the fp64 build is a diagnostic of the same device functions and takes the single-wave kernel, which keeps its
    // translation units three times smaller)
    // small batches of the DC machines behind a constant-speed load: dc_stream_kernel, up to one workgroup per TWO CUs (8192 envs): there it
    // already writes 5 TB/s, and beyond the pipelined kernel's 16-byte-aligned row stores use the write path better (same-box sweep,
    // PermExDc, us per 1000 steps at 4096 / 8192 / 12288 / 16384 envs: 39 / 41 / 88 / 94 against 71 / 72 / 72 / 72; tools/ab_dc_stream.py)
    if constexpr (sizeof(R) == 4 && LOAD == GEMX_LOAD_CONST_SPEED && !IL &&
                  (SYS == GEMX_SYS_DC_PERMEX || SYS == GEMX_SYS_DC_SERIES || SYS == GEMX_SYS_DC_SHUNT || SYS == GEMX_SYS_DC_EXTEX)) {
        bool dcs_ok = pipe_ok && fast_step && !h->cur_synth && !h->cur_half && (h->n % BLOCK) == 0 && a.coop && a.obs_vec && h->use_dc_stream != 0 && !need_full && delay == 0 && !(h->cfg.solver_flags & GEMX_SOLVER_ADAPTIVE) && h->cur_reward == nullptr && (h->omega_is_init || h->use_dc_stream >= 3) &&
                      params_of<R>(h).obs_layout == GEMX_OBS_AOS && params_of<R>(h).t_il == R(0) &&
                      (h->use_dc_stream > 1 || 2 * blocks <= (int64_t)h->n_cu) && dcs_smem_bytes<SYS, CONV>() <= h->lds_max &&
                      (int64_t)h->n * h->nout * 64 < ((int64_t)1 << 31);  // (SIGNED 32-bit store offsets: lane offsets across the rows of a (double)
                                                                          // group plus the per-group offset of up to 16 rows x N x NOUT x 4 bytes
                                                                          // of the two-state form -- advisor finding, round 3; only reachable with
                                                                          // GEMX_DC_STREAM >= 2 forcing the kernel beyond its 8192 envs)
        if (dcs_ok) {
            // never into a graph: `omega_is_init` is what the host knows NOW, a captured launch runs later, possibly behind a gemx_set_state.
            // The pipelined kernel decides on the device (lin_usable), so it is what a graph gets.
            hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(st, &capturing);
            dcs_ok = capturing == hipStreamCaptureStatusNone;
        }
        if (dcs_ok) {
            // 32 envs per workgroup -- twice the workgroups and CUs for the same batch, half the store instructions per CU -- whenever they
            // are few enough (GEMX_DCS_EPW=64: the 64-env form at every size)
            // (same box, PermExDc, Euler, us per 1000 steps: 4096 envs 28.6 -> 19.7, 1024 envs 28.9 -> 19.0; at 8192 envs -- 256 workgroups of 32, every
            // CU busy -- 33.0 against 31.2: so up to one workgroup per TWO CUs)
            const bool epw32 = h->dcs_epw != 64 && 4 * blocks <= (int64_t)h->n_cu;
            auto dkern = epw32 ? dc_stream_kernel<SYS, CONV, SOLVER, R, BLOCK / 2> : dc_stream_kernel<SYS, CONV, SOLVER, R, BLOCK>;
            if (!(h->dcs_attr_set & (epw32 ? 2 : 1))) {
                GEMX_HIP_TRY(hipFuncSetAttribute((const void *)dkern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_max));
                h->dcs_attr_set |= epw32 ? 2 : 1;
            }
            const int dd = dcs_depth<SYS>() * (epw32 ? 2 : 1);
            a.S = dd;
            a.D = dd;
            // (more than half the LDS: ONE workgroup per CU, else the dispatcher stacks two on one CU while others idle and their
            // integrator waves share issue slots -- 41 -> 88 us per 1000 steps between 8192 and 12288 envs)
            const size_t dneed = dcs_smem_bytes<SYS, CONV>(), dhalf = (size_t)h->lds_max / 2 + 1024, dsmem = dneed > dhalf ? dneed : dhalf;
            const long long dblocks = epw32 ? 2 * blocks : blocks;
            hipLaunchKernelGGL(dkern, dim3((unsigned)dblocks), dim3(dcs_waves<SYS>() * BLOCK), dsmem, st, a);
            GEMX_HIP_TRY(hipGetLastError());
            h->ll = {3, SYS, CONV, LOAD, SOLVER, (int)IL, (int)sizeof(R), dd, dcs_waves<SYS>() * BLOCK, K, dd, dblocks, dsmem};
            h->ll.shape = epw32 ? BLOCK / 2 : BLOCK;  // (the kernel's envs-per-workgroup template argument)
            return GEMX_OK;
        }
    }
    if constexpr (sizeof(R) == 4) if (pipe_ok) {
        using ST = Stepper<SYS, conv_base<CONV>(), LOAD, SOLVER, IL, R>;
        const int NHT = SysTraits<SYS>::ND + (SysTraits<SYS>::HAS_ANGLE ? 1 : 0) + ST::NH + 1 + (need_full ? 1 : 0);
        // LDS footprint of one workgroup at hand-off depth D (steps per barrier): ring + done ring + double-buffered hand-off rows + ...
        auto smem_of = [&](int D) {
            size_t b = (size_t)D * BLOCK * h->nout * sizeof(R) + (size_t)D * BLOCK + 2 * (size_t)D * BLOCK * NHT * sizeof(R);
            b += (size_t)pipe_queue_rows(D, delay, conv_dq<CONV>() && h->pf.dq_processor != 0, need_full) * BLOCK * conv_nact_c<CONV>() * sizeof(R);  // DeadTimeProcessor queue
            b += pipe_act_bufs(D) * (size_t)((D + 3) / 4 * 4) * BLOCK * ABYTES;  // action staging (global -> LDS direct, one or two blocks ahead)
            if (h->cur_reward != nullptr) b += pipe_ref_bufs(D) * (size_t)D * BLOCK * h->rw_n_ref * sizeof(R);  // reference staging for the fused reward
            if (ST::NVT > 0 && ConvTraits<CONV>::DISCRETE) b += (size_t)ConvTraits<CONV>::NACTIONS * 8 * sizeof(R);  // per-action voltage table
            if (h->cfg.init_kind != GEMX_INIT_CONST) b += (size_t)(prep_q<SYS>() * 8 + 1) * BLOCK * sizeof(uint32_t) + ((sizeof(InitDev) + 15) & ~(size_t)15);  // prepared draws (random initialisers) + the description
            return (b + 15) & ~(size_t)15;
        };
        // workgroups of a shape one CU holds: LDS, wave slots -- and REGISTERS (round 4: the arithmetic used to stop at the first two and
        // took four resident workgroups of a shallow shape for granted; a kernel at 129-136 VGPRs holds three, and a "one round" rule that
        // does not know runs a round plus a tail.  hipFuncGetAttributes once per handle and shape.)
        auto regs_limit = [&](int shape_, int waves_per_wg) -> int64_t {
            if (h->pipe_regs[shape_] == 0) {
                const void *kp = (const void *)pipe_kernel_of<SYS, CONV, LOAD, SOLVER, IL, R>(shape_);
                if (kp == nullptr) return 0;  // (a shape that is not built holds no workgroup)
                hipFuncAttributes fa;
                h->pipe_regs[shape_] = (hipFuncGetAttributes(&fa, kp) == hipSuccess && fa.numRegs > 0) ? fa.numRegs : 128;
            }
            int waves_per_simd = 512 / ((h->pipe_regs[shape_] + 7) & ~7);  // gfx950: 512 VGPRs per SIMD lane, allocation granule 8
            if (waves_per_simd < 1) waves_per_simd = 1;
            if (waves_per_simd > 8) waves_per_simd = 8;
            return (int64_t)(4 * waves_per_simd) / waves_per_wg;
        };
        // ... and what the runtime itself says a CU holds (hipOccupancyMaxActiveBlocksPerMultiprocessor with this launch's LDS bytes, once
        // per handle and shape): the rate limiter below prices every workgroup's interval with the number of workgroups that are
        // RUNNING, and the PermExDc <4, 2> kernel, eight per CU by LDS, wave slots and registers, runs five (r04m: 98304 envs paced for
        // 1536 concurrent workgroups with ~1280 resident ran 0.54 of the roofline against 0.90 at 65536)
        auto occ_limit = [&](int shape_, int threads, size_t smem_b) -> int64_t {
            if (h->pipe_occ[shape_] == 0 || h->pipe_occ_smem[shape_] != smem_b) {  // (the LDS bytes move with the fused reward / the queue depth)
                const void *kp = (const void *)pipe_kernel_of<SYS, CONV, LOAD, SOLVER, IL, R>(shape_);
                if (kp == nullptr) return 0;  // (a shape that is not built holds no workgroup)
                int nb_ = 0;
                h->pipe_occ_smem[shape_] = smem_b;
                if (!(h->pipe_attr_set & (1u << shape_))) {  // (the bit only on success: the launch path then repeats the call, checked, and reports the error)
                    if (hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_max) == hipSuccess) h->pipe_attr_set |= 1u << shape_;
                    else (void)hipGetLastError();
                }
                h->pipe_occ[shape_] = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, kp, threads, smem_b) == hipSuccess && nb_ > 0) ? nb_ : 64;
                (void)hipGetLastError();
            }
            return (int64_t)h->pipe_occ[shape_];
        };
        auto resident = [&](int D, int OW, int shape_full = -1) {  // (shape_full: one of the FULL / SLOW instantiations of <4, 2>, shapes 4-7)
            int64_t w = (int64_t)(h->lds_max / smem_of(D));
            const int waves_per_wg = 1 + OW + pipe_loader_waves(D);
            const int64_t wmax = 32 / waves_per_wg;
            const int shape_ = shape_full >= 0 ? shape_full : (D == PIPE_D ? (OW == PIPE_OUT_WAVES ? 0 : 3) : (D == PIPE_D2 ? 1 : 2));
            const int64_t wreg = regs_limit(shape_, waves_per_wg);
            w = w > wmax ? wmax : w;
            w = w > wreg ? wreg : w;
            if (smem_of(D) <= h->lds_max) {
                const int64_t wocc = occ_limit(shape_, waves_per_wg * BLOCK, smem_of(D));
                w = w > wocc ? wocc : w;
            }
            if (wreg == 0) return (int64_t)0;  // not built (pipe_deep_built)
            return (w < 1 ? 1 : w) * (int64_t)h->n_cu;
        };
        // one resident round of the 4-wave shape if N is that small, else the 3-wave shape at ANY N: measured at 131072 and 1048576
        // envs over all motor families (profiles/r01d_matrix.md) it is on par with or ahead of the single-wave kernel (PMSM
        // 1M envs: 100 vs 86 G env-steps/s) -- the split keeps stores fire-and-forget and the integrator free of vmcnt waits
        int D = 0, OW = 0, shape = 0;
        // six output waves (eight waves in all: still one workgroup per CU) where the output side carries more than three waves keep up
        // with: the fused reward, and the COMPACT hand-off rows of the synchronous machines behind a finite converter (advance_pipe_kernel,
        // COMPACT_K: the integrator is 8 % faster per block there, 3555 against 3875 cycles, and the output waves look the phase voltages
        // up themselves -- three of them need 4100 cycles per block, six 2500; PMSM headline 147.6 -> 140 us per launch)
        constexpr bool COMPACT_L = SYS == GEMX_SYS_SYNC && ST::NVT > 0 && ConvTraits<CONV>::DISCRETE && linable<SYS, LOAD, SOLVER, IL, R>();
        const bool compact_l = COMPACT_L && h->pf.lin_on != 0 && !need_full;
        // ... and with those rows the deep shape is ahead of <4, 2> at ANY N whose workgroups fill its rounds of one workgroup per CU: same box,
        // PMSM finite, of the roofline at 28672 / 32768 / 49152 / 65536 / 131072 envs: 0.71 / 0.72 / 0.70 / 0.77 / 0.79 against 0.68 / 0.68 / 0.67 /
        // 0.65 / 0.70 (profiles/r03s_shapes_compact.md); a last round less than ~85 % full loses instead (24576 envs: 0.59 against 0.65)
        // -- and a launch too short to amortise a workgroup's ~10 us outside its block loop, which <4, 2>'s four co-resident workgroups overlap
        // (1M envs x 100 steps: 0.53 against 0.68): K >= 400
        bool deep_rounds = false;
        // (not behind a DeadTimeProcessor: its delayed-read blocks are slower in the deep shape than <4, 2>'s queue: 131072 envs 0.57 against 0.69)
        // (nor with the fused reward, whose output waves are the bound at large N either way: 131072 envs 0.56 against 0.60)
        // (round 4: only while the rate limiter is OFF.  With it the shallow shapes hold 0.79 at every size from 32768 envs on, the deep shape in
        // rounds 0.64-0.77: profiles/r04m_pace_shapes.txt)
        const bool pacing_on = (h->pace_gbps < 0.0 ? GEMX_PACE_DEFAULT_ON != 0 : h->pace_gbps > 0.0) && K >= 64;
        if (pipe_deep_built<SYS, SOLVER, IL>() && compact_l && !pacing_on && K >= 400 && delay == 0 && h->cur_reward == nullptr && smem_of(PIPE_D) <= h->lds_max) {
            const int64_t res0 = resident(PIPE_D, PIPE_OUT_WAVES_RW), rounds = (blocks + res0 - 1) / res0;
            deep_rounds = blocks > res0 && 100 * blocks >= 85 * rounds * res0;
        }
        // (limiter on: the deep shape up to TWO workgroups per CU -- at three, 49152 envs, the DC machines it fits run 0.76 / 0.59 / 0.33 (PermExDc /
        // ShuntDc / SeriesDc SC) against 0.86 / 0.85 / 0.60 through <4, 2> under the limiter: profiles/r04p_shape_32768.txt)
        // LONG launches at one workgroup per CU lose their rate as well: the headline holds 0.80-0.83 of the roofline up to 1000 steps per launch,
        // 0.75 at 2000, 0.70 at 3000, 0.68 at 6000 -- workgroups that start in the same phase drift apart, and the stores of a CU's neighbours
        // stop landing in the same DRAM rows.  The limiter is also a clock that keeps them together: the same launches through <12, 3> at
        // the 32768-env target run 0.83 / 0.84 at 3000 / 6000 steps (PMSM cont 0.69 / 0.70 -> 0.77 / 0.79; profiles/r04r_probe5.txt).  Only the
        // rows whose integrator is faster than the target gain (the synchronous machines on the one-step map); an integrator-bound
        // row pays 3-6 % for the clock reads (SCIM finite 0.66 -> 0.62, ShuntDc 0.54 -> 0.52) and keeps its unpaced launch.  From 1200 steps on: at
        // 1000 the unpaced <12, 6> is ahead (0.84 against 0.81), at 1400 behind (0.77 against 0.81-0.83: profiles/r04q_probe3.txt A, r04r_probe5.txt).
        const bool long_one = pacing_on && h->pace_gbps < 0.0 && K >= 1200 && SYS == GEMX_SYS_SYNC && h->pf.lin_on != 0 && !need_full && h->cur_reward == nullptr &&
                              blocks <= (int64_t)h->n_cu;
        const int64_t deep_max = pacing_on && resident(PIPE_D, PIPE_OUT_WAVES) > 2 * (int64_t)h->n_cu ? 2 * (int64_t)h->n_cu : resident(PIPE_D, PIPE_OUT_WAVES);
        if (pipe_deep_built<SYS, SOLVER, IL>() && smem_of(PIPE_D) <= h->lds_max && (blocks <= deep_max || deep_rounds)) {
            D = PIPE_D; OW = PIPE_OUT_WAVES; shape = 0;
            // (long launches of the synchronous machines' one-step-map rows: <12, 3>, which carries the rate limiter -- see `long_one` below)
            if (h->cur_reward != nullptr || (compact_l && !long_one)) { OW = PIPE_OUT_WAVES_RW; shape = 3; }
        }
        else if (pipe_d3_built<SOLVER, IL>() && smem_of(PIPE_D3) <= h->lds_max && blocks <= resident(PIPE_D3, PIPE_OUT_WAVES3) &&
                 blocks > resident(PIPE_D2, PIPE_OUT_WAVES2) && 2 * blocks <= 3 * resident(PIPE_D2, PIPE_OUT_WAVES2)) {
            // (round 4: EVERY system.  Rounds 2-3 kept this to the induction machines on one sample -- ExtExDc at 131072 envs, which is not
            // this case at all (two full rounds of <4, 2>) -- but where <4, 2> runs a round plus a tail of at most half a round and <2, 2>
            // fits everything into one, the interleaved same-box A/B (tools/ab_matrix_shapes.py, profiles/r04c_shapes_ab.md, 65536 envs,
            // of the roofline) has it ahead almost everywhere: PMSM cont 0.57 -> 0.71, PMSM cont SC 0.50 -> 0.69, ShuntDc 0.61 -> 0.73,
            // EESM 0.57 -> 0.64 / 0.56 -> 0.61, DqToAbc + DeadTime 0.59 -> 0.75, PermExDc 0.78 -> 0.85, fused reward 0.56 -> 0.62;
            // level: SeriesDc SC, ExtExDc.)
            // (induction machines only: lighter steppers lose with the shallow shape -- ExtExDc 131072 envs 133 -> 104 G; PMSM finite at
            // 98304 envs, where this rule used to apply: 75.8 G against 84.2 G through <4, 2>, profiles/r03k_shapes_pmsm.md)
            // one resident round with the shallow shape where <4, 2> would run one round plus a tail of at most half a round: SCIM,
            // 65536 envs (BASELINE config 4) 49 -> 67 G env-steps/s.  Everywhere else <4, 2> is 5-20 % ahead of <2, 2> (fewer barriers,
            // fewer waves per SIMD): PMSM finite at 131072 envs = two FULL rounds of <4, 2>: 89-92 G against 79 G in one round of <2, 2>
            D = PIPE_D3; OW = PIPE_OUT_WAVES3; shape = 2;
        } else if (smem_of(PIPE_D2) <= h->lds_max) { D = PIPE_D2; OW = PIPE_OUT_WAVES2; shape = 1; }
        if (h->pipe_shape >= 0 && h->pipe_shape <= 3) {  // forced shape (tests)
            const int fd[4] = {PIPE_D, PIPE_D2, PIPE_D3, PIPE_D}, fo[4] = {PIPE_OUT_WAVES, PIPE_OUT_WAVES2, PIPE_OUT_WAVES3, PIPE_OUT_WAVES_RW};
            if (smem_of(fd[h->pipe_shape]) <= h->lds_max && pipe_kernel_of<SYS, CONV, LOAD, SOLVER, IL, R>(h->pipe_shape) != nullptr) { shape = h->pipe_shape; D = fd[shape]; OW = fo[shape]; }
        }
        if (need_full) {  // one instantiation serves these handles
            // ~180 VGPRs = two resident workgroups per CU.  RC supply: ahead of the single-wave kernel at every size (PMSM finite, same box:
            // 52 vs 14 G env-steps/s at 16384 envs, 89 vs 26 at 32768, 89 vs 47 at 65536).  Random initialisers (the fp64 draw sits in the
            // integrator's reset path): ahead up to 32768 envs (28 vs 16), level at 65536, behind beyond (28 vs 43 at 131072) --
            // tools/ab_full_variant.py
            // Round 5: random initialisers have their own instantiation (RINIT) whose loader wave prepares the draws; same box, PMSM finite:
            // 21 -> 54-57 G env-steps/s at 32768 / 65536 envs, and 57 against the single-wave kernel's 32 at 131072 -- the size rule
            // ("at most four workgroups per CU") is gone (profiles/r05r_ab_prep.txt).
            if (smem_of(PIPE_D2) <= h->lds_max) { D = PIPE_D2; OW = PIPE_OUT_WAVES2; shape = h->cfg.init_kind != GEMX_INIT_CONST ? 7 : 4; }
            else D = 0;
        }
        if (!fast_step && D != 0) {  // solver sub-steps / a custom constraint set: the SLOW instantiations of <4, 2> (no limiter: integrator-bound)
            if (smem_of(PIPE_D2) <= h->lds_max && h->cfg.init_kind == GEMX_INIT_CONST) { D = PIPE_D2; OW = PIPE_OUT_WAVES2; shape = 6; }
            else D = 0;  // (together with random initialisers: the single-wave kernel)
        }
        if (D != 0) {
            a.S = D;
            a.D = D;
            // the rate limiter (advance_pipe_kernel): only where the batch oversubscribes the write path -- more workgroups than CUs; at one
            // workgroup per CU in one round the integrator's instruction stream is the pacing, and an s_memrealtime per block would only cost
            // it time.  interval per row and workgroup = algorithmic bytes of a 64-env step x workgroups resident on the chip / target rate.
            int64_t pace_res = 0;
            int cal_slot = -1, cal_cand = 0;
            double pace_scale_used = 1.0;
            a.pace_block_ticks = 0;
            a.pace_tail_ticks = 0;
            a.pace_tail_from = 0xFFFFFFFFu;
            {
                // Target: a few per cent under the knee the sweeps over the real kernels found (profiles/r04l_pace_sweep.txt, r04n_pace_sweep2.txt;
                // same box, of the 8 TB/s).  Under the limiter a launch runs at ~0.98 x target / 8000 up to the knee and collapses beyond it:
                // rows of 14-24 values (3.5-6 KB per workgroup and step) at 32768 / 65536 / 131072 envs peak at 7000-7200 / 6800-7000 / 6400-6600
                // GB/s (PMSM finite 0.65 / 0.66 / 0.65 unpaced -> 0.85 / 0.84 / 0.80 at the knee, 0.66 / 0.79 / 0.76 one step beyond it; SCIM,
                // EESM, DFIM finite alike), the DC machines' 5-7 values (1.3-1.8 KB) at 7200-7600 / 7200 / 7000 (ShuntDc 0.62 / 0.61 / 0.65 ->
                // 0.85 / 0.87 / 0.80).  The knee moves down with the number of workgroups in flight.  Kernels whose integrators offer less
                // than the target (the speed-control rows, 16384 envs) never wait and are unchanged.
                const bool few = blocks <= 2 * (int64_t)h->n_cu, some = blocks <= 4 * (int64_t)h->n_cu;
                // Launches that also READ eight bytes or more per env-step (continuous duty cycles through the loader wave) have the knee lower and
                // a softer landing beyond it (back to the unpaced level, not below): 65536 / 131072 envs PMSM cont 0.66 / 0.63 unpaced -> 0.68 / 0.63
                // at 6000 / 5600, EESM cont 0.62 / 0.59 -> 0.73 / 0.66, DFIM cont 0.60 / 0.70 -> 0.72 / 0.69, control_space='dq' 0.62 / 0.62 -> 0.68 /
                // 0.65, ExtExDc cont 0.72 / 0.64 -> 0.77 / 0.64 at 6400 (profiles/r04p_pace_sweep3.txt).
                const int abytes_eff = h->cur_half ? ABYTES / 2 : ABYTES;  // (gemx_rollout_half: two bytes per value)
                const bool reads = abytes_eff >= 8 && !h->cur_synth;  // (synthetic actions are generated in the launch: nothing is read)
                const double dflt = h->nout <= 8 ? (reads ? (few ? 7000.0 : 6400.0) : (some ? 7000.0 : 6600.0))
                                    : reads      ? (few ? 6400.0 : (some ? 6000.0 : 5800.0))
                                                 : (few ? 6800.0 : (some ? 6600.0 : 6400.0));
                double target = h->pace_gbps < 0.0 ? (GEMX_PACE_DEFAULT_ON ? dflt : 0.0) : h->pace_gbps;
                // closed loop: which candidate this launch runs (gemx_handle::PaceCal)
                constexpr double CAL_SCALE[gemx_handle::PaceCal::NC] = {1.0, 0.93, 1.07, 0.86, 0.0};  // (0: unpaced)
                auto &pc = h->pcal;
                cal_slot = -1;
                const bool cal_eligible = h->pace_cal_on != 0 && h->pace_gbps < 0.0 && target > 0.0 && (blocks > (int64_t)h->n_cu || long_one) && K >= 64 && shape != 3 && shape < 5;
                if (cal_eligible) {
                    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
                    (void)hipStreamIsCapturing(st, &capturing);
                    // (the interval per hand-off block does not depend on the launch's length: K enters only through its class -- short launches,
                    // whose fixed costs weigh on the timings, and the long launches at one workgroup per CU, which pace a different shape.  A
                    // training loop that varies K within a class keeps its calibration; round 5 keyed it by K itself and re-calibrated, up to 240
                    // launches, at every change.)
                    const int k_class = long_one ? 2 : (K < 256 ? 0 : 1);
                    const long long sig = ((long long)k_class << 40) ^ ((long long)blocks << 8) ^ (long long)shape ^ (h->cur_reward != nullptr ? 0x80 : 0) ^ (h->cur_synth ? 0x40 : 0) ^ (h->cur_half ? 0x20 : 0);
                    if (pc.sig != sig) {  // a new kind of launch: start over (events are kept)
                        pc.sig = sig; pc.next = 0; pc.chosen = -1; pc.center = 1.0f; pc.recenters = 0;
                        for (int c = 0; c < gemx_handle::PaceCal::NC; ++c) { pc.best[c] = 1e30f; pc.count[c] = 0; pc.issued[c] = 0; }
                        if (pc.ev_init) for (int i = 0; i < gemx_handle::PaceCal::RING; ++i) pc.ev_cand[i] = -1;
                    }
                    if (pc.chosen < 0 && capturing == hipStreamCaptureStatusNone) {
                        if (!pc.ev_init) {
                            bool ok = true;
                            for (int i = 0; i < gemx_handle::PaceCal::RING && ok; ++i) {
                                ok = hipEventCreate((hipEvent_t *)&pc.ev0[i]) == hipSuccess && hipEventCreate((hipEvent_t *)&pc.ev1[i]) == hipSuccess;
                                pc.ev_cand[i] = -1;
                            }
                            pc.ev_init = ok;
                            if (!ok) { (void)hipGetLastError(); h->pace_cal_on = 0; }
                        }
                        if (pc.ev_init) {
                            // harvest the pairs whose launches have finished (never blocks)
                            for (int i = 0; i < gemx_handle::PaceCal::RING; ++i) {
                                if (pc.ev_cand[i] < 0 || hipEventQuery((hipEvent_t)pc.ev1[i]) != hipSuccess) continue;
                                float ms = 0.0f;
                                if (hipEventElapsedTime(&ms, (hipEvent_t)pc.ev0[i], (hipEvent_t)pc.ev1[i]) == hipSuccess && ms > 0.0f) {
                                    const int c = pc.ev_cand[i];
                                    pc.best[c] = ms < pc.best[c] ? ms : pc.best[c];
                                    pc.count[c]++;
                                }
                                pc.ev_cand[i] = -1;
                            }
                            (void)hipGetLastError();
                            bool done_cal = true;
                            for (int c = 0; c < gemx_handle::PaceCal::NC; ++c) done_cal &= pc.count[c] >= gemx_handle::PaceCal::SAMPLES;
                            if (done_cal) {
                                int bi = 0;
                                for (int c = 1; c < gemx_handle::PaceCal::NC; ++c) if (pc.best[c] < pc.best[bi]) bi = c;
                                // (a candidate must beat the starting point by more than 1 % to replace it: timing noise)
                                const int win = pc.best[bi] < 0.99f * pc.best[0] ? bi : 0;
                                // the winner at an EDGE of the bracket (x 1.07 or x 0.86): the optimum may lie beyond -- move the bracket there
                                // and calibrate again (the built-in targets are one box's knees; a launch that reads nothing, e.g. with
                                // synthetic actions, peaks well above them)
                                if ((win == 2 || win == 3) && pc.recenters < gemx_handle::PaceCal::MAX_RECENTER) {
                                    pc.center *= (float)CAL_SCALE[win];
                                    pc.recenters++;
                                    for (int c = 0; c < gemx_handle::PaceCal::NC; ++c) { pc.best[c] = 1e30f; pc.count[c] = 0; pc.issued[c] = 0; }
                                    // (pairs still in flight were timed at the OLD centre: they must not be harvested into the new bracket --
                                    // best[] keeps the minimum, one stale fast sample could pick the wrong candidate for good; advisor, round 5)
                                    for (int i = 0; i < gemx_handle::PaceCal::RING; ++i) pc.ev_cand[i] = -1;
                                } else {
                                    pc.chosen = win;
                                }
                            } else {
                                for (int i = 0; i < gemx_handle::PaceCal::RING; ++i)
                                    if (pc.ev_cand[i] < 0) { cal_slot = i; break; }
                                if (cal_slot >= 0) {  // (no free pair: this launch runs the starting point, untimed)
                                    // the candidate handed out least often so far goes next (round robin, starting with the built-in target)
                                    int c = 0;
                                    for (int q = 1; q < gemx_handle::PaceCal::NC; ++q) if (pc.issued[q] < pc.issued[c]) c = q;
                                    pc.issued[c]++;
                                    pc.next++;
                                    pc.ev_cand[cal_slot] = c;
                                    cal_cand = c;
                                }
                            }
                        }
                    }
                    const int use = pc.chosen >= 0 ? pc.chosen : (cal_slot >= 0 ? cal_cand : 0);
                    target *= CAL_SCALE[use] * pc.center;
                    pace_scale_used = CAL_SCALE[use] * pc.center;
                }
                // (round 5: the FULL instantiations are priced with their own registers and occupancy too -- without the draw code the RC-supply
                // kernel went from 207 to 118 VGPRs, four workgroups per CU instead of the two that used to be written here)
                int64_t res = shape >= 4 ? resident(D, OW, shape) : resident(D, OW);
                // More than four workgroups per CU (the DC machines' small rows) and a launch that needs the LAST slot of every CU to be one round:
                // count one slot less.  Registers, LDS, wave slots and the occupancy API said seven of the ShuntDc <4, 2> kernel; 114688 envs = 1792
                // workgroups were priced as one round of 1792 -- and ran 0.41 of the roofline against 0.63 unpaced: whichever workgroups do not
                // find their slot at once run alone afterwards at an interval meant for 1792.  Priced for six per CU they are a tail round,
                // 0.64.  (Only in that band: at 131072 envs, one round of 1792 and a tail of 256, the seven slots price right -- 0.79 against 0.65
                // with six; profiles/r04final_odd_sizes.txt, r04final_dc_residency.txt.)
                if (res > 4 * (int64_t)h->n_cu && blocks <= res && blocks > res - (int64_t)h->n_cu) res -= (int64_t)h->n_cu;
                pace_res = res;
                if (target > 0.0 && (blocks > (int64_t)h->n_cu || h->pace_gbps > 0.0 || long_one) && K >= 64 && shape != 3 && shape < 5) {  // (<12, 6> carries no limiter: see the kernel; nor do the SLOW ones)
                    const double wg_step_bytes = (double)BLOCK * ((h->cur_synth ? 0 : abytes_eff) + h->nout * sizeof(R) + 1 + (h->cur_reward != nullptr ? (h->rw_n_ref + 1) * sizeof(R) : 0));
                    auto ticks_for = [&](double active) {
                        const double t = wg_step_bytes * active / target * D / 10.0;  // bytes / (GB/s) = ns; 10 ns per tick; D rows per block
                        return t < 1.0 ? 1u : (t > 4.0e9 ? 0u : (uint32_t)(t + 0.5));
                    };
                    const int64_t full = blocks / res, tail = blocks - full * res;
                    a.pace_block_ticks = ticks_for((double)(blocks < res ? blocks : res));
                    if (full >= 1 && tail > 0) {  // the last round: `tail` workgroups on the chip
                        a.pace_tail_from = (uint32_t)(full * res);
                        a.pace_tail_ticks = ticks_for((double)(tail < (int64_t)h->n_cu ? (int64_t)h->n_cu : tail));
                    }
                }
            }
            const size_t psmem = smem_of(D);
            auto pkern = pipe_kernel_of<SYS, CONV, LOAD, SOLVER, IL, R>(shape);
            if (pkern == nullptr) return fail(GEMX_ERR_ARG, "internal: pipelined shape %d is not built for this solver", shape);
            if (!(h->pipe_attr_set & (1u << shape))) {
                GEMX_HIP_TRY(hipFuncSetAttribute((const void *)pkern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_max));
                h->pipe_attr_set |= 1u << shape;
            }
            const int threads = (1 + OW + pipe_loader_waves(D)) * BLOCK;
            if (cal_slot >= 0) (void)hipEventRecord((hipEvent_t)h->pcal.ev0[cal_slot], st);
            hipLaunchKernelGGL(pkern, dim3((unsigned)blocks), dim3(threads), psmem, st, a);
            if (cal_slot >= 0 && hipEventRecord((hipEvent_t)h->pcal.ev1[cal_slot], st) != hipSuccess) { h->pcal.ev_cand[cal_slot] = -1; (void)hipGetLastError(); }
            GEMX_HIP_TRY(hipGetLastError());
            h->pace_scale_last = pace_scale_used;
            h->pace_cal_state = h->pcal.chosen >= 0 ? 2 : (cal_slot >= 0 ? 1 : 0);
            h->ll = {1, SYS, CONV, LOAD, SOLVER, (int)IL, (int)sizeof(R), D, threads, K, D, (long long)blocks, psmem, a.pace_block_ticks, a.pace_tail_ticks, (long long)pace_res};
            h->ll.shape = shape;
            return GEMX_OK;
        }
    }
    if (h->cur_half)
        return fail(GEMX_ERR_ARG, "gemx_rollout_half needs the pipelined kernel: fp32, K >= 2, a continuous converter (and no random initial states together with solver sub-steps / a custom constraint set)");
    if (h->cur_synth)  // (only the pipelined kernel's loader wave generates actions; gemx_synthetic_actions writes the same stream for any other path)
        return fail(GEMX_ERR_ARG, "gemx_rollout_synthetic needs the pipelined kernel: fp32, K >= 2 (and no random initial states together with solver sub-steps / a custom constraint set)");
    if (K == 1 && h->cur_reward == nullptr && h->use_step_kernel != 0) {  // the closed-loop path: see step_kernel
        // one launch per control step is bound by the HOST's launch path: the function handle is resolved once per handle and the
        // arguments go as one buffer -- 3.36 against 3.55 us per launch through hipLaunchKernelGGL (tools/microbench_launch.hip)
        if (h->step_fn == nullptr && !h->step_fn_failed) {
            hipFunction_t f = nullptr;
            if (hipGetFuncBySymbol(&f, (const void *)step_kernel<SYS, CONV, LOAD, SOLVER, IL, R>) == hipSuccess && f != nullptr) h->step_fn = (void *)f;
            else { h->step_fn_failed = true; (void)hipGetLastError(); }
        }
        if (h->step_fn != nullptr) {
            size_t sz = sizeof(a);
            void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, (void *)&a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            GEMX_HIP_TRY(hipModuleLaunchKernel((hipFunction_t)h->step_fn, (unsigned)blocks, 1, 1, BLOCK, 1, 1, 0, st, nullptr, cfg));
        } else {
            hipLaunchKernelGGL((step_kernel<SYS, CONV, LOAD, SOLVER, IL, R>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, a);
            GEMX_HIP_TRY(hipGetLastError());
        }
        h->ll = {2, SYS, CONV, LOAD, SOLVER, (int)IL, (int)sizeof(R), 0, BLOCK, K, 1, (long long)blocks, 0};
        return GEMX_OK;
    }
    if (sizeof(R) == 4 && K >= 2 && obs_every && h->use_pipe != 0 && !h->warned_fallback) {
        // A fused fp32 rollout that lands HERE runs the single-wave kernel, several times slower than the pipelined one at the same size
        // (round 4 verdict: nothing but gemx_last_launch() told).  Said once per handle, with the reason; GEMX_QUIET=1 silences it.
        h->warned_fallback = true;
        const char *why = "random initial states together with solver sub-steps or a custom constraint set, or the LDS footprint of this configuration";
        const char *q = getenv("GEMX_QUIET");
        if (q == nullptr || atoi(q) == 0)
            fprintf(stderr, "gemx: note: this handle's fused rollouts run the single-wave fallback kernel (%s); expect a fraction of the pipelined kernel's rate\n", why);
    }
    if (!h->attr_set) {
        GEMX_HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_max));
        h->attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLOCK), smem, st, a);
    GEMX_HIP_TRY(hipGetLastError());
    h->ll = {0, SYS, CONV, LOAD, SOLVER, (int)IL, (int)sizeof(R), 0, BLOCK, K, a.S, (long long)blocks, smem};
    return GEMX_OK;
}

// run-time -> compile-time dispatch on (load, solver, interlocking) for one (system, converter, dtype) unit.
// IL = false only exists for fp32 (the product path); the fp64 diagnostic build always takes the general code.
template <int SYS, int CONV, class R>
int launch_advance_unit(gemx_handle *h, const void *actions, int K, void *obs, uint8_t *done, int obs_every, hipStream_t st) {
    // (finite converters behind an RC supply need the leg states of the previous step for i_sup: the dead-time-free instantiation
    // keeps them too, from Stepper::legs_of -- so an RC supply no longer forces the slower IL code)
    const bool il = sizeof(R) == 8 || h->cfg.interlocking_time > 0.0;
    const int ld = h->cfg.load_kind, sv = h->cfg.solver_kind;
    // (gemx_create refuses interlocking_time > 0 for the EESM -- the reference's dead-time branch for that system cannot execute,
    // physical_systems.py:634 -- so its fp32 units carry no dead-time code)
    constexpr bool IL_BUILT = sizeof(R) == 8 || SYS != GEMX_SYS_EESM;
#ifdef GEMX_DEV_ONLY  // tools/dev_build.py --only LOAD,SOLVER,IL: ONE combination of the unit (a variant build for probes: seconds, not minutes)
#define GEMX_CASE(LD, SV)                                                                                                  \
    if constexpr (LD == GEMX_DEV_LOAD && SV == GEMX_DEV_SOLVER)                                                             \
        if (ld == LD && sv == SV && il == (GEMX_DEV_IL != 0))                                                              \
            return launch_advance_t<SYS, CONV, LD, SV, GEMX_DEV_IL != 0, R>(h, actions, K, obs, done, obs_every, st);
#else
#define GEMX_CASE(LD, SV)                                                                                                  \
    if (ld == LD && sv == SV) {                                                                                            \
        if constexpr (IL_BUILT) if (il) return launch_advance_t<SYS, CONV, LD, SV, true, R>(h, actions, K, obs, done, obs_every, st); \
        if constexpr (sizeof(R) == 4) return launch_advance_t<SYS, CONV, LD, SV, false, R>(h, actions, K, obs, done, obs_every, st); \
    }
#endif
    GEMX_CASE(GEMX_LOAD_CONST_SPEED, GEMX_SOLVER_EULER)
    GEMX_CASE(GEMX_LOAD_CONST_SPEED, GEMX_SOLVER_RK4)
    GEMX_CASE(GEMX_LOAD_CONST_SPEED, GEMX_SOLVER_DP5)
    GEMX_CASE(GEMX_LOAD_POLY_STATIC, GEMX_SOLVER_EULER)
    GEMX_CASE(GEMX_LOAD_POLY_STATIC, GEMX_SOLVER_RK4)
    GEMX_CASE(GEMX_LOAD_POLY_STATIC, GEMX_SOLVER_DP5)
#undef GEMX_CASE
    return fail(GEMX_ERR_ARG, "unsupported load/solver combination %d/%d", ld, sv);
}

}  // namespace gemx
