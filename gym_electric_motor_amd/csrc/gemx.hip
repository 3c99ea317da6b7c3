// gemx.hip -- MI355X (gfx950 / CDNA4) batched SCML physical-system stepper and its C ABI (include/gemx.h).
//
// What it replaces (reference = upb-lea/gym-electric-motor 3.0.2, paths relative to src/gym_electric_motor/):
//   SCMLSystem.simulate                          physical_systems/physical_systems.py:171-203
//   SynchronousMotorSystem.simulate              physical_systems/physical_systems.py:487-525
//   SquirrelCageInductionMotorSystem.simulate    physical_systems/physical_systems.py:771-814
//   *.reset                                      physical_systems/physical_systems.py:256-287, 527-561, 816-847
//   converters / motors / loads / solvers / constraints on that path (cited at each device function).
//
// Design (MI355X-first, not a translation of the Python):
//   * one lane = one env; one 64-lane wavefront = one workgroup; N envs advance in lockstep.
//   * ODE state lives in HBM as SoA rows [S_ode][N] (coalesced dword loads/stores); in the fused K-step
//     launch it stays in VGPRs between steps, so a step moves only action-in + observation-out + done.
//   * all motor/load/converter/limit parameters are uniform across envs -> passed by value as kernel
//     arguments (s_load into SGPRs through the scalar cache): 0 HBM bytes per env, no LDS needed for them.
//   * LDS is used where lanes must exchange data: the [N, S_out] row-per-env observation (the reference's
//     contract) is transposed through LDS so that a wavefront writes its contiguous 64*S_out*4-byte span
//     with 16-byte-per-lane stores instead of S_out stride-S_out dword stores.
//   * the electrical angle is kept as a 32-bit fixed-point fraction of a turn in the fp32 path: wrap to
//     (-pi, pi] is integer overflow (exact, as sin/cos are periodic), resolution 1.5e-9 rad independent of
//     how long the rollout is (the reference integrates epsilon unwrapped in fp64, physical_systems.py:520-522).
//   * converter dead time (interlocking) is a per-lane 1- or 2-segment integration; the second segment is an
//     exec-masked branch that the wave skips entirely (s_cbranch_execz) when no lane of the wave switches.
//   * no MFMA: ~100-200 flops and 41-117 bytes per env-step; the path is HBM/latency bound.
//
// There is no CPU fallback in this file or anywhere in the product.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

#include "gemx.h"

namespace gemx {

constexpr int BLOCK = 64;  // one wavefront per workgroup (gfx950 wave64)
constexpr double kTwoPi = 6.283185307179586476925286766559;
constexpr double kPi = 3.141592653589793238462643383279;

// ------------------------------------------------------------------------------------------------
// compile-time system traits
// ------------------------------------------------------------------------------------------------
template <int SYS> struct SysTraits;
template <> struct SysTraits<GEMX_SYS_DC_PERMEX> { static constexpr int ND = 2, NOUT = 5, HAS_ANGLE = 0; };   // omega, i
template <> struct SysTraits<GEMX_SYS_SYNC>      { static constexpr int ND = 3, NOUT = 14, HAS_ANGLE = 1; };  // omega, i_sd, i_sq (+eps)
template <> struct SysTraits<GEMX_SYS_SCIM>      { static constexpr int ND = 5, NOUT = 14, HAS_ANGLE = 1; };  // omega, i_sa, i_sb, psi_ra, psi_rb (+eps)

template <int CONV> struct ConvTraits;
template <> struct ConvTraits<GEMX_CONV_CONT_4QC>  { static constexpr int NACT = 1, DISCRETE = 0; };
template <> struct ConvTraits<GEMX_CONV_FINITE_B6> { static constexpr int NACT = 1, DISCRETE = 1; };
template <> struct ConvTraits<GEMX_CONV_CONT_B6>   { static constexpr int NACT = 3, DISCRETE = 0; };

// ------------------------------------------------------------------------------------------------
// uniform parameters (kernel argument, by value -> SGPRs)
// ------------------------------------------------------------------------------------------------
template <class R> struct DevParams {
    R m[16];      // non-zero entries of motor._model_constants, see pack_model()
    R tc0, tc1;   // torque coefficients
    R pole;       // d(eps)/dt = pole * omega
    R inv_j, la, lb, lc, omega_lim, lin_factor;  // PolynomialStaticLoad
    R u_sup;      // IdealVoltageSupply
    R il_ratio;   // interlocking_time / tau (continuous converters)
    R tau, t_il;  // control step, dead time
    R inv_ns;     // 1 / solver_nsteps
    R inv_lim[GEMX_MAX_OUT];
    R init[GEMX_MAX_ODE];  // [omega, motor states...] (angle separately)
    const R *cw;           // device array [2][GEMX_MAX_OUT]: generic constraint path 0/1 weights (limit | squared)
    int64_t init_angle_rep; // initial angle in Angle<R>::T representation (bit pattern)
    int32_t solver, nsteps, has_il, auto_reset, obs_layout;
    int32_t constr_kind;    // 0 none, 1 the system's default constraint (fast path), 2 generic weights
};

// ------------------------------------------------------------------------------------------------
// angle representation
// ------------------------------------------------------------------------------------------------
template <class R> struct Angle;

template <> struct Angle<float> {
    using T = int32_t;  // 2*pi / 2^32 rad per count, wraps by integer overflow
    static constexpr float kCountsPerRad = 683565275.57643158978229477811f;  // 2^32 / (2 pi)
    static constexpr float kRadPerCount = 1.4629180792671596e-9f;            // 2 pi / 2^32
    static __host__ __device__ T from_rad(double a) {
        double t = a / kTwoPi;
        t -= floor(t + 0.5);  // [-0.5, 0.5)
        long long c = llrint(t * 4294967296.0);
        return (T)(uint32_t)(unsigned long long)c;
    }
    static __host__ __device__ T from_bits(int64_t b) { return (T)(int32_t)b; }
    static __host__ int64_t to_bits(T a) { return (int64_t)a; }
    static __device__ __forceinline__ T advance(T a, float d_rad) {
        int32_t inc = __float2int_rn(d_rad * kCountsPerRad);
        return (T)((uint32_t)a + (uint32_t)inc);
    }
    static __device__ __forceinline__ float wrapped(T a) { return (float)a * kRadPerCount; }  // [-pi, pi]
    static __device__ __forceinline__ float to_rad(T a) { return wrapped(a); }
    // sin/cos of a fixed-point angle: quadrant from the top bits, Cephes single-precision minimax
    // polynomials on [-pi/4, pi/4] (abs error < 1.2e-7); ~20 VALU ops for both, no range-reduction branches.
    static __device__ __forceinline__ void sincos(T a, float &s, float &c) {
        uint32_t ua = (uint32_t)a + 0x20000000u;                    // + 1/8 turn
        uint32_t q = ua >> 30;                                      // quadrant 0..3
        int32_t r = (int32_t)(ua & 0x3FFFFFFFu) - 0x20000000;       // [-2^29, 2^29) counts == [-pi/4, pi/4)
        float x = (float)r * kRadPerCount;
        float z = x * x;
        float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, x, x);
        float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                        fmaf(-0.5f, z, 1.0f));
        float s1 = (q & 1u) ? cp : sp;
        float c1 = (q & 1u) ? sp : cp;
        s = (q & 2u) ? -s1 : s1;
        c = ((q + 1u) & 2u) ? -c1 : c1;
    }
};

template <> struct Angle<double> {
    using T = double;  // unwrapped radians, exactly as the reference integrates it
    static __host__ __device__ T from_rad(double a) { return a; }
    static __host__ __device__ T from_bits(int64_t b) { T r; memcpy(&r, &b, 8); return r; }
    static __host__ int64_t to_bits(T a) { int64_t b; memcpy(&b, &a, 8); return b; }
    static __device__ __forceinline__ T advance(T a, double d) { return a + d; }
    static __device__ __forceinline__ double wrapped(T a) {  // physical_systems.py:520-522
        double e = fmod(a, kTwoPi);
        if (e < 0) e += kTwoPi;
        if (e > kPi) e -= kTwoPi;
        return e;
    }
    static __device__ __forceinline__ double to_rad(T a) { return a; }
    static __device__ __forceinline__ void sincos(T a, double &s, double &c) { ::sincos(a, &s, &c); }
};

// ------------------------------------------------------------------------------------------------
// small math helpers
// ------------------------------------------------------------------------------------------------
template <class R> __device__ __forceinline__ R clip01(R x) { return fmin(fmax(x, R(0)), R(1)); }
template <class R> __device__ __forceinline__ R sgn(R x) { return x > R(0) ? R(1) : (x < R(0) ? R(-1) : R(0)); }  // np.sign

// Clarke / inverse Clarke (three_phase_motor.py:18-28, 31-54) and Park rotation (56-88)
template <class R> __device__ __forceinline__ void t23(R a, R b, R c, R &al, R &be) {
    al = R(2.0 / 3.0) * (a - R(0.5) * b - R(0.5) * c);
    be = R(0.57735026918962576451) * (b - c);  // 2/3 * sqrt(3)/2
}
template <class R> __device__ __forceinline__ void t32(R al, R be, R &a, R &b, R &c) {
    const R h = R(0.86602540378443864676) * be;
    a = al;
    b = R(-0.5) * al + h;
    c = R(-0.5) * al - h;
}

// ------------------------------------------------------------------------------------------------
// load: dω/dt (constant_speed_load.py:40-42; polynomial_static_load.py:62-66, 87-99)
// ------------------------------------------------------------------------------------------------
template <int LOAD, class R> __device__ __forceinline__ R load_ode(const DevParams<R> &P, R omega, R torque) {
    if (LOAD == GEMX_LOAD_CONST_SPEED) return R(0);
    R sign = sgn(omega);
    R a = fabs(omega) > P.omega_lim ? sign * P.la : P.lin_factor * omega;
    R tl = sign * P.lc * omega * omega + P.lb * omega + a;
    return (torque - tl) * P.inv_j;
}

// ------------------------------------------------------------------------------------------------
// motor torque and right-hand side.  y = [omega, motor states w/o angle], u = segment-constant input.
// The reference evaluates matmul(model_constants, feature_vector); here only the structurally non-zero
// entries are used (pack_model() rejects a matrix with any other non-zero entry).
// ------------------------------------------------------------------------------------------------
template <int SYS, int LOAD, class R> struct Motor;

template <int LOAD, class R> struct Motor<GEMX_SYS_DC_PERMEX, LOAD, R> {  // dc_permanently_excited_motor.py:67-84
    static __device__ __forceinline__ R torque(const DevParams<R> &P, const R (&y)[2]) { return P.tc0 * y[1]; }
    static __device__ __forceinline__ void rhs(const DevParams<R> &P, const R (&y)[2], const R (&u)[2], R (&dy)[2]) {
        dy[0] = load_ode<LOAD, R>(P, y[0], torque(P, y));
        dy[1] = P.m[0] * y[0] + P.m[1] * y[1] + P.m[2] * u[0];
    }
};
template <int LOAD, class R> struct Motor<GEMX_SYS_SYNC, LOAD, R> {  // synchronous_motor.py:143-168, permanent_magnet_synchronous_motor.py:107-139
    static __device__ __forceinline__ R torque(const DevParams<R> &P, const R (&y)[3]) { return (P.tc0 + P.tc1 * y[1]) * y[2]; }
    static __device__ __forceinline__ void rhs(const DevParams<R> &P, const R (&y)[3], const R (&u)[2], R (&dy)[3]) {
        const R w = y[0], id = y[1], iq = y[2];
        dy[0] = load_ode<LOAD, R>(P, w, torque(P, y));
        dy[1] = P.m[0] * id + P.m[1] * u[0] + P.m[2] * (w * iq);
        dy[2] = P.m[3] * w + P.m[4] * iq + P.m[5] * u[1] + P.m[6] * (w * id);
    }
};
template <int LOAD, class R> struct Motor<GEMX_SYS_SCIM, LOAD, R> {  // induction_motor.py:187-217, 236-248, 287-312; squirrel_cage_induction_motor.py:121-129
    static __device__ __forceinline__ R torque(const DevParams<R> &P, const R (&y)[5]) { return P.tc0 * (y[3] * y[2] - y[4] * y[1]); }
    static __device__ __forceinline__ void rhs(const DevParams<R> &P, const R (&y)[5], const R (&u)[2], R (&dy)[5]) {
        const R w = y[0], ia = y[1], ib = y[2], pa = y[3], pb = y[4];
        dy[0] = load_ode<LOAD, R>(P, w, torque(P, y));
        dy[1] = P.m[0] * ia + P.m[1] * pa + P.m[2] * (w * pb) + P.m[3] * u[0];
        dy[2] = P.m[4] * ib + P.m[5] * pb + P.m[6] * (w * pa) + P.m[7] * u[1];
        dy[3] = P.m[8] * ia + P.m[9] * pa + P.m[10] * (w * pb);
        dy[4] = P.m[11] * ib + P.m[12] * pb + P.m[13] * (w * pa);
    }
};

// ------------------------------------------------------------------------------------------------
// integrate one segment of length h; returns the angle increment  ∫ pole*omega dt  of the scheme
// (solvers.py:103-136 Euler; classical RK4; Dormand-Prince 5th-order weights).  The solver kind and the
// sub-step count are wave-uniform run-time values (scalar branches).
// ------------------------------------------------------------------------------------------------
template <int SYS, int LOAD, class R>
__device__ __forceinline__ R integrate(const DevParams<R> &P, R (&y)[SysTraits<SYS>::ND], const R (&u)[2], R h) {
    constexpr int ND = SysTraits<SYS>::ND;
    using M = Motor<SYS, LOAD, R>;
    const int ns = P.nsteps;
    const R hs = h * P.inv_ns;
    R wsum = R(0);  // sum over sub-steps of the omega quadrature
    for (int s = 0; s < ns; ++s) {
        R k1[ND], k2[ND], k3[ND], k4[ND], yt[ND];
        M::rhs(P, y, u, k1);
        if (P.solver == GEMX_SOLVER_EULER) {
            wsum += y[0];
#pragma unroll
            for (int i = 0; i < ND; ++i) y[i] = y[i] + k1[i] * hs;
        } else if (P.solver == GEMX_SOLVER_RK4) {
            R wq = y[0];
            const R hh = R(0.5) * hs;
#pragma unroll
            for (int i = 0; i < ND; ++i) yt[i] = y[i] + hh * k1[i];
            M::rhs(P, yt, u, k2);
            wq += R(2) * yt[0];
#pragma unroll
            for (int i = 0; i < ND; ++i) yt[i] = y[i] + hh * k2[i];
            M::rhs(P, yt, u, k3);
            wq += R(2) * yt[0];
#pragma unroll
            for (int i = 0; i < ND; ++i) yt[i] = y[i] + hs * k3[i];
            M::rhs(P, yt, u, k4);
            wq += yt[0];
            wsum += wq * R(1.0 / 6.0);
            const R h6 = hs * R(1.0 / 6.0);
#pragma unroll
            for (int i = 0; i < ND; ++i) y[i] = y[i] + h6 * (k1[i] + R(2) * (k2[i] + k3[i]) + k4[i]);
        } else {  // GEMX_SOLVER_DP5: one Dormand-Prince step, 5th-order solution, no error control
            R k5[ND], k6[ND];
            R wq = R(35.0 / 384.0) * y[0];
#pragma unroll
            for (int i = 0; i < ND; ++i) yt[i] = y[i] + hs * (R(1.0 / 5.0) * k1[i]);
            M::rhs(P, yt, u, k2);
#pragma unroll
            for (int i = 0; i < ND; ++i) yt[i] = y[i] + hs * (R(3.0 / 40.0) * k1[i] + R(9.0 / 40.0) * k2[i]);
            M::rhs(P, yt, u, k3);
            wq += R(500.0 / 1113.0) * yt[0];
#pragma unroll
            for (int i = 0; i < ND; ++i)
                yt[i] = y[i] + hs * (R(44.0 / 45.0) * k1[i] - R(56.0 / 15.0) * k2[i] + R(32.0 / 9.0) * k3[i]);
            M::rhs(P, yt, u, k4);
            wq += R(125.0 / 192.0) * yt[0];
#pragma unroll
            for (int i = 0; i < ND; ++i)
                yt[i] = y[i] + hs * (R(19372.0 / 6561.0) * k1[i] - R(25360.0 / 2187.0) * k2[i] +
                                     R(64448.0 / 6561.0) * k3[i] - R(212.0 / 729.0) * k4[i]);
            M::rhs(P, yt, u, k5);
            wq -= R(2187.0 / 6784.0) * yt[0];
#pragma unroll
            for (int i = 0; i < ND; ++i)
                yt[i] = y[i] + hs * (R(9017.0 / 3168.0) * k1[i] - R(355.0 / 33.0) * k2[i] + R(46732.0 / 5247.0) * k3[i] +
                                     R(49.0 / 176.0) * k4[i] - R(5103.0 / 18656.0) * k5[i]);
            M::rhs(P, yt, u, k6);
            wq += R(11.0 / 84.0) * yt[0];
            wsum += wq;
#pragma unroll
            for (int i = 0; i < ND; ++i)
                y[i] = y[i] + hs * (R(35.0 / 384.0) * k1[i] + R(500.0 / 1113.0) * k3[i] + R(125.0 / 192.0) * k4[i] -
                                    R(2187.0 / 6784.0) * k5[i] + R(11.0 / 84.0) * k6[i]);
        }
    }
    return P.pole * hs * wsum;
}

// ------------------------------------------------------------------------------------------------
// converters: normalised phase voltages for one segment
// ------------------------------------------------------------------------------------------------
// ContTwoQuadrantConverter via ContDynamicallyAveragedConverter (converters.py:144-158, 177-184, 425-427)
template <class R> __device__ __forceinline__ R cont_leg(const DevParams<R> &P, R duty, R i) {
    return clip01(duty - sgn(i) * P.il_ratio);
}
// FiniteTwoQuadrantConverter.convert (converters.py:277-285): leg state 1 -> 1, 2 -> 0, 0 (dead) -> freewheeling diode
// (conducts to the upper rail while i < 0).  Branch-free: returns +-0.5 * u_sup directly.
template <class R> __device__ __forceinline__ R fin_leg_u(uint32_t st, R i, R half_us) {
    const bool upper = (st == 1u) | ((st == 0u) & (i < R(0)));
    return upper ? half_us : -half_us;
}

// phase voltages of the B6 bridges for one segment
template <int CONV, class R>
__device__ __forceinline__ void b6_voltages(const DevParams<R> &P, const R (&act)[3], uint32_t leg_state, R ia, R ib, R ic,
                                            R &ua, R &ub, R &uc) {
    if (CONV == GEMX_CONV_CONT_B6) {  // converters.py:888-903
        ua = (cont_leg(P, clip01(R(0.5) * (act[0] + R(1))), ia) - R(0.5)) * P.u_sup;
        ub = (cont_leg(P, clip01(R(0.5) * (act[1] + R(1))), ib) - R(0.5)) * P.u_sup;
        uc = (cont_leg(P, clip01(R(0.5) * (act[2] + R(1))), ic) - R(0.5)) * P.u_sup;
    } else {  // converters.py:816-823; leg_state: 2 bits per leg, leg 0 in bits 0-1
        const R hu = R(0.5) * P.u_sup;
        ua = fin_leg_u<R>(leg_state & 3u, ia, hu);
        ub = fin_leg_u<R>((leg_state >> 2) & 3u, ib, hu);
        uc = fin_leg_u<R>((leg_state >> 4) & 3u, ic, hu);
    }
}

// Finite-B6C action -> per-leg sub-action (1 = upper, 2 = lower), converters.py:788-797, packed 2 bits per leg.
__device__ __forceinline__ uint32_t b6_subactions(uint32_t a) {
    return ((a & 4u) ? 1u : 2u) | (((a & 2u) ? 1u : 2u) << 2) | (((a & 1u) ? 1u : 2u) << 4);
}
// Interlocking (FiniteTwoQuadrantConverter._set_switching_pattern 300-310 + convert 270-276 as driven by
// *.simulate(), which passes the segment START time): a leg that changes between upper and lower goes to the
// dead state 0 for the WHOLE step (two segments [t, t+t_il], [t+t_il, t+tau]) and takes the new state on the
// next step.  Returns the leg states used during this step; `two` = this env integrates two segments.
__device__ __forceinline__ uint32_t b6_interlock(uint32_t prev, uint32_t want, bool &two) {
    uint32_t used = 0;
    two = false;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        uint32_t s = (prev >> (2 * l)) & 3u, a = (want >> (2 * l)) & 3u;
        bool trans = (s != 0u) && (a != s);
        two |= trans;
        used |= (trans ? 0u : a) << (2 * l);
    }
    return used;
}

// ------------------------------------------------------------------------------------------------
// one control step of one env: advances (y, ang, sw) and fills the normalised observation row.
// ------------------------------------------------------------------------------------------------
template <int SYS, int CONV, int LOAD, class R> struct Stepper;

// ---- DcMotorSystem + Cont-4QC (physical_systems.py:171-203; converters.py:481-491) --------------------------
template <int LOAD, class R> struct Stepper<GEMX_SYS_DC_PERMEX, GEMX_CONV_CONT_4QC, LOAD, R> {
    using AngT = typename Angle<R>::T;
    static __device__ __forceinline__ void step(const DevParams<R> &P, R (&y)[2], AngT &, uint32_t &, const R (&act)[3],
                                                uint32_t, R (&obs)[5]) {
        const R d0 = clip01(R(0.5) * (act[0] + R(1)));
        const R d1 = clip01(R(-0.5) * (act[0] - R(1)));
        const R un = cont_leg(P, d0, y[1]) - cont_leg(P, d1, y[1]);  // both sub-converters see the same i (line 483)
        R u[2] = {un * P.u_sup, R(0)};
        integrate<GEMX_SYS_DC_PERMEX, LOAD, R>(P, y, u, P.tau);
        obs[0] = y[0] * P.inv_lim[0];
        obs[1] = Motor<GEMX_SYS_DC_PERMEX, LOAD, R>::torque(P, y) * P.inv_lim[1];
        obs[2] = y[1] * P.inv_lim[2];
        obs[3] = u[0] * P.inv_lim[3];
        obs[4] = P.u_sup * P.inv_lim[4];
    }
    // default constraint of the DC envs: LimitConstraint('i') (cont_cc_permex_dc_env.py:104)
    static __device__ __forceinline__ bool default_done(const R (&obs)[5]) { return fabs(obs[2]) > R(1); }
};

// ---- SynchronousMotorSystem (physical_systems.py:487-525), control_space 'abc' ---------------------------------
template <int CONV, int LOAD, class R> struct Stepper<GEMX_SYS_SYNC, CONV, LOAD, R> {
    using AngT = typename Angle<R>::T;
    static __device__ __forceinline__ void step(const DevParams<R> &P, R (&y)[3], AngT &ang, uint32_t &sw, const R (&act)[3],
                                                uint32_t dact, R (&obs)[14]) {
        R s, c;
        Angle<R>::sincos(ang, s, c);
        uint32_t legs = 0;
        bool two = false;
        if (CONV == GEMX_CONV_FINITE_B6) {
            legs = b6_subactions(dact);
            if (P.has_il) { legs = b6_interlock(sw, legs, two); sw = legs; }
        }
        R ua, ub, uc, u[2];
        auto segment = [&](R h) {
            // i_in = T32(Q(i_dq, eps)) (line 493/505); only its sign matters (dead legs / cont. interlocking)
            R ial = c * y[1] - s * y[2], ibe = s * y[1] + c * y[2], ia, ib, ic;
            t32(ial, ibe, ia, ib, ic);
            b6_voltages<CONV, R>(P, act, legs, ia, ib, ic, ua, ub, uc);
            R ual, ube;
            t23(ua, ub, uc, ual, ube);
            u[0] = c * ual + s * ube;  // Q^-1(., eps): u_dq frozen at the segment-start angle (line 501/511)
            u[1] = -s * ual + c * ube;
            R deps = integrate<GEMX_SYS_SYNC, LOAD, R>(P, y, u, h);
            ang = Angle<R>::advance(ang, deps);
        };
        segment(two ? P.t_il : P.tau);
        if (two) {  // exec-masked; the whole wave skips it when no lane switches (s_cbranch_execz)
            Angle<R>::sincos(ang, s, c);  // eps / i_in refreshed at the switching instant (lines 504-505)
            segment(P.tau - P.t_il);
        }
        // outputs: i_abc from the NEW i_dq with the angle of the last segment start (line 519, reference quirk)
        R ial = c * y[1] - s * y[2], ibe = s * y[1] + c * y[2], ia, ib, ic;
        t32(ial, ibe, ia, ib, ic);
        obs[0] = y[0] * P.inv_lim[0];
        obs[1] = Motor<GEMX_SYS_SYNC, LOAD, R>::torque(P, y) * P.inv_lim[1];
        obs[2] = ia * P.inv_lim[2];
        obs[3] = ib * P.inv_lim[3];
        obs[4] = ic * P.inv_lim[4];
        obs[5] = y[1] * P.inv_lim[5];
        obs[6] = y[2] * P.inv_lim[6];
        obs[7] = ua * P.inv_lim[7];
        obs[8] = ub * P.inv_lim[8];
        obs[9] = uc * P.inv_lim[9];
        obs[10] = u[0] * P.inv_lim[10];
        obs[11] = u[1] * P.inv_lim[11];
        obs[12] = Angle<R>::wrapped(ang) * P.inv_lim[12];
        obs[13] = P.u_sup * P.inv_lim[13];
    }
    // default constraint: SquaredConstraint(('i_sq','i_sd')) (finite_cc_pmsm_env.py:106)
    static __device__ __forceinline__ bool default_done(const R (&obs)[14]) { return obs[5] * obs[5] + obs[6] * obs[6] > R(1); }
};

// ---- SquirrelCageInductionMotorSystem (physical_systems.py:771-814), control_space 'abc' -----------------------
template <int CONV, int LOAD, class R> struct Stepper<GEMX_SYS_SCIM, CONV, LOAD, R> {
    using AngT = typename Angle<R>::T;
    // cos/sin of the rotor-flux angle eps_fs = atan2(psi_b, psi_a) (calculate_field_angle, 765-769) without atan2
    static __device__ __forceinline__ void field_angle(R pa, R pb, R &s, R &c) {
        R n2 = pa * pa + pb * pb;
        if (n2 < R(1e-30)) { pa *= R(1e18); pb *= R(1e18); n2 = pa * pa + pb * pb; }
        if (n2 > R(0)) {
            R rn = R(1) / sqrt(n2);
            c = pa * rn;
            s = pb * rn;
        } else {
            c = R(1);
            s = R(0);  // atan2(0, 0) = 0
        }
    }
    static __device__ __forceinline__ void step(const DevParams<R> &P, R (&y)[5], AngT &ang, uint32_t &sw, const R (&act)[3],
                                                uint32_t dact, R (&obs)[14]) {
        R s, c;
        field_angle(y[3], y[4], s, c);
        uint32_t legs = 0;
        bool two = false;
        if (CONV == GEMX_CONV_FINITE_B6) {
            legs = b6_subactions(dact);
            if (P.has_il) { legs = b6_interlock(sw, legs, two); sw = legs; }
        }
        R ua, ub, uc, u[2];
        auto segment = [&](R h) {
            R ia, ib, ic;
            t32(y[1], y[2], ia, ib, ic);  // i_in = T32(i_alphabeta) (line 780/792)
            b6_voltages<CONV, R>(P, act, legs, ia, ib, ic, ua, ub, uc);
            t23(ua, ub, uc, u[0], u[1]);  // u_alphabeta constant over the segment (line 788/799)
            R deps = integrate<GEMX_SYS_SCIM, LOAD, R>(P, y, u, h);
            ang = Angle<R>::advance(ang, deps);
        };
        segment(two ? P.t_il : P.tau);
        if (two) {
            field_angle(y[3], y[4], s, c);  // line 791
            segment(P.tau - P.t_il);
        }
        // i_dq = Q^-1(i_alphabeta_new, eps_fs of the last segment start) (line 806, reference quirk);
        // i_abc = T32(Q(i_dq, eps_fs)) == T32(i_alphabeta_new) (line 807); u_dq = Q^-1(u_alphabeta, eps_fs) (798)
        R ia, ib, ic;
        t32(y[1], y[2], ia, ib, ic);
        obs[0] = y[0] * P.inv_lim[0];
        obs[1] = Motor<GEMX_SYS_SCIM, LOAD, R>::torque(P, y) * P.inv_lim[1];
        obs[2] = ia * P.inv_lim[2];
        obs[3] = ib * P.inv_lim[3];
        obs[4] = ic * P.inv_lim[4];
        obs[5] = (c * y[1] + s * y[2]) * P.inv_lim[5];
        obs[6] = (-s * y[1] + c * y[2]) * P.inv_lim[6];
        obs[7] = ua * P.inv_lim[7];
        obs[8] = ub * P.inv_lim[8];
        obs[9] = uc * P.inv_lim[9];
        obs[10] = (c * u[0] + s * u[1]) * P.inv_lim[10];
        obs[11] = (-s * u[0] + c * u[1]) * P.inv_lim[11];
        obs[12] = Angle<R>::wrapped(ang) * P.inv_lim[12];
        obs[13] = P.u_sup * P.inv_lim[13];
    }
    // default constraint: SquaredConstraint(('i_sq','i_sd')) (cont_sc_scim_env.py:111)
    static __device__ __forceinline__ bool default_done(const R (&obs)[14]) { return obs[5] * obs[5] + obs[6] * obs[6] > R(1); }
};

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

// ------------------------------------------------------------------------------------------------
// kernel arguments
// ------------------------------------------------------------------------------------------------
template <class R> struct KArgs {
    DevParams<R> P;
    R *state;                       // [ND][N]
    typename Angle<R>::T *angle;    // [N] (systems with an angle)
    uint8_t *sw;                    // [N] packed leg states (Finite-B6C with interlocking)
    const unsigned char *actions;   // [K][N][A] R  |  [K][N] uint8
    R *obs;                         // [K][N][NOUT] | [K][NOUT][N]  (or a single step's worth if !obs_every)
    uint8_t *done;                  // [K][N] | [N]
    uint32_t *err;                  // device error word (bit 0: discrete action out of range)
    int64_t N;
    int32_t K, obs_every;
    int32_t S;                      // control steps per I/O block (LDS ring depth)
    int32_t coop;                   // 1: action rows / done rows of full blocks are 16-byte aligned -> cooperative staging
    int32_t obs_vec;                // 1: observation rows of full blocks are 16-byte aligned -> 16-byte stores
};

extern __shared__ __attribute__((aligned(16))) unsigned char gemx_smem[];

constexpr int MAX_ACT_CHUNKS = 12;  // upper bound of 16-byte chunks of staged actions per lane and I/O block
constexpr int MAX_STEPS_PER_BLOCK = 32;
// chunks per lane needed to stage MAX_STEPS_PER_BLOCK steps of a row made of `cpr` 16-byte chunks
__host__ __device__ constexpr int act_chunks(int cpr) {
    return (MAX_STEPS_PER_BLOCK * cpr + BLOCK - 1) / BLOCK < MAX_ACT_CHUNKS ? (MAX_STEPS_PER_BLOCK * cpr + BLOCK - 1) / BLOCK : MAX_ACT_CHUNKS;
}

// S control steps of one I/O block.  COOP: actions come from the LDS tile (no global memory access at all in
// this loop); otherwise straight from global memory (K == 1, tail workgroup, unaligned tensors).  The two variants
// are separate instantiations on purpose: a pointer that may be LDS or global would compile to FLAT loads, whose
// s_waitcnt covers vmcnt as well and would wait for every outstanding observation store.
template <bool COOP, int SYS, int CONV, int LOAD, class R>
__device__ __forceinline__ void compute_block(const KArgs<R> &a, R (&y)[SysTraits<SYS>::ND], typename Angle<R>::T &ang, uint32_t &sw,
                                              R (&obs)[SysTraits<SYS>::NOUT], uint32_t &done_or, uint32_t &bad_action, R *ring,
                                              const unsigned char *atile, unsigned char *donebuf, int k0, int sb, int tid,
                                              int64_t e, typename Angle<R>::T init_ang) {
    constexpr int ND = SysTraits<SYS>::ND, NOUT = SysTraits<SYS>::NOUT, NACT = ConvTraits<CONV>::NACT;
    constexpr bool DISCRETE = ConvTraits<CONV>::DISCRETE;
    constexpr int ABYTES = DISCRETE ? 1 : NACT * (int)sizeof(R);
    constexpr int ROWB = BLOCK * ABYTES;
    using ST = Stepper<SYS, CONV, LOAD, R>;
    const DevParams<R> &P = a.P;
    for (int s = 0; s < sb; ++s) {
        R act[3] = {R(0), R(0), R(0)};
        uint32_t dact = 0;
        if (COOP) {
            if (DISCRETE) dact = atile[s * ROWB + tid];
            else {
#pragma unroll
                for (int i = 0; i < NACT; ++i) act[i] = reinterpret_cast<const R *>(atile + s * ROWB)[tid * NACT + i];
            }
        } else {
            const unsigned char *g = a.actions + ((int64_t)(k0 + s) * a.N + e) * ABYTES;
            if (DISCRETE) dact = *g;
            else {
#pragma unroll
                for (int i = 0; i < NACT; ++i) act[i] = reinterpret_cast<const R *>(g)[i];
            }
        }
        if (DISCRETE) { bad_action |= dact > 7u; dact &= 7u; }
        ST::step(P, y, ang, sw, act, dact, obs);
        const bool done = constraint_done<ST, NOUT, R>(P, obs);
        done_or |= done ? 1u : 0u;
        if (a.obs_every) {
            if (P.obs_layout == GEMX_OBS_AOS) {
#pragma unroll
                for (int j = 0; j < NOUT; ++j) ring[(s * BLOCK + tid) * NOUT + j] = obs[j];
            } else {
#pragma unroll
                for (int j = 0; j < NOUT; ++j) ring[(s * NOUT + j) * BLOCK + tid] = obs[j];
            }
            donebuf[s * BLOCK + tid] = done ? 1 : 0;
        }
        if (done && P.auto_reset) {  // `if terminated: env.reset()`; switching state survives (converters.py:45-54)
#pragma unroll
            for (int j = 0; j < ND; ++j) y[j] = P.init[j];
            ang = init_ang;
        }
    }
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
template <int SYS, int CONV, int LOAD, class R>
__global__ __launch_bounds__(BLOCK) void advance_kernel(const KArgs<R> a) {
    constexpr int ND = SysTraits<SYS>::ND, NOUT = SysTraits<SYS>::NOUT, NACT = ConvTraits<CONV>::NACT;
    constexpr bool DISCRETE = ConvTraits<CONV>::DISCRETE;
    constexpr int ABYTES = DISCRETE ? 1 : NACT * (int)sizeof(R);  // action bytes per env and step
    constexpr int ROWB = BLOCK * ABYTES;                           // action bytes per 64-env row
    constexpr int CPR = ROWB / 16;                                 // 16-byte chunks per action row
    constexpr int VEC = 16 / sizeof(R);
    constexpr int ROWV = BLOCK * NOUT / VEC;                       // 16-byte chunks per observation row
    constexpr int NCH = act_chunks(CPR);                           // 16-byte action chunks per lane and I/O block
    using AngT = typename Angle<R>::T;
    using ST = Stepper<SYS, CONV, LOAD, R>;
    using V = typename std::conditional<sizeof(R) == 4, float4, double2>::type;

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
    const bool coop = a.coop && full && K > 1;   // cooperative action staging for this workgroup (uniform)

    R y[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) y[j] = a.state[(int64_t)j * N + e];
    AngT ang = AngT(0);
    if (SysTraits<SYS>::HAS_ANGLE) ang = a.angle[e];
    uint32_t sw = 0;
    const bool use_sw = (CONV == GEMX_CONV_FINITE_B6) && P.has_il;
    if (use_sw) sw = a.sw[e];
    const AngT init_ang = Angle<R>::from_bits(P.init_angle_rep);

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

        // 2. compute: no global memory traffic in here when coop
        const unsigned char *atile = actbuf + (size_t)half * S * ROWB;
        if (coop) compute_block<true, SYS, CONV, LOAD, R>(a, y, ang, sw, obs, done_or, bad_action, ring, atile, donebuf, k0, sb, tid, e, init_ang);
        else compute_block<false, SYS, CONV, LOAD, R>(a, y, ang, sw, obs, done_or, bad_action, ring, atile, donebuf, k0, sb, tid, e, init_ang);
        __syncthreads();

        // 3. park the prefetched tile
        if (coop && sb_next > 0) GEMX_TILE_PARK(half ^ 1, sb_next);

        // 4. flush the rings
        if (a.obs_every) {
            if (P.obs_layout == GEMX_OBS_AOS) {
                if (a.obs_vec) {
                    const int nvec = rows * NOUT / VEC;  // == ROWV for full blocks
                    for (int s = 0; s < sb; ++s) {
                        V *gv = reinterpret_cast<V *>(a.obs + ((int64_t)(k0 + s) * N + blk0) * NOUT);
                        const V *lv = reinterpret_cast<const V *>(ring + (size_t)s * BLOCK * NOUT);
#pragma unroll
                        for (int i = 0; i < (ROWV + BLOCK - 1) / BLOCK; ++i) {
                            const int idx = tid + i * BLOCK;
                            if (idx < nvec) gv[idx] = lv[idx];
                        }
                        for (int idx = nvec * VEC + tid; idx < rows * NOUT; idx += BLOCK)
                            a.obs[((int64_t)(k0 + s) * N + blk0) * NOUT + idx] = ring[(size_t)s * BLOCK * NOUT + idx];
                    }
                } else {
                    for (int s = 0; s < sb; ++s)
                        for (int idx = tid; idx < rows * NOUT; idx += BLOCK)
                            a.obs[((int64_t)(k0 + s) * N + blk0) * NOUT + idx] = ring[(size_t)s * BLOCK * NOUT + idx];
                }
            } else if (valid) {
                for (int s = 0; s < sb; ++s) {
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) a.obs[((int64_t)(k0 + s) * NOUT + j) * N + env] = ring[(s * NOUT + j) * BLOCK + tid];
                }
            }
            if (a.done != nullptr) {
                if (a.coop && full) {  // done rows are 64 contiguous bytes: 4 x 16-byte chunks per row
                    const int nchunk = sb * (BLOCK / 16);
                    for (int idx = tid; idx < nchunk; idx += BLOCK) {
                        const int row = idx >> 2, col = idx & 3;
                        *reinterpret_cast<uint4 *>(a.done + (int64_t)(k0 + row) * N + blk0 + col * 16) =
                            *reinterpret_cast<const uint4 *>(donebuf + row * BLOCK + col * 16);
                    }
                } else if (valid) {
                    for (int s = 0; s < sb; ++s) a.done[(int64_t)(k0 + s) * N + env] = donebuf[s * BLOCK + tid];
                }
            }
        }
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
        if (use_sw) a.sw[env] = (uint8_t)sw;
    }
    if (bad_action && valid) atomicOr(a.err, 1u);
#undef GEMX_TILE_LOAD
#undef GEMX_TILE_PARK
}

// reset: masked envs back to the initial ODE state; optional broadcast of the reset observation
template <class R>
__global__ void reset_kernel(R *state, typename Angle<R>::T *angle, const uint8_t *mask, R *obs, int64_t N, int nd, int nout,
                             int has_angle, int obs_layout, DevParams<R> P, const R *reset_obs) {
    int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    if (mask != nullptr && mask[env] == 0) return;
    for (int j = 0; j < nd; ++j) state[(int64_t)j * N + env] = P.init[j];
    if (has_angle) angle[env] = Angle<R>::from_bits(P.init_angle_rep);
    if (obs != nullptr) {
        for (int j = 0; j < nout; ++j) {
            if (obs_layout == GEMX_OBS_AOS) obs[env * nout + j] = reset_obs[j];
            else obs[(int64_t)j * N + env] = reset_obs[j];
        }
    }
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

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(x)                                                                                             \
    do {                                                                                                       \
        hipError_t _e = (x);                                                                                   \
        if (_e != hipSuccess) return fail(GEMX_ERR_DEVICE, "%s failed: %s", #x, hipGetErrorString(_e));        \
    } while (0)

struct gemx_handle {
    gemx_config cfg;
    int64_t n;
    int device;
    int nd, nout, nact, has_angle;
    DevParams<float> pf;
    DevParams<double> pd;
    void *state = nullptr;   // [nd][n] R
    void *angle = nullptr;   // [n] int32 | double
    uint8_t *sw = nullptr;   // [n]
    uint32_t *err = nullptr;
    void *reset_obs_dev = nullptr;  // [nout] R
    void *cw_dev = nullptr;         // [2][GEMX_MAX_OUT] R constraint weights
    double reset_obs[GEMX_MAX_OUT];
    int n_cu = 256;
    size_t lds_max = 160 * 1024;
    int steps_per_block = 0;  // 0 = heuristic
};

template <class R> static const DevParams<R> &params_of(const gemx_handle *h);
template <> const DevParams<float> &params_of<float>(const gemx_handle *h) { return h->pf; }
template <> const DevParams<double> &params_of<double>(const gemx_handle *h) { return h->pd; }

// which entries of the reference's model-constant matrix each system uses, in the order of DevParams::m
static const int DC_IDX[][2] = {{0, 0}, {0, 1}, {0, 2}};
static const int SYNC_IDX[][2] = {{0, 1}, {0, 3}, {0, 6}, {1, 0}, {1, 2}, {1, 4}, {1, 5}};
static const int SCIM_IDX[][2] = {{0, 1}, {0, 3}, {0, 6}, {0, 7}, {1, 2}, {1, 4}, {1, 5}, {1, 8},
                                  {2, 1}, {2, 3}, {2, 6}, {3, 2}, {3, 4}, {3, 5}};

static int pack_model(const gemx_config &c, double *m, double *pole) {
    const int(*idx)[2];
    int n, pole_row, rows, cols;
    switch (c.system_kind) {
        case GEMX_SYS_DC_PERMEX: idx = DC_IDX; n = 3; pole_row = -1; rows = 1; cols = 3; break;
        case GEMX_SYS_SYNC: idx = SYNC_IDX; n = 7; pole_row = 2; rows = 3; cols = 7; break;
        case GEMX_SYS_SCIM: idx = SCIM_IDX; n = 14; pole_row = 4; rows = 5; cols = 9; break;  // u_r columns: zero rotor voltage
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
    for (int i = 0; i < 16; ++i) P.m[i] = (R)m[i];
    P.tc0 = (R)c.torque_coef[0];
    P.tc1 = (R)c.torque_coef[1];
    P.pole = (R)pole;
    P.inv_j = (R)(c.j_total > 0 ? 1.0 / c.j_total : 0.0);
    P.la = (R)c.load_a; P.lb = (R)c.load_b; P.lc = (R)c.load_c;
    // PolynomialStaticLoad.set_j_rotor, polynomial_static_load.py:62-66
    P.omega_lim = (R)(c.j_total > 0 ? c.load_a / c.j_total * c.tau_decay : 0.0);
    P.lin_factor = (R)(c.tau_decay > 0 ? c.j_total / c.tau_decay : 0.0);
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
    P.solver = c.solver_kind;
    P.nsteps = c.solver_nsteps;
    P.has_il = (c.converter_kind == GEMX_CONV_FINITE_B6 && c.interlocking_time > 0.0) ? 1 : 0;
    P.auto_reset = c.auto_reset;
    P.obs_layout = c.obs_layout;
    // the env's default constraint gets the 3-instruction fast path (Stepper::default_done)
    const uint32_t def_limit = c.system_kind == GEMX_SYS_DC_PERMEX ? (1u << 2) : 0u;
    const uint32_t def_sq = c.system_kind == GEMX_SYS_DC_PERMEX ? 0u : ((1u << 5) | (1u << 6));
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
    if (c.system_kind == GEMX_SYS_DC_PERMEX) {
        o[0] = y[0]; o[1] = c.torque_coef[0] * y[1]; o[2] = y[1]; o[3] = 0.0 * us; o[4] = us;
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

template <class R> static int launch_reset(gemx_handle *h, const uint8_t *mask, void *obs, hipStream_t st) {
    using AngT = typename Angle<R>::T;
    const DevParams<R> &P = params_of<R>(h);
    int64_t blocks = (h->n + 255) / 256;
    hipLaunchKernelGGL(reset_kernel<R>, dim3((unsigned)blocks), dim3(256), 0, st, (R *)h->state, (AngT *)h->angle, mask, (R *)obs, h->n,
                       h->nd, h->nout, h->has_angle, h->cfg.obs_layout, P, (const R *)h->reset_obs_dev);
    HIP_TRY(hipGetLastError());
    return GEMX_OK;
}

// I/O block depth S (control steps staged in LDS between global-memory bursts): as deep as the LDS allows for the
// number of workgroups that should be co-resident per CU, capped by the per-lane action-prefetch registers.
static int choose_steps_per_block(const gemx_handle *h, int K, int es, int abytes) {
    if (K <= 1) return 1;
    const size_t per_step = (size_t)BLOCK * h->nout * es + 2 * (size_t)BLOCK * abytes + BLOCK;
    int S = h->steps_per_block;
    if (S <= 0) {
        const int64_t nblocks = (h->n + BLOCK - 1) / BLOCK;
        int64_t per_cu = (nblocks + h->n_cu - 1) / h->n_cu;
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        S = (int)((h->lds_max - 1024) / per_cu / per_step);
        if (S > MAX_STEPS_PER_BLOCK) S = MAX_STEPS_PER_BLOCK;
    }
    const int cpr = BLOCK * abytes / 16;
    const int s_regs = act_chunks(cpr) * BLOCK / cpr;  // steps whose action rows fit the per-lane prefetch registers
    if (S > s_regs) S = s_regs;
    const int s_lds = (int)((h->lds_max - 256) / per_step);
    if (S > s_lds) S = s_lds;
    if (S > K) S = K;
    if (S < 1) S = 1;
    return S;
}

template <int SYS, int CONV, int LOAD, class R>
static int launch_advance_t(gemx_handle *h, const void *actions, int K, void *obs, uint8_t *done, int obs_every, hipStream_t st) {
    constexpr int ABYTES = ConvTraits<CONV>::DISCRETE ? 1 : ConvTraits<CONV>::NACT * (int)sizeof(R);
    KArgs<R> a;
    a.P = params_of<R>(h);
    a.state = (R *)h->state;
    a.angle = (typename Angle<R>::T *)h->angle;
    a.sw = h->sw;
    a.actions = (const unsigned char *)actions;
    a.obs = (R *)obs;
    a.done = done;
    a.err = h->err;
    a.N = h->n;
    a.K = K;
    a.obs_every = obs_every;
    a.S = choose_steps_per_block(h, K, (int)sizeof(R), ABYTES);
    // 16-byte alignment of every full block's rows: row starts are (k*N + blk0) * bytes_per_env with blk0 % 64 == 0
    a.coop = (((uintptr_t)actions & 15u) == 0 && ((size_t)h->n * ABYTES) % 16 == 0 &&
              (done == nullptr || (((uintptr_t)done & 15u) == 0 && (size_t)h->n % 16 == 0))) ? 1 : 0;
    a.obs_vec = (((size_t)h->n * h->nout * sizeof(R)) % 16 == 0) ? 1 : 0;
    size_t smem = (size_t)a.S * BLOCK * h->nout * sizeof(R) + 2 * (size_t)a.S * BLOCK * ABYTES + (size_t)a.S * BLOCK;
    smem = (smem + 15) & ~(size_t)15;
    auto kern = advance_kernel<SYS, CONV, LOAD, R>;
    static size_t attr_set = 0;  // per instantiation: largest dynamic-LDS size enabled so far
    if (smem > attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_max));
        attr_set = h->lds_max;
    }
    int64_t blocks = (h->n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BLOCK), smem, st, a);
    HIP_TRY(hipGetLastError());
    return GEMX_OK;
}

template <int SYS, int CONV, class R>
static int launch_advance_l(gemx_handle *h, const void *actions, int K, void *obs, uint8_t *done, int obs_every, hipStream_t st) {
    if (h->cfg.load_kind == GEMX_LOAD_CONST_SPEED)
        return launch_advance_t<SYS, CONV, GEMX_LOAD_CONST_SPEED, R>(h, actions, K, obs, done, obs_every, st);
    return launch_advance_t<SYS, CONV, GEMX_LOAD_POLY_STATIC, R>(h, actions, K, obs, done, obs_every, st);
}

template <class R>
static int launch_advance(gemx_handle *h, const void *actions, int K, void *obs, uint8_t *done, int obs_every, hipStream_t st) {
    const int s = h->cfg.system_kind, c = h->cfg.converter_kind;
    if (s == GEMX_SYS_DC_PERMEX && c == GEMX_CONV_CONT_4QC) return launch_advance_l<GEMX_SYS_DC_PERMEX, GEMX_CONV_CONT_4QC, R>(h, actions, K, obs, done, obs_every, st);
    if (s == GEMX_SYS_SYNC && c == GEMX_CONV_FINITE_B6) return launch_advance_l<GEMX_SYS_SYNC, GEMX_CONV_FINITE_B6, R>(h, actions, K, obs, done, obs_every, st);
    if (s == GEMX_SYS_SYNC && c == GEMX_CONV_CONT_B6) return launch_advance_l<GEMX_SYS_SYNC, GEMX_CONV_CONT_B6, R>(h, actions, K, obs, done, obs_every, st);
    if (s == GEMX_SYS_SCIM && c == GEMX_CONV_FINITE_B6) return launch_advance_l<GEMX_SYS_SCIM, GEMX_CONV_FINITE_B6, R>(h, actions, K, obs, done, obs_every, st);
    if (s == GEMX_SYS_SCIM && c == GEMX_CONV_CONT_B6) return launch_advance_l<GEMX_SYS_SCIM, GEMX_CONV_CONT_B6, R>(h, actions, K, obs, done, obs_every, st);
    return fail(GEMX_ERR_ARG, "unsupported system/converter combination %d/%d", s, c);
}

static int elem_size(const gemx_handle *h) { return h->cfg.dtype == GEMX_F64 ? 8 : 4; }

extern "C" {

int gemx_abi_version(void) { return GEMX_ABI_VERSION; }
int gemx_sizeof_config(void) { return (int)sizeof(gemx_config); }
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
    if (!(cfg->tau > 0)) return fail(GEMX_ERR_ARG, "tau must be positive");
    if (cfg->interlocking_time < 0 || cfg->interlocking_time >= cfg->tau)
        return fail(GEMX_ERR_ARG, "interlocking_time must be in [0, tau)");
    if (cfg->solver_kind < GEMX_SOLVER_EULER || cfg->solver_kind > GEMX_SOLVER_DP5) return fail(GEMX_ERR_ARG, "unknown solver_kind");
    if (cfg->solver_nsteps < 1 || cfg->solver_nsteps > 1024) return fail(GEMX_ERR_ARG, "solver_nsteps must be in [1, 1024]");
    if (cfg->dtype != GEMX_F32 && cfg->dtype != GEMX_F64) return fail(GEMX_ERR_ARG, "unknown dtype");
    if (cfg->obs_layout != GEMX_OBS_AOS && cfg->obs_layout != GEMX_OBS_SOA) return fail(GEMX_ERR_ARG, "unknown obs_layout");
    if (cfg->load_kind != GEMX_LOAD_CONST_SPEED && cfg->load_kind != GEMX_LOAD_POLY_STATIC) return fail(GEMX_ERR_ARG, "unknown load_kind");
    if (cfg->load_kind == GEMX_LOAD_POLY_STATIC && !(cfg->j_total > 0 && cfg->tau_decay > 0))
        return fail(GEMX_ERR_ARG, "PolynomialStaticLoad needs j_total > 0 and tau_decay > 0");
    const int s = cfg->system_kind, c = cfg->converter_kind;
    const bool combo = (s == GEMX_SYS_DC_PERMEX && c == GEMX_CONV_CONT_4QC) ||
                       ((s == GEMX_SYS_SYNC || s == GEMX_SYS_SCIM) && (c == GEMX_CONV_FINITE_B6 || c == GEMX_CONV_CONT_B6));
    if (!combo) return fail(GEMX_ERR_ARG, "unsupported system/converter combination %d/%d", s, c);

    gemx_handle *h = new (std::nothrow) gemx_handle();
    if (!h) return fail(GEMX_ERR_ALLOC, "out of host memory");
    h->cfg = *cfg;
    h->n = n_envs;
    h->device = device;
    h->nd = s == GEMX_SYS_DC_PERMEX ? 2 : (s == GEMX_SYS_SYNC ? 3 : 5);
    h->nout = s == GEMX_SYS_DC_PERMEX ? 5 : 14;
    h->has_angle = s != GEMX_SYS_DC_PERMEX;
    h->nact = c == GEMX_CONV_CONT_B6 ? 3 : 1;
    for (int i = 0; i < h->nout; ++i)
        if (!(cfg->limits[i] > 0)) { delete h; return fail(GEMX_ERR_ARG, "limits[%d] must be positive", i); }
    if ((cfg->limit_mask | cfg->squared_mask) >> h->nout) { delete h; return fail(GEMX_ERR_ARG, "constraint mask has bits beyond S_out=%d", h->nout); }

    double m[16] = {0}, pole = 0;
    int rc = pack_model(*cfg, m, &pole);
    if (rc != GEMX_OK) { delete h; return rc; }

    // all argument validation is done; from here on a device is required
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        delete h;
        return fail(GEMX_ERR_DEVICE, "no HIP device visible: the gemx stepper has no CPU fallback");
    }
    if (device < 0 || device >= ndev) { delete h; return fail(GEMX_ERR_ARG, "device %d out of range (0..%d)", device, ndev - 1); }
    if (hipSetDevice(device) != hipSuccess) { delete h; return fail(GEMX_ERR_DEVICE, "hipSetDevice(%d) failed", device); }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
            if (prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
            if (prop.sharedMemPerBlock >= 64 * 1024) h->lds_max = prop.sharedMemPerBlock;
        }
        const char *ev = getenv("GEMX_STEPS_PER_BLOCK");
        if (ev) h->steps_per_block = atoi(ev);
    }
    host_reset_obs(*h, m);

    const size_t es = (size_t)elem_size(h);
    auto cleanup = [&](int code) { gemx_destroy(h); return code; };
    if (hipMalloc(&h->state, es * h->nd * (size_t)h->n) != hipSuccess) return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(state) failed"));
    if (h->has_angle && hipMalloc(&h->angle, (cfg->dtype == GEMX_F64 ? 8 : 4) * (size_t)h->n) != hipSuccess)
        return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(angle) failed"));
    if (hipMalloc((void **)&h->sw, (size_t)h->n) != hipSuccess) return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(sw) failed"));
    if (hipMalloc((void **)&h->err, sizeof(uint32_t)) != hipSuccess) return cleanup(fail(GEMX_ERR_ALLOC, "hipMalloc(err) failed"));
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
    if (hipMemset(h->sw, 0, (size_t)h->n) != hipSuccess || hipMemset(h->err, 0, sizeof(uint32_t)) != hipSuccess)
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
    *out = h;
    return GEMX_OK;
}

int gemx_destroy(gemx_handle *h) {
    if (!h) return GEMX_OK;
    (void)hipSetDevice(h->device);
    if (h->state) (void)hipFree(h->state);
    if (h->angle) (void)hipFree(h->angle);
    if (h->sw) (void)hipFree(h->sw);
    if (h->err) (void)hipFree(h->err);
    if (h->reset_obs_dev) (void)hipFree(h->reset_obs_dev);
    if (h->cw_dev) (void)hipFree(h->cw_dev);
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
    return h->cfg.converter_kind == GEMX_CONV_FINITE_B6 ? 1 : elem_size(h);
}
int gemx_reset_observation(const gemx_handle *h, double *obs_host) {
    if (!h || !obs_host) return fail(GEMX_ERR_ARG, "null argument");
    memcpy(obs_host, h->reset_obs, sizeof(double) * h->nout);
    return GEMX_OK;
}

int gemx_reset(gemx_handle *h, const uint8_t *mask_dev, void *obs_out_dev, void *stream) {
    if (!h) return fail(GEMX_ERR_ARG, "null handle");
    hipStream_t st = (hipStream_t)stream;
    return h->cfg.dtype == GEMX_F64 ? launch_reset<double>(h, mask_dev, obs_out_dev, st) : launch_reset<float>(h, mask_dev, obs_out_dev, st);
}

int gemx_rollout(gemx_handle *h, const void *actions_dev, int32_t K, void *obs_out_dev, uint8_t *done_out_dev, int32_t obs_every,
                 void *stream) {
    if (!h) return fail(GEMX_ERR_ARG, "null handle");
    if (!actions_dev || !obs_out_dev) return fail(GEMX_ERR_ARG, "actions_dev and obs_out_dev must not be null");
    if (K < 1) return fail(GEMX_ERR_ARG, "K must be >= 1");
    if (((uintptr_t)obs_out_dev & 15u) != 0) return fail(GEMX_ERR_ARG, "obs_out_dev must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    return h->cfg.dtype == GEMX_F64 ? launch_advance<double>(h, actions_dev, K, obs_out_dev, done_out_dev, obs_every ? 1 : 0, st)
                                    : launch_advance<float>(h, actions_dev, K, obs_out_dev, done_out_dev, obs_every ? 1 : 0, st);
}

int gemx_step(gemx_handle *h, const void *actions_dev, void *obs_out_dev, uint8_t *done_out_dev, void *stream) {
    return gemx_rollout(h, actions_dev, 1, obs_out_dev, done_out_dev, 1, stream);
}

int gemx_get_state(gemx_handle *h, void *soa_out_dev, void *stream) {
    if (!h || !soa_out_dev) return fail(GEMX_ERR_ARG, "null argument");
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = (h->n + 255) / 256;
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
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = (h->n + 255) / 256;
    if (h->cfg.dtype == GEMX_F64)
        hipLaunchKernelGGL(set_state_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, (double *)h->state, (double *)h->angle,
                           (const double *)soa_in_dev, h->n, h->nd, h->has_angle);
    else
        hipLaunchKernelGGL(set_state_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (float *)h->state, (int32_t *)h->angle,
                           (const float *)soa_in_dev, h->n, h->nd, h->has_angle);
    HIP_TRY(hipGetLastError());
    return GEMX_OK;
}
int gemx_get_switch_state(gemx_handle *h, uint8_t *out_dev, void *stream) {
    if (!h || !out_dev) return fail(GEMX_ERR_ARG, "null argument");
    HIP_TRY(hipMemcpyAsync(out_dev, h->sw, (size_t)h->n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GEMX_OK;
}
int gemx_set_switch_state(gemx_handle *h, const uint8_t *in_dev, void *stream) {
    if (!h || !in_dev) return fail(GEMX_ERR_ARG, "null argument");
    HIP_TRY(hipMemcpyAsync(h->sw, in_dev, (size_t)h->n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GEMX_OK;
}
int gemx_set_steps_per_block(gemx_handle *h, int32_t steps) {
    if (!h || steps < 0) return fail(GEMX_ERR_ARG, "invalid argument");
    h->steps_per_block = steps;
    return GEMX_OK;
}
int gemx_error_flags(gemx_handle *h, uint32_t *flags_host, void *stream) {
    if (!h || !flags_host) return fail(GEMX_ERR_ARG, "null argument");
    HIP_TRY(hipMemcpyAsync(flags_host, h->err, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return GEMX_OK;
}

}  // extern "C"
