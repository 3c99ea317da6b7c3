// gemx_common.hpp -- types shared by the kernel instantiation units and the C-ABI unit of libgemx.so.
// See gemx_kernels.hpp for the design notes.  gfx950 only; no CPU fallback anywhere in the product.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
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
template <> struct SysTraits<GEMX_SYS_DC_SERIES> { static constexpr int ND = 2, NOUT = 5, HAS_ANGLE = 0; };   // omega, i
template <> struct SysTraits<GEMX_SYS_DC_SHUNT>  { static constexpr int ND = 3, NOUT = 6, HAS_ANGLE = 0; };   // omega, i_a, i_e
template <> struct SysTraits<GEMX_SYS_DC_EXTEX>  { static constexpr int ND = 3, NOUT = 7, HAS_ANGLE = 0; };   // omega, i_a, i_e
template <> struct SysTraits<GEMX_SYS_EESM>      { static constexpr int ND = 4, NOUT = 16, HAS_ANGLE = 1; };  // omega, i_sd, i_sq, i_e (+eps)
template <> struct SysTraits<GEMX_SYS_DFIM>      { static constexpr int ND = 5, NOUT = 24, HAS_ANGLE = 1; };  // as SCIM (+eps), rotor voltage live

constexpr int MAX_ACT = 6;  // continuous action entries per env (2 x Cont-B6C of the DFIM envs)
constexpr int MAX_U = 4;    // motor input voltages per segment (DFIM: u_s alpha/beta, u_r alpha/beta)

template <int CONV> struct ConvTraits;
template <> struct ConvTraits<GEMX_CONV_CONT_4QC>   { static constexpr int NACT = 1, DISCRETE = 0, NACTIONS = 0; };
template <> struct ConvTraits<GEMX_CONV_FINITE_B6>  { static constexpr int NACT = 1, DISCRETE = 1, NACTIONS = 8; };
template <> struct ConvTraits<GEMX_CONV_CONT_B6>    { static constexpr int NACT = 3, DISCRETE = 0, NACTIONS = 0; };
template <> struct ConvTraits<GEMX_CONV_FINITE_4QC> { static constexpr int NACT = 1, DISCRETE = 1, NACTIONS = 4; };
// two-sub-converter MultiConverters; discrete: one byte = flat index of MultiDiscrete([n0, n1])
template <> struct ConvTraits<GEMX_CONV_CONT_2X4QC>    { static constexpr int NACT = 2, DISCRETE = 0, NACTIONS = 0; };
template <> struct ConvTraits<GEMX_CONV_FINITE_2X4QC>  { static constexpr int NACT = 1, DISCRETE = 1, NACTIONS = 16; };
template <> struct ConvTraits<GEMX_CONV_CONT_B6_4QC>   { static constexpr int NACT = 4, DISCRETE = 0, NACTIONS = 0; };
template <> struct ConvTraits<GEMX_CONV_FINITE_B6_4QC> { static constexpr int NACT = 1, DISCRETE = 1, NACTIONS = 32; };
template <> struct ConvTraits<GEMX_CONV_CONT_2XB6>     { static constexpr int NACT = 6, DISCRETE = 0, NACTIONS = 0; };
template <> struct ConvTraits<GEMX_CONV_FINITE_2XB6>   { static constexpr int NACT = 1, DISCRETE = 1, NACTIONS = 64; };
// internal kinds: a continuous B6 bridge whose caller-side action is (u_d, u_q[, u_e]) -- gemx_config.action_frame != ABC
constexpr int CONV_CONT_B6_DQ = 10;      // SYNC / SCIM + Cont-B6C
constexpr int CONV_CONT_B6_4QC_DQ = 11;  // EESM + MultiConverter(Cont-B6C, Cont-4QC)
template <> struct ConvTraits<CONV_CONT_B6_DQ>     { static constexpr int NACT = 2, DISCRETE = 0, NACTIONS = 0; };
template <> struct ConvTraits<CONV_CONT_B6_4QC_DQ> { static constexpr int NACT = 3, DISCRETE = 0, NACTIONS = 0; };
template <int CONV> constexpr bool conv_dq() { return CONV == CONV_CONT_B6_DQ || CONV == CONV_CONT_B6_4QC_DQ; }
// the converter behind the action stage, and the width of ITS action
template <int CONV> constexpr int conv_base() {
    return CONV == CONV_CONT_B6_DQ ? (int)GEMX_CONV_CONT_B6 : (CONV == CONV_CONT_B6_4QC_DQ ? (int)GEMX_CONV_CONT_B6_4QC : CONV);
}
template <int CONV> constexpr int conv_nact_c() { return ConvTraits<conv_base<CONV>()>::NACT; }
// converters that keep per-leg switching state between steps (dead time)
template <int CONV> constexpr bool conv_has_legs() {
    return CONV == GEMX_CONV_FINITE_B6 || CONV == GEMX_CONV_FINITE_4QC || CONV == GEMX_CONV_FINITE_2X4QC || CONV == GEMX_CONV_FINITE_2XB6;
}
// bytes of packed leg state per env (2 bits per half-bridge): 6 half-bridges need two rows of the [rows][N] array
template <int CONV> constexpr int conv_sw_bytes() { return CONV == GEMX_CONV_FINITE_2XB6 ? 2 : 1; }

// ------------------------------------------------------------------------------------------------
// uniform parameters (kernel argument, by value -> SGPRs)
// ------------------------------------------------------------------------------------------------
template <class R> struct DevParams {
    R m[20];      // non-zero entries of motor._model_constants, see pack_model()
    R tc0, tc1;   // torque coefficients
    R tc2, tc3;   // DFIM rotor current reconstruction: i_r = tc2 * psi_r - tc3 * i_s
    R pole;       // d(eps)/dt = pole * omega
    R inv_j, la, lb, lc, omega_lim, lin_factor;  // PolynomialStaticLoad
    R inv_tau_decay;                             //   lin_factor / J
    R u_sup;      // IdealVoltageSupply
    R il_ratio;   // interlocking_time / tau (continuous converters)
    R tau, t_il;  // control step, dead time
    R inv_ns;     // 1 / solver_nsteps
    R inv_lim[GEMX_MAX_OUT];
    R init[GEMX_MAX_ODE];  // [omega, motor states...] (angle separately)
    const R *cw;           // device array [2][GEMX_MAX_OUT]: generic constraint path 0/1 weights (limit | squared)
    int64_t init_angle_rep; // initial angle in Angle<R>::T representation (bit pattern)
    int32_t nsteps, auto_reset, obs_layout;
    int32_t constr_kind;    // 0 none, 1 the system's default constraint (fast path), 2 generic weights
    const R *lin;           // constant-speed loads: the solver's one-step map of the (linear) electrical subsystem at omega = init[0]:
    int32_t lin_on;         //   x1 = Phi x0 + S g, Phi [NM][NM] then S [NM][NG] (built once per handle by linmap_kernel)
    int32_t init_kind;      // != 0: initial states are drawn per reset (rinit / reset counters in KArgs)
    int32_t rc_supply;      // RCVoltageSupply: u_sup is a per-env state (rows ND, ND+1 of the state array: u, time since last update)
    R sup_r, sup_inv_rc;    // R, 1 / (R C)
    int32_t dq_processor;   // dq action frames: 0 control_space='dq' (step-start angle), 1 DqToAbcActionProcessor (advanced angle)
    int32_t delay;          // DeadTimeProcessor steps
    R dreset[MAX_ACT];      //   the action every reset refills its queue with (gemx_config.action_delay_reset; a discrete index as R) ...
    uint32_t dreset_d;      //   ... and that index as an integer (0 for continuous actions)
    R dq_adv;               // (0.5 + delay) * tau * pole: angle advance per rad/s of omega
    int32_t kink_split;     // GEMX_SOLVER_SPLIT_KINKS: the PolynomialStaticLoad's kinks are corrected for in closed form (integrate<>)
    int32_t adaptive;       // GEMX_SOLVER_ADAPTIVE (DP5 only): error-controlled sub-stepping, dp5_adaptive()
    R rtol, atol;           //   its tolerances (gemx_config.solver_rtol / solver_atol)
    R atol_w;               //   ... and omega's absolute tolerance (gemx_config.solver_atol_omega; default solver_atol x limits[omega])
    uint32_t *errw;         //   the handle's device error word (GEMX_ERRFLAG_TOLERANCE)
    // DC machines: the default LimitConstraint on current c as a threshold in AMPERES, dc_thr[c] = the smallest R with
    // fl(dc_thr[c] * inv_lim[2 + c]) > 1 (viol_threshold(), host): |i| >= dc_thr[c]  <=>  |i * inv_lim| > 1 for EVERY i, because
    // rounding is monotonic -- the same done / reset decisions bit for bit, without the multiply on the step-to-step chain
    R dc_thr[2];
};
// smallest t >= 0 with fl(t * c) > 1 (c > 0): see DevParams::dc_thr
template <class R> inline R viol_threshold(R c) {
    if (!(c > R(0)) || !std::isfinite(c)) return (R)INFINITY;  // (no finite current violates such a limit; gemx_create rejects it anyway)
    volatile R t = R(1) / c, p = t * c;
    while (p > R(1)) { t = std::nextafter((R)t, R(0)); p = t * c; }
    while (!(p > R(1))) { t = std::nextafter((R)t, (R)INFINITY); p = t * c; }
    return t;
}

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
    // (the float -> int conversion SATURATES at +-2^31 counts = half a turn, so an increment beyond pi -- the DqToAbcActionProcessor's
    // (0.5 + dead time) * tau * p * omega at 8 dead-time steps near the speed limit, or a large tau -- is first reduced modulo one turn:
    // x - 2^32 rint(x 2^-32) is exact, both terms being multiples of x's ulp)
    using Inc = int32_t;  // an increment in counts
    static __device__ __forceinline__ Inc increment(float d_rad) {
        const float x = d_rad * kCountsPerRad;
        const float y = fmaf(-4294967296.0f, rintf(x * 2.3283064365386963e-10f), x);  // in [-2^31, 2^31]
        return __float2int_rn(y);
    }
    static __device__ __forceinline__ T add(T a, Inc inc) { return (T)((uint32_t)a + (uint32_t)inc); }
    static __device__ __forceinline__ T advance(T a, float d_rad) { return add(a, increment(d_rad)); }
    static __device__ __forceinline__ float wrapped(T a) { return (float)a * kRadPerCount; }  // [-pi, pi]
    static __device__ __forceinline__ float to_rad(T a) { return wrapped(a); }
    // sin/cos of a fixed-point angle with the hardware's V_SIN_F32 / V_COS_F32, whose argument is in REVOLUTIONS: the count times
    // 2^-32 is the argument, no range reduction at all.  Measured on gfx950 over 2^22 angles (tools/microbench_hwsin.hip): max abs
    // error 1.8e-7 for both, i.e. that of the single-precision minimax polynomials used before (1.2e-7, ~28 VALU instructions with
    // the quadrant logic) at 4 instructions -- two of them quarter-rate.
    // Its errors are not centred, though (mean s^2+c^2-1 = -6.4e-8 against -1e-9, rms 4.5e-8 against 2.4e-8), and the doubly fed
    // induction motor system, whose field-oriented transforms amplify angle noise ~1000x, keeps the polynomials (sincos_precise).
    static __device__ __forceinline__ void sincos(T a, float &s, float &c) {
        const float x = (float)a * 2.3283064365386963e-10f;  // [-0.5, 0.5) revolutions
        s = __builtin_amdgcn_sinf(x);
        c = __builtin_amdgcn_cosf(x);
    }
    // quadrant from the top bits, Cephes single-precision minimax polynomials on [-pi/4, pi/4] (abs error < 1.2e-7)
    static __device__ __forceinline__ void sincos_precise(T a, float &s, float &c) {
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
    using Inc = double;
    static __device__ __forceinline__ Inc increment(double d) { return d; }
    static __device__ __forceinline__ T add(T a, Inc inc) { return a + inc; }
    static __device__ __forceinline__ T advance(T a, double d) { return a + d; }
    static __device__ __forceinline__ double wrapped(T a) {  // physical_systems.py:520-522
        double e = fmod(a, kTwoPi);
        if (e < 0) e += kTwoPi;
        if (e > kPi) e -= kTwoPi;
        return e;
    }
    static __device__ __forceinline__ double to_rad(T a) { return a; }
    static __device__ __forceinline__ void sincos(T a, double &s, double &c) { ::sincos(a, &s, &c); }
    static __device__ __forceinline__ void sincos_precise(T a, double &s, double &c) { ::sincos(a, &s, &c); }
};

// ------------------------------------------------------------------------------------------------
// small math helpers
// ------------------------------------------------------------------------------------------------
template <class R> __device__ __forceinline__ R clip01(R x) { return fmin(fmax(x, R(0)), R(1)); }
// duty cycles of the continuous converters, clip(0.5 (a + 1)) and clip(-0.5 (a - 1)) (converters.py:489-490, 900-902), each as ONE fused
// multiply-add with the clamp modifier: scaling by a power of two commutes with rounding and |a + 1| is never subnormal, so
// fma(a, 0.5, 0.5) == 0.5 * fl(a + 1) bit for bit
template <class R> __device__ __forceinline__ R duty_pos(R a) {
    if constexpr (sizeof(R) == 4) return clip01(__builtin_fmaf(a, 0.5f, 0.5f));
    else return clip01(__builtin_fma(a, 0.5, 0.5));
}
template <class R> __device__ __forceinline__ R duty_neg(R a) {
    if constexpr (sizeof(R) == 4) return clip01(__builtin_fmaf(a, -0.5f, 0.5f));
    else return clip01(__builtin_fma(a, -0.5, 0.5));
}
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
// Philox4x32-10 (Salmon et al., SC'11), counter = (env lo, env hi, reset count, block), key = seed: the reference generators' streams
// (gemx_refgen.hip); the random INITIAL STATES used it until round 5 and take InitRng below since.
// Counter-based, so a reset needs no RNG state beyond the env's reset count.  env = the GLOBAL env index (gemx_config.env_base + i).
// ------------------------------------------------------------------------------------------------
struct Philox {
    static __host__ __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    static __host__ __device__ __forceinline__ void block(uint64_t seed, uint64_t env, uint32_t count, uint32_t blk, uint32_t (&out)[4]) {
        uint32_t c[4] = {(uint32_t)env, (uint32_t)(env >> 32), count, blk};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
        for (int r = 0; r < 10; ++r) {
            round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        for (int i = 0; i < 4; ++i) out[i] = c[i];
    }
    // uniform in the open interval (0, 1), 32 bits of resolution
    static __host__ __device__ __forceinline__ double u01(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }
};
// INITIAL-STATE streams since round 6: Threefry-4x32 with 12 rounds (the add-rotate-xor generator of the same paper; counter = (env lo, env hi,
// reset count, block), key = (seed lo, seed hi, 0, 0)).  Philox's rounds are two 32 x 32 -> 64-bit multiplies each, quarter-rate instructions
// on gfx950 (~900 cycles per block of four values for a wave); Threefry's are adds, rotates and xors at full rate (~340).  That matters twice:
// the pipelined kernel's LOADER wave prepares every lane's next draws while the integrators of the other workgroups on its SIMD compete for
// the same issue slots (Cont-SC-PMSM with random initial states at 131072 envs was bound by the loader's 6000 cycles per block), and a lane
// that outruns its queue draws INLINE on the integrator's own stream (SCIM: 7.7 % of the resets, 7000 of the integrator's 10000 cycles per
// block: profiles/r06_rinit_probe.md).  The reference generators (gemx_refgen.hip) keep Philox.
struct InitRng {
    static __host__ __device__ __forceinline__ uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
    static __host__ __device__ __forceinline__ void block(uint64_t seed, uint64_t env, uint32_t count, uint32_t blk, uint32_t (&out)[4]) {
#ifdef GEMX_INIT_RNG_PHILOX  // A/B builds: round 5's generator
        Philox::block(seed, env, count, blk, out);
        return;
#endif
        const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), k2 = 0u, k3 = 0u;
        const uint32_t ks[5] = {k0, k1, k2, k3, 0x1BD11BDAu ^ k0 ^ k1 ^ k2 ^ k3};
        uint32_t x0 = (uint32_t)env + ks[0], x1 = (uint32_t)(env >> 32) + ks[1], x2 = count + ks[2], x3 = blk + ks[3];
        constexpr int R0[8] = {10, 11, 13, 23, 6, 17, 25, 18}, R1[8] = {26, 21, 27, 5, 20, 11, 10, 20};
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            if ((r & 1) == 0) {
                x0 += x1; x1 = rotl(x1, R0[r & 7]); x1 ^= x0;
                x2 += x3; x3 = rotl(x3, R1[r & 7]); x3 ^= x2;
            } else {
                x0 += x3; x3 = rotl(x3, R0[r & 7]); x3 ^= x0;
                x2 += x1; x1 = rotl(x1, R1[r & 7]); x1 ^= x2;
            }
            if ((r & 3) == 3) {  // key injection after every fourth round
                const int s_ = (r + 1) >> 2;
                x0 += ks[s_ % 5]; x1 += ks[(s_ + 1) % 5]; x2 += ks[(s_ + 2) % 5]; x3 += ks[(s_ + 3) % 5] + (uint32_t)s_;
            }
        }
        out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
    }
};
// per-state sampling description (device array read in the rare reset path only)
struct InitDev {
    int32_t kind, n;  // gemx_init_kind; number of ODE states incl. the angle
    uint64_t seed;
    int64_t env_base;  // gemx_config.env_base: the streams are keyed by the GLOBAL env index env_base + i (a shard draws what the unsharded job draws)
    double lo[GEMX_MAX_ODE], hi[GEMX_MAX_ODE], mu[GEMX_MAX_ODE], sigma[GEMX_MAX_ODE], constant[GEMX_MAX_ODE];
    double cdf_lo[GEMX_MAX_ODE], cdf_hi[GEMX_MAX_ODE];  // gaussian: Phi((lo - mu) / sigma), Phi((hi - mu) / sigma), computed on the host
    // induction machines (gemx_config.init_flux_mode): the two flux states' bounds are re-derived at every reset from a random field
    // angle; lo / hi of their slots then hold the user's `interval` (+-HUGE_VAL = none), flux[] = gemx_config.init_flux
    int32_t flux_mode, flux_slot;  // first flux slot (ODE index of psi_r alpha)
    double flux[8];
};
// inverse of the standard normal CDF: Wichura's algorithm AS 241 (PPND16, |rel err| < 1e-16) -- a few rational polynomials instead
// of the library's normcdfinv expansion, because this code is inlined into every advance kernel's (rare) auto-reset path
__host__ __device__ inline double inv_norm_cdf(double p) {
    const double q = p - 0.5;
    if (fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        const double num = (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r + 4.5921953931549871457e4) * r +
                               1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r + 1.3314166789178437745e2) * r + 3.3871328727963666080e0);
        const double den = (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r + 2.1213794301586595867e4) * r +
                               5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r + 4.2313330701600911252e1) * r + 1.0);
        return q * num / den;
    }
    double r = q < 0 ? p : 1.0 - p;
    r = sqrt(-log(r));
    double val;
    if (r <= 5.0) {
        r -= 1.6;
        const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r + 1.27045825245236838258e0) * r +
                               3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r + 4.63033784615654529590e0) * r + 1.42343711074968357734e0);
        const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r + 1.48103976427480074590e-1) * r +
                               6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r + 2.05319162663775882187e0) * r + 1.0);
        val = num / den;
    } else {
        r -= 5.0;
        const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r + 2.65321895265761230930e-2) * r +
                               2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r + 5.46378491116411436990e0) * r + 6.65790464350110377720e0);
        const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r + 7.86869131145613259100e-4) * r +
                               1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r + 5.99832206555887937690e-1) * r + 1.0);
        val = num / den;
    }
    return q < 0 ? -val : val;
}
// The truncated normal's arithmetic -- the inverse CDF's three rational polynomials, erfc for per-reset bounds: ~600 fp64 instructions -- as
// REAL CALLS (round 6).  Inlined, it sat in the middle of every draw, also the uniform ones that never execute it, and a draw on the
// integrator's stream (a lane that outran its queue of prepared draws) fetched its way through all of it from a cold instruction cache:
// ~13 000 cycles per event for a PMSM draw of ~200 executed instructions (profiles/r06_rinit_probe.md).
__device__ __attribute__((noinline)) double init_gauss_state(double mu, double sigma, double cdf_lo, double cdf_hi, double lo, double hi, double u) {
    const double x = mu + sigma * inv_norm_cdf(cdf_lo + (cdf_hi - cdf_lo) * u);
    return fmin(fmax(x, lo), hi);
}
__device__ __attribute__((noinline)) double init_gauss_bounded(double mu_cfg, double sg, double lo, double hi, double u) {
    // electric_motor.py:236-249: mue = random_params[0] or the middle of the interval, sigma = random_params[1] or 1 (mu = NaN: middle)
    const double mu = mu_cfg == mu_cfg ? mu_cfg : 0.5 * (hi - lo) + lo;
    const double cl = 0.5 * erfc(-(lo - mu) / sg * 0.70710678118654752440), ch = 0.5 * erfc(-(hi - mu) / sg * 0.70710678118654752440);
    return fmin(fmax(mu + sg * inv_norm_cdf(cl + (ch - cl) * u), lo), hi);
}
// the j-th initial state from its uniform u in (0, 1): uniform in [lo, hi], or normal(mu, sigma) truncated to [lo, hi] by inverse CDF
__device__ __forceinline__ double init_state_from_uniform(const InitDev *I, int j, double u) {
    if (!(I->lo[j] < I->hi[j])) return I->constant[j];
    if (I->kind == GEMX_INIT_UNIFORM) return fma(I->hi[j] - I->lo[j], u, I->lo[j]);  // (one rounding, spelled out: the same bits from every call site)
    return init_gauss_state(I->mu[j], I->sigma[j], I->cdf_lo[j], I->cdf_hi[j], I->lo[j], I->hi[j], u);
}
// all (<= 8) uniforms of one (env, reset count): two Philox blocks -- the second one only where something reads it (more than four ODE
// states, or the induction machines' field angle, uniform 7); wave-uniform
__device__ __forceinline__ bool init_needs_block1(const InitDev *I) { return I->n > 4 || I->flux_mode != 0; }
__device__ __forceinline__ void init_uniforms_from(const uint32_t (&r0)[4], const uint32_t (&r1)[4], double (&u)[GEMX_MAX_ODE]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { u[i] = Philox::u01(r0[i]); u[4 + i] = Philox::u01(r1[i]); }
}
__device__ __forceinline__ void init_uniforms(const InitDev *I, int64_t env, uint32_t count, double (&u)[GEMX_MAX_ODE], bool block1 = true) {
    uint32_t r0[4], r1[4] = {0u, 0u, 0u, 0u};
    const uint64_t genv = (uint64_t)(I->env_base + env);  // global env index
    InitRng::block(I->seed, genv, count, 0u, r0);
    if (block1 && init_needs_block1(I)) InitRng::block(I->seed, genv, count, 1u, r1);
    init_uniforms_from(r0, r1, u);
}
// one state from its uniform with explicit bounds (the induction machines' per-reset flux bounds)
__device__ __forceinline__ double init_draw_bounded(const InitDev *I, int j, double lo, double hi, double u) {
    if (!(lo < hi)) return lo;  // (upper - lower) * u + lower of a degenerate interval
    if (I->kind == GEMX_INIT_UNIFORM) return fma(hi - lo, u, lo);
    return init_gauss_bounded(I->mu[j], I->sigma[j], lo, hi, u);
}
// THE draw of one reset: out[0 .. n-1] = the ODE states (+ the angle in slot n - 1 when the system has one) of (env, reset count).
// Induction machines (flux_mode; induction_motor.py:174-185, 250-285, squirrel_cage_induction_motor.py:146-157): a field angle
// eps_mag ~ U(-pi, pi) from uniform 7, psi_d_max from this reset's omega -- and, for omega != 0, from the stator currents of the
// PREVIOUS reset's draw (the reference reads the motor's stale _initial_states; the counter-based stream lets the kernel recompute
// them instead of storing them) --, flux bounds +-psi_d_max (|cos|, |sin|) clipped to the user's interval.
// FLUX: compiled in for the induction-machine systems only (the code sits in every kernel's auto-reset path).
// `init_draw_from`: the draw from its uniforms -- `u` of (env, count) and, where the induction machines read the previous reset's currents
// (count > 1, omega != 0), `up` of (env, count - 1), fetched through `prev` only then.  The pipelined kernel's loader wave computes the
// Philox blocks one per pass and finishes with this (prepared draws); init_draw_all is the draw in one go.
// NS: the number of ODE states incl. the angle where the caller knows it at compile time (the kernels of one system: the loop unrolls, and a
// descriptor the caller copied into registers -- the pipelined kernel's loader wave -- is never indexed dynamically); 0: I->n.
template <bool FLUX, int NS = 0, class Prev>
__device__ __forceinline__ void init_draw_from(const InitDev *I, uint32_t count, const double (&u)[GEMX_MAX_ODE], Prev prev, double (&out)[GEMX_MAX_ODE]) {
    if constexpr (NS > 0) {
#pragma unroll
        for (int j = 0; j < NS; ++j) out[j] = init_state_from_uniform(I, j, u[j]);
    } else {
        for (int j = 0; j < I->n && j < GEMX_MAX_ODE; ++j) out[j] = init_state_from_uniform(I, j, u[j]);
    }
    if (FLUX && I->flux_mode) {
        constexpr int FS_STATIC = 3;  // (gemx_create: [omega, i_s alpha, i_s beta, psi_r alpha, psi_r beta, epsilon] for both induction machines)
        const int fs = NS > 0 ? FS_STATIC : I->flux_slot;  // slots fs - 2, fs - 1: i_s alpha, i_s beta
        // eps_mag = 2 pi u7 - pi ~ U(-pi, pi): cos / sin through the fp32 hardware functions, which take their argument in TURNS
        // (cos(2 pi u - pi) = -cos(2 pi u)); 1e-6 absolute is all a random field angle needs (round 5: the fp64 library sincos, a few
        // hundred instructions on the integrator's stream whenever a lane drew inline).  The draw is distributional by contract (KS tests).
        const float u7 = (float)u[7];
        const double ce = -(double)__builtin_amdgcn_cosf(u7), se = -(double)__builtin_amdgcn_sinf(u7), om = out[0];
        double psi = I->flux[0];
        if (om != 0.0) {
            double ia = I->constant[fs - 2], ib = I->constant[fs - 1];
            if (count > 1u) {
                double up[GEMX_MAX_ODE];
                prev(up);
                ia = init_state_from_uniform(I, fs - 2, up[fs - 2]);
                ib = init_state_from_uniform(I, fs - 1, up[fs - 1]);
            }
            const double id = ce * ia + se * ib, iq = -se * ia + ce * ib;  // q_inv(i_alphabeta, eps_mag)
            psi = (I->flux[1] * om * I->flux[2] * id + I->flux[3] * iq + I->flux[4]) / (-I->flux[1] * om * I->flux[5]);
            psi = 0.9 * fmin(fmax(psi, 0.0), fabs(I->flux[6] * id));
        }
        const double ha = fabs(psi * ce), hb = fabs(psi * se);
        out[fs] = init_draw_bounded(I, fs, fmax(-ha, I->lo[fs]), fmin(ha, I->hi[fs]), u[fs]);
        out[fs + 1] = init_draw_bounded(I, fs + 1, fmax(-hb, I->lo[fs + 1]), fmin(hb, I->hi[fs + 1]), u[fs + 1]);
    }
}
template <bool FLUX, int NS = 0>
__device__ __forceinline__ void init_draw_all(const InitDev *I, int64_t env, uint32_t count, double (&out)[GEMX_MAX_ODE]) {
    double u[GEMX_MAX_ODE];
    init_uniforms(I, env, count, u);
    // (of the previous draw only the stator currents' uniforms are read, slots flux_slot - 2 and - 1: block 1 only if they reach into it)
    init_draw_from<FLUX, NS>(I, count, u, [&](double (&up)[GEMX_MAX_ODE]) { init_uniforms(I, env, count - 1u, up, I->flux_slot > 4); }, out);
}

// ------------------------------------------------------------------------------------------------
// SYNTHETIC ACTIONS (round 5; SURVEY.md 8e: "actions ... can be generated on-device").  A counter-based stream: the action of env e at
// control step t (position in the stream) is a pure function of (seed, e, t, component) -- no state, the same bits on the host, in the
// generator kernel (gemx_synthetic_actions) and in the pipelined kernel's loader wave (gemx_rollout_synthetic), which then reads NO action
// tensor at all: random-action rollouts of the continuous converters lose the 8-24 B per env-step whose trip from the HBM between the row
// stores costs them 15-20 % of their rate.  One 32-bit mix per value (lowbias32: two 32-bit multiplies, three xor-shifts;
// the ten rounds of Philox used for the initial states would be 80 quarter-rate multiplies per env-step on a wave that shares its SIMD).
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t synth_u32(uint64_t seed, int64_t env, uint32_t t, uint32_t comp) {
    uint32_t x = (uint32_t)env * 0x9E3779B9u + t * 0x85EBCA6Bu + comp * 0xC2B2AE35u + (uint32_t)seed;
    x ^= (uint32_t)(seed >> 32) ^ ((uint32_t)((uint64_t)env >> 32) * 0x27D4EB2Fu);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
// continuous: uniform on the 2^24 midpoints of (-1, 1), exact in fp32 and fp64; discrete: the top bits (every action count is a power of two)
__host__ __device__ __forceinline__ float synth_unit(uint32_t x) { return (float)(x >> 8) * 1.1920928955078125e-7f + (-1.0f + 5.9604644775390625e-8f); }
__host__ __device__ __forceinline__ uint32_t synth_index(uint32_t x, uint32_t n_actions) { return (uint32_t)(((uint64_t)x * n_actions) >> 32); }

// ------------------------------------------------------------------------------------------------
// fused reward (WeightedSumOfErrors): device-resident description, read through scalar loads when a reward is requested
// ------------------------------------------------------------------------------------------------
template <class R> struct RewardDev {
    // terms 0 .. n_ref-1 compare state column col[t] with reference column t; terms n_ref .. n_term-1 compare with 0
    int32_t n_ref, n_term;
    int32_t col[GEMX_MAX_OUT];
    int32_t kind[GEMX_MAX_OUT];   // 1: power 1, 2: power 2, 3: general power (pow)
    R coef[GEMX_MAX_OUT];         // weight (applied after the power)
    R inv_len[GEMX_MAX_OUT];      // 1 / state_length
    R power[GEMX_MAX_OUT];
    R bias, violation_reward;
};

// The first GEMX_REWARD_HOT terms by value, inside the kernel arguments: kernel arguments are read with SCALAR loads (lgkmcnt).  Read
// through the device pointer (KArgs::rw) the same words are VECTOR loads with a uniform address -- the compiler cannot prove that the
// kernel's own stores do not alias them -- and their `s_waitcnt vmcnt(0)` waits, vmcnt retiring in order, for every observation store
// the wave has in flight: ~5000 cycles per block in the pipelined kernel's output waves (s_memtime probe).
constexpr int GEMX_REWARD_HOT = 4;
template <class R> struct RewardHot {
    int32_t n_ref, n_term, general;  // general: some term has a power other than 1 or 2 (pow() path through KArgs::rw)
    int32_t col[GEMX_REWARD_HOT], kind[GEMX_REWARD_HOT];  // unused terms: column 0, kind 1, weight 0
    R coef[GEMX_REWARD_HOT], inv_len[GEMX_REWARD_HOT], power[GEMX_REWARD_HOT];
    R bias, violation_reward;
};
template <class R> inline void reward_hot_from(const RewardDev<R> &W, RewardHot<R> &H) {
    H.n_ref = W.n_ref;
    H.n_term = W.n_term;
    H.general = 0;
    for (int t = 0; t < W.n_term; ++t) H.general |= W.kind[t] == 3;
    for (int t = 0; t < GEMX_REWARD_HOT; ++t) {
        const bool used = t < W.n_term;
        H.col[t] = used ? W.col[t] : 0;
        H.kind[t] = used ? W.kind[t] : 1;
        H.coef[t] = used ? W.coef[t] : R(0);
        H.inv_len[t] = used ? W.inv_len[t] : R(0);
        H.power[t] = used ? W.power[t] : R(1);
    }
    H.bias = W.bias;
    H.violation_reward = W.violation_reward;
}

// ------------------------------------------------------------------------------------------------
// kernel arguments
// ------------------------------------------------------------------------------------------------
template <class R> struct KArgs {
    DevParams<R> P;
    RewardHot<R> rh;                // valid when rw != nullptr
    R *state;                       // [ND][N]
    typename Angle<R>::T *angle;    // [N] (systems with an angle)
    uint8_t *sw;                    // [rows][N] packed leg states, 2 bits per half-bridge (finite converters with interlocking)
    const unsigned char *actions;   // [K][N][A] R  |  [K][N] uint8
    R *obs;                         // [K][N][NOUT] | [K][NOUT][N]  (or a single step's worth if !obs_every)
    uint8_t *done;                  // [K][N] | [N]
    unsigned char *ring;            // [delay][N][A_conv] R | [delay][N] uint8: DeadTimeProcessor FIFO between launches
    // DeadTimeProcessor FIFO phase IN DEVICE MEMORY: [0] = FIFO slot of this launch's first step (control steps so far mod delay), read by
    // every workgroup at its start; [1] = a ticket counter: the LAST workgroup to finish advances [0] by K (fifo_phase_advance).  Kept on
    // the device so that a launch replayed from a captured HIP graph continues the queue where the previous launch left it (a phase
    // computed on the host at capture time would be frozen into the graph).
    uint32_t *fifo_phase;
    uint32_t *err;                  // device error word (GEMX_ERRFLAG_*: bit 0 discrete action out of range, bit 1 dc_stream_kernel met a moved omega)
    const InitDev *rinit;           // random initial states: description, per-env reset counters [N]
    uint32_t *rcnt;
    const RewardDev<R> *rw;         // fused reward: description (nullptr: no reward), references [K][N][n_ref], output [K][N]
    const R *refs;
    R *reward;
    int64_t N;
    int32_t K, obs_every;
    int32_t S;                      // control steps per I/O block (LDS ring depth)
    int32_t D;                      // pipelined kernel: control steps per hand-off block (one barrier per D steps)
    int32_t coop;                   // 1: action rows / done rows of full blocks are 16-byte aligned -> cooperative staging
    int32_t obs_vec;                // 1: observation rows of full blocks are 16-byte aligned -> 16-byte stores
    // pipelined kernel, large batches: a RATE LIMIT on every workgroup -- hand-off block b starts no earlier than b * pace_block_ticks
    // ticks of the constant 100 MHz clock (s_memrealtime) after the workgroup's start; 0 = none.  See launch_advance_t.
    uint32_t pace_block_ticks;
    uint32_t pace_tail_ticks;       // ... of the workgroups from index pace_tail_from on: the LAST round, which has the chip to itself with fewer
    uint32_t pace_tail_from;        //     workgroups than a full one and may run that much faster each
    // synthetic actions (gemx_rollout_synthetic): act_synth != 0 -> `actions` is not read; the loader wave generates synth_u32(act_seed, env,
    // act_step0 + k, component) instead
    uint64_t act_seed;
    int64_t act_env_base;  // gemx_config.env_base: the stream is keyed by the global env index
    int32_t prep_phases;   // random initial states: phases of the loader's prepared-draw state machine per hand-off block (launch_advance_t)
    int32_t act_half;      // gemx_rollout_half: `actions` is [K][N][A] IEEE half (continuous converters, fp32 kernels): widened while staged
    uint32_t act_step0;
    int32_t act_synth;
};

// FIFO slot of this launch's first step (0 without a DeadTimeProcessor) / its advance by the last workgroup to finish: every workgroup
// has read the phase at its start before it takes its ticket at its end, so the new value is written only when nobody reads the old one
template <class R> __device__ __forceinline__ int fifo_phase_read(const KArgs<R> &a) {
    return a.P.delay > 0 ? (int)*a.fifo_phase : 0;
}
template <class R> __device__ __forceinline__ void fifo_phase_advance(const KArgs<R> &a, int phase, bool one_lane) {
    if (a.P.delay > 0 && one_lane) {
        const uint32_t ticket = atomicAdd(a.fifo_phase + 1, 1u);
        if (ticket == gridDim.x - 1) {
            a.fifo_phase[1] = 0u;
            a.fifo_phase[0] = (uint32_t)((phase + a.K) % a.P.delay);
        }
    }
}

constexpr int MAX_ACT_CHUNKS = 12;  // upper bound of 16-byte chunks of staged actions per lane and I/O block
constexpr int MAX_STEPS_PER_BLOCK = 32;
// pipelined kernel: <control steps per hand-off block (one barrier per block), output/store waves per workgroup>.
//   <12, 3>: N small enough for one resident workgroup per CU: integrator + 3 output waves = one wave on each of the CU's 4 SIMDs;
//   < 4, 2>: up to ~2 rounds of 4 resident workgroups per CU (a third of the LDS per workgroup).
constexpr int PIPE_D = 12, PIPE_OUT_WAVES = 3;
constexpr int PIPE_D2 = 4, PIPE_OUT_WAVES2 = 2;
// deep shape with six output waves: launches that evaluate the fused reward, whose output waves otherwise bound the launch
constexpr int PIPE_OUT_WAVES_RW = 6;
// shallow shape: half the LDS of <4, 2> again, for the N at which only IT fits all workgroups into one resident round
constexpr int PIPE_D3 = 2, PIPE_OUT_WAVES3 = 2;
// both shapes carry a LOADER wave that stages actions / references global -> LDS (0: the integrator wave stages them itself).  Measured
// at 131072 envs over all motor families (same box A/B): -5 .. +22 %, the heavier steppers and the continuous-action ones gain most
constexpr int pipe_loader_waves(int) { return 1; }
// action staging buffers of the pipelined kernel: the loader wave runs one block ahead.  (Two ahead through a third buffer: round 2 saw no
// change; round 4 tried again for the launches whose action tensor no longer fits the 256 MB Infinity Cache -- PMSM cont from 22 M
// env-steps per launch, where the rate falls from 0.89 to 0.68 of the roofline -- and again nothing moved (16384 envs x 2000 steps 0.687 ->
// 0.683, PermExDc cont 131072 x 1000 0.573 -> 0.570) while the extra LDS cost EESM cont its deep shape: profiles/r04r_probe4.txt.  The
// loss there is the HBM serving reads between the writes, not the latency of this wave's loads.)
constexpr int PIPE_ACT_BUFS = 2;
// Round 5 tried once more, with a reason: a timing build at 65536 envs (<2, 2>, four workgroups per CU) shows the loader wave's trip to the
// HBM for one block's actions taking 2500-3300 cycles, longer than the integrator's 2-step block for every machine but the induction
// machines behind a PolynomialStaticLoad.  Two blocks ahead (GEMX_PIPE_AHEAD2=1: a third action buffer, a fourth reference buffer, the loader
// waits with `s_waitcnt vmcnt(NSTAGE)` for the OLDER block only) takes the loader's wait from ~2800 to ~600 cycles per block -- and the
// launch time does not move: those launches are held by the rate limiter (1620 ns per block = the measured period; the calibration finds
// nothing faster, paced or not), i.e. by the write path, not by this latency.  Same-box A/B over six rows x three sizes: +3 % on one row
// (Cont-SC-PMSM 65536), -14 % on another (PermExDc cont 65536: the extra buffer costs a resident workgroup), the rest within 1 %
// (profiles/r05h_ab_loader_two_ahead.md, r05h_pipe_probe.txt).  Default off; the code stays as an A/B build.
#ifndef GEMX_PIPE_AHEAD2  // 1: the shallow shapes (D <= 4) stage two blocks ahead (A/B builds)
#define GEMX_PIPE_AHEAD2 0
#endif
constexpr int pipe_act_ahead(int D) { return (D <= 4 && GEMX_PIPE_AHEAD2) ? 2 : 1; }
constexpr int pipe_act_bufs(int D) { return pipe_act_ahead(D) + 1; }
constexpr int pipe_ref_bufs(int D) { return pipe_act_ahead(D) + 2; }  // (the output waves read a block's references one block behind the integrator)
// rows of the pipelined kernel's per-lane action queue in LDS: `delay` (FIFO / carry rows), or D + delay where the queue of TRANSFORMED
// actions behind a DqToAbcActionProcessor is kept as a row buffer indexed by the step of the block (deep shape only)
__host__ __device__ constexpr int pipe_queue_rows(int D, int delay, bool dq_processor, bool full) {
    return (D == PIPE_D && !full && dq_processor && delay > 0) ? D + delay : delay;
}
// chunks per lane needed to stage MAX_STEPS_PER_BLOCK steps of a row made of `cpr` 16-byte chunks
__host__ __device__ constexpr int act_chunks(int cpr) {
    return (MAX_STEPS_PER_BLOCK * cpr + BLOCK - 1) / BLOCK < MAX_ACT_CHUNKS ? (MAX_STEPS_PER_BLOCK * cpr + BLOCK - 1) / BLOCK : MAX_ACT_CHUNKS;
}

}  // namespace gemx

// ------------------------------------------------------------------------------------------------
// host-side handle (shared by gemx_capi.hip and the instantiation units)
// ------------------------------------------------------------------------------------------------
struct gemx_handle {
    gemx_config cfg;
    int64_t n;
    int device;
    int nd, nout, nact, has_angle;
    gemx::DevParams<float> pf;
    gemx::DevParams<double> pd;
    void *state = nullptr;   // [nd + extra_rows][n] R
    int extra_rows = 0;      // rows behind the ODE states: RC supply (2) / + the error-controlled solver's carried step size (3)
    void *angle = nullptr;   // [n] int32 | double
    uint8_t *sw = nullptr;   // [sw_rows][n]
    int sw_rows = 1;
    void *step_fn = nullptr;      // hipFunction_t of this handle's step_kernel instantiation (resolved at the first gemx_step)
    bool step_fn_failed = false;
    void *linmap_dev = nullptr;  // one-step maps of the electrical subsystem (constant-speed loads), R[4][lin_count]: Phi(tau) | D(t_il), D(tau - t_il), D(tau) (linmap_kernel)
    int linmap_state = 0;        // 0: not built yet, 1: built and enabled, -1: not applicable
    void *rinit_dev = nullptr;  // InitDev (random initial states)
    uint32_t *rcnt = nullptr;   // [n] resets so far per env
    void *rw_dev = nullptr;  // RewardDev<R> (gemx_set_reward)
    gemx::RewardHot<float> rh_f = {};   // its first terms by value (kernel argument)
    gemx::RewardHot<double> rh_d = {};
    int rw_n_ref = -1;       // -1: no reward installed
    bool cur_half = false;           // set by gemx_rollout_half around the launch: the action tensor holds IEEE halves
    bool cur_synth = false;          // set by gemx_rollout_synthetic around the launch: actions come from synth_u32(cur_seed, env, cur_step0 + k, i)
    uint64_t cur_seed = 0;
    uint32_t cur_step0 = 0;
    const void *cur_refs = nullptr;  // set by gemx_rollout_reward around the launch
    void *cur_reward = nullptr;
    void *ring = nullptr;    // DeadTimeProcessor FIFO [delay][n][nact_conv] R | [delay][n] uint8
    size_t ring_bytes = 0;
    int nact_conv = 1;       // converter-side action width (== nact unless a dq action frame is configured)
    int conv_unit = 0;       // converter kind of the kernel unit (internal dq kinds included)
    void *unit_launch = nullptr;  // gemx_unit_launch of libgemx_u<system>_<conv_unit>_<dtype>.so, loaded by gemx_create (gemx_capi.hip: load_unit)
    unsigned long long steps_total = 0;  // control steps launched since creation (informational; the FIFO phase lives on the device)
    uint32_t *fifo_phase = nullptr;      // device words [phase, ticket] of the DeadTimeProcessor FIFO (KArgs::fifo_phase)
    uint32_t *err = nullptr;
    void *reset_obs_dev = nullptr;  // [nout] R
    void *cw_dev = nullptr;         // [2][GEMX_MAX_OUT] R constraint weights
    double reset_obs[GEMX_MAX_OUT];
    int n_cu = 256;
    size_t lds_max = 160 * 1024;
    int steps_per_block = 0;  // 0 = heuristic
    struct LastLaunch { int pipe, sys, conv, load, solver, il, real_size, d, threads, k, s; long long blocks; size_t lds; unsigned pace, pace_tail; long long pace_res; int shape; };  // shape: pipelined kernel: pipe_kernel_of's index; dc_stream_kernel: envs per workgroup
    LastLaunch ll = {};            // most recent advance launch (formatted lazily by gemx_last_launch)
    mutable char last_launch[512] = "";
    char overrides[192] = "";  // the GEMX_* environment switches that were set when the handle was created ("NAME=value ..."): gemx_last_launch() names them
    double pace_gbps = -1.0;  // target chip-wide algorithmic rate of the rate limiter [GB/s]; < 0: the built-in default, 0: off (GEMX_PACE_GBPS)
    // CLOSED LOOP (round 5): the built-in target is only where the calibration starts.  The first paced launches of a handle (per launch
    // signature: steps, workgroups, shape) take turns at target x {1, 0.93, 1.07, 0.86} and unpaced, each timed with a pair of HIP events on
    // the launch stream (harvested without ever blocking: hipEventQuery at later launches); when every candidate has PACE_CAL_SAMPLES
    // completed launches the fastest (by its best time) is kept for the rest of the handle's life.  GEMX_PACE_CAL=0 / an explicit
    // GEMX_PACE_GBPS: no calibration.  Results are unaffected either way (the limiter only delays block starts).
    struct PaceCal {
        static constexpr int NC = 5, SAMPLES = 6, RING = 16;  // (six rounds: the first ones run while the clocks still ramp)
        long long sig = -1;        // launch signature the state belongs to
        float center = 1.0f;       // the candidates are center x {1, 0.93, 1.07, 0.86} (+ unpaced): re-centred when the winner sits at an edge
        int recenters = 0;         //   ... at most MAX_RECENTER times (a hill climb from the built-in target)
        static constexpr int MAX_RECENTER = 4;
        int next = 0;              // launches handed out so far (candidate = next % NC)
        int chosen = -1;           // >= 0: calibration done, candidate index kept
        float best[NC] = {1e30f, 1e30f, 1e30f, 1e30f, 1e30f};
        int count[NC] = {0, 0, 0, 0, 0};   // completed, timed launches per candidate
        int issued[NC] = {0, 0, 0, 0, 0};  // launches handed out per candidate
        void *ev0[RING] = {}, *ev1[RING] = {};  // hipEvent_t pairs
        int ev_cand[RING];         // candidate of the launch a pair brackets, -1: free
        bool ev_init = false;
    } pcal;
    int pace_cal_on = 1;  // GEMX_PACE_CAL
    double pace_scale_last = 1.0;  // factor on the built-in target the last paced launch ran at (0: unpaced)
    int pace_cal_state = 0;        // of the last launch: 0 no calibration, 1 a calibration launch, 2 calibrated
    size_t pipe_occ_smem[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int pipe_occ[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ... and the workgroups per CU the runtime reports for it with this handle's LDS bytes (occupancy API, once)
    int pipe_regs[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // VGPRs of the pipelined kernel's shape k (hipFuncGetAttributes, once): the launcher's residency arithmetic
    unsigned pipe_attr_set = 0;  // bit k: hipFuncSetAttribute(max dynamic LDS) done for pipelined shape k (per handle = per device:
    bool attr_set = false;       //   the attribute is per device, and a handle is bound to one device and one kernel instantiation)
    int wg_per_cu = 0;           // single-wave kernel: resident workgroups per CU from its VGPR count (0: not queried yet)
    int pipe_shape = -1;      // GEMX_PIPE_SHAPE=0/1/2 forces <12,3> / <4,2> / <2,2> whenever it fits (tests: every shape on small N)
    int use_step_kernel = 1;  // K = 1 launches take step_kernel (GEMX_STEP_KERNEL=0: advance_kernel, for A/B runs and bit-identity tests)
    int use_pipe = -1;        // pipelined kernel: -1 / 1 whenever eligible (default), 0 never (GEMX_PIPE=0: A/B and bit-identity tests)
    int dcs_epw = 32;         // dc_stream_kernel: envs per workgroup where twice the workgroups still find a CU each (GEMX_DCS_EPW=64: always 64)
    int use_dc_stream = 1;    // dc_stream_kernel: 1 when eligible and N <= 64 * CUs (default), 2 at any N, 0 never (GEMX_DC_STREAM)
    int dcs_attr_set = 0;     // bit 0 / 1: the 64- / 32-env instantiation's dynamic-LDS attribute is set
    bool warned_fallback = false;  // the one-time note that fused rollouts of this handle take the single-wave kernel has been printed
    bool omega_is_init = true;  // every env's omega equals init[0] (constant-speed loads): false between gemx_set_state and the next full reset
    bool omega_unknown = false; // a state-changing call was CAPTURED into a graph: the host cannot know when it runs -> omega_is_init stays false
};

namespace gemx {
template <class R> inline const DevParams<R> &params_of(const gemx_handle *h);
template <> inline const DevParams<float> &params_of<float>(const gemx_handle *h) { return h->pf; }
template <> inline const DevParams<double> &params_of<double>(const gemx_handle *h) { return h->pd; }

int fail(int code, const char *fmt, ...);  // sets gemx_last_error(); defined in gemx_capi.hip

// one launcher per (system, converter, dtype) instantiation unit; dispatches on load / solver / interlocking
typedef int (*advance_fn)(gemx_handle *h, const void *actions, int K, void *obs, uint8_t *done, int obs_every, hipStream_t st);
}  // namespace gemx

namespace gemx {
// Every entry point that launches, allocates or copies runs on ITS handle's device and leaves the caller's current device (torch's,
// when called from Python) as it found it.
struct DeviceGuard {
    int prev = -1, dev;
    explicit DeviceGuard(int device) : dev(device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
}  // namespace gemx

#define GEMX_HIP_TRY(x)                                                                                              \
    do {                                                                                                             \
        hipError_t _e = (x);                                                                                         \
        if (_e != hipSuccess) return gemx::fail(GEMX_ERR_DEVICE, "%s failed: %s", #x, hipGetErrorString(_e));        \
    } while (0)
