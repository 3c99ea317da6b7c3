// gemx_refgen.hip -- device-side reference generation (second half): N x n_ref independent
// WienerProcessReferenceGenerator streams, i.e. what `MultipleReferenceGenerator([WienerProcessReferenceGenerator(...)] * n_ref)`
// produces for N envs.  Reference (paths relative to src/gym_electric_motor/reference_generators/):
//   SubepisodedReferenceGenerator  subepisoded_reference_generator.py:66-119  (sub-episodes of int(U(len_lo, len_hi)) steps;
//                                  reset(): reference value := initial reference, a new sub-episode starts at once)
//   WienerProcessReferenceGenerator wiener_process_reference_generator.py:30-49 (per sub-episode sigma = 10 ** U(log10 sigma_range),
//                                  value += N(0, sigma) per step, clipped to the limit margin; reset(): initial value ~ U(initial_range))
//   MultipleReferenceGenerator      multiple_reference_generator.py:77-92       (independent sub-generators, concatenated)
// numpy's PCG64 streams cannot be reproduced on a device: parity is DISTRIBUTIONAL (tests/test_gpu_parity.py); the arithmetic of the
// clipped random walk is the reference's.  Randomness: counter-based Philox4x32-10 (gemx_common.hpp) indexed by
// (seed; env, generator, draw kind, draw index), so chunked == one-shot generation and no RNG state is stored.
//
// Two kernels per call: (1) all K*N*n_ref standard normals in parallel, written into the output tensor; (2) one lane per
// (env, generator) walks its K steps sequentially: scale by the sub-episode's sigma, accumulate, clip, restart on `done`.
#include "gemx_common.hpp"

void gemx_cov_note(const char *key);  // gemx_capi.hip: instantiation coverage (GEMX_COVERAGE_FILE)

struct gemx_refgen {
    gemx_refgen_config cfg;
    int64_t n;
    int device, f64;
    // per (generator, env): current reference value, steps left in the sub-episode, sigma, sub-episode / reset counters
    double *value = nullptr, *sigma = nullptr;
    int32_t *left = nullptr;
    uint32_t *n_sub = nullptr, *n_reset = nullptr;
    unsigned long long t_total = 0;  // steps generated so far (index of the per-step normal draws)
};

namespace {

enum { DRAW_STEP = 0, DRAW_SUB = 1, DRAW_RESET = 2 };

__device__ inline void refgen_block(uint64_t seed, int64_t env, int gen, int kind, uint64_t index, uint32_t (&r)[4]) {
    // key: seed; counter: (env lo, env hi | gen << 24 | kind << 28, index lo, index hi)
    const uint64_t env_word = (uint64_t)env | ((uint64_t)gen << 56) | ((uint64_t)kind << 60);
    uint32_t c[4] = {(uint32_t)env_word, (uint32_t)(env_word >> 32), (uint32_t)index, (uint32_t)(index >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int i = 0; i < 10; ++i) {
        gemx::Philox::round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    for (int i = 0; i < 4; ++i) r[i] = c[i];
}

// (1) standard normals for steps t0 .. t0+K-1 of every (env, generator): Box-Muller on two Philox words
template <class R>
__global__ void refgen_normals_kernel(R *out, int64_t N, int n_ref, int K, uint64_t seed, uint64_t t0, int64_t env_base) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)K * N * n_ref;
    if (idx >= total) return;
    const int g = (int)(idx % n_ref);
    const int64_t env = (idx / n_ref) % N;
    const int64_t k = idx / ((int64_t)n_ref * N);
    uint32_t r[4];
    refgen_block(seed, env_base + env, g, DRAW_STEP, t0 + (uint64_t)k, r);  // (the GLOBAL env index keys the stream)
    const double u1 = gemx::Philox::u01(r[0]), u2 = gemx::Philox::u01(r[1]);
    out[idx] = (R)(sqrt(-2.0 * log(u1)) * cos(gemx::kTwoPi * u2));
}

struct RefgenDev {
    int32_t n_ref, len_lo, len_hi;
    uint64_t seed;
    int64_t env_base;  // gemx_refgen_config.env_base
    double log_sig_lo[GEMX_MAX_REF], log_sig_hi[GEMX_MAX_REF], m_lo[GEMX_MAX_REF], m_hi[GEMX_MAX_REF], i_lo[GEMX_MAX_REF], i_hi[GEMX_MAX_REF];
};

// SubepisodedReferenceGenerator.get_reference_observation, lines 104-111: a new sub-episode draws its length and its sigma
__device__ inline void new_subepisode(const RefgenDev &G, int64_t env, int g, uint32_t &n_sub, int32_t &left, double &sigma) {
    uint32_t r[4];
    refgen_block(G.seed, G.env_base + env, g, DRAW_SUB, n_sub++, r);
    // int((hi - lo) * U + lo), _get_current_value lines 116-119; then 10 ** U(log10 sigma_range), wiener ... line 31
    left = (int32_t)((double)(G.len_hi - G.len_lo) * gemx::Philox::u01(r[0]) + (double)G.len_lo);
    sigma = pow(10.0, (G.log_sig_hi[g] - G.log_sig_lo[g]) * gemx::Philox::u01(r[1]) + G.log_sig_lo[g]);
}
// WienerProcessReferenceGenerator.reset, lines 43-49: initial reference ~ U(initial_range); SubepisodedReferenceGenerator.reset 86-93
__device__ inline void reset_generator(const RefgenDev &G, int64_t env, int g, uint32_t &n_reset, uint32_t &n_sub, int32_t &left, double &sigma,
                                       double &value) {
    uint32_t r[4];
    refgen_block(G.seed, G.env_base + env, g, DRAW_RESET, n_reset++, r);
    value = (G.i_hi[g] - G.i_lo[g]) * gemx::Philox::u01(r[0]) + G.i_lo[g];
    left = 0;  // `_current_episode_length = -1`: the next get_reference_observation starts a sub-episode
    (void)n_sub; (void)sigma;
}
// one step of the clipped walk, wiener_process_reference_generator.py:35-41
__device__ inline double walk_step(const RefgenDev &G, int g, double value, double sigma, double z) {
    value += sigma * z;
    if (value > G.m_hi[g]) value = G.m_hi[g];
    if (value < G.m_lo[g]) value = G.m_lo[g];
    return value;
}

// (2) out[k][env][g] holds z on entry and the reference of step k on exit.  done[k][env] != 0: the env terminated in step k, its
// generators are reset before the reference of step k+1 is produced (env.reset() -> reference_generator.reset(), core.py:312-313).
template <class R>
__global__ void refgen_walk_kernel(R *out, const uint8_t *done, const uint8_t *reset_mask, int reset_all, int64_t N, int K, RefgenDev G,
                                   double *value, double *sigma, int32_t *left, uint32_t *n_sub, uint32_t *n_reset) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * G.n_ref) return;
    const int g = (int)(idx % G.n_ref);
    const int64_t env = idx / G.n_ref;
    const int64_t si = (int64_t)g * N + env;
    double v = value[si], sg = sigma[si];
    int32_t lf = left[si];
    uint32_t ns = n_sub[si], nr = n_reset[si];
    if (K == 0) {  // gemx_refgen_reset: the masked envs, or all of them (reset_all: no mask buffer at all)
        if (reset_all || (reset_mask != nullptr && reset_mask[env])) reset_generator(G, env, g, nr, ns, lf, sg, v);
    }
    for (int k = 0; k < K; ++k) {
        const int64_t o = ((int64_t)k * N + env) * G.n_ref + g;
        if (lf <= 0) new_subepisode(G, env, g, ns, lf, sg);
        v = walk_step(G, g, v, sg, (double)out[o]);
        --lf;
        out[o] = (R)v;
        if (done != nullptr && done[(int64_t)k * N + env]) reset_generator(G, env, g, nr, ns, lf, sg, v);
    }
    value[si] = v; sigma[si] = sg; left[si] = lf; n_sub[si] = ns; n_reset[si] = nr;
}

RefgenDev make_dev(const gemx_refgen_config &c) {
    RefgenDev G;
    memset(&G, 0, sizeof(G));
    G.n_ref = c.n_ref; G.len_lo = c.episode_len_lo; G.len_hi = c.episode_len_hi; G.seed = c.seed; G.env_base = c.env_base;
    for (int g = 0; g < c.n_ref; ++g) {
        G.log_sig_lo[g] = log10(c.sigma_lo[g]); G.log_sig_hi[g] = log10(c.sigma_hi[g]);
        G.m_lo[g] = c.margin_lo[g]; G.m_hi[g] = c.margin_hi[g]; G.i_lo[g] = c.initial_lo[g]; G.i_hi[g] = c.initial_hi[g];
    }
    return G;
}

template <class R> int walk(gemx_refgen *r, void *out, const uint8_t *done, const uint8_t *mask, int reset_all, int K, hipStream_t st) {
    const int64_t lanes = r->n * r->cfg.n_ref;
    gemx_cov_note(sizeof(R) == 4 ? "refgen_walk_kernel<float>" : "refgen_walk_kernel<double>");
    hipLaunchKernelGGL(refgen_walk_kernel<R>, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, (R *)out, done, mask, reset_all, r->n, K, make_dev(r->cfg),
                       r->value, r->sigma, r->left, r->n_sub, r->n_reset);
    GEMX_HIP_TRY(hipGetLastError());
    return GEMX_OK;
}

}  // namespace

extern "C" {

int gemx_refgen_create(const gemx_refgen_config *cfg, int64_t n_envs, int device, int dtype, gemx_refgen **out) {
    if (!cfg || !out) return gemx::fail(GEMX_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(gemx_refgen_config)) return gemx::fail(GEMX_ERR_ARG, "gemx_refgen_config size mismatch");
    if (cfg->env_base < 0) return gemx::fail(GEMX_ERR_ARG, "env_base must be >= 0");
    if (cfg->n_ref < 1 || cfg->n_ref > GEMX_MAX_REF) return gemx::fail(GEMX_ERR_ARG, "n_ref must be in [1, %d]", GEMX_MAX_REF);
    if (n_envs <= 0) return gemx::fail(GEMX_ERR_ARG, "n_envs must be positive");
    if (cfg->episode_len_lo < 1 || cfg->episode_len_hi < cfg->episode_len_lo) return gemx::fail(GEMX_ERR_ARG, "episode lengths must satisfy 1 <= lo <= hi");
    for (int g = 0; g < cfg->n_ref; ++g) {
        if (!(cfg->sigma_lo[g] > 0) || cfg->sigma_hi[g] < cfg->sigma_lo[g]) return gemx::fail(GEMX_ERR_ARG, "sigma range %d must satisfy 0 < lo <= hi", g);
        if (cfg->margin_hi[g] < cfg->margin_lo[g] || cfg->initial_hi[g] < cfg->initial_lo[g]) return gemx::fail(GEMX_ERR_ARG, "empty margin / initial range %d", g);
    }
    if (dtype != GEMX_F32 && dtype != GEMX_F64) return gemx::fail(GEMX_ERR_ARG, "unknown dtype");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return gemx::fail(GEMX_ERR_DEVICE, "no HIP device visible: there is no CPU fallback");
    if (device < 0 || device >= ndev) return gemx::fail(GEMX_ERR_ARG, "device %d out of range", device);
    gemx::DeviceGuard guard(device);
    gemx_refgen *r = new (std::nothrow) gemx_refgen();
    if (!r) return gemx::fail(GEMX_ERR_ALLOC, "out of host memory");
    r->cfg = *cfg; r->n = n_envs; r->device = device; r->f64 = dtype == GEMX_F64;
    const size_t m = (size_t)n_envs * cfg->n_ref;
    if (hipMalloc((void **)&r->value, m * 8) != hipSuccess || hipMalloc((void **)&r->sigma, m * 8) != hipSuccess ||
        hipMalloc((void **)&r->left, m * 4) != hipSuccess || hipMalloc((void **)&r->n_sub, m * 4) != hipSuccess ||
        hipMalloc((void **)&r->n_reset, m * 4) != hipSuccess) {
        gemx_refgen_destroy(r);
        return gemx::fail(GEMX_ERR_ALLOC, "hipMalloc(refgen) failed");
    }
    (void)hipMemset(r->value, 0, m * 8); (void)hipMemset(r->sigma, 0, m * 8); (void)hipMemset(r->left, 0, m * 4);
    (void)hipMemset(r->n_sub, 0, m * 4); (void)hipMemset(r->n_reset, 0, m * 4);
    *out = r;
    return GEMX_OK;
}

int gemx_refgen_destroy(gemx_refgen *r) {
    if (!r) return GEMX_OK;
    gemx::DeviceGuard guard(r->device);
    if (r->value) (void)hipFree(r->value);
    if (r->sigma) (void)hipFree(r->sigma);
    if (r->left) (void)hipFree(r->left);
    if (r->n_sub) (void)hipFree(r->n_sub);
    if (r->n_reset) (void)hipFree(r->n_reset);
    delete r;
    return GEMX_OK;
}

// reference_generator.reset() for the envs with mask != 0 (all if NULL): new initial reference value, a new sub-episode starts with
// the next generated step.
int gemx_refgen_reset(gemx_refgen *r, const uint8_t *mask_dev, void *stream) {
    if (!r) return gemx::fail(GEMX_ERR_ARG, "null handle");
    gemx::DeviceGuard guard(r->device);
    hipStream_t st = (hipStream_t)stream;
    const int all = mask_dev == nullptr;
    return r->f64 ? walk<double>(r, nullptr, nullptr, mask_dev, all, 0, st) : walk<float>(r, nullptr, nullptr, mask_dev, all, 0, st);
}

int gemx_refgen_rollout(gemx_refgen *r, const uint8_t *done_dev, int32_t K, void *refs_out_dev, void *stream) {
    if (!r || !refs_out_dev) return gemx::fail(GEMX_ERR_ARG, "null argument");
    if (K < 1) return gemx::fail(GEMX_ERR_ARG, "K must be >= 1");
    gemx::DeviceGuard guard(r->device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)K * r->n * r->cfg.n_ref;
    gemx_cov_note(r->f64 ? "refgen_normals_kernel<double>" : "refgen_normals_kernel<float>");
    if (r->f64)
        hipLaunchKernelGGL(refgen_normals_kernel<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (double *)refs_out_dev, r->n, r->cfg.n_ref, K,
                           r->cfg.seed, (uint64_t)r->t_total, r->cfg.env_base);
    else
        hipLaunchKernelGGL(refgen_normals_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (float *)refs_out_dev, r->n, r->cfg.n_ref, K,
                           r->cfg.seed, (uint64_t)r->t_total, r->cfg.env_base);
    GEMX_HIP_TRY(hipGetLastError());
    r->t_total += (unsigned long long)K;
    return r->f64 ? walk<double>(r, refs_out_dev, done_dev, nullptr, 0, K, st) : walk<float>(r, refs_out_dev, done_dev, nullptr, 0, K, st);
}

// debug / test access: per (generator, env) arrays [n_ref][N]: value (double), sigma (double), steps left (int32)
int gemx_refgen_get_state(gemx_refgen *r, double *value_out_dev, double *sigma_out_dev, int32_t *left_out_dev, void *stream) {
    if (!r) return gemx::fail(GEMX_ERR_ARG, "null handle");
    gemx::DeviceGuard guard(r->device);
    const size_t m = (size_t)r->n * r->cfg.n_ref;
    hipStream_t st = (hipStream_t)stream;
    if (value_out_dev) GEMX_HIP_TRY(hipMemcpyAsync(value_out_dev, r->value, m * 8, hipMemcpyDeviceToDevice, st));
    if (sigma_out_dev) GEMX_HIP_TRY(hipMemcpyAsync(sigma_out_dev, r->sigma, m * 8, hipMemcpyDeviceToDevice, st));
    if (left_out_dev) GEMX_HIP_TRY(hipMemcpyAsync(left_out_dev, r->left, m * 4, hipMemcpyDeviceToDevice, st));
    return GEMX_OK;
}

}  // extern "C"
