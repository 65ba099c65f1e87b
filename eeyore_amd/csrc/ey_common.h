// Shared host/device definitions for the eeyore_amd HIP library (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/eeyore_amd.h"

#define EY_MAX_LAYERS 8
#define EY_VERSION 200  // 0.2.0

// ----------------------------------------------------------------------------------------------- errors
void ey_set_error(const std::string& msg);
#define EY_FAIL(code, msg) \
  do {                     \
    ey_set_error(msg);     \
    return (code);         \
  } while (0)
#define EY_HIP(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      ey_set_error(std::string(#call) + ": " + hipGetErrorString(e_));                            \
      return EY_ERR_HIP;                                                                          \
    }                                                                                             \
  } while (0)

// ----------------------------------------------------------------------------------------------- plan
// Device-visible model description, passed to kernels by value.
struct EyModel {
  int nl;                        // number of layers K
  int dims[EY_MAX_LAYERS + 1];   // d_0..d_K
  int bias[EY_MAX_LAYERS];
  int act[EY_MAX_LAYERS];
  int woff[EY_MAX_LAYERS];       // offset of W_l in theta
  int boff[EY_MAX_LAYERS];       // offset of b_l in theta, -1 if none
  int hoff[EY_MAX_LAYERS + 1];   // first activation row of layer l in the per-tile activation store
  int hrows;                     // sum of d_l
  int dmax;                      // max d_l
  int lik;
  int P;
  int N;
  const void* x;        // [N, d0]
  const void* y;        // [N, dK]
  const int* labels;    // [N] argmax(y,1) (CE)
  const void* mu;       // [P]
  const void* inv_var;  // [P] 1/sigma^2
  double prior_const;   // sum_i (-log sigma_i - 0.5 log 2pi)
};

struct ey_plan {
  EyModel m;
  int dtype;
  int device;
  bool has_data, has_prior;
  void *d_x, *d_y, *d_mu, *d_inv_var;
  bool prior_uniform = false;  // every parameter has the same (mu, sigma)
  double prior_mu0 = 0.0, prior_iv0 = 0.0;
  // running moments attached with ey_plan_attach_moments (caller-owned device memory)
  double *mom_s1 = nullptr, *mom_s2 = nullptr, *mom_acc = nullptr;
  int64_t mom_C = 0;
  // per-chain dual averaging of the HMC step size, attached with ey_plan_attach_da (caller-owned device memory)
  double* da_state = nullptr;        // [C,3]: barh, logbare, mu
  void* da_step = nullptr;           // [C] of the plan's dtype: the step of the next iteration
  const double* da_table = nullptr;  // [da_n,3]: d_w, sqrt(t)/gamma, e_w of the k-th adapting iteration
  int64_t da_n = 0, da_done = 0, da_C = 0;
  double da_d = 0.65, da_logeub = 0.0;
  bool da_has_eub = false, da_final_avg = false;
  int* d_labels;
  // mfma32 path (4-32-32-3-like models, f32): padded/packed data image
  bool mfma32_ok;        // the model is one the fused kernel serves
  int mfma32_kind = 0;   // 1: the headline model; 2: served in the bf16x3 form only (ey_mfma32_kind)
  bool mfma32_data_ok;   // ... and the current batch fits its LDS image (recomputed by every ey_plan_set_data)
  void* d_xpack;
  int64_t cap_N;         // rows the data buffers below were allocated for (they only grow)
  // fused16 path (d0-H-H-dK with H in {16, 32, 64}, f32 and f64): operand-order data image in global memory
  bool fused16_ok = false;
  void* d_xpack16 = nullptr;
  size_t xpack16_bytes = 0;
  int n_cu;
  int variant = 0;   // diagnostic switches (ey_plan_set_variant; bits as documented in include/eeyore_amd.h)
  int products = 0;  // EY_OPT_F32_PRODUCTS: EY_PRODUCTS_BF16X3 (0) or EY_PRODUCTS_EXACT (1)
  int row_waves = 0; // EY_OPT_ROW_WAVES: EY_ROW_WAVES_OFF (0, default), _ON (1), _AUTO (2)
  // layerwise batched-GEMM path for models whose parameters do not fit LDS (ey_large.hip): workspace it owns
  void* d_work;
  size_t work_bytes;
  // ... and the data matrix split into bf16 pieces once per batch, in both orientations (ey_large.hip: ensure_xpre)
  void* d_xpre = nullptr;
  size_t xpre_bytes = 0;
  hipEvent_t xpre_event = nullptr;   // recorded behind the split on the stream that ran it: a consumer on another stream waits for it
  hipStream_t xpre_stream = nullptr;
  uint64_t data_version = 0, xpre_version = ~0ull;  // ey_plan_set_data counts; the images remember which batch they hold
};

// The diagnostic switches of the plan a C-ABI call is serving, for the dispatch code below the entry points (thread-local:
// plans on different threads are independent; an entry point sets it for its own duration).
extern thread_local int t_ey_variant, t_ey_products;
struct EyVariantScope {
  int old, oldp;
  explicit EyVariantScope(int v, int products = -1) : old(t_ey_variant), oldp(t_ey_products) {
    t_ey_variant = v;
    if (products >= 0) t_ey_products = products;
  }
  explicit EyVariantScope(const struct ey_plan* pl);
  ~EyVariantScope() { t_ey_variant = old; t_ey_products = oldp; }
};
#define EY_VBIT(b) ((t_ey_variant >> (b)) & 1)
int ey_default_variant();   // what plans created now start with (ey_debug_set_variant, EY_VARIANT)
int ey_default_products();  // ... and their EY_OPT_F32_PRODUCTS (EY_F32_PRODUCTS; bit 10 of ey_debug_set_variant: exact)

inline EyVariantScope::EyVariantScope(const ey_plan* pl) : EyVariantScope(pl->variant, pl->products) {}

// what a launch needs of the attached dual averaging: the rows of the table for its iterations
struct EyDA {
  double* state = nullptr;
  void* step = nullptr;
  const double* table = nullptr;  // first row = this launch's first iteration
  int n = 0;                      // adapting iterations in this launch (the rest of the launch keeps the last step)
  int final_it = -1;              // launch-local iteration that stores exp(logbare) instead of exp(loge); -1 = none
  double d = 0.65, logeub = 0.0;
  int has_eub = 0;
};

// One update of the recurrence of eeyore/tuners/hmcda_tuner.py:43-59 for one chain (Hoffman & Gelman 2014, algorithm 5):
// returns the step of the next iteration.  row = (d_w, sqrt(t)/gamma, e_w), worked out on the host for iteration t.
__host__ __device__ inline double ey_da_update(double* st, const double* row, double rate, double d, bool has_eub,
                                               double logeub, bool averaged) {
#pragma clang fp contract(off)  // separate multiplies and adds, as the host form of the recurrence rounds them
  double barh = st[0], logbare = st[1];
  const double m = st[2];
  const double r = rate == rate ? rate : 0.0;  // a NaN rate (a rejected non-finite trajectory) counts as 0
  barh += row[0] * ((d - r) - barh);
  double loge = m - row[1] * barh;
  if (has_eub && loge > logeub) loge = logeub;
  logbare += row[2] * (loge - logbare);
  st[0] = barh;
  st[1] = logbare;
  return exp(averaged ? logbare : loge);
}

// ey_hmc_run: n_iters consecutive draws inside one launch, with optional per-iteration records
struct EyRun {
  int n_iters;
  void* samples;    // [n_iters, C, P] of the plan's dtype, or null
  void* targets;    // [n_iters, C], or null
  void* accepted;   // [n_iters, C] uint8, or null
  int* accept_count;  // [C] int32 (+=), or null
};

// the attached moments from the records of a launch, in one streaming pass (ey_api.hip)
int ey_stats_update_run(const void* samples, const void* accepted_rec, int n_it, int64_t C, int64_t P, int dtype, void* s1,
                        void* s2, void* acc, void* stream);

// fused value + gradient of mid-size MLPs, f32 (ey_mid.hip)
bool ey_mid_supports(const ey_plan* pl);
int ey_mid_eval(ey_plan* pl, const float* theta, const float* temp, int C, float* lik_o, float* grad, hipStream_t s);
bool ey_mid32_supports(const ey_plan* pl);  // narrow deeper models: every hidden width <= 32, up to three hidden layers, d_0 <= 64
int ey_mid32_eval(ey_plan* pl, const float* theta, const float* temp, int C, float* lik_o, float* grad, void* scratch,
                  hipStream_t s);

// generic kernels (ey_generic.hip)
int ey_generic_log_target(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* lik, void* prior,
                          void* target, void* grad, hipStream_t s);
int ey_generic_log_lik_rows(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* rows, hipStream_t s);
int ey_generic_hmc(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                   const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                   uint64_t chain_offset, uint32_t flags, void* accepted, void* rate, void* hcur, void* hprop,
                   hipStream_t s, const EyRun* run = nullptr);
int ey_generic_leapfrog(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                        int64_t C, void* target, void* grad, hipStream_t s);
int ey_generic_mala(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                    const void* step_vec, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                    uint64_t chain_offset, void* accepted, void* log_rate, hipStream_t s, const EyRun* run = nullptr);
int ey_generic_mh(ey_plan* pl, void* theta, void* target, const void* z, const void* u, const void* scale,
                  const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, void* accepted,
                  void* log_rate, hipStream_t s, const EyRun* run = nullptr);

// layerwise batched-GEMM kernels for large models, f32 (ey_large.hip)
bool ey_large_needed(const ey_plan* pl, int nvec = 3);  // true when the generic kernels cannot hold the model in LDS
int ey_large_log_target(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* lik, void* prior,
                        void* target, void* grad, hipStream_t s);
int ey_large_hmc(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                 const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                 uint64_t chain_offset, uint32_t flags, void* accepted, void* rate, void* hcur, void* hprop,
                 hipStream_t s);
int ey_large_mala_mh(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                     const void* step_vec, const void* scale, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                     uint64_t chain_offset, void* accepted, void* log_rate, hipStream_t s);
int ey_large_leapfrog(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                      int64_t C, void* target, void* grad, hipStream_t s);
int ey_large_log_lik_rows(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* rows, hipStream_t s);
void ey_large_free(ey_plan* pl);

// mfma32 kernels (ey_mfma32.hip)
bool ey_mfma32_supports(const ey_plan* pl);
int ey_mfma32_kind(const ey_plan* pl);  // 1 the headline model, 2 served in the bf16x3 form only, 0 not served
int ey_mfma32_set_data(ey_plan* pl, hipStream_t s);
int ey_mfma32_hmc(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                  const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                  uint64_t chain_offset, uint32_t flags, void* accepted, void* rate, void* hcur, void* hprop,
                  hipStream_t s, const EyRun* run = nullptr, const EyDA* da = nullptr);
int ey_mfma32_log_target_grad(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* target, void* grad,
                              hipStream_t s);
int ey_mfma32_mala(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                   const void* step_vec, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                   uint64_t chain_offset, void* accepted, void* log_rate, hipStream_t s, const EyRun* run = nullptr);
int ey_mfma32_mh(ey_plan* pl, void* theta, void* target, const void* z, const void* u, const void* scale,
                 const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, void* accepted,
                 void* log_rate, hipStream_t s, const EyRun* run = nullptr);
int ey_mfma32_leapfrog(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                       int64_t C, void* target, void* grad, hipStream_t s);

// fused kernels on 16x16x4 tiles (ey_fused16.hip)
bool ey_fused16_supports(const ey_plan* pl);
int ey_fused16_set_data(ey_plan* pl, hipStream_t s);
int ey_fused16_hmc(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                   const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                   uint64_t chain_offset, uint32_t flags, void* accepted, void* rate, void* hcur, void* hprop,
                   hipStream_t s, const EyRun* run = nullptr, const EyDA* da = nullptr);
int ey_fused16_mala(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                    const void* step_vec, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                    uint64_t chain_offset, void* accepted, void* log_rate, hipStream_t s, const EyRun* run = nullptr);
int ey_fused16_mh(ey_plan* pl, void* theta, void* target, const void* z, const void* u, const void* scale,
                  const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, void* accepted,
                  void* log_rate, hipStream_t s, const EyRun* run = nullptr);
int ey_fused16_log_target(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* lik, void* prior,
                          void* target, void* grad, hipStream_t s);
int ey_fused16_leapfrog(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                        int64_t C, void* target, void* grad, hipStream_t s);

// ----------------------------------------------------------------------------------------------- Philox4x32-10
// Counter-based generator (Salmon et al. 2011).  One call per (element, chain, iteration, stream): the value a
// lane needs never depends on which lane asks, so every kernel layout reproduces the same stream.
#define EY_STREAM_NORMAL 0u
#define EY_STREAM_UNIFORM 1u

struct EyRng {
  uint32_t k0, k1;  // key = seed
  uint32_t c1, c2, c3;  // chain, iter_lo, (iter_hi << 8) | stream
};

__host__ __device__ inline void ey_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += W0;
    k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__host__ __device__ inline EyRng ey_rng_make(uint64_t seed, uint64_t chain, uint64_t iter, uint32_t stream) {
  EyRng r;
  r.k0 = (uint32_t)seed;
  r.k1 = (uint32_t)(seed >> 32);
  r.c1 = (uint32_t)chain;
  r.c2 = (uint32_t)iter;
  r.c3 = ((uint32_t)(iter >> 32) << 8) | (stream & 0xffu) | ((uint32_t)(chain >> 32) << 20);
  return r;
}

// N(0,1) for the four elements 4b .. 4b+3 of a chain's stream ("block" b): element idx is component idx & 3 of block
// idx >> 2, whoever computes it, so every kernel layout reproduces the same stream.  f32: ONE Philox call per block,
// two Box-Muller pairs on 24-bit uniforms (o0,o1 -> elements 0,1 as r cos, r sin; o2,o3 -> elements 2,3).  f64: two
// calls (counters 2b and 2b+1), one Box-Muller pair on 32+21-bit uniforms each.  The 32-bit integer multiplies of
// Philox are quarter rate and the generator is a visible share of a draw (all of a MALA / MH draw's prologue), so
// producers generate whole blocks.  Contraction is off so that the fused kernels and ey_philox_normal produce
// bit-identical values.
template <typename T>
__device__ inline void ey_rng_normal4(const EyRng& r, uint32_t block, T out[4]);

template <>
__device__ inline void ey_rng_normal4<float>(const EyRng& r, uint32_t block, float out[4]) {
#pragma clang fp contract(off)
  uint32_t o[4];
  ey_philox4x32_10(block, r.c1, r.c2, r.c3, r.k0, r.k1, o);
#pragma unroll
  for (int pair = 0; pair < 2; ++pair) {
    const float u1 = ((float)(o[2 * pair] >> 8) + 0.5f) * 5.9604644775390625e-08f;  // (0,1)
    const float u2 = (float)(o[2 * pair + 1] >> 8) * 5.9604644775390625e-08f;       // [0,1)
    const float rad = sqrtf(-2.0f * logf(u1));
    out[2 * pair] = rad * cospif(2.0f * u2);
    out[2 * pair + 1] = rad * sinpif(2.0f * u2);
  }
}

template <>
__device__ inline void ey_rng_normal4<double>(const EyRng& r, uint32_t block, double out[4]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int pair = 0; pair < 2; ++pair) {
    uint32_t o[4];
    ey_philox4x32_10(2u * block + (uint32_t)pair, r.c1, r.c2, r.c3, r.k0, r.k1, o);
    const double u1 = ((double)(((uint64_t)o[0] << 21) | (o[2] >> 11)) + 0.5) * 1.1102230246251565e-16;  // 2^-53
    const double u2 = (double)(((uint64_t)o[1] << 21) | (o[3] >> 11)) * 1.1102230246251565e-16;
    const double rad = sqrt(-2.0 * log(u1));
    out[2 * pair] = rad * cospi(2.0 * u2);
    out[2 * pair + 1] = rad * sinpi(2.0 * u2);
  }
}

// one element (for consumers that are handed single indices): the same arithmetic, the other components discarded
template <typename T>
__device__ inline T ey_rng_normal(const EyRng& r, uint32_t idx) {
  T o[4];
  ey_rng_normal4<T>(r, idx >> 2, o);
  return o[idx & 3u];
}

template <typename T>
__device__ inline T ey_rng_uniform(const EyRng& r);

template <>
__device__ inline float ey_rng_uniform<float>(const EyRng& r) {
  uint32_t o[4];
  ey_philox4x32_10(0u, r.c1, r.c2, r.c3, r.k0, r.k1, o);
  return (float)(o[0] >> 8) * 5.9604644775390625e-08f;
}
template <>
__device__ inline double ey_rng_uniform<double>(const EyRng& r) {
  uint32_t o[4];
  ey_philox4x32_10(0u, r.c1, r.c2, r.c3, r.k0, r.k1, o);
  return (double)(((uint64_t)o[0] << 21) | (o[1] >> 11)) * 1.1102230246251565e-16;
}
