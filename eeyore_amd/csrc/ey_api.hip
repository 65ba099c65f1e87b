// C-ABI entry points (include/eeyore_amd.h): argument validation, plan management, dispatch to the kernel
// families.  No torch types cross this boundary.
#include <math.h>

#include <atomic>
#include <cstring>
#include <vector>

#include "ey_common.h"

static thread_local std::string g_err;
void ey_set_error(const std::string& msg) { g_err = msg; }

// labels = argmax(y, 1), first maximal index (eeyore/constants/constants.py:17)
template <typename T>
__global__ void k_labels(const T* __restrict__ y, int* __restrict__ labels, int64_t N, int dK) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int best = 0;
  T bv = y[n * dK];
  for (int j = 1; j < dK; ++j) {
    const T v = y[n * dK + j];
    if (v > bv) { bv = v; best = j; }
  }
  labels[n] = best;
}

thread_local int t_ey_variant = 0, t_ey_products = EY_PRODUCTS_BF16X3;
// what a new plan starts with: EY_VARIANT / EY_F32_PRODUCTS in the environment, or ey_debug_set_variant
static std::atomic<int> g_ey_default_variant{[] { const char* e = getenv("EY_VARIANT"); return e ? atoi(e) & 32767 : 0; }()};
int ey_default_variant() { return g_ey_default_variant.load() & (32767 & ~1024); }
int ey_default_products() {
  if (g_ey_default_variant.load() & 1024) return EY_PRODUCTS_EXACT;
  const char* e = getenv("EY_F32_PRODUCTS");
  return (e && (!strcmp(e, "exact") || !strcmp(e, "1"))) ? EY_PRODUCTS_EXACT : EY_PRODUCTS_BF16X3;
}

extern "C" {

int ey_version(void) { return EY_VERSION; }
const char* ey_last_error(void) { return g_err.c_str(); }

int ey_plan_create(ey_plan** out, int n_layers, const int* dims, const int* bias, const int* act, int likelihood,
                   int dtype, int device_id) {
  if (!out || !dims || !act) EY_FAIL(EY_ERR_INVALID, "ey_plan_create: null argument");
  // Hyperparameters: len(dims) >= 3 and len(dims) == len(activations)+1 (eeyore/models/mlp.py:15-19).  A
  // single layer (logistic regression, eeyore/models/logistic_regression.py) is accepted as the K=1 case.
  if (n_layers < 1 || n_layers > EY_MAX_LAYERS) EY_FAIL(EY_ERR_INVALID, "ey_plan_create: n_layers must be in 1..8");
  if (dtype != EY_F32 && dtype != EY_F64) EY_FAIL(EY_ERR_INVALID, "ey_plan_create: dtype must be EY_F32 or EY_F64");
  if (likelihood != EY_LIK_BCE_SUM && likelihood != EY_LIK_CE_SUM)
    EY_FAIL(EY_ERR_INVALID, "ey_plan_create: unknown likelihood");
  for (int l = 0; l <= n_layers; ++l)
    if (dims[l] < 1) EY_FAIL(EY_ERR_INVALID, "ey_plan_create: dims must be positive");
  for (int l = 0; l < n_layers; ++l)
    if (act[l] < EY_ACT_NONE || act[l] > EY_ACT_RELU) EY_FAIL(EY_ERR_INVALID, "ey_plan_create: unknown activation");
  int ndev = 0;
  EY_HIP(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) EY_FAIL(EY_ERR_INVALID, "ey_plan_create: no such device");
  ey_plan* pl = new ey_plan();
  EyModel& m = pl->m;
  m.nl = n_layers;
  int P = 0, hr = 0, dmax = 0;
  for (int l = 0; l <= n_layers; ++l) {
    m.dims[l] = dims[l];
    m.hoff[l] = hr;
    hr += dims[l];
    if (dims[l] > dmax) dmax = dims[l];
  }
  for (int l = 0; l < n_layers; ++l) {
    m.bias[l] = bias ? (bias[l] != 0) : 1;
    m.act[l] = act[l];
    m.woff[l] = P;
    P += dims[l + 1] * dims[l];
    m.boff[l] = m.bias[l] ? P : -1;
    if (m.bias[l]) P += dims[l + 1];
  }
  m.hrows = hr;
  m.dmax = dmax;
  m.lik = likelihood;
  m.P = P;
  m.N = 0;
  m.x = m.y = m.mu = m.inv_var = nullptr;
  m.labels = nullptr;
  m.prior_const = 0.0;
  pl->dtype = dtype;
  pl->device = device_id;
  pl->has_data = pl->has_prior = false;
  pl->d_x = pl->d_y = pl->d_mu = pl->d_inv_var = nullptr;
  pl->d_labels = nullptr;
  pl->d_xpack = nullptr;
  pl->cap_N = 0;
  pl->mfma32_data_ok = false;
  pl->d_work = nullptr;
  pl->work_bytes = 0;
  hipDeviceProp_t prop;
  EY_HIP(hipGetDeviceProperties(&prop, device_id));
  pl->n_cu = prop.multiProcessorCount;
  pl->variant = g_ey_default_variant.load() & (32767 & ~1024);
  pl->products = (g_ey_default_variant.load() & 1024) ? EY_PRODUCTS_EXACT : ey_default_products();
  {
    const char* e = getenv("EY_ROW_WAVES");
    pl->row_waves = (e && e[0] >= '0' && e[0] <= '2' && !e[1]) ? e[0] - '0' : EY_ROW_WAVES_OFF;
  }
  pl->mfma32_kind = ey_mfma32_kind(pl);
  pl->mfma32_ok = pl->mfma32_kind != 0;
  pl->fused16_ok = ey_fused16_supports(pl);  // also the headline model's second choice (batches beyond mfma32's row limit)
  *out = pl;
  return EY_OK;
}

int ey_plan_destroy(ey_plan* pl) {
  if (!pl) return EY_OK;
  (void)hipSetDevice(pl->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(pl->d_x);
  (void)hipFree(pl->d_y);
  (void)hipFree(pl->d_mu);
  (void)hipFree(pl->d_inv_var);
  (void)hipFree(pl->d_labels);
  (void)hipFree(pl->d_xpack);
  (void)hipFree(pl->d_xpack16);
  ey_large_free(pl);
  delete pl;
  return EY_OK;
}

int ey_plan_num_params(const ey_plan* pl, int64_t* P) {
  if (!pl || !P) EY_FAIL(EY_ERR_INVALID, "ey_plan_num_params: null argument");
  *P = pl->m.P;
  return EY_OK;
}

// Diagnostic switches for A/B runs and tests (not part of the reference-facing surface).  They belong to the plan:
// ey_plan_set_variant changes one plan, ey_debug_set_variant the value plans created afterwards start with (and what
// ey_debug_bgemm, which has no plan, runs under).  Both return the previous value.
extern "C" int ey_debug_set_variant(int v) { return g_ey_default_variant.exchange(v & 32767); }
extern "C" int ey_plan_set_variant(ey_plan* pl, int v) {
  if (!pl) return -1;
  const int old = pl->variant;
  pl->variant = v & (32767 & ~1024);
  return old;
}
extern "C" int ey_plan_set_option(ey_plan* pl, int option, int value) {
  if (!pl) EY_FAIL(EY_ERR_INVALID, "ey_plan_set_option: null plan");
  if (option == EY_OPT_F32_PRODUCTS) {
    if (value != EY_PRODUCTS_BF16X3 && value != EY_PRODUCTS_EXACT)
      EY_FAIL(EY_ERR_INVALID, "ey_plan_set_option: EY_OPT_F32_PRODUCTS takes EY_PRODUCTS_BF16X3 or EY_PRODUCTS_EXACT");
    pl->products = value;
    return EY_OK;
  }
  if (option == EY_OPT_ROW_WAVES) {
    if (value < EY_ROW_WAVES_OFF || value > EY_ROW_WAVES_AUTO)
      EY_FAIL(EY_ERR_INVALID, "ey_plan_set_option: EY_OPT_ROW_WAVES takes EY_ROW_WAVES_OFF, _ON or _AUTO");
    pl->row_waves = value;
    return EY_OK;
  }
  EY_FAIL(EY_ERR_INVALID, "ey_plan_set_option: unknown option");
}
extern "C" int ey_plan_get_option(const ey_plan* pl, int option, int* value) {
  if (!pl || !value) EY_FAIL(EY_ERR_INVALID, "ey_plan_get_option: null argument");
  if (option == EY_OPT_F32_PRODUCTS) { *value = pl->products; return EY_OK; }
  if (option == EY_OPT_ROW_WAVES) { *value = pl->row_waves; return EY_OK; }
  EY_FAIL(EY_ERR_INVALID, "ey_plan_get_option: unknown option");
}
// the fused MFMA kernel serves this plan with the batch it currently holds
static bool use_mfma32(const ey_plan* pl) {
  if (!pl->mfma32_ok || !(pl->mfma32_data_ok || !pl->has_data)) return false;
  if (pl->mfma32_kind == 2)  // the other 4-32-32 models: the bf16x3 form only, whose LDS image holds 16 row tiles
    return pl->products == EY_PRODUCTS_BF16X3 && !(t_ey_variant & 1) && (!pl->has_data || pl->m.N <= 512);
  return true;
}
// nvec: state vectors the generic kernel of the calling operation keeps in LDS (2 value/MH, 3 HMC, 4 MALA)
// The layerwise path is NEEDED when the generic kernel's LDS image does not fit, and PREFERRED for models that fit but
// are wide enough for 32-wide matrix tiles to beat one wave's vector ALUs: measured (tools/route_probe.py, any number
// of chains, any row count) the crossover is at sum_l d_l d_{l+1} ~ 600 in f32 (MLP(4-70-3) 0.9 x, MLP(12-48-6) 2 x,
// MLP(10-100-10) 6-7 x the generic kernel) and below 240 in f64 (MLP(6-24-4) 1.3 x, MLP(4-70-3) 2.3 x).  Re-measured
// over 64 .. 16384 chains with the blocked generic loop (tools/route_probe_chains.py): MLP(10-30-10), sum 600, is
// 1.6-2.9 x faster layerwise in f32 at every chain count, MLP(4-70-3), sum 490, 0.7 x from 1024 chains up: 560.
// EY_FORCE_GENERIC overrides the preference, not the need.
static bool prefer_large(const ey_plan* pl) {
  long w = 0;
  for (int l = 0; l < pl->m.nl; ++l) w += (long)pl->m.dims[l] * pl->m.dims[l + 1];
  return w >= (pl->dtype == EY_F32 ? 560 : 200);
}
static bool use_large(const ey_plan* pl, int nvec = 3, uint32_t flags = 0) {
  if (ey_large_needed(pl, nvec)) return true;
  if (flags & EY_FORCE_GENERIC) return false;
  if (EY_VBIT(4) && !pl->mfma32_ok) return true;
  return prefer_large(pl);
}
// the fused 16x16x4 kernels serve this plan (any batch size: their data image lives in global memory)
static bool use_fused16(const ey_plan* pl) { return pl->fused16_ok && !use_mfma32(pl) && !EY_VBIT(4); }
const char* ey_plan_kernel(const ey_plan* pl) {
  if (!pl) return "generic";
  EyVariantScope vs(pl);
  if (use_mfma32(pl)) return "mfma32";
  if (use_fused16(pl)) return "fused16";
  return use_large(pl) ? "bgemm" : "generic";
}

static size_t esize(const ey_plan* pl) { return pl->dtype == EY_F32 ? 4 : 8; }

// Asynchronous on `stream` (a sampler calls this once per minibatch, eeyore/samplers/serial_sampler.py:41-46): the
// plan's copies of the batch are written by device-to-device copies and two small kernels ordered on the stream; the
// buffers only grow, so the only synchronisation is the reallocation when a batch is larger than any before it.
int ey_plan_set_data(ey_plan* pl, const void* x, const void* y, int64_t N, void* stream) {
  if (!pl || !x || !y) EY_FAIL(EY_ERR_INVALID, "ey_plan_set_data: null argument");
  if (N < 1 || N > (1 << 24)) EY_FAIL(EY_ERR_INVALID, "ey_plan_set_data: N out of range");
  hipStream_t s = (hipStream_t)stream;
  EY_HIP(hipSetDevice(pl->device));
  EyModel& m = pl->m;
  const size_t es = esize(pl);
  const int d0 = m.dims[0], dK = m.dims[m.nl];
  if (N > pl->cap_N) {
    EY_HIP(hipDeviceSynchronize());  // launches on any stream may still read the old buffers
    (void)hipFree(pl->d_x); (void)hipFree(pl->d_y); (void)hipFree(pl->d_labels);
    pl->d_x = pl->d_y = nullptr; pl->d_labels = nullptr;
    pl->cap_N = 0;
    EY_HIP(hipMalloc(&pl->d_x, es * N * d0));
    EY_HIP(hipMalloc(&pl->d_y, es * N * dK));
    EY_HIP(hipMalloc((void**)&pl->d_labels, sizeof(int) * N));
    pl->cap_N = N;
  }
  EY_HIP(hipMemcpyAsync(pl->d_x, x, es * N * d0, hipMemcpyDeviceToDevice, s));
  EY_HIP(hipMemcpyAsync(pl->d_y, y, es * N * dK, hipMemcpyDeviceToDevice, s));
  const dim3 grid((unsigned)((N + 255) / 256));
  if (es == 4) hipLaunchKernelGGL(k_labels<float>, grid, dim3(256), 0, s, (const float*)pl->d_y, pl->d_labels, N, dK);
  else hipLaunchKernelGGL(k_labels<double>, grid, dim3(256), 0, s, (const double*)pl->d_y, pl->d_labels, N, dK);
  EY_HIP(hipGetLastError());
  m.N = (int)N;
  ++pl->data_version;
  m.x = pl->d_x;
  m.y = pl->d_y;
  m.labels = pl->d_labels;
  pl->has_data = true;
  if (pl->mfma32_ok) {
    int rc = ey_mfma32_set_data(pl, s);
    if (rc) return rc;
  }
  if (pl->fused16_ok && (pl->mfma32_kind == 2 || !(pl->mfma32_ok && pl->mfma32_data_ok))) {  // kind 2: either may serve
    int rc = ey_fused16_set_data(pl, s);
    if (rc) return rc;
  }
  return EY_OK;
}

int ey_plan_set_prior(ey_plan* pl, const void* mu, const void* sigma, void* stream) {
  if (!pl || !mu || !sigma) EY_FAIL(EY_ERR_INVALID, "ey_plan_set_prior: null argument");
  hipStream_t s = (hipStream_t)stream;
  EY_HIP(hipSetDevice(pl->device));
  EyModel& m = pl->m;
  const size_t es = esize(pl);
  const int P = m.P;
  EY_HIP(hipStreamSynchronize(s));
  if (!pl->d_mu) EY_HIP(hipMalloc(&pl->d_mu, es * P));
  if (!pl->d_inv_var) EY_HIP(hipMalloc(&pl->d_inv_var, es * P));
  std::vector<unsigned char> hs(es * P);
  EY_HIP(hipMemcpyAsync(pl->d_mu, mu, es * P, hipMemcpyDeviceToDevice, s));
  EY_HIP(hipMemcpyAsync(hs.data(), sigma, es * P, hipMemcpyDeviceToHost, s));
  std::vector<unsigned char> hm(es * P);
  EY_HIP(hipMemcpyAsync(hm.data(), mu, es * P, hipMemcpyDeviceToHost, s));
  EY_HIP(hipStreamSynchronize(s));
  // Normal.log_prob = -(v-mu)^2/(2 sigma^2) - log(sigma) - log(sqrt(2 pi)); the theta-independent part is
  // summed once here (in double), the quadratic part is evaluated per call with 1/sigma^2.
  double c = 0.0;
  std::vector<unsigned char> hiv(es * P);
  for (int i = 0; i < P; ++i) {
    const double sg = es == 4 ? (double)((const float*)hs.data())[i] : ((const double*)hs.data())[i];
    if (!(sg > 0.0)) EY_FAIL(EY_ERR_INVALID, "ey_plan_set_prior: sigma must be positive");
    c += -log(sg) - 0.91893853320467274178;
    if (es == 4) ((float*)hiv.data())[i] = 1.0f / ((float)sg * (float)sg);
    else ((double*)hiv.data())[i] = 1.0 / (sg * sg);
  }
  EY_HIP(hipMemcpy(pl->d_inv_var, hiv.data(), es * P, hipMemcpyHostToDevice));
  // one (mu, sigma) for every parameter (the usual N(0, s) prior): the fused kernel then needs no per-element loads
  pl->prior_uniform = P > 0 && memcmp(hm.data(), hm.data() + es, es * (P - 1)) == 0 &&
                      memcmp(hiv.data(), hiv.data() + es, es * (P - 1)) == 0;
  pl->prior_mu0 = es == 4 ? (double)((const float*)hm.data())[0] : ((const double*)hm.data())[0];
  pl->prior_iv0 = es == 4 ? (double)((const float*)hiv.data())[0] : ((const double*)hiv.data())[0];
  m.mu = pl->d_mu;
  m.inv_var = pl->d_inv_var;
  m.prior_const = c;
  pl->has_prior = true;
  return EY_OK;
}

extern "C" int ey_stats_update(const void* theta, const void* accepted, int64_t C, int64_t P, int dtype, void* s1,
                               void* s2, void* acc, void* stream);
static int moments_check(const ey_plan* pl, int64_t C, const char* who) {
  if (pl && pl->mom_s1 && pl->mom_C != C) {
    ey_set_error(std::string(who) + ": the attached moments were sized for " + std::to_string(pl->mom_C) +
                 " chains, called with " + std::to_string(C));
    return EY_ERR_INVALID;
  }
  return EY_OK;
}
// trailing pass for the kernel families that do not fuse the accumulation
static int moments_trailing(ey_plan* pl, int rc, const void* theta, const void* accepted, int64_t C, void* stream) {
  if (rc != EY_OK || !pl->mom_s1) return rc;
  return ey_stats_update(theta, accepted, C, pl->m.P, pl->dtype, pl->mom_s1, pl->mom_s2, pl->mom_acc, stream);
}

// The window of the attached dual averaging that a launch of n_iters HMC iterations covers (nothing once the table is
// used up); `advance` moves the plan's position in the table past it.
static EyDA da_window(ey_plan* pl, int n_iters) {
  EyDA w;
  if (!pl->da_state) return w;
  w.step = pl->da_step;  // the attached step vector stays the step of every launch until it is detached
  if (pl->da_done >= pl->da_n) return w;  // the table is used up: nothing adapts any more
  w.state = pl->da_state;
  w.table = pl->da_table + 3 * pl->da_done;
  w.n = (int)std::min<int64_t>(n_iters, pl->da_n - pl->da_done);
  w.final_it = (pl->da_final_avg && pl->da_done + n_iters >= pl->da_n) ? (int)(pl->da_n - 1 - pl->da_done) : -1;
  w.d = pl->da_d;
  w.logeub = pl->da_logeub;
  w.has_eub = pl->da_has_eub ? 1 : 0;
  return w;
}
static int da_check(const ey_plan* pl, int64_t C, bool fused, const char* who) {
  if (!pl->da_state) return EY_OK;
  if (pl->da_C != C)
    EY_FAIL(EY_ERR_INVALID, std::string(who) + ": the attached dual averaging was sized for " + std::to_string(pl->da_C) +
                            " chains, called with " + std::to_string(C));
  if (!fused && pl->da_done < pl->da_n)
    EY_FAIL(EY_ERR_UNSUPPORTED, std::string(who) + ": in-kernel dual averaging needs one of the fused kernel families "
                                                   "(mfma32, fused16); detach it and adapt on the host for this plan");
  return EY_OK;
}

static int check_ready(const ey_plan* pl, int64_t C, const char* who) {
  if (!pl) EY_FAIL(EY_ERR_INVALID, std::string(who) + ": null plan");
  if (!pl->has_data) EY_FAIL(EY_ERR_STATE, std::string(who) + ": ey_plan_set_data has not been called");
  if (!pl->has_prior) EY_FAIL(EY_ERR_STATE, std::string(who) + ": ey_plan_set_prior has not been called");
  if (C < 0 || C > 0x7fffffffLL) EY_FAIL(EY_ERR_INVALID, std::string(who) + ": chain count out of range");
  return C == 0 ? 1 : EY_OK;  // 1 = nothing to do (an empty chain batch has null buffers)
}

int ey_log_target(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* log_lik, void* log_prior,
                  void* stream) {
  if (!pl) EY_FAIL(EY_ERR_INVALID, "ey_log_target: null plan");
  EyVariantScope vs(pl);
  // log_prior alone (bayesian_model.py:46-50) needs no data: with no rows attached the likelihood sum is empty
  if (!log_lik && !pl->has_data && pl->has_prior) {
    if (!theta) EY_FAIL(EY_ERR_INVALID, "ey_log_target: null theta");
    if (C <= 0) return EY_OK;
    EY_HIP(hipSetDevice(pl->device));
    return ey_generic_log_target(pl, theta, temp, C, nullptr, log_prior, nullptr, nullptr, (hipStream_t)stream);
  }
  int rc = check_ready(pl, C, "ey_log_target");
  if (rc) return rc < 0 ? rc : EY_OK;
  if (!theta) EY_FAIL(EY_ERR_INVALID, "ey_log_target: null theta");
  if (C == 0) return EY_OK;
  EY_HIP(hipSetDevice(pl->device));
  if (use_fused16(pl)) return ey_fused16_log_target(pl, theta, temp, C, log_lik, log_prior, nullptr, nullptr, (hipStream_t)stream);
  if (use_large(pl)) return ey_large_log_target(pl, theta, temp, C, log_lik, log_prior, nullptr, nullptr, (hipStream_t)stream);
  return ey_generic_log_target(pl, theta, temp, C, log_lik, log_prior, nullptr, nullptr, (hipStream_t)stream);
}

int ey_log_lik_rows(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* rows, void* stream) {
  int rc = check_ready(pl, C, "ey_log_lik_rows");
  if (rc) return rc < 0 ? rc : EY_OK;
  EyVariantScope vs(pl);
  if (!theta || !rows) EY_FAIL(EY_ERR_INVALID, "ey_log_lik_rows: null argument");
  if (C == 0) return EY_OK;
  EY_HIP(hipSetDevice(pl->device));
  if (use_large(pl)) return ey_large_log_lik_rows(pl, theta, temp, C, rows, (hipStream_t)stream);
  return ey_generic_log_lik_rows(pl, theta, temp, C, rows, (hipStream_t)stream);
}

int ey_log_target_grad(ey_plan* pl, const void* theta, const void* temp, int64_t C, void* target, void* grad,
                       void* stream) {
  int rc = check_ready(pl, C, "ey_log_target_grad");
  if (rc) return rc < 0 ? rc : EY_OK;
  EyVariantScope vs(pl);
  if (!theta || !target || !grad) EY_FAIL(EY_ERR_INVALID, "ey_log_target_grad: null argument");
  if (C == 0) return EY_OK;
  EY_HIP(hipSetDevice(pl->device));
  if (use_mfma32(pl)) return ey_mfma32_log_target_grad(pl, theta, temp, C, target, grad, (hipStream_t)stream);
  if (use_fused16(pl)) return ey_fused16_log_target(pl, theta, temp, C, nullptr, nullptr, target, grad, (hipStream_t)stream);
  if (use_large(pl)) return ey_large_log_target(pl, theta, temp, C, nullptr, nullptr, target, grad, (hipStream_t)stream);
  return ey_generic_log_target(pl, theta, temp, C, nullptr, nullptr, target, grad, (hipStream_t)stream);
}

int ey_hmc_step(ey_plan* pl, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                uint64_t chain_offset, uint32_t flags, void* accepted, void* accept_rate, void* H_cur, void* H_prop,
                void* stream) {
  int rc = check_ready(pl, C, "ey_hmc_step");
  if (rc) return rc < 0 ? rc : EY_OK;
  EyVariantScope vs(pl);
  if (!theta || !target || !grad || !accepted) EY_FAIL(EY_ERR_INVALID, "ey_hmc_step: null argument");
  if (L < 1) EY_FAIL(EY_ERR_INVALID, "ey_hmc_step: num_steps must be >= 1");
  if (C == 0) return EY_OK;
  if ((rc = moments_check(pl, C, "ey_hmc_step"))) return rc;
  const bool fused = (use_mfma32(pl) || use_fused16(pl)) && !(flags & EY_FORCE_GENERIC);
  if ((rc = da_check(pl, C, fused, "ey_hmc_step"))) return rc;
  EY_HIP(hipSetDevice(pl->device));
  const EyDA da = da_window(pl, 1);
  if (da.step) step_vec = da.step;  // every kernel family: the attached step vector is the step
  if (use_mfma32(pl) && !(flags & EY_FORCE_GENERIC)) {  // accumulates attached moments itself
    rc = ey_mfma32_hmc(pl, theta, target, grad, p0, u, step, step_vec, L, temp, C, seed, iter, chain_offset, flags,
                       accepted, accept_rate, H_cur, H_prop, (hipStream_t)stream, nullptr, &da);
    if (rc == EY_OK && da.state) pl->da_done += 1;  // the table advances only past iterations that were launched
    return rc;
  }
  if (use_fused16(pl) && !(flags & EY_FORCE_GENERIC)) {
    rc = ey_fused16_hmc(pl, theta, target, grad, p0, u, step, step_vec, L, temp, C, seed, iter, chain_offset, flags,
                        accepted, accept_rate, H_cur, H_prop, (hipStream_t)stream, nullptr, &da);
    if (rc == EY_OK && da.state) pl->da_done += 1;
  }
  else if (use_large(pl, 3, flags))
    rc = ey_large_hmc(pl, theta, target, grad, p0, u, step, step_vec, L, temp, C, seed, iter, chain_offset, flags,
                      accepted, accept_rate, H_cur, H_prop, (hipStream_t)stream);
  else
    rc = ey_generic_hmc(pl, theta, target, grad, p0, u, step, step_vec, L, temp, C, seed, iter, chain_offset, flags,
                        accepted, accept_rate, H_cur, H_prop, (hipStream_t)stream);
  return moments_trailing(pl, rc, theta, accepted, C, stream);
}

__global__ void k_count_accepts(const unsigned char* __restrict__ accepted, int* __restrict__ count, int64_t C) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C && accepted[c]) count[c] += 1;
}

int ey_hmc_run(ey_plan* pl, void* theta, void* target, void* grad, double step, const void* step_vec, int L,
               const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, uint32_t flags,
               int n_iters, void* samples, void* targets, void* accepted_rec, void* accept_count, void* accepted,
               void* stream) {
  int rc = check_ready(pl, C, "ey_hmc_run");
  if (rc) return rc < 0 ? rc : EY_OK;
  EyVariantScope vs(pl);
  if (!theta || !target || !grad || !accepted) EY_FAIL(EY_ERR_INVALID, "ey_hmc_run: null argument");
  if (L < 1) EY_FAIL(EY_ERR_INVALID, "ey_hmc_run: num_steps must be >= 1");
  if (n_iters < 1) EY_FAIL(EY_ERR_INVALID, "ey_hmc_run: n_iters must be >= 1");
  if (C == 0) return EY_OK;
  if ((rc = moments_check(pl, C, "ey_hmc_run"))) return rc;
  const bool f16 = use_fused16(pl) && !(flags & EY_FORCE_GENERIC);
  const bool m32 = use_mfma32(pl) && !(flags & EY_FORCE_GENERIC);
  if ((rc = da_check(pl, C, f16 || m32, "ey_hmc_run"))) return rc;
  EY_HIP(hipSetDevice(pl->device));
  hipStream_t s = (hipStream_t)stream;
  EyRun run = {n_iters, samples, targets, accepted_rec, (int*)accept_count};
  const EyDA da = da_window(pl, n_iters);
  if (da.step) step_vec = da.step;
  if (m32) {
    rc = ey_mfma32_hmc(pl, theta, target, grad, nullptr, nullptr, step, step_vec, L, temp, C, seed, iter, chain_offset,
                       flags, accepted, nullptr, nullptr, nullptr, s, &run, &da);
    if (rc == EY_OK && da.state) pl->da_done += da.n;
    return rc;
  }
  if (f16 || !use_large(pl, 3, flags)) {
    // the generic kernels do not fuse the moments: they are replayed from the recorded samples, which must then exist
    // (checked BEFORE the launch: a failure must leave the chains where they were)
    if (pl->mom_s1 && n_iters > 1 && (!samples || !accepted_rec))
      EY_FAIL(EY_ERR_UNSUPPORTED, "ey_hmc_run: attached moments with n_iters > 1 need the samples and accepted records "
                                  "on this kernel family");
    if (f16) {
      rc = ey_fused16_hmc(pl, theta, target, grad, nullptr, nullptr, step, step_vec, L, temp, C, seed, iter, chain_offset,
                          flags, accepted, nullptr, nullptr, nullptr, s, &run, &da);
      if (rc == EY_OK && da.state) pl->da_done += da.n;
    } else
      rc = ey_generic_hmc(pl, theta, target, grad, nullptr, nullptr, step, step_vec, L, temp, C, seed, iter,
                          chain_offset, flags, accepted, nullptr, nullptr, nullptr, s, &run);
    if (rc != EY_OK || !pl->mom_s1) return rc;
    if (n_iters == 1)
      return ey_stats_update(theta, accepted_rec ? accepted_rec : accepted, C, pl->m.P, pl->dtype, pl->mom_s1, pl->mom_s2,
                             pl->mom_acc, stream);
    return ey_stats_update_run(samples, accepted_rec, n_iters, C, pl->m.P, pl->dtype, pl->mom_s1, pl->mom_s2, pl->mom_acc,
                               stream);
  }
  // models beyond LDS: the layerwise path has no in-kernel iteration loop; the launches are queued back to back
  const size_t es = esize(pl);
  for (int it = 0; it < n_iters; ++it) {
    rc = ey_large_hmc(pl, theta, target, grad, nullptr, nullptr, step, step_vec, L, temp, C, seed, iter + it,
                      chain_offset, flags, accepted, nullptr, nullptr, nullptr, s);
    if (rc) return rc;
    if (samples)
      EY_HIP(hipMemcpyAsync((char*)samples + (size_t)it * C * pl->m.P * es, theta, (size_t)C * pl->m.P * es,
                            hipMemcpyDeviceToDevice, s));
    if (targets)
      EY_HIP(hipMemcpyAsync((char*)targets + (size_t)it * C * es, target, (size_t)C * es, hipMemcpyDeviceToDevice, s));
    if (accepted_rec)
      EY_HIP(hipMemcpyAsync((char*)accepted_rec + (size_t)it * C, accepted, (size_t)C, hipMemcpyDeviceToDevice, s));
    if (accept_count)
      hipLaunchKernelGGL(k_count_accepts, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s,
                         (const unsigned char*)accepted, (int*)accept_count, C);
    if ((rc = moments_trailing(pl, EY_OK, theta, accepted, C, stream))) return rc;
  }
  EY_HIP(hipGetLastError());
  return EY_OK;
}

int ey_hmc_leapfrog(ey_plan* pl, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                    int64_t C, void* target, void* grad, void* stream) {
  int rc = check_ready(pl, C, "ey_hmc_leapfrog");
  if (rc) return rc < 0 ? rc : EY_OK;
  EyVariantScope vs(pl);
  if (!theta || !p || !target || !grad) EY_FAIL(EY_ERR_INVALID, "ey_hmc_leapfrog: null argument");
  if (L < 1) EY_FAIL(EY_ERR_INVALID, "ey_hmc_leapfrog: num_steps must be >= 1");
  if (C == 0) return EY_OK;
  EY_HIP(hipSetDevice(pl->device));
  if (use_mfma32(pl)) return ey_mfma32_leapfrog(pl, theta, p, step, step_vec, L, temp, C, target, grad, (hipStream_t)stream);
  if (use_fused16(pl)) return ey_fused16_leapfrog(pl, theta, p, step, step_vec, L, temp, C, target, grad, (hipStream_t)stream);
  if (use_large(pl)) return ey_large_leapfrog(pl, theta, p, step, step_vec, L, temp, C, target, grad, (hipStream_t)stream);
  return ey_generic_leapfrog(pl, theta, p, step, step_vec, L, temp, C, target, grad, (hipStream_t)stream);
}

// the kernel families that do not fuse attached moments replay them from the per-iteration records of a *_run call
static int moments_replay_check(const ey_plan* pl, const EyRun* run, const char* who) {
  if (pl->mom_s1 && run && run->n_iters > 1 && (!run->samples || !run->accepted))
    EY_FAIL(EY_ERR_UNSUPPORTED, std::string(who) + ": attached moments with n_iters > 1 need the samples and accepted "
                                                   "records on this kernel family");
  return EY_OK;
}
static int moments_replay(ey_plan* pl, const EyRun* run, const void* theta, const void* accepted, int64_t C,
                          const char* who, void* stream) {
  if (!pl->mom_s1) return EY_OK;
  if (!run || run->n_iters == 1) return ey_stats_update(theta, accepted, C, pl->m.P, pl->dtype, pl->mom_s1, pl->mom_s2,
                                                        pl->mom_acc, stream);
  int rcc = moments_replay_check(pl, run, who);
  if (rcc) return rcc;
  return ey_stats_update_run(run->samples, run->accepted, run->n_iters, C, pl->m.P, pl->dtype, pl->mom_s1, pl->mom_s2,
                             pl->mom_acc, stream);
}

// copy the state after one iteration of a host-looped run into the per-iteration records
static int record_iteration(ey_plan* pl, const EyRun* run, int it, const void* theta, const void* target,
                            const void* accepted, int64_t C, hipStream_t s) {
  const size_t es = esize(pl);
  if (run->samples)
    EY_HIP(hipMemcpyAsync((char*)run->samples + (size_t)it * C * pl->m.P * es, theta, (size_t)C * pl->m.P * es,
                          hipMemcpyDeviceToDevice, s));
  if (run->targets)
    EY_HIP(hipMemcpyAsync((char*)run->targets + (size_t)it * C * es, target, (size_t)C * es, hipMemcpyDeviceToDevice, s));
  if (run->accepted)
    EY_HIP(hipMemcpyAsync((char*)run->accepted + (size_t)it * C, accepted, (size_t)C, hipMemcpyDeviceToDevice, s));
  if (run->accept_count)
    hipLaunchKernelGGL(k_count_accepts, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s,
                       (const unsigned char*)accepted, run->accept_count, C);
  return EY_OK;
}

static int mala_impl(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                     const void* step_vec, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                     uint64_t chain_offset, uint32_t flags, void* accepted, void* log_rate, void* stream,
                     const EyRun* run, const char* who) {
  int rc = check_ready(pl, C, who);
  if (rc) return rc < 0 ? rc : EY_OK;
  EyVariantScope vs(pl);
  if (!theta || !target || !grad || !accepted) EY_FAIL(EY_ERR_INVALID, std::string(who) + ": null argument");
  if (!(step > 0.0) && !step_vec) EY_FAIL(EY_ERR_INVALID, std::string(who) + ": step must be positive");
  if (run && run->n_iters < 1) EY_FAIL(EY_ERR_INVALID, std::string(who) + ": n_iters must be >= 1");
  if (C == 0) return EY_OK;
  if ((rc = moments_check(pl, C, who))) return rc;
  EY_HIP(hipSetDevice(pl->device));
  hipStream_t s = (hipStream_t)stream;
  if (use_mfma32(pl) && !(flags & EY_FORCE_GENERIC))
    return ey_mfma32_mala(pl, theta, target, grad, z, u, step, step_vec, temp, C, seed, iter, chain_offset, accepted,
                          log_rate, s, run);
  const bool f16 = use_fused16(pl) && !(flags & EY_FORCE_GENERIC);
  if (f16 || !use_large(pl, 4, flags)) {  // k_mala carves four state vectors from LDS
    if ((rc = moments_replay_check(pl, run, who))) return rc;  // before the launch: a failure leaves the chains alone
    rc = f16 ? ey_fused16_mala(pl, theta, target, grad, z, u, step, step_vec, temp, C, seed, iter, chain_offset, accepted,
                               log_rate, s, run)
             : ey_generic_mala(pl, theta, target, grad, z, u, step, step_vec, temp, C, seed, iter, chain_offset, accepted,
                               log_rate, s, run);
    return rc ? rc : moments_replay(pl, run, theta, accepted, C, who, stream);
  }
  const int n = run ? run->n_iters : 1;
  for (int it = 0; it < n; ++it) {  // models beyond LDS: launches queued back to back
    rc = ey_large_mala_mh(pl, theta, target, grad, z, u, step, step_vec, nullptr, temp, C, seed, iter + it,
                          chain_offset, accepted, log_rate, s);
    if (rc) return rc;
    if (run && (rc = record_iteration(pl, run, it, theta, target, accepted, C, s))) return rc;
    if ((rc = moments_trailing(pl, EY_OK, theta, accepted, C, stream))) return rc;
  }
  EY_HIP(hipGetLastError());
  return EY_OK;
}

static int mh_impl(ey_plan* pl, void* theta, void* target, const void* z, const void* u, const void* scale,
                   const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, uint32_t flags,
                   void* accepted, void* log_rate, void* stream, const EyRun* run, const char* who) {
  int rc = check_ready(pl, C, who);
  if (rc) return rc < 0 ? rc : EY_OK;
  EyVariantScope vs(pl);
  if (!theta || !target || !scale || !accepted) EY_FAIL(EY_ERR_INVALID, std::string(who) + ": null argument");
  if (run && run->n_iters < 1) EY_FAIL(EY_ERR_INVALID, std::string(who) + ": n_iters must be >= 1");
  if (C == 0) return EY_OK;
  if ((rc = moments_check(pl, C, who))) return rc;
  EY_HIP(hipSetDevice(pl->device));
  hipStream_t s = (hipStream_t)stream;
  if (use_mfma32(pl) && !(flags & EY_FORCE_GENERIC))
    return ey_mfma32_mh(pl, theta, target, z, u, scale, temp, C, seed, iter, chain_offset, accepted, log_rate, s, run);
  const bool f16 = use_fused16(pl) && !(flags & EY_FORCE_GENERIC);
  if (f16 || !use_large(pl, 2, flags)) {
    if ((rc = moments_replay_check(pl, run, who))) return rc;
    rc = f16 ? ey_fused16_mh(pl, theta, target, z, u, scale, temp, C, seed, iter, chain_offset, accepted, log_rate, s, run)
             : ey_generic_mh(pl, theta, target, z, u, scale, temp, C, seed, iter, chain_offset, accepted, log_rate, s, run);
    return rc ? rc : moments_replay(pl, run, theta, accepted, C, who, stream);
  }
  const int n = run ? run->n_iters : 1;
  for (int it = 0; it < n; ++it) {
    rc = ey_large_mala_mh(pl, theta, target, nullptr, z, u, 0.0, nullptr, scale, temp, C, seed, iter + it, chain_offset,
                          accepted, log_rate, s);
    if (rc) return rc;
    if (run && (rc = record_iteration(pl, run, it, theta, target, accepted, C, s))) return rc;
    if ((rc = moments_trailing(pl, EY_OK, theta, accepted, C, stream))) return rc;
  }
  EY_HIP(hipGetLastError());
  return EY_OK;
}

int ey_mala_step(ey_plan* pl, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                 const void* step_vec, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                 uint64_t chain_offset, uint32_t flags, void* accepted, void* log_rate, void* stream) {
  return mala_impl(pl, theta, target, grad, z, u, step, step_vec, temp, C, seed, iter, chain_offset, flags, accepted,
                   log_rate, stream, nullptr, "ey_mala_step");
}

int ey_mala_run(ey_plan* pl, void* theta, void* target, void* grad, double step, const void* step_vec, const void* temp,
                int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, uint32_t flags, int n_iters,
                void* samples, void* targets, void* accepted_rec, void* accept_count, void* accepted, void* stream) {
  const EyRun run = {n_iters, samples, targets, accepted_rec, (int*)accept_count};
  return mala_impl(pl, theta, target, grad, nullptr, nullptr, step, step_vec, temp, C, seed, iter, chain_offset, flags,
                   accepted, nullptr, stream, &run, "ey_mala_run");
}

int ey_mh_step(ey_plan* pl, void* theta, void* target, const void* z, const void* u, const void* scale,
               const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, uint32_t flags,
               void* accepted, void* log_rate, void* stream) {
  return mh_impl(pl, theta, target, z, u, scale, temp, C, seed, iter, chain_offset, flags, accepted, log_rate, stream,
                 nullptr, "ey_mh_step");
}

int ey_mh_run(ey_plan* pl, void* theta, void* target, const void* scale, const void* temp, int64_t C, uint64_t seed,
              uint64_t iter, uint64_t chain_offset, uint32_t flags, int n_iters, void* samples, void* targets,
              void* accepted_rec, void* accept_count, void* accepted, void* stream) {
  const EyRun run = {n_iters, samples, targets, accepted_rec, (int*)accept_count};
  return mh_impl(pl, theta, target, nullptr, nullptr, scale, temp, C, seed, iter, chain_offset, flags, accepted, nullptr,
                 stream, &run, "ey_mh_run");
}

}  // extern "C"

// ----------------------------------------------------------------------------------------------- small kernels
template <typename T>
__global__ void k_pt_swap(const T* ell_i, const T* ell_j, const T* t_i, const T* t_j, const T* dlogq, const T* u,
                          int64_t C, unsigned char* swap, T* log_rate) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  // power_posterior_sampler.py:135-141 with T_k = t_k * ell:  dlogq + (t_i - t_j) * (ell_j - ell_i)
  T lr = (t_i[c] - t_j[c]) * (ell_j[c] - ell_i[c]);
  if (dlogq) lr += dlogq[c];
  const T lu = sizeof(T) == 4 ? (T)logf((float)u[c]) : (T)log((double)u[c]);
  swap[c] = lu < lr ? 1 : 0;  // :160
  if (log_rate) log_rate[c] = lr;
}

template <typename T>
__global__ void k_philox_normal(T* out, int64_t C, int64_t P, uint64_t seed, uint64_t iter, uint64_t chain_offset) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nb = (P + 3) / 4;  // blocks of four stream elements per chain
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < C * nb; k += stride) {
    const int64_t c = k / nb, b = k - c * nb;
    const EyRng r = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_NORMAL);
    T o[4];
    ey_rng_normal4<T>(r, (uint32_t)b, o);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * b + j < P) out[c * P + 4 * b + j] = o[j];
  }
}

template <typename T>
__global__ void k_philox_uniform(T* out, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const EyRng r = ey_rng_make(seed, chain_offset + (uint64_t)c, iter, EY_STREAM_UNIFORM);
  out[c] = ey_rng_uniform<T>(r);
}

extern "C" {

int ey_pt_swap_decide(const void* ell_i, const void* ell_j, const void* t_i, const void* t_j, const void* dlogq,
                      const void* u, int64_t C, int dtype, void* swap, void* log_rate, void* stream) {
  if (!ell_i || !ell_j || !t_i || !t_j || !u || !swap) EY_FAIL(EY_ERR_INVALID, "ey_pt_swap_decide: null argument");
  if (dtype != EY_F32 && dtype != EY_F64) EY_FAIL(EY_ERR_INVALID, "ey_pt_swap_decide: bad dtype");
  if (C <= 0) return EY_OK;
  const unsigned nb = (unsigned)((C + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EY_F32)
    hipLaunchKernelGGL(k_pt_swap<float>, dim3(nb), dim3(256), 0, s, (const float*)ell_i, (const float*)ell_j,
                       (const float*)t_i, (const float*)t_j, (const float*)dlogq, (const float*)u, C,
                       (unsigned char*)swap, (float*)log_rate);
  else
    hipLaunchKernelGGL(k_pt_swap<double>, dim3(nb), dim3(256), 0, s, (const double*)ell_i, (const double*)ell_j,
                       (const double*)t_i, (const double*)t_j, (const double*)dlogq, (const double*)u, C,
                       (unsigned char*)swap, (double*)log_rate);
  EY_HIP(hipGetLastError());
  return EY_OK;
}

int ey_philox_normal(void* out, int64_t C, int64_t P, uint64_t seed, uint64_t iter, uint64_t chain_offset, int dtype,
                     void* stream) {
  if (!out) EY_FAIL(EY_ERR_INVALID, "ey_philox_normal: null argument");
  if (dtype != EY_F32 && dtype != EY_F64) EY_FAIL(EY_ERR_INVALID, "ey_philox_normal: bad dtype");
  if (C <= 0 || P <= 0) return EY_OK;
  const int64_t nblk = (C * P + 255) / 256;
  const dim3 grid((unsigned)(nblk < 65536 ? nblk : 65536));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EY_F32)
    hipLaunchKernelGGL(k_philox_normal<float>, grid, dim3(256), 0, s, (float*)out, C, P, seed, iter, chain_offset);
  else
    hipLaunchKernelGGL(k_philox_normal<double>, grid, dim3(256), 0, s, (double*)out, C, P, seed, iter, chain_offset);
  EY_HIP(hipGetLastError());
  return EY_OK;
}

int ey_philox_block(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
  if (!counter || !key || !out) EY_FAIL(EY_ERR_INVALID, "ey_philox_block: null argument");
  ey_philox4x32_10(counter[0], counter[1], counter[2], counter[3], key[0], key[1], out);
  return EY_OK;
}

int ey_philox_uniform(void* out, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, int dtype,
                      void* stream) {
  if (!out) EY_FAIL(EY_ERR_INVALID, "ey_philox_uniform: null argument");
  if (dtype != EY_F32 && dtype != EY_F64) EY_FAIL(EY_ERR_INVALID, "ey_philox_uniform: bad dtype");
  if (C <= 0) return EY_OK;
  const unsigned nb = (unsigned)((C + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EY_F32)
    hipLaunchKernelGGL(k_philox_uniform<float>, dim3(nb), dim3(256), 0, s, (float*)out, C, seed, iter, chain_offset);
  else
    hipLaunchKernelGGL(k_philox_uniform<double>, dim3(nb), dim3(256), 0, s, (double*)out, C, seed, iter, chain_offset);
  EY_HIP(hipGetLastError());
  return EY_OK;
}

}  // extern "C"

// ----------------------------------------------------------------------------------------------- chain moments
// Running per-chain sums for ChainLists.mean / multi_rhat-style summaries (eeyore/chains/chain_lists.py:65-66,
// eeyore/stats/multi_rhat.py): s1 += theta, s2 += theta^2 in double, acc += accepted; one streaming pass.
template <typename T>
__global__ void k_stats_update(const T* __restrict__ theta, double* __restrict__ s1, double* __restrict__ s2,
                               int64_t n4, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int64_t b = 4 * i;
    if (b + 3 < n && sizeof(T) == 4) {
      const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(theta) + b);
      double2* p1 = reinterpret_cast<double2*>(s1 + b);
      double2* p2 = reinterpret_cast<double2*>(s2 + b);
      double2 a0 = p1[0], a1 = p1[1], q0 = p2[0], q1 = p2[1];
      a0.x += t.x; a0.y += t.y; a1.x += t.z; a1.y += t.w;
      q0.x += (double)t.x * t.x; q0.y += (double)t.y * t.y; q1.x += (double)t.z * t.z; q1.y += (double)t.w * t.w;
      p1[0] = a0; p1[1] = a1; p2[0] = q0; p2[1] = q1;
    } else {
      for (int64_t k = b; k < n && k < b + 4; ++k) {
        const double t = (double)theta[k];
        s1[k] += t;
        s2[k] += t * t;
      }
    }
  }
}

__global__ void k_acc_update(const unsigned char* __restrict__ accepted, double* __restrict__ acc, int64_t C) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) acc[c] += (double)accepted[c];
}

// The same sums over the n_it recorded iterations of a launch in ONE pass: every element's accumulators are read once,
// take the element's n_it states in iteration order (the additions ey_stats_update would make launch after launch, or the
// step kernel draw after draw: the same bits) and are written once -- the record buffer streams through (4 bytes per
// state) instead of the f64 accumulators being read and written once per iteration (32 bytes per state).
template <typename T>
__global__ void k_stats_update_run(const T* __restrict__ samples, int n_it, double* __restrict__ s1, double* __restrict__ s2,
                                   int64_t n4, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int64_t b = 4 * i;
    if (b + 3 < n && sizeof(T) == 4 && (n & 3) == 0) {
      double2* p1 = reinterpret_cast<double2*>(s1 + b);
      double2* p2 = reinterpret_cast<double2*>(s2 + b);
      double2 a0 = p1[0], a1 = p1[1], q0 = p2[0], q1 = p2[1];
      const float* src = reinterpret_cast<const float*>(samples) + b;
      int it = 0;
      for (; it + 4 <= n_it; it += 4) {  // four iterations' loads in flight
        float4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const float4*>(src + (int64_t)(it + k) * n);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          a0.x += t[k].x; a0.y += t[k].y; a1.x += t[k].z; a1.y += t[k].w;
          q0.x += (double)t[k].x * t[k].x; q0.y += (double)t[k].y * t[k].y;
          q1.x += (double)t[k].z * t[k].z; q1.y += (double)t[k].w * t[k].w;
        }
      }
      for (; it < n_it; ++it) {
        const float4 t = *reinterpret_cast<const float4*>(src + (int64_t)it * n);
        a0.x += t.x; a0.y += t.y; a1.x += t.z; a1.y += t.w;
        q0.x += (double)t.x * t.x; q0.y += (double)t.y * t.y; q1.x += (double)t.z * t.z; q1.y += (double)t.w * t.w;
      }
      p1[0] = a0; p1[1] = a1; p2[0] = q0; p2[1] = q1;
    } else {
      for (int64_t k = b; k < n && k < b + 4; ++k) {
        double a = s1[k], q = s2[k];
        for (int it = 0; it < n_it; ++it) {
          const double t = (double)samples[(int64_t)it * n + k];
          a += t;
          q += t * t;
        }
        s1[k] = a;
        s2[k] = q;
      }
    }
  }
}
__global__ void k_acc_update_run(const unsigned char* __restrict__ accepted, int n_it, double* __restrict__ acc, int64_t C) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = acc[c];
  for (int it = 0; it < n_it; ++it) a += (double)accepted[(int64_t)it * C + c];
  acc[c] = a;
}
int ey_stats_update_run(const void* samples, const void* accepted_rec, int n_it, int64_t C, int64_t P, int dtype, void* s1,
                        void* s2, void* acc, void* stream) {
  if (!samples || !s1 || !s2) EY_FAIL(EY_ERR_INVALID, "ey_stats_update_run: null argument");
  if (C <= 0 || P <= 0 || n_it <= 0) return EY_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = C * P, n4 = (n + 3) / 4;
  const unsigned nb = (unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  if (dtype == EY_F32)
    hipLaunchKernelGGL(k_stats_update_run<float>, dim3(nb), dim3(256), 0, s, (const float*)samples, n_it, (double*)s1,
                       (double*)s2, n4, n);
  else
    hipLaunchKernelGGL(k_stats_update_run<double>, dim3(nb), dim3(256), 0, s, (const double*)samples, n_it, (double*)s1,
                       (double*)s2, n4, n);
  if (accepted_rec && acc)
    hipLaunchKernelGGL(k_acc_update_run, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s,
                       (const unsigned char*)accepted_rec, n_it, (double*)acc, C);
  EY_HIP(hipGetLastError());
  return EY_OK;
}

extern "C" int ey_stats_update(const void* theta, const void* accepted, int64_t C, int64_t P, int dtype, void* s1,
                               void* s2, void* acc, void* stream) {
  if (!theta || !s1 || !s2) EY_FAIL(EY_ERR_INVALID, "ey_stats_update: null argument");
  if (dtype != EY_F32 && dtype != EY_F64) EY_FAIL(EY_ERR_INVALID, "ey_stats_update: bad dtype");
  if (C <= 0 || P <= 0) return EY_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = C * P, n4 = (n + 3) / 4;
  const unsigned nb = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  if (dtype == EY_F32)
    hipLaunchKernelGGL(k_stats_update<float>, dim3(nb), dim3(256), 0, s, (const float*)theta, (double*)s1, (double*)s2,
                       n4, n);
  else
    hipLaunchKernelGGL(k_stats_update<double>, dim3(nb), dim3(256), 0, s, (const double*)theta, (double*)s1,
                       (double*)s2, n4, n);
  if (accepted && acc)
    hipLaunchKernelGGL(k_acc_update, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s,
                       (const unsigned char*)accepted, (double*)acc, C);
  EY_HIP(hipGetLastError());
  return EY_OK;
}

extern "C" int ey_plan_attach_da(ey_plan* pl, void* state, void* step_vec, const void* table, int64_t n, int64_t C,
                                 double d, double log_eub, int final_avg) {
  if (!pl) EY_FAIL(EY_ERR_INVALID, "ey_plan_attach_da: null plan");
  if (!state) {
    pl->da_state = nullptr; pl->da_step = nullptr; pl->da_table = nullptr;
    pl->da_n = pl->da_done = pl->da_C = 0;
    return EY_OK;
  }
  if (!step_vec || !table || n <= 0 || C <= 0)
    EY_FAIL(EY_ERR_INVALID, "ey_plan_attach_da: state, step_vec, table, n > 0 and C > 0 are required");
  if (!(d > 0.0 && d < 1.0)) EY_FAIL(EY_ERR_INVALID, "ey_plan_attach_da: the target acceptance must lie in (0, 1)");
  pl->da_state = (double*)state;
  pl->da_step = step_vec;
  pl->da_table = (const double*)table;
  pl->da_n = n;
  pl->da_done = 0;
  pl->da_C = C;
  pl->da_d = d;
  pl->da_has_eub = log_eub == log_eub;  // NaN = no upper bound
  pl->da_logeub = pl->da_has_eub ? log_eub : 0.0;
  pl->da_final_avg = final_avg != 0;
  return EY_OK;
}

extern "C" int ey_plan_attach_moments(ey_plan* pl, void* s1, void* s2, void* acc, int64_t C) {
  if (!pl) EY_FAIL(EY_ERR_INVALID, "ey_plan_attach_moments: null plan");
  if (!s1) {
    pl->mom_s1 = pl->mom_s2 = pl->mom_acc = nullptr;
    pl->mom_C = 0;
    return EY_OK;
  }
  if (!s2 || !acc || C <= 0) EY_FAIL(EY_ERR_INVALID, "ey_plan_attach_moments: s1, s2, acc and C > 0 are required");
  pl->mom_s1 = (double*)s1;
  pl->mom_s2 = (double*)s2;
  pl->mom_acc = (double*)acc;
  pl->mom_C = C;
  return EY_OK;
}
