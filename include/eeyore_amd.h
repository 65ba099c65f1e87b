/* eeyore_amd C ABI -- the drop-in boundary for the chain-batched MCMC hot path on MI355X (gfx950).
 *
 * The reference (papamarkou/eeyore v0.0.20) is pure Python and has no FFI: its boundary is the duck-typed
 * protocol between a sampler and a model.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - every `const void*` / `void*` data argument is a DEVICE pointer (torch.Tensor.data_ptr()) of the plan's
 *    dtype unless stated otherwise; the caller owns every buffer, the library never returns memory;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); compute entry points AND ey_plan_set_data
 *    are asynchronous on it (set_data: device-to-device copies and packing kernels on `stream`; it allocates -- and
 *    then synchronises the device -- only when a batch is larger than any before it); ey_plan_create /
 *    ey_plan_set_prior / ey_plan_destroy synchronise;
 *  - return value 0 = EY_OK, negative = ey_status; ey_last_error() gives a thread-local message;
 *  - parameters are flat `theta[C, P]`, chain-major, each row in nn.Module.parameters() order: per layer
 *    W_l [d_{l+1} x d_l] row-major then b_l (eeyore/models/model.py:38-55, eeyore/models/mlp.py:37-43);
 *  - C = 1 is the reference's single-chain case.
 *  - a plan is not thread-safe; distinct plans on distinct streams/devices are independent.
 */
#ifndef EEYORE_AMD_H
#define EEYORE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ey_plan ey_plan;

enum ey_status {
  EY_OK = 0,
  EY_ERR_INVALID = -1,      /* bad argument (ValueError in the reference, eeyore/models/mlp.py:15-19) */
  EY_ERR_UNSUPPORTED = -2,  /* model/shape outside what the kernels cover */
  EY_ERR_HIP = -3,          /* HIP runtime error (RuntimeError in the reference's benchmark(), serial_sampler.py:112) */
  EY_ERR_STATE = -4         /* data or prior not set */
};

/* activation codes: eeyore/models/mlp.py:9-13,45-50 (`None` or torch.sigmoid in every reference test/example) */
enum ey_act { EY_ACT_NONE = 0, EY_ACT_SIGMOID = 1, EY_ACT_TANH = 2, EY_ACT_RELU = 3 };

/* likelihood codes: eeyore/constants/constants.py:15-18 */
enum ey_lik {
  EY_LIK_BCE_SUM = 0, /* 'binary_classification': naive BCE(sum) on probabilities, eeyore/stats/loss.py:1-11 */
  EY_LIK_CE_SUM = 1   /* 'multiclass_classification': CrossEntropyLoss(sum)(logits, argmax(y,1)) */
};

enum ey_dtype { EY_F32 = 0, EY_F64 = 1 }; /* model.dtype, eeyore/models/model.py:7-10 */

/* Plan options (ey_plan_set_option).
 * EY_OPT_F32_PRODUCTS: how the fused f32 trajectory kernel ("mfma32": MLP(4-32-32-3), BASELINE configs 3/4) forms its three
 * 32x32x32 products per row tile, and the layerwise path ("bgemm") its 128-wide batched products.  The reference computes them with torch's f32 matmul (eeyore/models/mlp.py:45-50 ->
 * nn.Linear); both forms below are f32 in, f32 out, f32 accumulate, and both pass every f32 parity test at the same
 * tolerances:
 *   EY_PRODUCTS_BF16X3 (default): each f32 operand is split EXACTLY into three bf16 pieces (hi + mid + lo = x) and the
 *     product is summed from the six piece products of relative size >= 2^-18 on v_mfma_f32_32x32x16_bf16 (bf16 x bf16 is
 *     exact in f32, accumulation in f32, smallest terms first); measured error against f64 (profiles/r03_bf3_error_probe.txt,
 *     tests/test_bf16x3.py): rms within 0.85 .. 1.06 x the exact form's, maximum within 0.74 .. 1.48 x (1.30 x on the
 *     golden fixtures; the bar the tests hold it to is 1.5 x); the fused kernel takes it for batches of up to 512
 *     rows, and in this form it also serves the other MLP(4-32-32-dK) models with one hidden activation (sigmoid / tanh /
 *     relu; CE-sum on 3 logits or BCE-sum on one sigmoid output), which otherwise run on "fused16";
 *   EY_PRODUCTS_EXACT: v_mfma_f32_32x32x2_f32 (16x16x4 on "fused16"), bit for bit a k-ordered f32 fma chain.
 * The environment variable EY_F32_PRODUCTS=exact|bf16x3 sets what new plans start with. */
enum ey_option { EY_OPT_F32_PRODUCTS = 1, EY_OPT_ROW_WAVES = 2 };
enum ey_products { EY_PRODUCTS_BF16X3 = 0, EY_PRODUCTS_EXACT = 1 };
/* EY_OPT_ROW_WAVES: tiny models (at most three layers, eight inputs, other widths <= 4: the reference's own test and example
 * models) on batches of 128 rows or more may give a chain up to four waves, each taking every fourth 64-row tile of an
 * evaluation; the waves' partial gradients are added in a fixed order, which is not the order one wave adds them in, so the
 * two differ in the last bits.  EY_ROW_WAVES_OFF is the default (environment EY_ROW_WAVES=0|1|2 sets what new plans start
 * with): a chain's bits then depend on (seed, chain_offset + chain, iteration) alone -- not on how many chains share the
 * launch, how they are sharded over ranks, or the device's CU count.  _ON pins the waves (same guarantee, the other
 * summation order).  EY_ROW_WAVES_AUTO is the opt-in latency setting: the waves while one wave per chain would leave the
 * chip idle (chains <= 4 x CUs), so a chain's last bits follow the launch's chain count: BASELINE config 2 (MALA, 256
 * chains, N = 256) 7.7 -> 5.4 us per draw (tools/bench_configs.py opts in). */
enum ey_row_waves { EY_ROW_WAVES_OFF = 0, EY_ROW_WAVES_ON = 1, EY_ROW_WAVES_AUTO = 2 };

enum ey_flags {
  EY_RECOMPUTE_INITIAL_GRAD = 1, /* HMC: re-evaluate the gradient at the start of the trajectory exactly as
                                    hmc.py:104 does (L+1 evaluations) instead of using the cached `grad` (L) */
  EY_FORCE_GENERIC = 2           /* route to the generic VALU kernels even when an MFMA kernel covers the plan */
};

int ey_version(void);
const char* ey_last_error(void);

/* Replaces mlp.Hyperparameters + MLP.__init__/set_fc_layers (eeyore/models/mlp.py:9-43) and the choice of
 * loss_functions[...] (eeyore/constants/constants.py:15-18).  dims has n_layers+1 entries; bias/act n_layers. */
int ey_plan_create(ey_plan** out, int n_layers, const int* dims, const int* bias, const int* act, int likelihood,
                   int dtype, int device_id);
int ey_plan_destroy(ey_plan* plan);
/* Model.num_params (eeyore/models/model.py:34-36) */
int ey_plan_num_params(const ey_plan* plan, int64_t* P);
/* name of the kernel family that serves ey_hmc_step for this plan: "mfma32" (fused f32 trajectory, 4-32-32-dK), "fused16"
 * (fused 16x16x4 trajectory, f32 and f64: one or two hidden layers of at most 64 units (32 in f64), at most 16 inputs,
 * CE-sum on at most 16 logits or BCE-sum on at most 4 sigmoid outputs, every layer with or without a bias), "bgemm"
 * (layerwise batched GEMMs for models beyond LDS and for wide ones that fit, f32 and f64) or "generic" (anything mlp.py
 * builds) */
const char* ey_plan_kernel(const ey_plan* plan);
int ey_plan_set_option(ey_plan* plan, int option, int value);
int ey_plan_get_option(const ey_plan* plan, int option, int* value);

/* The (x, y) full batch the samplers receive from their DataLoader (eeyore/samplers/serial_sampler.py:41-46).
 * x [N, d_0]; y [N, d_K] (one-hot for CE as XYDataset(yonehot=True) yields, {0,1} for BCE).  Copied into the plan. */
int ey_plan_set_data(ey_plan* plan, const void* x, const void* y, int64_t N, void* stream);
/* model.prior = Normal(mu, sigma) elementwise (eeyore/models/mlp.py:31-35).  mu, sigma [P].  Copied. */
int ey_plan_set_prior(ey_plan* plan, const void* mu, const void* sigma, void* stream);

/* BayesianModel.log_lik / log_prior / log_target (eeyore/models/bayesian_model.py:30-56) for C chains.
 * temp: per-chain temperature [C] or NULL (model.temperature = None); multiplies BOTH outputs (:33-34,48-49).
 * log_lik, log_prior: [C] outputs (either may be NULL). */
int ey_log_target(ey_plan* plan, const void* theta, const void* temp, int64_t C, void* log_lik, void* log_prior,
                  void* stream);
/* The N terms of BayesianModel.log_lik's sum, one per data row (the loss of eeyore/constants/constants.py:15-18 is a
 * sum over rows): rows [C, N], rows[c, n] = log-likelihood of row n under theta[c] (times temp[c] if given).  This is
 * the integrand of BayesianModel.predictive_posterior for N points and C posterior samples at once
 * (eeyore/models/bayesian_model.py:58-67: exp of the log-likelihood of ONE point, averaged over samples by MCIntegrator,
 * eeyore/integrators/mcintegrator.py:16-36). */
int ey_log_lik_rows(ey_plan* plan, const void* theta, const void* temp, int64_t C, void* rows, void* stream);
/* LogTargetModel.upto_grad_log_target (eeyore/models/log_target_model.py:15-23): target [C], grad [C,P]. */
int ey_log_target_grad(ey_plan* plan, const void* theta, const void* temp, int64_t C, void* target, void* grad,
                       void* stream);

/* One HMC.draw (eeyore/samplers/hmc.py:126-156, full-batch path) for C chains: momentum draw, HMC.leapfrog
 * (:100-124), Hamiltonians (:91-98), accept `u < min(exp(H_cur-H_prop),1)` (:143-148), state update.
 *  theta [C,P], target [C], grad [C,P]: current state, updated in place for accepted chains;
 *  p0 [C,P] replaces torch.randn (:134) and u [C] replaces torch.rand(1) (:148); NULL => the in-kernel
 *    Philox4x32-10 stream keyed by (seed, chain_offset + chain, iter) -- see ey_philox_normal/uniform;
 *  step: scalar step size; step_vec [C] overrides it per chain when not NULL; L = num_steps;
 *  accepted [C] uint8, accept_rate [C], H_cur [C], H_prop [C]: outputs (the last three may be NULL). */
int ey_hmc_step(ey_plan* plan, void* theta, void* target, void* grad, const void* p0, const void* u, double step,
                const void* step_vec, int L, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                uint64_t chain_offset, uint32_t flags, void* accepted, void* accept_rate, void* H_cur, void* H_prop,
                void* stream);

/* n_iters consecutive iterations (iter, iter + 1, ...) of HMC.draw for every chain inside ONE launch: exactly what
 * n_iters calls of ey_hmc_step with p0 = u = NULL do (same Philox streams, bit-identical states), without the launches
 * in between and with every chain looping on its own.  This is SerialSampler.run's inner loop
 * (eeyore/samplers/serial_sampler.py:41-52) for the iterations in which nothing on the host looks at the state
 * (no tuner step, no minibatch change).  Records, each nullable: samples [n_iters, C, P], targets [n_iters, C] and
 * accepted_rec [n_iters, C] uint8 = the state of every chain after each iteration, i.e. what ChainList.update stores
 * (eeyore/chains/chain_list.py:64-67); accept_count [C] int32 is incremented per accepted iteration.
 * accepted [C] uint8 receives the last iteration's flags.  Attached moments are accumulated every iteration. */
int ey_hmc_run(ey_plan* plan, void* theta, void* target, void* grad, double step, const void* step_vec, int L,
               const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, uint32_t flags,
               int n_iters, void* samples, void* targets, void* accepted_rec, void* accept_count, void* accepted,
               void* stream);

/* HMC.leapfrog(position0, momentum0, x, y) (eeyore/samplers/hmc.py:100-124) for C chains, exactly as the
 * reference runs it: L steps, L+1 gradient evaluations, final momentum negated.  theta [C,P] and p [C,P] are
 * in/out (position_L, momentum_L); target [C] and grad [C,P] receive the log-target and its gradient at
 * position_L.  Used by HMC.init_step (:38-77) and by callers of the public leapfrog method. */
int ey_hmc_leapfrog(ey_plan* plan, void* theta, void* p, double step, const void* step_vec, int L, const void* temp,
                    int64_t C, void* target, void* grad, void* stream);

/* One MALA.draw (eeyore/samplers/mala.py:46-82) with the default NormalKernel(theta + step/2 grad, sqrt(step))
 * (:35-41; eeyore/kernels/normal_kernel.py:5-23): z [C,P] standard normals (NULL => Philox), u [C].
 * Accept iff log(u) < log_rate (:66).  log_rate [C] output may be NULL. */
int ey_mala_step(ey_plan* plan, void* theta, void* target, void* grad, const void* z, const void* u, double step,
                 const void* step_vec, const void* temp, int64_t C, uint64_t seed, uint64_t iter,
                 uint64_t chain_offset, uint32_t flags, void* accepted, void* log_rate, void* stream);

/* One MetropolisHastings.draw (eeyore/samplers/metropolis_hastings.py:41-73), symmetric NormalKernel(theta,
 * scale): prop = theta + scale * z; scale [P] device array.  log_rate = target(prop) - target(theta) (:50). */
int ey_mh_step(ey_plan* plan, void* theta, void* target, const void* z, const void* u, const void* scale,
               const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, uint32_t flags,
               void* accepted, void* log_rate, void* stream);

/* n_iters consecutive MALA.draw / MetropolisHastings.draw iterations inside one launch, with the same records as
 * ey_hmc_run: exactly what n_iters calls of ey_mala_step / ey_mh_step with z = u = NULL do. */
int ey_mala_run(ey_plan* plan, void* theta, void* target, void* grad, double step, const void* step_vec,
                const void* temp, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, uint32_t flags,
                int n_iters, void* samples, void* targets, void* accepted_rec, void* accept_count, void* accepted,
                void* stream);
int ey_mh_run(ey_plan* plan, void* theta, void* target, const void* scale, const void* temp, int64_t C, uint64_t seed,
              uint64_t iter, uint64_t chain_offset, uint32_t flags, int n_iters, void* samples, void* targets,
              void* accepted_rec, void* accept_count, void* accepted, void* stream);

/* PowerPosteriorSampler.between_chain_move (eeyore/samplers/power_posterior_sampler.py:135-163) decision for C
 * chain pairs: log_rate = dlogq + (t_i - t_j) * (ell_j - ell_i) with ell the UNTEMPERED log-target; swap iff
 * log(u) < log_rate (:160).  All arrays [C] of `dtype`; dlogq may be NULL (symmetric partner choice). */
int ey_pt_swap_decide(const void* ell_i, const void* ell_j, const void* t_i, const void* t_j, const void* dlogq,
                      const void* u, int64_t C, int dtype, void* swap /* uint8 [C] */, void* log_rate, void* stream);

/* The in-kernel random streams, exposed so a caller (or a test) can reproduce them:
 *  normal  out[c, i] = N(0,1) for parameter i of chain chain_offset + c at iteration iter (what p0/z = NULL uses)
 *  uniform out[c]    = U[0,1) accept variate of that chain and iteration (what u = NULL uses). */
int ey_philox_normal(void* out, int64_t C, int64_t P, uint64_t seed, uint64_t iter, uint64_t chain_offset, int dtype,
                     void* stream);
int ey_philox_uniform(void* out, int64_t C, uint64_t seed, uint64_t iter, uint64_t chain_offset, int dtype,
                      void* stream);
/* One Philox4x32-10 block on the HOST (no device needed): out[4] = philox(counter[4], key[2]), the function the device
 * streams above are built on (Salmon et al. 2011; checked against Random123's known-answer vectors in the tests).
 * counter = (block, chain_lo, iter_lo, iter_hi << 8 | stream | chain_hi << 20), key = (seed_lo, seed_hi), stream 0 =
 * normals, 1 = accept uniform. */
int ey_philox_block(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);

/* Running per-chain moments for ChainLists.mean / R-hat style summaries (eeyore/chains/chain_lists.py:65-66,
 * eeyore/stats/multi_rhat.py:10-40): s1 += theta, s2 += theta^2 ([C,P] double accumulators), acc += accepted
 * ([C] double; `accepted` uint8 [C]; both may be NULL).  theta [C,P] of `dtype`.  One streaming pass. */
int ey_stats_update(const void* theta, const void* accepted, int64_t C, int64_t P, int dtype, void* s1, void* s2,
                    void* acc, void* stream);

/* Initial-sequence estimate of the asymptotic variance (eeyore/stats/inse_mc_cov.py:9-83) of every column of
 * x [n, S] at once, each column on its own (p = 1): S = chains * parameters of a stored run, n iterations, row-major as a
 * chain buffer [iterations, C, P] is laid out.  sig2 [S] double: the estimate (NaN where the reference raises 'Not
 * enough samples', :45-46); var [S] double: the unbiased sample variance (eeyore/stats/cov.py:5-15), so that multi_ess'
 * n * (det cov / det mc_cov)^(1/p) (eeyore/stats/multi_ess.py:6-14) is n * var / sig2; num_pairs [S] int32 or NULL: lag
 * pairs that entered the sum (-1 where not enough).  The reference's adjust=True changes nothing for p = 1. */
int ey_inse_univariate(const void* x, int64_t n, int64_t S, int dtype, void* sig2, void* var, void* num_pairs,
                       void* stream);

/* The reference's MULTIVARIATE initial-sequence estimator (eeyore/stats/inse_mc_cov.py:9-83, adjust=False) for C
 * chains of p <= 64 parameters at once (up to 16, with n p doubles inside 144 KiB, the chain lies in LDS; beyond that the
 * centred chains go through a workspace the call allocates and frees on the stream): x is addressed as x[i * stride_n + c * stride_c + j] (elements; a chain buffer
 * [iterations, C, P] has stride_n = C*P, stride_c = P; [C, n, p] has stride_n = p, stride_c = n*p).  sig [C,p,p]
 * double: the estimate (NaN where the reference raises 'Not enough samples'); cov [C,p,p] double or NULL: the unbiased
 * sample covariance (eeyore/stats/cov.py:5-15); mean [C,p] double or NULL; num_pairs [C] int32 or NULL.  With these,
 * multi_ess (eeyore/stats/multi_ess.py:6-14) and both parts of multi_rhat (eeyore/stats/multi_rhat.py:10-40: W = the
 * mean of sig over chains, B = the covariance of the chain means) follow from [C,p,p]- and [C,p]-sized arrays. */
int ey_inse_multivariate(const void* x, int64_t n, int64_t C, int64_t p, int64_t stride_n, int64_t stride_c, int dtype,
                         void* sig, void* cov, void* mean, void* num_pairs, void* stream);

/* Attach running-moment accumulators to a plan: from now on every ey_hmc_step / ey_mala_step / ey_mh_step on it also
 * performs, for the state each chain is left in, exactly what ey_stats_update does (s1 += theta, s2 += theta^2,
 * acc += accepted) -- inside the fused kernel where there is one (no extra pass over [C,P]), as a trailing pass on
 * the same stream otherwise.  s1, s2 [C,P] double, acc [C] double, all three required; steps must then be called with
 * that same C.  s1 = NULL detaches.  (The reference keeps every sample and reduces afterwards,
 * eeyore/chains/chain_list.py:64-67, chain_lists.py:65-66; with thousands of chains the moments are kept instead.) */
int ey_plan_attach_moments(ey_plan* plan, void* s1, void* s2, void* acc, int64_t C);

/* Per-chain dual averaging of the HMC step size INSIDE the step kernels (Hoffman & Gelman 2014, algorithm 5: the
 * recurrence of eeyore/tuners/hmcda_tuner.py:43-59, which HMC.draw runs on the host after every burn-in iteration,
 * eeyore/samplers/hmc.py:158-163), so that burn-in too can run as blocks of iterations per launch (ey_hmc_run).
 * state [C,3] double, in/out: (barh, logbare, mu = log(10 e0)) per chain.  step_vec [C] of the plan's dtype: the step
 * every chain takes in its next iteration -- the kernels read it and, after each adapting iteration, write the next
 * one (it replaces the step / step_vec arguments of ey_hmc_step / ey_hmc_run while attached).  table [n,3] double on
 * the device: (1/(t + t0), sqrt(t)/gamma, t^-kappa) for the t = 1st..n-th adapting iteration, worked out by the caller
 * so that host and device agree to the last bit; the plan counts the iterations it has adapted and stops after n.
 * d: target acceptance; log_eub: log of the upper bound on the step or NaN; final_avg != 0: the n-th iteration leaves
 * the AVERAGED step exp(logbare) (hmcda_tuner.py's return_e=False).  The number of leapfrog steps stays what the calls
 * pass.  For as long as a state is attached, step_vec is the step of every HMC launch on the plan -- also after the n
 * adapting iterations are used up (the launches then read it and adapt nothing) -- and the step / step_vec arguments of
 * the calls are ignored; the plan's position in the table advances only past iterations whose launch succeeded.
 * state = NULL detaches.  Served by the fused kernel families (mfma32, fused16); EY_ERR_UNSUPPORTED otherwise. */
int ey_plan_attach_da(ey_plan* plan, void* state, void* step_vec, const void* table, int64_t n, int64_t C, double d,
                      double log_eub, int final_avg);

/* Diagnostic switches for A/B runs and tests (not part of the drop-in surface).  They are state of the PLAN:
 * ey_plan_set_variant changes one plan and returns its previous value; ey_debug_set_variant sets what plans created
 * afterwards start with (also EY_VARIANT in the environment) and returns the previous default.  Bits 0..2: workgroup
 * shape / issue-priority / parking variants of the fused f32 trajectory kernel; 4: f32 plans through the layerwise path;
 * 5, 6, 7: that path without LDS-DMA staging / fused last layer / fused leapfrog update; 8, 9: tiny models never / always
 * through the register-resident evaluation of the generic kernels; 10 (ey_debug_set_variant only): new plans, and
 * ey_debug_bgemm, start with EY_PRODUCTS_EXACT; 11: the layerwise path's bf16x3 products split the data matrix in every
 * workgroup instead of taking it pre-split; 12: the layerwise path's epilogues that read per element (prior gradient, fused
 * leapfrog update, act'(H)) element by element instead of in batches of loads; 3: the fused f32 trajectory kernel's HMC draw with
 * the pipelined tile loop at one wave per SIMD (same bits, slower: DESIGN.md 4.1.3); 13: value + gradient of mid-size models
 * (hidden widths 33 .. 128, at most two hidden layers, d_K <= 16, f32) by the fused workgroup-per-chain kernel instead of one
 * product launch per layer and direction; 14: narrow deeper models (every hidden width <= 32, up to three
 * hidden layers, up to 64 inputs, f32) through those product launches instead of the fused kernel k_mid32 that serves them
 * by default.  Results agree to rounding across them (bit for bit across bits 3, 11 and 12). */
int ey_plan_set_variant(ey_plan* plan, int variant);
int ey_debug_set_variant(int variant);

/* Test / measurement entry of the layerwise path's batched f32 product (not part of the drop-in surface):
 * C[b] = act(A[b] B[b] + bias[b]) for b < batch through the same dispatcher the evaluations use.  Element strides
 * (sAm, sAk), (sBk, sBn), (sCm, sCn), batch strides bA, bB, bC (0 = shared operand); bias may be NULL; act is an
 * EY_ACT_* code.  Parity against torch.bmm: tests/test_gpu_parity.py::test_batched_gemm_vs_torch_bmm. */
int ey_debug_bgemm(const float* A, const float* B, float* C, int M, int N, int K, long sAm, long sAk, long sBk, long sBn,
                   long sCm, long sCn, long bA, long bB, long bC, const float* bias, long bBias, int act, int batch,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EEYORE_AMD_H */
