/* C restatement of eeyore's MLP log-target / gradient / HMC / MALA / MH step -- TEST INFRASTRUCTURE ONLY.
 * Included twice by mlp_oracle.c with REAL = double and REAL = float.  Each function cites the
 * reference file:line it follows (paths relative to the papamarkou/eeyore checkout).
 */
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(CAT(oc_, SUFFIX), CAT(_, name))

/* activation on pre-activation g; derivative from the OUTPUT h (what autograd's sigmoid/tanh backward uses) */
static inline REAL FN(act)(int code, REAL g) {
  switch (code) {
    case 1: return (REAL)1 / ((REAL)1 + EXP(-g));
    case 2: return TANH(g);
    case 3: return g > 0 ? g : 0;
    default: return g;
  }
}
static inline REAL FN(dact)(int code, REAL h) {
  switch (code) {
    case 1: return h * ((REAL)1 - h);
    case 2: return (REAL)1 - h * h;
    case 3: return h > 0 ? (REAL)1 : (REAL)0;
    default: return (REAL)1;
  }
}

/* BayesianModel.log_target (eeyore/models/bayesian_model.py:30-56) + hand-coded backward replacing
 * LogTargetModel.grad_log_target (eeyore/models/log_target_model.py:15-18).
 * work: caller scratch of oc_work_size(spec) REALs.  grad may be NULL (value only).
 * temp: NAN => temperature None.  Returns log_target; *lik_out / *prior_out receive the tempered parts. */
REAL FN(log_target_grad)(const oc_spec* s, const REAL* theta, const REAL* x, const REAL* y, int N,
                         const REAL* mu, const REAL* sigma, double temp, REAL* grad, REAL* lik_out,
                         REAL* prior_out, REAL* work) {
  const int nl = s->nl;
  int woff[OC_MAX_LAYERS], boff[OC_MAX_LAYERS], hoff[OC_MAX_LAYERS + 1];
  int P = 0, hsz = 0, dmax = 0;
  for (int l = 0; l < nl; ++l) {
    woff[l] = P; P += s->dims[l + 1] * s->dims[l];
    boff[l] = s->bias[l] ? P : -1; if (s->bias[l]) P += s->dims[l + 1];
  }
  for (int l = 0; l <= nl; ++l) { hoff[l] = hsz; hsz += s->dims[l]; if (s->dims[l] > dmax) dmax = s->dims[l]; }
  REAL* h = work;            /* activations of one row, all layers */
  REAL* d0 = work + hsz;     /* delta ping */
  REAL* d1 = d0 + dmax;      /* delta pong */
  const int dK = s->dims[nl];
  REAL lik = 0;
  if (grad) for (int i = 0; i < P; ++i) grad[i] = 0;
  for (int n = 0; n < N; ++n) {
    /* MLP.forward, eeyore/models/mlp.py:45-50 */
    for (int i = 0; i < s->dims[0]; ++i) h[i] = x[(size_t)n * s->dims[0] + i];
    for (int l = 0; l < nl; ++l) {
      const int din = s->dims[l], dout = s->dims[l + 1];
      const REAL* W = theta + woff[l];
      for (int j = 0; j < dout; ++j) {
        REAL g = 0;
        for (int i = 0; i < din; ++i) g += h[hoff[l] + i] * W[j * din + i];
        if (boff[l] >= 0) g += theta[boff[l] + j];
        h[hoff[l + 1] + j] = FN(act)(s->acts[l], g);
      }
    }
    const REAL* out = h + hoff[nl];
    const REAL* yn = y + (size_t)n * dK;
    REAL* delta = d0;
    if (s->lik == 0) {
      /* eeyore/stats/loss.py:2 -- naive logs on probabilities */
      for (int j = 0; j < dK; ++j) {
        lik += LOG(out[j]) * yn[j] + LOG((REAL)1 - out[j]) * ((REAL)1 - yn[j]);
        delta[j] = (yn[j] / out[j] - ((REAL)1 - yn[j]) / ((REAL)1 - out[j])) * FN(dact)(s->acts[nl - 1], out[j]);
      }
    } else {
      /* eeyore/constants/constants.py:17 -- CrossEntropyLoss(sum) with argmax(y,1) labels */
      int lab = 0; REAL m = out[0];
      for (int j = 1; j < dK; ++j) { if (yn[j] > yn[lab]) lab = j; if (out[j] > m) m = out[j]; }
      REAL ssum = 0;
      for (int j = 0; j < dK; ++j) ssum += EXP(out[j] - m);
      lik += out[lab] - (m + LOG(ssum));
      for (int j = 0; j < dK; ++j)
        delta[j] = ((j == lab ? (REAL)1 : (REAL)0) - EXP(out[j] - m) / ssum) * FN(dact)(s->acts[nl - 1], out[j]);
    }
    if (!grad) continue;
    for (int l = nl - 1; l >= 0; --l) {
      const int din = s->dims[l], dout = s->dims[l + 1];
      const REAL* W = theta + woff[l];
      const REAL* hin = h + hoff[l];
      for (int j = 0; j < dout; ++j) {
        for (int i = 0; i < din; ++i) grad[woff[l] + j * din + i] += delta[j] * hin[i];
        if (boff[l] >= 0) grad[boff[l] + j] += delta[j];
      }
      if (l > 0) {
        REAL* dn = (delta == d0) ? d1 : d0;
        for (int i = 0; i < din; ++i) {
          REAL a = 0;
          for (int j = 0; j < dout; ++j) a += delta[j] * W[j * din + i];
          dn[i] = a * FN(dact)(s->acts[l - 1], hin[i]);
        }
        delta = dn;
      }
    }
  }
  /* BayesianModel.log_prior, eeyore/models/bayesian_model.py:46-50, elementwise Normal */
  REAL prior = 0;
  for (int i = 0; i < P; ++i) {
    const REAL dlt = theta[i] - mu[i], var = sigma[i] * sigma[i];
    prior += -(dlt * dlt) / ((REAL)2 * var) - LOG(sigma[i]) - (REAL)0.9189385332046727;
    if (grad) grad[i] += -dlt / var;
  }
  if (temp == temp) { /* not NaN: temperature multiplies BOTH parts (bayesian_model.py:33-34,48-49) */
    const REAL t = (REAL)temp;
    lik *= t; prior *= t;
    if (grad) for (int i = 0; i < P; ++i) grad[i] *= t;
  }
  if (lik_out) *lik_out = lik;
  if (prior_out) *prior_out = prior;
  return lik + prior;
}

/* HMC.leapfrog (eeyore/samplers/hmc.py:100-124): L steps, L+1 gradient evaluations, momentum negated.
 * theta, p: in/out [P]; target/grad out. */
void FN(leapfrog)(const oc_spec* s, REAL* theta, REAL* p, const REAL* x, const REAL* y, int N, const REAL* mu,
                  const REAL* sigma, double temp, double step, int L, REAL* target, REAL* grad, REAL* work) {
  const int P = oc_num_params(s);
  const REAL eps = (REAL)step, half = (REAL)0.5;
  REAL t = FN(log_target_grad)(s, theta, x, y, N, mu, sigma, temp, grad, 0, 0, work);
  for (int i = 0; i < P; ++i) p[i] = p[i] - half * eps * (-grad[i]);
  for (int k = 0; k < L - 1; ++k) {
    for (int i = 0; i < P; ++i) theta[i] = theta[i] + eps * p[i];
    t = FN(log_target_grad)(s, theta, x, y, N, mu, sigma, temp, grad, 0, 0, work);
    for (int i = 0; i < P; ++i) p[i] = p[i] - eps * (-grad[i]);
  }
  for (int i = 0; i < P; ++i) theta[i] = theta[i] + eps * p[i];
  t = FN(log_target_grad)(s, theta, x, y, N, mu, sigma, temp, grad, 0, 0, work);
  for (int i = 0; i < P; ++i) { p[i] = p[i] - half * eps * (-grad[i]); p[i] = -p[i]; }
  *target = t;
}

/* HMC.draw (eeyore/samplers/hmc.py:126-156), full-batch path, for C independent chains (OpenMP over chains).
 * theta/target/grad are the chains' current state (in/out); p0 [C,P] replaces torch.randn (:134), u [C]
 * replaces torch.rand(1) (:148).  Returns the number of accepted chains. */
int FN(hmc_draw_chains)(const oc_spec* s, int C, REAL* theta, REAL* target, REAL* grad, const REAL* p0,
                        const REAL* u, const REAL* x, const REAL* y, int N, const REAL* mu, const REAL* sigma,
                        double temp, double step, int L, unsigned char* accepted, REAL* h_cur_out,
                        REAL* h_prop_out, int nthreads) {
  const int P = oc_num_params(s);
  const int wsz = oc_work_size(s);
  int nacc = 0;
#pragma omp parallel for num_threads(nthreads) reduction(+ : nacc) schedule(static)
  for (int c = 0; c < C; ++c) {
    REAL* buf = (REAL*)malloc(sizeof(REAL) * (size_t)(3 * P + wsz));
    REAL *th = buf, *p = buf + P, *g = buf + 2 * P, *work = buf + 3 * P;
    REAL kin = 0;
    for (int i = 0; i < P; ++i) { th[i] = theta[(size_t)c * P + i]; p[i] = p0[(size_t)c * P + i]; kin += p[i] * p[i]; }
    const REAL h_cur = -target[c] + (REAL)0.5 * kin;
    REAL tv;
    FN(leapfrog)(s, th, p, x, y, N, mu, sigma, temp, step, L, &tv, g, work);
    kin = 0;
    for (int i = 0; i < P; ++i) kin += p[i] * p[i];
    const REAL h_prop = -tv + (REAL)0.5 * kin;
    REAL rate = EXP(h_cur - h_prop);
    if (rate > (REAL)1) rate = (REAL)1; /* torch.min(exp(.), 1); NaN stays NaN => reject */
    const int acc = u[c] < rate;
    if (acc) {
      for (int i = 0; i < P; ++i) { theta[(size_t)c * P + i] = th[i]; grad[(size_t)c * P + i] = g[i]; }
      target[c] = tv;
    }
    accepted[c] = (unsigned char)acc;
    if (h_cur_out) h_cur_out[c] = h_cur;
    if (h_prop_out) h_prop_out[c] = h_prop;
    nacc += acc;
    free(buf);
  }
  return nacc;
}

static REAL FN(normal_logprob_sum)(const REAL* v, const REAL* loc, REAL scale, int P) {
  /* NormalizedKernel.log_prob, eeyore/kernels/normalized_kernel.py:14-15 */
  REAL a = 0;
  for (int i = 0; i < P; ++i) {
    const REAL d = v[i] - loc[i];
    a += -(d * d) / ((REAL)2 * scale * scale) - LOG(scale) - (REAL)0.9189385332046727;
  }
  return a;
}

/* MALA.draw (eeyore/samplers/mala.py:46-82), full batch, C chains; z [C,P] standard normals, u [C]. */
int FN(mala_draw_chains)(const oc_spec* s, int C, REAL* theta, REAL* target, REAL* grad, const REAL* z,
                         const REAL* u, const REAL* x, const REAL* y, int N, const REAL* mu, const REAL* sigma,
                         double temp, double step, unsigned char* accepted, REAL* log_rate_out, int nthreads) {
  const int P = oc_num_params(s);
  const int wsz = oc_work_size(s);
  const REAL eps = (REAL)step, half = (REAL)0.5, scale = (REAL)sqrt(step);
  int nacc = 0;
#pragma omp parallel for num_threads(nthreads) reduction(+ : nacc) schedule(static)
  for (int c = 0; c < C; ++c) {
    REAL* buf = (REAL*)malloc(sizeof(REAL) * (size_t)(4 * P + wsz));
    REAL *loc = buf, *prop = buf + P, *g = buf + 2 * P, *loc2 = buf + 3 * P, *work = buf + 4 * P;
    const REAL* th = theta + (size_t)c * P;
    const REAL* gc = grad + (size_t)c * P;
    for (int i = 0; i < P; ++i) { loc[i] = th[i] + half * eps * gc[i]; prop[i] = loc[i] + scale * z[(size_t)c * P + i]; }
    const REAL tv = FN(log_target_grad)(s, prop, x, y, N, mu, sigma, temp, g, 0, 0, work);
    REAL lr = tv - target[c];
    lr = lr - FN(normal_logprob_sum)(prop, loc, scale, P);
    for (int i = 0; i < P; ++i) loc2[i] = prop[i] + half * eps * g[i];
    lr = lr + FN(normal_logprob_sum)(th, loc2, scale, P);
    const int acc = LOG(u[c]) < lr;
    if (acc) {
      for (int i = 0; i < P; ++i) { theta[(size_t)c * P + i] = prop[i]; grad[(size_t)c * P + i] = g[i]; }
      target[c] = tv;
    }
    accepted[c] = (unsigned char)acc;
    if (log_rate_out) log_rate_out[c] = lr;
    nacc += acc;
    free(buf);
  }
  return nacc;
}

/* MetropolisHastings.draw (eeyore/samplers/metropolis_hastings.py:41-73), symmetric NormalKernel(scale[P]). */
int FN(mh_draw_chains)(const oc_spec* s, int C, REAL* theta, REAL* target, const REAL* z, const REAL* u,
                       const REAL* scale, const REAL* x, const REAL* y, int N, const REAL* mu, const REAL* sigma,
                       double temp, unsigned char* accepted, REAL* log_rate_out, int nthreads) {
  const int P = oc_num_params(s);
  const int wsz = oc_work_size(s);
  int nacc = 0;
#pragma omp parallel for num_threads(nthreads) reduction(+ : nacc) schedule(static)
  for (int c = 0; c < C; ++c) {
    REAL* buf = (REAL*)malloc(sizeof(REAL) * (size_t)(P + wsz));
    REAL *prop = buf, *work = buf + P;
    for (int i = 0; i < P; ++i) prop[i] = theta[(size_t)c * P + i] + scale[i] * z[(size_t)c * P + i];
    const REAL tv = FN(log_target_grad)(s, prop, x, y, N, mu, sigma, temp, 0, 0, 0, work);
    const REAL lr = tv - target[c];
    const int acc = LOG(u[c]) < lr;
    if (acc) { for (int i = 0; i < P; ++i) theta[(size_t)c * P + i] = prop[i]; target[c] = tv; }
    accepted[c] = (unsigned char)acc;
    if (log_rate_out) log_rate_out[c] = lr;
    nacc += acc;
    free(buf);
  }
  return nacc;
}

#undef FN
#undef CAT
#undef CAT_
