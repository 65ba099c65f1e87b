"""numpy restatement of eeyore's per-iteration MCMC step for an MLP Bayesian net.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Every function cites
the reference file:line (paths relative to the reference checkout) whose
behaviour it restates.  The arithmetic type is the dtype of ``theta``
(float64 is the reference default, ``eeyore/models/model.py:7``).

Conventions
-----------
* ``theta`` is the flat parameter vector in ``nn.Module.parameters()`` order:
  per layer ``W_l`` (``[d_{l+1}, d_l]`` row-major) then ``b_l`` if that layer
  has a bias (``eeyore/models/model.py:38-55``, ``eeyore/models/mlp.py:37-43``,
  pinned by ``tests/test_binary_classif_mlp2321_log_lik.py:50-64``).
* ``acts`` is a list of activation codes, one per layer:
  0 = None/identity, 1 = sigmoid, 2 = tanh, 3 = relu.
* ``lik`` is 0 for 'binary_classification' (BCE-sum on probabilities,
  ``eeyore/constants/constants.py:16`` + ``eeyore/stats/loss.py:1-11``) and
  1 for 'multiclass_classification' (CrossEntropyLoss(sum) on logits with
  ``argmax(y, 1)`` labels, ``eeyore/constants/constants.py:17``).
"""
import numpy as np

ACT_NONE, ACT_SIGMOID, ACT_TANH, ACT_RELU = 0, 1, 2, 3
LIK_BCE, LIK_CE = 0, 1

_HALF_LOG_2PI = 0.9189385332046727  # 0.5*log(2*pi), torch.distributions.Normal.log_prob


class Spec:
    """Static description of an MLP target: dims/bias/acts/likelihood (+prior, temperature)."""

    def __init__(self, dims, acts, lik, bias=None, mu=None, sigma=None, temperature=None):
        self.dims = [int(d) for d in dims]
        self.nl = len(self.dims) - 1
        self.acts = [int(a) for a in acts]
        self.bias = [1] * self.nl if bias is None else [int(b) for b in bias]
        assert len(self.acts) == self.nl and len(self.bias) == self.nl
        self.lik = int(lik)
        self.w_off, self.b_off = [], []
        off = 0
        for l in range(self.nl):
            self.w_off.append(off)
            off += self.dims[l + 1] * self.dims[l]
            self.b_off.append(off if self.bias[l] else -1)
            if self.bias[l]:
                off += self.dims[l + 1]
        self.P = off  # eeyore/models/model.py:34-36
        self.mu = np.zeros(self.P) if mu is None else np.asarray(mu)
        self.sigma = np.ones(self.P) if sigma is None else np.asarray(sigma)
        self.temperature = temperature


def _act(code, g):
    if code == ACT_NONE:
        return g
    if code == ACT_SIGMOID:
        return 1.0 / (1.0 + np.exp(-g))
    if code == ACT_TANH:
        return np.tanh(g)
    if code == ACT_RELU:
        return np.maximum(g, 0)
    raise ValueError(code)


def _act_deriv_from_output(code, h):
    one = h.dtype.type(1)
    if code == ACT_NONE:
        return np.ones_like(h)
    if code == ACT_SIGMOID:
        return h * (one - h)
    if code == ACT_TANH:
        return one - h * h
    if code == ACT_RELU:
        return (h > 0).astype(h.dtype)
    raise ValueError(code)


def forward(spec, theta, x):
    """``MLP.forward`` (eeyore/models/mlp.py:45-50). Returns the list [h_0=x, h_1, ..., h_K]."""
    dt = theta.dtype
    hs = [np.asarray(x, dtype=dt)]
    for l in range(spec.nl):
        W = theta[spec.w_off[l]:spec.w_off[l] + spec.dims[l + 1] * spec.dims[l]].reshape(spec.dims[l + 1], spec.dims[l])
        g = hs[-1] @ W.T
        if spec.bias[l]:
            g = g + theta[spec.b_off[l]:spec.b_off[l] + spec.dims[l + 1]]
        hs.append(_act(spec.acts[l], g).astype(dt))
    return hs


def labels_from_y(y):
    """``torch.argmax(y, 1)`` (eeyore/constants/constants.py:17); first maximal index."""
    return np.argmax(np.asarray(y), axis=1)


def log_lik(spec, theta, x, y, temperature="spec"):
    """``BayesianModel.log_lik`` (eeyore/models/bayesian_model.py:30-35)."""
    dt = theta.dtype
    out = forward(spec, theta, x)[-1]
    y = np.asarray(y, dtype=dt)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if spec.lik == LIK_BCE:
            # naive logs, eeyore/stats/loss.py:2 -- NaN once a sigmoid saturates to exactly 0/1
            val = np.sum(np.log(out) * y + np.log(dt.type(1) - out) * (dt.type(1) - y))
        else:
            lab = labels_from_y(y)
            m = out.max(axis=1, keepdims=True)
            lse = m[:, 0] + np.log(np.sum(np.exp(out - m), axis=1))
            val = np.sum(out[np.arange(out.shape[0]), lab] - lse)
    val = dt.type(val)
    t = spec.temperature if isinstance(temperature, str) else temperature
    if t is not None:
        val = dt.type(t) * val
    return val


def log_prior(spec, theta, temperature="spec"):
    """``BayesianModel.log_prior`` (eeyore/models/bayesian_model.py:46-50) with an elementwise Normal prior."""
    dt = theta.dtype
    mu = spec.mu.astype(dt)
    sg = spec.sigma.astype(dt)
    # torch.distributions.Normal.log_prob: -((v - loc)**2) / (2*var) - log(scale) - log(sqrt(2*pi))
    lp = -((theta - mu) ** 2) / (dt.type(2) * sg * sg) - np.log(sg) - dt.type(_HALF_LOG_2PI)
    val = dt.type(np.sum(lp))
    t = spec.temperature if isinstance(temperature, str) else temperature
    if t is not None:
        val = dt.type(t) * val
    return val


def log_target(spec, theta, x, y, temperature="spec"):
    """``BayesianModel.log_target`` (eeyore/models/bayesian_model.py:52-56)."""
    return theta.dtype.type(log_lik(spec, theta, x, y, temperature) + log_prior(spec, theta, temperature))


def upto_grad_log_target(spec, theta, x, y, temperature="spec"):
    """``LogTargetModel.upto_grad_log_target`` (eeyore/models/log_target_model.py:15-23).

    The reference differentiates with autograd; this is the hand-coded backward
    of SURVEY.md section 8(a8): output delta = dL/dh_K * act'(h_K), then
    dW_l = delta_l^T h_{l-1}, db_l = sum_n delta_l, delta_{l-1} = (delta_l W_l) * act'(h_{l-1}).
    """
    dt = theta.dtype
    one = dt.type(1)
    hs = forward(spec, theta, x)
    out = hs[-1]
    y = np.asarray(y, dtype=dt)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if spec.lik == LIK_BCE:
            lik = np.sum(np.log(out) * y + np.log(one - out) * (one - y))
            dout = y / out - (one - y) / (one - out)  # d/dh of the naive BCE log-lik
        else:
            lab = labels_from_y(y)
            m = out.max(axis=1, keepdims=True)
            e = np.exp(out - m)
            s = np.sum(e, axis=1, keepdims=True)
            lik = np.sum(out[np.arange(out.shape[0]), lab] - (m[:, 0] + np.log(s[:, 0])))
            onehot = np.zeros_like(out)
            onehot[np.arange(out.shape[0]), lab] = one
            dout = onehot - e / s
        grad = np.zeros(spec.P, dtype=dt)
        delta = dout * _act_deriv_from_output(spec.acts[-1], out)
        for l in range(spec.nl - 1, -1, -1):
            dout_l, din_l = spec.dims[l + 1], spec.dims[l]
            grad[spec.w_off[l]:spec.w_off[l] + dout_l * din_l] = (delta.T @ hs[l]).reshape(-1)
            if spec.bias[l]:
                grad[spec.b_off[l]:spec.b_off[l] + dout_l] = delta.sum(axis=0)
            if l > 0:
                W = theta[spec.w_off[l]:spec.w_off[l] + dout_l * din_l].reshape(dout_l, din_l)
                delta = (delta @ W) * _act_deriv_from_output(spec.acts[l - 1], hs[l])
    mu = spec.mu.astype(dt)
    sg = spec.sigma.astype(dt)
    lp = -((theta - mu) ** 2) / (dt.type(2) * sg * sg) - np.log(sg) - dt.type(_HALF_LOG_2PI)
    prior = dt.type(np.sum(lp))
    gprior = -(theta - mu) / (sg * sg)
    lik = dt.type(lik)
    t = spec.temperature if isinstance(temperature, str) else temperature
    if t is not None:
        t = dt.type(t)
        return dt.type(t * lik + t * prior), (t * (grad + gprior)).astype(dt)
    return dt.type(lik + prior), (grad + gprior).astype(dt)


# --------------------------------------------------------------------------- HMC

def hamiltonian(target_val, momentum):
    """``HMC.hamiltonian`` (eeyore/samplers/hmc.py:84-98): -target + 0.5*sum(p**2)."""
    dt = momentum.dtype
    return dt.type(-target_val + dt.type(0.5) * np.sum(momentum ** 2))


def leapfrog(spec, theta0, p0, x, y, step, num_steps, temperature="spec"):
    """``HMC.leapfrog`` (eeyore/samplers/hmc.py:100-124): L steps, L+1 gradient evaluations,
    final momentum negated. Returns (theta_L, p_L, target_L, grad_L)."""
    dt = theta0.dtype
    eps = dt.type(step)
    half = dt.type(0.5)
    theta = theta0.copy()
    target, g = upto_grad_log_target(spec, theta, x, y, temperature)
    # grad_potential = -g  =>  p - 0.5*eps*grad_potential = p + 0.5*eps*g
    p = p0 - half * eps * (-g)
    for _ in range(num_steps - 1):
        theta = theta + eps * p
        target, g = upto_grad_log_target(spec, theta, x, y, temperature)
        p = p - eps * (-g)
    theta = theta + eps * p
    target, g = upto_grad_log_target(spec, theta, x, y, temperature)
    p = p - half * eps * (-g)
    p = -p
    return theta, p, target, g


def hmc_draw(spec, cur, p0, u, x, y, step, num_steps, temperature="spec"):
    """``HMC.draw`` (eeyore/samplers/hmc.py:126-156), full-batch path (num_batches == 1).

    ``cur`` = dict(sample, target_val, grad_val); ``p0`` replaces ``torch.randn(P)`` (:134),
    ``u`` replaces ``torch.rand(1)`` (:148). Returns (new_cur, info)."""
    dt = cur["sample"].dtype
    h_cur = hamiltonian(cur["target_val"], p0)
    th, p, tv, gv = leapfrog(spec, cur["sample"], p0, x, y, step, num_steps, temperature)
    h_prop = hamiltonian(tv, p)
    with np.errstate(over="ignore", invalid="ignore"):
        rate = np.minimum(np.exp(dt.type(h_cur - h_prop)), dt.type(1))
    accepted = bool(dt.type(u) < rate)  # strict <, NaN => reject (:148)
    if accepted:
        new = dict(sample=th, target_val=tv, grad_val=gv)
    else:
        new = dict(sample=cur["sample"], target_val=cur["target_val"], grad_val=cur["grad_val"])
    info = dict(h_cur=h_cur, h_prop=h_prop, rate=dt.type(rate), accepted=int(accepted),
                prop_sample=th, prop_momentum=p, prop_target=tv)
    return new, info


def init_step(spec, theta, momentum, x, y):
    """``HMC.init_step`` (eeyore/samplers/hmc.py:38-77) on one full batch, AS THE REFERENCE BEHAVES: from step 1 with
    one leapfrog step per trial and the acceptance ratio r = exp(H_cur - H_prop); ``momentum`` replaces the
    ``torch.randn`` of :48 and is kept through all trials (:46-49, :71-72).  The direction a = 2 (r > 1/2) - 1 (:58) is an
    INTEGER tensor, so ``torch.pow(2, -a)`` (:60) and ``torch.pow(2, a)`` (:67) are integer powers and 2**(-1) is 0:
      a = +1: the loop condition r > pow(2, -1) is r > 0 -- the step doubles until exp underflows to 0 (or r is NaN);
      a = -1: the condition 1/r > 2 holds, the step is multiplied by pow(2, -1) = 0, the next ratio is exp(0) = 1 and
              the loop ends with step 0 (``tuner.num_steps`` then raises ZeroDivisionError, hmc.py:27).
    (What :58-77 evidently intend -- Hoffman & Gelman 2014, algorithm 4: double or halve until r crosses 1/2 -- is
    ``init_step_intended`` below, the form the per-chain extension uses.)"""
    dt = theta.dtype
    h_cur = hamiltonian(log_target(spec, theta, x, y), momentum)

    def ratio(step):
        _, p, tv, _ = leapfrog(spec, theta, momentum, x, y, step, 1)
        with np.errstate(over="ignore", invalid="ignore"):
            return np.exp(dt.type(h_cur - hamiltonian(tv, p)))

    step = 1.0
    r = ratio(step)
    a = 1 if r > 0.5 else -1
    threshold, factor = (0, 2) if a == 1 else (2, 0)  # integer pow(2, -a), pow(2, a)
    with np.errstate(divide="ignore"):
        while r ** a > threshold:
            step = factor * step
            r = ratio(step)
    return float(step)


def init_step_intended(spec, theta, momentum, x, y, max_doublings=60):
    """Hoffman & Gelman's heuristic for a first step size (2014, algorithm 4), which hmc.py:38-77 set out to write:
    double the step while the one-step acceptance ratio stays above 1/2, or halve it while it stays below."""
    dt = theta.dtype
    h_cur = hamiltonian(log_target(spec, theta, x, y), momentum)

    def ratio(step):
        _, p, tv, _ = leapfrog(spec, theta, momentum, x, y, step, 1)
        with np.errstate(over="ignore", invalid="ignore"):
            return np.exp(dt.type(h_cur - hamiltonian(tv, p)))

    step = 1.0
    r = ratio(step)
    a = 1 if r > 0.5 else -1
    for _ in range(max_doublings):
        if not r ** a > 2.0 ** (-a):
            break
        step = step * 2.0 ** a
        r = ratio(step)
    return float(step)


def dual_averaging(l, e0, d, eub, rates):
    """``HMCDATuner.tune`` (eeyore/tuners/hmcda_tuner.py:43-59; Hoffman & Gelman 2014, algorithms 4-5) fed with the
    acceptance rates of iterations 0, 1, ...: the exploring step exp(log e) and ``num_steps = max(1, round(l / e))``
    after each, the averaged step for the last one (``return_e=False``, as hmc.py:158-163 asks at the end of burn-in)."""
    g, t0, k = 0.05, 10, 0.75
    m = np.log(10 * e0)
    barh, logbare = 0.0, 0.0
    steps, nsteps = [], []
    for idx, rate in enumerate(rates):
        it = idx + 1
        d_w, e_w = 1 / (it + t0), 1 / it ** k
        barh = (1 - d_w) * barh + d_w * (d - rate)
        loge = m - np.sqrt(it) * barh / g
        if eub is not None:
            loge = min(loge, np.log(eub))
        logbare = e_w * loge + (1 - e_w) * logbare
        e = np.exp(loge) if idx < len(rates) - 1 else np.exp(logbare)
        steps.append(e)
        nsteps.append(max(1, round(l / e)))
    return np.array(steps), np.array(nsteps), (barh, logbare, m)


# --------------------------------------------------------------------------- MALA / MH

def normal_log_prob_sum(v, loc, scale):
    """``NormalizedKernel.log_prob`` (eeyore/kernels/normalized_kernel.py:14-15) for NormalKernel."""
    dt = v.dtype
    scale = np.asarray(scale, dtype=dt)
    lp = -((v - loc) ** 2) / (dt.type(2) * scale * scale) - np.log(scale) - dt.type(_HALF_LOG_2PI)
    return dt.type(np.sum(lp))


def mala_draw(spec, cur, z, u, x, y, step, temperature="spec"):
    """``MALA.draw`` (eeyore/samplers/mala.py:46-82), full-batch path.

    ``z`` replaces the standard-normal draw inside ``Normal(loc, scale).sample()``
    (loc + scale*z with scale = sqrt(step), mala.py:35-41); ``u`` replaces ``torch.rand(1)`` (:66)."""
    dt = cur["sample"].dtype
    eps = dt.type(step)
    half = dt.type(0.5)
    scale = dt.type(np.sqrt(step))  # np.sqrt on the python float, then cast (mala.py:39)
    loc_cur = cur["sample"] + half * eps * cur["grad_val"]
    prop = loc_cur + scale * z
    tv, gv = upto_grad_log_target(spec, prop, x, y, temperature)
    log_rate = dt.type(tv - cur["target_val"])
    log_q_fwd = normal_log_prob_sum(prop, loc_cur, scale)
    log_rate = dt.type(log_rate - log_q_fwd)
    loc_prop = prop + half * eps * gv
    log_q_bwd = normal_log_prob_sum(cur["sample"], loc_prop, scale)
    log_rate = dt.type(log_rate + log_q_bwd)
    with np.errstate(divide="ignore"):
        accepted = bool(np.log(dt.type(u)) < log_rate)
    if accepted:
        new = dict(sample=prop, target_val=tv, grad_val=gv)
    else:
        new = dict(sample=cur["sample"], target_val=cur["target_val"], grad_val=cur["grad_val"])
    info = dict(log_rate=log_rate, log_q_fwd=log_q_fwd, log_q_bwd=log_q_bwd, accepted=int(accepted),
                prop_sample=prop, prop_target=tv)
    return new, info


def mh_draw(spec, cur, z, u, x, y, scale, temperature="spec"):
    """``MetropolisHastings.draw`` (eeyore/samplers/metropolis_hastings.py:41-73), symmetric
    NormalKernel(loc=current, scale): prop = cur + scale*z, log_rate = delta target (:50)."""
    dt = cur["sample"].dtype
    scale = np.asarray(scale, dtype=dt)
    prop = cur["sample"] + scale * z
    tv = log_target(spec, prop, x, y, temperature)
    log_rate = dt.type(tv - cur["target_val"])
    with np.errstate(divide="ignore"):
        accepted = bool(np.log(dt.type(u)) < log_rate)
    if accepted:
        new = dict(sample=prop, target_val=tv)
    else:
        new = dict(sample=cur["sample"], target_val=cur["target_val"])
    info = dict(log_rate=log_rate, accepted=int(accepted), prop_sample=prop, prop_target=tv)
    return new, info


# --------------------------------------------------------------------------- parallel tempering

def pt_ladder(num_chains):
    """Default ladder ``t_i = (i/K)**4`` (eeyore/samplers/power_posterior_sampler.py:92)."""
    return [(i / num_chains) ** 4 for i in range(1, num_chains + 1)]


def pt_categorical_probs(i, num_chains, b):
    """``eval_categorical_probs`` (power_posterior_sampler.py:107-117): partner j != i with
    probability proportional to exp(-b|i-j|)."""
    eb = np.exp(-b)
    den = eb * (2 - eb ** i - eb ** (num_chains - 1 - i)) / (1 - eb)
    return np.array([eb ** abs(j - i) / den for j in list(range(i)) + list(range(i + 1, num_chains))])


def pt_swap_log_rate(log_q_i_given_j, log_q_j_given_i, ell_i, ell_j, t_i, t_j):
    """``between_chain_move_log_rate`` (power_posterior_sampler.py:135-141) rewritten with the
    untempered log-target ell(theta) = log_lik + log_prior:
    log q(i|j) - log q(j|i) - t_i ell_i - t_j ell_j + t_i ell_j + t_j ell_i."""
    return log_q_i_given_j - log_q_j_given_i + (t_i - t_j) * (ell_j - ell_i)
