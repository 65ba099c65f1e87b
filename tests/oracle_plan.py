"""A test double with the interface of ``eeyore_amd.plan.Plan`` backed by the C oracle on CPU tensors.

It lets the ``-m "not gpu"`` suite exercise the host logic (sampler run loop, burn-in gating, chain storage,
counters) in this GPU-less container.  It lives under tests/ and is never imported by the product.
"""
import numpy as np
import torch

from oracle.c_oracle import COracle


class OraclePlan:
    kernel = "oracle-test-double"

    def __init__(self, dims, bias, acts, likelihood, dtype):
        self.dims, self.bias, self.acts, self.lik = list(dims), list(bias), list(acts), int(likelihood)
        self.dtype = dtype
        self.np_dtype = np.float64 if dtype == torch.float64 else np.float32
        self.device = torch.device("cpu")
        self.P = sum((dims[l] + (1 if bias[l] else 0)) * dims[l + 1] for l in range(len(dims) - 1))
        self.mu = np.zeros(self.P)
        self.sigma = np.ones(self.P)
        self._data_key = None
        self.x = self.y = None
        self._co = {}

    @classmethod
    def for_model(cls, model):
        from eeyore_amd.models.mlp import activation_code
        return cls(model.hp.dims, model.hp.bias, [activation_code(a) for a in model.hp.activations], model.loss.code,
                   model.dtype)

    def set_data(self, x, y):
        self.x = x.detach().cpu().numpy().astype(self.np_dtype)
        self.y = y.detach().cpu().numpy().astype(self.np_dtype).reshape(self.x.shape[0], -1)
        self._data_key = 1
        self._co = {}

    def set_prior(self, mu, sigma):
        self.mu = np.broadcast_to(torch.as_tensor(mu).detach().cpu().numpy(), (self.P,)).copy()
        self.sigma = np.broadcast_to(torch.as_tensor(sigma).detach().cpu().numpy(), (self.P,)).copy()
        self._co = {}

    def _oracle(self, temp):
        key = None if temp is None else float(temp)
        if key not in self._co:
            self._co[key] = COracle(self.dims, self.acts, self.lik, self.x, self.y, self.mu, self.sigma,
                                    dtype=self.np_dtype, bias=self.bias, temperature=key)
        return self._co[key]

    def _t(self, a, dtype=None):
        return torch.as_tensor(np.asarray(a), dtype=dtype or self.dtype)

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dtype)

    def log_target(self, theta, temp=None, prior_only=False):
        co = self._oracle(temp)
        out = [co.log_target_grad(t.numpy(), want_grad=False) for t in theta]
        return self._t([o[2] for o in out]), self._t([o[3] for o in out])

    def log_target_grad(self, theta, temp=None):
        co = self._oracle(temp)
        out = [co.log_target_grad(t.numpy()) for t in theta]
        return self._t([o[0] for o in out]), self._t(np.stack([o[1] for o in out]))

    def leapfrog(self, theta, p, step, num_steps, step_vec=None, temp=None):
        co = self._oracle(temp)
        ts, gs = [], []
        for c in range(theta.shape[0]):
            th, pp, t, g = co.leapfrog(theta[c].numpy(), p[c].numpy(), step, num_steps)
            theta[c] = self._t(th); p[c] = self._t(pp)
            ts.append(t); gs.append(g)
        return self._t(ts), self._t(np.stack(gs))

    def hmc_step(self, theta, target, grad, step, num_steps, p0=None, u=None, step_vec=None, temp=None, seed=0, it=0,
                 chain_offset=0, flags=0, out=None):
        assert p0 is not None and u is not None, "the test double has no in-kernel RNG"
        co = self._oracle(temp)
        th, tv, g = theta.numpy(), target.numpy(), grad.numpy()  # share memory with the tensors: updated in place
        if step_vec is not None:  # per-chain step sizes: one oracle call per chain
            C = th.shape[0]
            acc, hc, hp = np.zeros(C, np.uint8), np.zeros(C, self.np_dtype), np.zeros(C, self.np_dtype)
            for c in range(C):
                tc, vc, gc = th[c:c + 1].copy(), tv[c:c + 1].copy(), g[c:c + 1].copy()
                a, b, d = co.hmc_draw(tc, vc, gc, np.ascontiguousarray(p0.numpy()[c:c + 1]),
                                      np.ascontiguousarray(u.numpy()[c:c + 1]), float(step_vec[c]), int(num_steps))
                th[c], tv[c], g[c], acc[c], hc[c], hp[c] = tc[0], vc[0], gc[0], a[0], b[0], d[0]
        else:
            acc, hc, hp = co.hmc_draw(th, tv, g, np.ascontiguousarray(p0.numpy()), np.ascontiguousarray(u.numpy()),
                                      float(step), int(num_steps))
        with np.errstate(over="ignore", invalid="ignore"):
            rate = np.minimum(np.exp(hc - hp), 1)
        return dict(accepted=torch.as_tensor(acc), rate=self._t(rate), h_cur=self._t(hc), h_prop=self._t(hp))

    def mala_step(self, theta, target, grad, step, z=None, u=None, step_vec=None, temp=None, seed=0, it=0,
                  chain_offset=0, flags=0, out=None):
        assert z is not None and u is not None
        co = self._oracle(temp)
        acc, lr = co.mala_draw(theta.numpy(), target.numpy(), grad.numpy(), np.ascontiguousarray(z.numpy()),
                               np.ascontiguousarray(u.numpy()), float(step))
        return dict(accepted=torch.as_tensor(acc), log_rate=self._t(lr))

    def mh_step(self, theta, target, scale, z=None, u=None, temp=None, seed=0, it=0, chain_offset=0, flags=0, out=None):
        assert z is not None and u is not None
        co = self._oracle(temp)
        acc, lr = co.mh_draw(theta.numpy(), target.numpy(), np.ascontiguousarray(z.numpy()),
                             np.ascontiguousarray(u.numpy()), torch.as_tensor(scale).numpy())
        return dict(accepted=torch.as_tensor(acc), log_rate=self._t(lr))


def _groups(temp, C):
    """Split a per-chain temperature vector into (value, index list) groups; None/scalar -> one group."""
    if temp is None or not isinstance(temp, torch.Tensor) or temp.dim() == 0:
        return [(None if temp is None else float(temp), list(range(C)))]
    vals = temp.detach().cpu().numpy()
    out = {}
    for c, v in enumerate(vals):
        out.setdefault(float(v), []).append(c)
    return list(out.items())


class VectorTempOraclePlan(OraclePlan):
    """OraclePlan accepting a per-chain temperature vector (parallel tempering): chains are processed per
    temperature group and the results scattered back in place."""

    def log_target(self, theta, temp=None, prior_only=False):
        lik, prior = self.empty(theta.shape[0]), self.empty(theta.shape[0])
        for t, idx in _groups(temp, theta.shape[0]):
            a, b = super().log_target(theta[idx].contiguous(), temp=t)
            lik[idx], prior[idx] = a, b
        return lik, prior

    def log_target_grad(self, theta, temp=None):
        tv, g = self.empty(theta.shape[0]), self.empty(*theta.shape)
        for t, idx in _groups(temp, theta.shape[0]):
            a, b = super().log_target_grad(theta[idx].contiguous(), temp=t)
            tv[idx], g[idx] = a, b
        return tv, g

    def mala_step(self, theta, target, grad, step, z=None, u=None, step_vec=None, temp=None, **kw):
        C = theta.shape[0]
        out = dict(accepted=torch.zeros(C, dtype=torch.uint8), log_rate=self.empty(C))
        for t, idx in _groups(temp, C):
            th, tv, g = theta[idx].contiguous(), target[idx].contiguous(), grad[idx].contiguous()
            st = float(step_vec[idx[0]]) if step_vec is not None else step
            o = super().mala_step(th, tv, g, st, z=z[idx].contiguous(), u=u[idx].contiguous(), temp=t)
            theta[idx], target[idx], grad[idx] = th, tv, g
            out["accepted"][idx], out["log_rate"][idx] = o["accepted"], o["log_rate"]
        return out

    def pt_swap_decide(self, ell_i, ell_j, t_i, t_j, u, dlogq=None):
        lr = (t_i - t_j) * (ell_j - ell_i)
        if dlogq is not None:
            lr = lr + dlogq
        return (torch.log(u) < lr).to(torch.uint8), lr


def attach(model, vector_temp=False):
    if vector_temp:
        object.__setattr__(model, "_hip_plan", VectorTempOraclePlan.for_model(model))
        return model
    return _attach(model)


def _attach(model):
    """Give ``model`` (device='cpu') the oracle test double in place of a HIP plan."""
    object.__setattr__(model, "_hip_plan", OraclePlan.for_model(model))
    return model
