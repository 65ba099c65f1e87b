"""Model base classes behind the reference's names (``Model``, ``LogTargetModel``, ``BayesianModel``).

A model is an ``nn.Module`` whose flattened parameters -- ``nn.Module.parameters()`` order, i.e. per layer the weight
matrix row-major and then the bias (eeyore/models/model.py:38-55) -- are the MCMC state ``theta``.  The reference
evaluates ``log_target`` with torch ops and differentiates it with autograd (eeyore/models/log_target_model.py:15-23,
eeyore/models/bayesian_model.py:30-56); here every evaluation is one call into the HIP library through the model's
``Plan`` (``_plan(x, y)``, provided by the concrete model), for one chain (``theta`` of shape [P]) or for C chains
([C, P]).
"""
import hashlib

import torch
import torch.nn as nn


class Model(nn.Module):
    def __init__(self, dtype=torch.float64, device='cpu'):
        super().__init__()
        self.dtype = dtype
        self.device = device

    # -- flat parameter vector <-> layer parameters
    def num_params(self):
        return sum(p.numel() for p in self.parameters())

    def get_params(self):
        return torch.cat([p.reshape(-1) for p in self.parameters()])

    def get_grad(self):
        return torch.cat([p.grad.reshape(-1) for p in self.parameters()])

    def set_params(self, theta, grad_val=None):
        """Point every layer parameter at its slice of ``theta`` (views, no copy); optionally install gradient slices."""
        offset = 0
        for p in self.parameters():
            end = offset + p.numel()
            p.data = theta[offset:end].view(p.shape)
            if p.grad is not None:
                p.grad.detach_()
                p.grad.zero_()
            if grad_val is not None:
                p.grad = grad_val[offset:end].view(p.shape)
            offset = end

    # -- reporting
    def hashsummary(self):
        """SHA-256 of every parameter tensor, layer by layer (model.py:23-32)."""
        return [hashlib.sha256(p.detach().cpu().numpy().tobytes()).hexdigest()
                for child in self.children() for p in child.parameters()]

    def _summary_lines(self):
        return [f"Number of model parameters: {self.num_params()}"]

    def summary(self, hashsummary=False):
        rule = "-" * 80
        print(self)
        for line in self._summary_lines():
            print(rule)
            print(line)
        print(rule)
        if hashsummary:
            print('Hash Summary:')
            for idx, digest in enumerate(self.hashsummary()):
                print(f"{idx}: {digest}")


class LogTargetModel(Model):
    """A model with a log-target density and its gradient with respect to theta."""

    def __init__(self, temperature=None, dtype=torch.float64, device='cpu'):
        super().__init__(dtype=dtype, device=device)
        self.temperature = temperature

    def log_target(self, theta, x, y):
        raise NotImplementedError

    def upto_grad_log_target(self, theta, x, y):
        raise NotImplementedError

    def grad_log_target(self, theta, x, y):
        """Gradient of the log-target at ``theta``.  (The reference takes the autograd graph of a log-target value,
        log_target_model.py:15-18; with a fused kernel the natural argument is the position.)"""
        return self.upto_grad_log_target(theta, x, y)[1]


class BayesianModel(LogTargetModel):
    """log_target = log_lik + log_prior, with log_lik = -loss(forward(x), y) and log_prior = sum prior.log_prob(theta);
    ``temperature`` (if set) multiplies both (bayesian_model.py:33-34,48-49)."""

    def __init__(self, loss, temperature=None, dtype=torch.float64, device='cpu'):
        super().__init__(temperature=temperature, dtype=dtype, device=device)
        self.loss = loss

    def default_prior(self):
        raise NotImplementedError

    def _summary_lines(self):
        return super()._summary_lines() + [f"Prior: {self.prior}"]

    # -- plumbing shared by the evaluations below
    def _chains(self, theta):
        """theta as a contiguous [C, P] batch on the model's device, and whether it was a single chain."""
        th = theta.detach()
        single = th.dim() == 1
        if single:
            th = th.unsqueeze(0)
        return th.to(device=self.device, dtype=self.dtype).contiguous(), single

    def _evaluate(self, theta, x, y, what, track=True):
        if track:  # the reference's log_target also leaves the model's parameters at theta (bayesian_model.py:53)
            self.set_params(theta if theta.dim() == 1 else theta[0])
        th, single = self._chains(theta)
        plan = self._plan(x, y)
        if what == 'grad':
            out = plan.log_target_grad(th, temp=self.temperature)
        elif what == 'prior':
            out = (plan.log_target(th, temp=self.temperature, prior_only=True)[1],)
        else:
            lik, prior = plan.log_target(th, temp=self.temperature)
            out = (lik,) if what == 'lik' else (lik + prior,)
        return tuple(o[0] for o in out) if single else out

    # -- the reference's evaluation surface
    def log_lik(self, x, y):
        """Log-likelihood at the model's current parameters."""
        return self._evaluate(self.get_params(), x, y, 'lik', track=False)[0]

    def set_params_and_log_lik(self, theta, x, y):
        return self._evaluate(theta, x, y, 'lik')[0]

    def set_params_and_lik(self, theta, x, y):
        return torch.exp(self.set_params_and_log_lik(theta, x, y))

    def log_prior(self, theta=None):
        """Log-prior at the model's current parameters (or at ``theta``)."""
        return self._evaluate(self.get_params() if theta is None else theta, None, None, 'prior', track=False)[0]

    def log_target(self, theta, x, y):
        """theta [P] -> 0-d tensor; theta [C, P] -> [C]."""
        return self._evaluate(theta, x, y, 'target')[0]

    def upto_grad_log_target(self, theta, x, y):
        """theta [P] -> (0-d, [P]); theta [C, P] -> ([C], [C, P])."""
        return self._evaluate(theta, x, y, 'grad')

    # -- posterior predictive by Monte Carlo integration over stored samples (bayesian_model.py:58-67)
    def _predictive_integrator(self, samples):
        from eeyore_amd.integrators import MCIntegrator
        return MCIntegrator(f=lambda s, x, y: self.set_params_and_lik(s.clone().detach(), x, y), samples=samples)

    def predictive_posterior(self, theta, x, y):
        return self._predictive_integrator(theta).integrate(x, y)

    def _stack_samples(self, theta):
        stack = theta if torch.is_tensor(theta) else torch.stack(list(theta))
        return self._chains(stack if stack.dim() == 2 else stack.unsqueeze(0))[0]

    def predictive_posterior_batched(self, theta, x, y):
        """The reference's predictive_posterior (bayesian_model.py:58-62) for K points at once: x [K, d_0], y [K, d_K],
        theta the stored samples ([S, P] tensor or a list of [P]).  Returns (estimates [K], dropped [K]): per point the
        mean over samples of the likelihood of that ONE point, NaN integrands dropped and counted
        (eeyore/integrators/mcintegrator.py:24-28).  One pass of the row log-likelihood kernel over S x K."""
        th = self._stack_samples(theta)
        rows = self._plan(x, y).log_lik_rows(th, temp=self.temperature)  # [S, K]
        lik = torch.exp(rows)
        ok = ~torch.isnan(lik)
        kept = ok.sum(0)
        est = torch.where(ok, lik, torch.zeros_like(lik)).sum(0) / kept.clamp(min=1)
        est = torch.where(kept > 0, est, torch.full_like(est, float('nan')))
        return est, (~ok).sum(0)

    def predictive_posterior_from_dataset(self, theta, dataset, num_points, shuffle=True, verbose=False, verbose_step=1):
        """(integrals, indices, numbers of dropped samples) as bayesian_model.py:64-67 /
        mcintegrator.py:38-63: points are drawn one at a time from a DataLoader(dataset, batch_size=1, shuffle) exactly
        as there (same draws from the torch generator), then all of them are integrated in one device pass."""
        from torch.utils.data import DataLoader
        xs, ys, idxs = [], [], []
        while len(xs) < num_points:
            for item in DataLoader(dataset, batch_size=1, shuffle=shuffle):
                if len(xs) == num_points:
                    break
                xs.append(item[0].reshape(1, -1))
                ys.append(item[1].reshape(1, -1))
                idxs.append(int(item[2]) if len(item) > 2 else -1)
                if verbose and len(xs) % verbose_step == 0:
                    print(f"Iteration {len(xs)} out of {num_points}")
        est, dropped = self.predictive_posterior_batched(theta, torch.cat(xs), torch.cat(ys))
        return (est.to(self.dtype), torch.tensor(idxs, dtype=torch.int64, device=est.device),
                dropped.to(torch.int64))
