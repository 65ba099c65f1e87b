"""The reference's CPU path restated op for op on torch-CPU autograd.  TEST / BASELINE INFRASTRUCTURE ONLY: used by
tests/ (checked against the golden fixtures) and by bench.py's ``cpu_baseline`` leg, never by eeyore_amd/.

The reference itself cannot travel to the GPU box, so this is the stand-in that is timed there beside the GPU run
(BASELINE.md section 3, item 2).  It performs the reference's own sequence of torch operations per gradient
evaluation -- flat theta scattered into per-layer views (eeyore/models/model.py:44-55), ``nn.Linear`` + activation
per layer (eeyore/models/mlp.py:45-50), ``CrossEntropyLoss(reduction='sum')`` on ``argmax(y, 1)`` or the naive
BCE-sum (eeyore/constants/constants.py:15-18, eeyore/stats/loss.py:1-11), ``Normal.log_prob`` summed
(eeyore/models/bayesian_model.py:46-50), ``autograd.grad(create_graph=True)`` concatenated flat
(eeyore/models/log_target_model.py:15-23) -- and the reference's HMC draw with its L + 1 evaluations
(eeyore/samplers/hmc.py:100-156), chains one after the other as the reference runs them
(eeyore/samplers/power_posterior_sampler.py:131-133).
"""
import torch
import torch.nn as nn
from torch.distributions import Normal


class TorchReferencePath:
    def __init__(self, dims, acts, lik, x, y, mu, sigma, dtype=torch.float64, temperature=None):
        self.dims, self.dtype, self.lik, self.temperature = list(dims), dtype, int(lik), temperature
        self.acts = [None if a == 0 else {1: torch.sigmoid, 2: torch.tanh, 3: torch.relu}[int(a)] for a in acts]
        self.layers = nn.ModuleList(nn.Linear(dims[k], dims[k + 1]).to(dtype) for k in range(len(dims) - 1))
        self.P = sum(p.numel() for p in self.layers.parameters())
        self.x = torch.as_tensor(x, dtype=dtype)
        self.y = torch.as_tensor(y, dtype=dtype)
        self.prior = Normal(torch.as_tensor(mu, dtype=dtype).expand(self.P).clone(),
                            torch.as_tensor(sigma, dtype=dtype).expand(self.P).clone())
        self.ce = nn.CrossEntropyLoss(reduction='sum')

    def _set_params(self, theta):
        i = 0
        for p in self.layers.parameters():
            j = i + p.numel()
            p.data = theta[i:j].view(p.shape)
            if p.grad is not None:
                p.grad.detach_()
                p.grad.zero_()
            i = j

    def _get_params(self):
        return torch.cat([p.view(-1) for p in self.layers.parameters()])

    def _forward(self, h):
        for layer, act in zip(self.layers, self.acts):
            h = layer(h)
            if act is not None:
                h = act(h)
        return h

    def log_target(self, theta):
        self._set_params(theta)
        out = self._forward(self.x)
        if self.lik == 1:
            loss = self.ce(out, torch.argmax(self.y, 1))
        else:
            loss = -(torch.log(out) * self.y + torch.log(1 - out) * (1 - self.y)).sum()
        ll = -loss
        lp = torch.sum(self.prior.log_prob(self._get_params()))
        if self.temperature is not None:
            ll, lp = self.temperature * ll, self.temperature * lp
        return ll + lp

    def upto_grad_log_target(self, theta):
        val = self.log_target(theta)
        grads = torch.autograd.grad(val, list(self.layers.parameters()), create_graph=True)
        return val, torch.cat([g.view(-1) for g in grads])

    def leapfrog(self, position0, momentum0, step, num_steps):
        position = position0.clone().detach()
        t, g = self.upto_grad_log_target(position)
        momentum = momentum0 + 0.5 * step * g
        for _ in range(num_steps - 1):
            position = position + step * momentum
            t, g = self.upto_grad_log_target(position.clone().detach())
            momentum = momentum + step * g
        position = position + step * momentum
        t, g = self.upto_grad_log_target(position.clone().detach())
        momentum = -(momentum + 0.5 * step * g)
        return position, momentum, t, g

    def hmc_draw(self, cur, step, num_steps, p0=None, u=None):
        """One HMC.draw on the state dict ``cur`` (sample, target_val, grad_val); ``p0`` / ``u`` default to fresh draws."""
        momentum = torch.randn(self.P, dtype=self.dtype) if p0 is None else p0
        h_cur = -cur['target_val'] + 0.5 * torch.sum(momentum ** 2)
        pos, mom, t, g = self.leapfrog(cur['sample'], momentum, step, num_steps)
        h_prop = -t + 0.5 * torch.sum(mom ** 2)
        rate = torch.exp(h_cur - h_prop)
        rate = torch.min(rate, torch.ones_like(rate))
        draw = torch.rand(1, dtype=self.dtype) if u is None else u
        if draw < rate:
            cur['sample'], cur['target_val'], cur['grad_val'] = pos.clone().detach(), t.clone().detach(), g.clone().detach()
            cur['accepted'] = 1
        else:
            self._set_params(cur['sample'].clone().detach())
            cur['accepted'] = 0
        return cur

    def start(self, theta0):
        t, g = self.upto_grad_log_target(theta0.clone().detach())
        return dict(sample=theta0.clone().detach(), target_val=t.detach(), grad_val=g.detach(), accepted=None)
