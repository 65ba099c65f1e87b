import torch

from .single_chain_serial_sampler import SingleChainSerialSampler
from eeyore_amd.datasets import DataCounter
from eeyore_amd.kernels import NormalKernel


class MetropolisHastings(SingleChainSerialSampler):
    """Random-walk Metropolis-Hastings (eeyore/samplers/metropolis_hastings.py:8-73) as one ``ey_mh_step`` per
    draw.  ``kernel`` may be a ``NormalKernel``; its ``density.scale`` is the proposal scale (default ones, :26-29).
    For a Normal random walk q(a|b) = q(b|a), so ``symmetric=False`` gives the same log-rate (:51-54)."""

    def __init__(self, model, theta0=None, dataloader=None, data0=None, counter=None, symmetric=True, kernel=None,
                 chain=None, rng=None, seed=0, chain_offset=0, temperature=None):
        super(MetropolisHastings, self).__init__(counter or DataCounter.from_dataloader(dataloader))
        self.model = model
        self.temperature = temperature
        self.dataloader = dataloader
        self.symmetric = symmetric
        if kernel is not None and not isinstance(kernel, NormalKernel):
            raise ValueError("MetropolisHastings: only a NormalKernel proposal is fused into the HIP step")
        self._init_mode(theta0, chain, rng, seed, chain_offset)
        self.keys = ['sample', 'target_val', 'accepted']
        self._iter = 0

        if theta0 is not None:
            self.set_current(theta0.clone().detach(), data=data0)

        self.kernel = kernel or self.default_kernel(self.current)

    def default_kernel(self, state):
        loc = state['sample']
        scale = torch.ones(self.model.num_params(), dtype=self.model.dtype, device=self.model.device)
        return NormalKernel(loc, scale)

    def set_current(self, theta, data=None):
        x, y = super().set_current(theta, data=data)
        self._theta = self._state_tensor(theta)
        plan = self.model._plan(x, y)
        lik, prior = plan.log_target(self._theta, temp=self._temp())
        self._target = lik + prior
        self._publish(torch.zeros(self.num_chains, dtype=torch.uint8))
        self.current['accepted'] = None

    def set_kernel(self, state, scale=None, scale_tril=None):
        self.kernel.set_density_params(state['sample'].clone().detach())

    def draw(self, x, y, savestate=False):
        """metropolis_hastings.py:41-73."""
        plan = self.model._plan(x, y)
        C, P = self._theta.shape
        temp = self._temp()
        if self.counter.num_batches != 1:
            lik, prior = plan.log_target(self._theta, temp=temp)
            self._target = lik + prior
        z = u = None
        if self.rng == 'torch':
            z = self._randn(C, P)
            u = self._rand(C)
        scale = self.kernel.density.scale
        if scale.dim() > 1:
            scale = scale[0]
        out = plan.mh_step(self._theta, self._target, scale, z=z, u=u, temp=temp, seed=self.seed, it=self._iter,
                           chain_offset=self.chain_offset)
        self._iter += 1
        self._publish(out['accepted'])
        self.kernel.set_density_params(self.current['sample'])
        self.last = out
        if savestate:
            self.chain.detach_and_update(self.current)
