import torch

from .base import SingleChainSerialSampler, default_counter
from eeyore_amd.kernels import NormalKernel


class MetropolisHastings(SingleChainSerialSampler):
    """Random-walk Metropolis-Hastings (eeyore/samplers/metropolis_hastings.py:8-73) as one ``ey_mh_step`` per draw.
    ``kernel`` may be a ``NormalKernel``: its ``density.scale`` is the proposal scale (ones by default, :26-29) and its
    location follows the chain.  For a Normal random walk q(a|b) = q(b|a), so ``symmetric=False`` yields the same
    log-rate (:51-54) and the flag is accepted for compatibility only."""

    keys = ['sample', 'target_val', 'accepted']

    def __init__(self, model, theta0=None, dataloader=None, data0=None, counter=None, symmetric=True, kernel=None,
                 chain=None, rng=None, seed=0, chain_offset=0, temperature=None):
        super().__init__(default_counter(counter, dataloader))
        if kernel is not None and not isinstance(kernel, NormalKernel):
            raise ValueError("MetropolisHastings: only a NormalKernel proposal is fused into the HIP step")
        self._configure(model, dataloader, theta0, chain, rng, seed, chain_offset, temperature)
        self.symmetric = symmetric
        if theta0 is not None:
            self.set_current(theta0.clone().detach(), data=data0)
        self.kernel = kernel or self.default_kernel(self.current)

    def default_kernel(self, state):
        unit = torch.ones(self.model.num_params(), dtype=self.model.dtype, device=self.model.device)
        return NormalKernel(state['sample'], unit)

    def _evaluate_target(self, plan):
        lik, prior = plan.log_target(self._theta, temp=self._temp())
        self._target = lik + prior

    def set_current(self, theta, data=None):
        x, y = super().set_current(theta, data=data)
        self._theta = self._state_tensor(theta)
        self._evaluate_target(self.model._plan(x, y))
        self._publish(torch.zeros(self.num_chains, dtype=torch.uint8))
        self.current['accepted'] = None

    def set_kernel(self, state, scale=None, scale_tril=None):
        self.kernel.set_density_params(state['sample'].clone().detach())

    def _scale(self):
        scale = self.kernel.density.scale
        return scale[0] if scale.dim() > 1 else scale

    def _run_block(self, plan, k, rec):
        out = plan.mh_run(self._theta, self._target, self._scale(), k, temp=self._temp(), seed=self.seed, it=self._iter,
                          chain_offset=self.chain_offset, **rec)
        return out

    def _draw_block(self, x, y, k, savestate):
        super()._draw_block(x, y, k, savestate)
        self.kernel.set_density_params(self.current['sample'])

    def draw(self, x, y, savestate=False):
        plan = self.model._plan(x, y)
        if self.counter.num_batches != 1:  # metropolis_hastings.py:44-45
            self._evaluate_target(plan)
        z, u = self._draw_randoms(*self._theta.shape)
        scale = self.kernel.density.scale
        out = plan.mh_step(self._theta, self._target, scale[0] if scale.dim() > 1 else scale, z=z, u=u,
                           temp=self._temp(), seed=self.seed, it=self._iter, chain_offset=self.chain_offset)
        self._finish_draw(out, savestate)
        self.kernel.set_density_params(self.current['sample'])
