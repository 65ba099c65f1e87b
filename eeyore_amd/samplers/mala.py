import math

import torch

from .base import SingleChainSerialSampler, default_counter
from eeyore_amd.kernels import NormalKernel


class MALA(SingleChainSerialSampler):
    """Metropolis-adjusted Langevin algorithm (eeyore/samplers/mala.py:9-82) as one ``ey_mala_step`` per draw:
    proposal theta + step/2 grad + sqrt(step) z (the reference's default ``NormalKernel``, mala.py:35-41), one
    evaluation at the proposal, log-rate with both proposal densities, accept iff log u < log-rate.
    A user-supplied ``kernel`` has no HIP counterpart and is rejected."""

    keys = ['sample', 'target_val', 'grad_val', 'accepted']

    def __init__(self, model, theta0=None, dataloader=None, data0=None, counter=None, step=0.1, kernel=None,
                 chain=None, rng=None, seed=0, chain_offset=0, temperature=None):
        super().__init__(default_counter(counter, dataloader))
        if kernel is not None:
            raise ValueError("MALA: only the default NormalKernel(theta + step/2 grad, sqrt(step)) proposal is fused "
                             "into the HIP step")
        self._configure(model, dataloader, theta0, chain, rng, seed, chain_offset, temperature)
        self.step = step
        if theta0 is not None:
            self.set_current(theta0.clone().detach(), data=data0)

    def set_current(self, theta, data=None):
        x, y = super().set_current(theta, data=data)
        self._theta = self._state_tensor(theta)
        self._target, self._grad = self.model._plan(x, y).log_target_grad(self._theta, temp=self._temp())
        self._publish(torch.zeros(self.num_chains, dtype=torch.uint8))
        self.current['accepted'] = None

    def kernel_mean(self, state):
        return state['sample'] + 0.5 * self.step * state['grad_val']

    @property
    def kernel(self):
        """The proposal density at the current state, materialised on demand (the step itself never builds it)."""
        loc = self.kernel_mean(self.current)
        return NormalKernel(loc, torch.full_like(loc, math.sqrt(self.step)))

    def _run_block(self, plan, k, rec):
        step, step_vec = self._step_args()
        return plan.mala_run(self._theta, self._target, self._grad, step, k, step_vec=step_vec, temp=self._temp(),
                             seed=self.seed, it=self._iter, chain_offset=self.chain_offset, **rec)

    def draw(self, x, y, savestate=False):
        plan = self.model._plan(x, y)
        temp = self._temp()
        if self.counter.num_batches != 1:  # mala.py:49-51
            self._target, self._grad = plan.log_target_grad(self._theta, temp=temp)
        z, u = self._draw_randoms(*self._theta.shape)
        step, step_vec = self._step_args()
        out = plan.mala_step(self._theta, self._target, self._grad, step, z=z, u=u, step_vec=step_vec, temp=temp,
                             seed=self.seed, it=self._iter, chain_offset=self.chain_offset)
        self._finish_draw(out, savestate)
