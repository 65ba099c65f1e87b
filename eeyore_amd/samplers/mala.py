import numpy as np
import torch

from .single_chain_serial_sampler import SingleChainSerialSampler
from eeyore_amd.datasets import DataCounter
from eeyore_amd.kernels import NormalKernel


class MALA(SingleChainSerialSampler):
    """Metropolis-adjusted Langevin algorithm (eeyore/samplers/mala.py:9-82) as one ``ey_mala_step`` per draw.
    The proposal is the reference's default ``NormalKernel(theta + step/2 grad, sqrt(step))`` (mala.py:35-41);
    a user-supplied ``kernel`` other than that has no HIP counterpart and is rejected."""

    def __init__(self, model, theta0=None, dataloader=None, data0=None, counter=None, step=0.1, kernel=None,
                 chain=None, rng=None, seed=0, chain_offset=0, temperature=None):
        super().__init__(counter or DataCounter.from_dataloader(dataloader))
        self.model = model
        self.temperature = temperature
        self.dataloader = dataloader
        self.step = step
        if kernel is not None:
            raise ValueError("MALA: only the default NormalKernel(theta + step/2 grad, sqrt(step)) proposal is fused "
                             "into the HIP step")
        self._init_mode(theta0, chain, rng, seed, chain_offset)
        self.keys = ['sample', 'target_val', 'grad_val', 'accepted']
        self._iter = 0

        if theta0 is not None:
            self.set_current(theta0.clone().detach(), data=data0)

    def set_current(self, theta, data=None):
        x, y = super().set_current(theta, data=data)
        self._theta = self._state_tensor(theta)
        plan = self.model._plan(x, y)
        self._target, self._grad = plan.log_target_grad(self._theta, temp=self._temp())
        self._publish(torch.zeros(self.num_chains, dtype=torch.uint8))
        self.current['accepted'] = None

    def reset(self, theta, data=None, reset_counter=True, reset_chain=True):
        super().reset(theta, data=data, reset_counter=reset_counter, reset_chain=reset_chain)

    def kernel_mean(self, state):
        return state['sample'] + 0.5 * self.step * state['grad_val']

    @property
    def kernel(self):
        """The proposal density at the current state (mala.py:38-44), materialised on demand."""
        loc = self.kernel_mean(self.current)
        scale = torch.full_like(loc, np.sqrt(self.step))
        return NormalKernel(loc, scale)

    def draw(self, x, y, savestate=False):
        """mala.py:46-82."""
        plan = self.model._plan(x, y)
        C, P = self._theta.shape
        temp = self._temp()
        step, step_vec = self._step_args()
        if self.counter.num_batches != 1:
            self._target, self._grad = plan.log_target_grad(self._theta, temp=temp)
        z = u = None
        if self.rng == 'torch':
            z = self._randn(C, P)
            u = self._rand(C)
        out = plan.mala_step(self._theta, self._target, self._grad, step, z=z, u=u, step_vec=step_vec, temp=temp,
                             seed=self.seed,
                             it=self._iter, chain_offset=self.chain_offset)
        self._iter += 1
        self._publish(out['accepted'])
        self.last = out
        if savestate:
            self.chain.detach_and_update(self.current)
