from pathlib import Path

import torch

from eeyore_amd.chains import ChainBuffer, ChainList

from .serial_sampler import SerialSampler


class SingleChainSerialSampler(SerialSampler):
    """eeyore/samplers/single_chain_serial_sampler.py:5-41, extended to hold C chains advanced together.

    ``theta0`` of shape [P] is the reference's single chain: ``current['sample']`` is [P], ``accepted`` a python int,
    the chain a ``ChainList``.  ``theta0`` of shape [C, P] selects the chain-batched mode: state tensors carry a
    leading chain axis, ``accepted`` is a uint8 tensor [C] that stays on the device, the chain a ``ChainBuffer``."""

    def __init__(self, counter):
        super().__init__(counter=counter)

    # ---- state layout helpers shared by HMC / MALA / MetropolisHastings
    def _init_mode(self, theta0, chain, rng, seed, chain_offset):
        self.batched = theta0 is not None and theta0.dim() == 2
        self.num_chains = theta0.shape[0] if self.batched else 1
        if chain is None:
            chain = ChainBuffer() if self.batched else ChainList()
        self.chain = chain
        self.rng = rng or ('philox' if self.batched else 'torch')
        if self.rng not in ('philox', 'torch'):
            raise ValueError("rng must be 'philox' (in-kernel counter-based stream) or 'torch' (global torch generator)")
        self.seed = int(seed)
        self.chain_offset = int(chain_offset)

    def _temp(self):
        """Temperature(s) of the chains: the sampler's own per-chain vector if given, else model.temperature."""
        t = getattr(self, 'temperature', None)
        return t if t is not None else self.model.temperature

    def _step_args(self):
        """(scalar step, per-chain step vector or None) for the C ABI."""
        if isinstance(self.step, torch.Tensor) and self.step.dim() == 1:
            return 0.0, self.step
        return float(self.step), None

    def _state_tensor(self, theta):
        th = theta.detach().to(device=self.model.device, dtype=self.model.dtype)
        return (th if th.dim() == 2 else th[None]).contiguous().clone()

    def _expose(self, t):
        """[C, ...] device state -> what ``current[...]`` holds ([...] for the single-chain mode)."""
        return t if self.batched else t[0]

    def _publish(self, accepted):
        c = self.current
        c['sample'] = self._expose(self._theta)
        c['target_val'] = self._expose(self._target)
        if hasattr(self, '_grad') and 'grad_val' in c:
            c['grad_val'] = self._expose(self._grad)
        c['accepted'] = accepted if self.batched else int(accepted[0].item())
        if not self.batched:
            self.model.set_params(c['sample'])  # the model's parameters follow the chain (hmc.py:149-155)

    def _randn(self, C, P):
        return torch.randn(C, P, dtype=self.model.dtype, device=self.model.device) if self.batched else \
            torch.randn(P, dtype=self.model.dtype, device=self.model.device)[None]

    def _rand(self, C):
        return torch.rand(C, dtype=self.model.dtype, device=self.model.device) if self.batched else \
            torch.rand(1, dtype=self.model.dtype, device=self.model.device)

    # ---- reference surface
    def get_model(self):
        return self.model

    def get_chain(self):
        return self.chain

    def get_param(self, idx):
        return self.get_chain().get_param(idx)

    def get_sample(self, idx):
        return self.get_chain().get_sample(idx)

    def set_current(self, theta, data=None):
        self.current = {key: None for key in self.keys}
        self.current['sample'] = theta
        x, y = data or next(iter(self.dataloader))
        return x, y

    def set_all(self, theta, data=None):
        self.set_current(theta, data=data)

    def reset(self, theta, data=None, reset_counter=True, reset_chain=True):
        if reset_counter:
            self.counter.reset()
        if reset_chain:
            self.chain.reset(keys=self.chain.vals.keys())
        self.set_all(theta, data=data)

    def to_chainfile(self, path=Path.cwd(), mode='a'):
        self.chain.to_chainfile(path=path, mode=mode)
