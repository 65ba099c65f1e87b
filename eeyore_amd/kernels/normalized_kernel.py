import torch

from .kernel import Kernel


class NormalizedKernel(Kernel):
    """eeyore/kernels/normalized_kernel.py:5-19."""

    def log_prob(self, state):
        return torch.sum(self.density.log_prob(state))

    def sample(self):
        return self.density.sample()
