import hashlib

import torch
import torch.nn as nn


class Model(nn.Module):
    """Sampleable neural network model (eeyore/models/model.py:5-55): the flat parameter vector theta in
    ``parameters()`` order is the MCMC state."""

    def __init__(self, dtype=torch.float64, device='cpu'):
        super().__init__()
        self.dtype = dtype
        self.device = device

    def summary(self, hashsummary=False):
        print(self)
        print("-" * 80)
        print(f"Number of model parameters: {self.num_params()}")
        print("-" * 80)
        if hashsummary:
            print('Hash Summary:')
            for idx, hashvalue in enumerate(self.hashsummary()):
                print(f"{idx}: {hashvalue}")

    def hashsummary(self):
        result = []
        for child in self.children():
            result.extend(hashlib.sha256(x.detach().cpu().numpy().tobytes()).hexdigest() for x in child.parameters())
        return result

    def num_params(self):
        return sum(p.numel() for p in self.parameters())

    def get_params(self):
        return torch.cat([p.view(-1) for p in self.parameters()])

    def get_grad(self):
        return torch.cat([p.grad.view(-1) for p in self.parameters()])

    def set_params(self, theta, grad_val=None):
        """Scatter flat theta into the layer parameters as views (model.py:44-55)."""
        i = 0
        for p in self.parameters():
            j = p.numel()
            p.data = theta[i:i+j].view(p.size())
            if p.grad is not None:
                p.grad.detach_()
                p.grad.zero_()
            if grad_val is not None:
                p.grad = grad_val[i:i+j].view(p.size())
            i += j
