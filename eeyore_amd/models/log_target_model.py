import torch

from .model import Model


class LogTargetModel(Model):
    """eeyore/models/log_target_model.py:7-23.  The reference differentiates ``log_target`` with autograd;
    here ``upto_grad_log_target`` is one fused HIP evaluation (value + hand-coded backward)."""

    def __init__(self, temperature=None, dtype=torch.float64, device='cpu'):
        super().__init__(dtype=dtype, device=device)
        self.temperature = temperature

    def log_target(self, theta, x, y):
        raise NotImplementedError

    def upto_grad_log_target(self, theta, x, y):
        raise NotImplementedError

    def grad_log_target(self, theta, x, y):
        """Gradient of the log-target at theta.  (The reference takes the autograd graph of a log_target value,
        log_target_model.py:15-18; with a fused kernel the natural argument is the position.)"""
        return self.upto_grad_log_target(theta, x, y)[1]
