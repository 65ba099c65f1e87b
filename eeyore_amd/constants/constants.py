"""``loss_functions`` and ``torch_to_np_types`` (eeyore/constants/constants.py:7,15-18).

Each loss is a small object: calling it evaluates the reference's formula with torch ops (so ``model.loss(out, y)``
keeps working in user code), while the samplers only read ``.code`` -- the likelihood code of the C ABI
(include/eeyore_amd.h: enum ey_lik) -- and run the fused kernels."""
import numpy as np
import torch
import torch.nn.functional as F

from eeyore_amd.stats.loss import binary_cross_entropy

torch_to_np_types = {torch.float32: np.float32, torch.float64: np.float64}


class Loss:
    def __init__(self, name, code, fn):
        self.name, self.code, self._fn = name, code, fn

    def __call__(self, output, target):
        return self._fn(output, target)

    def __repr__(self):
        return f"Loss({self.name!r}, code={self.code})"


def _bce_sum(probabilities, y):
    return binary_cross_entropy(probabilities, y, reduction='sum')


def _ce_sum(logits, y_onehot):
    return F.cross_entropy(logits, torch.argmax(y_onehot, 1), reduction='sum')


loss_functions = {
    'binary_classification': Loss('binary_classification', 0, _bce_sum),
    'multiclass_classification': Loss('multiclass_classification', 1, _ce_sum),
}
