"""``eeyore.constants`` counterpart (eeyore/constants/constants.py:7,15-18)."""
import numpy as np
import torch
import torch.nn as nn

from eeyore_amd.stats.loss import binary_cross_entropy

torch_to_np_types = {torch.float32: np.float32, torch.float64: np.float64}


class Loss:
    """A loss the HIP library knows by code.  Calling it evaluates the same formula with torch ops (API
    compatibility for user code that calls ``model.loss`` directly); the samplers never call it -- they
    read ``.code`` and run the fused kernels."""

    def __init__(self, name, code, fn):
        self.name, self.code, self._fn = name, code, fn

    def __call__(self, x, y):
        return self._fn(x, y)

    def __repr__(self):
        return f"Loss({self.name!r})"


loss_functions = {
    'binary_classification': Loss('binary_classification', 0, lambda x, y: binary_cross_entropy(x, y, reduction='sum')),
    'multiclass_classification': Loss('multiclass_classification', 1,
                                      lambda x, y: nn.CrossEntropyLoss(reduction='sum')(x, torch.argmax(y, 1))),
}
