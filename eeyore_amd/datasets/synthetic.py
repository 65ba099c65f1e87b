"""Seeded synthetic datasets of the shapes BASELINE.json names (no network, no bundled benchmarks)."""
import numpy as np
import torch

from .core import XYDataset

_IRIS_MEANS = np.array([[5.0, 3.4, 1.5, 0.2], [5.9, 2.8, 4.3, 1.3], [6.6, 3.0, 5.6, 2.0]])
_IRIS_SDS = np.array([[0.35, 0.38, 0.17, 0.10], [0.52, 0.31, 0.47, 0.20], [0.64, 0.32, 0.55, 0.27]])


def iris_shaped_arrays(seed=0, per_class=50):
    """Iris-shaped data: 3 balanced classes, 4 Gaussian features with Iris-like class means
    and standard deviations; y one-hot [N, 3].  Returns float64 numpy arrays."""
    rng = np.random.default_rng(seed)
    xs, ys = [], []
    for k in range(3):
        xs.append(_IRIS_MEANS[k] + _IRIS_SDS[k] * rng.standard_normal((per_class, 4)))
        ys.append(np.full(per_class, k))
    x = np.concatenate(xs)
    lab = np.concatenate(ys)
    y = np.zeros((x.shape[0], 3))
    y[np.arange(x.shape[0]), lab] = 1.0
    return x, y


def iris_shaped(seed=0, per_class=50, dtype=torch.float32, device='cpu'):
    x, y = iris_shaped_arrays(seed, per_class)
    return XYDataset(torch.tensor(x, dtype=dtype, device=device), torch.tensor(y, dtype=dtype, device=device))


def binary_xor_like_arrays(n=256, seed=0):
    """cfg2 throughput data: x ~ U(0,1)^2, y = (x1 > .5) xor (x2 > .5)."""
    rng = np.random.default_rng(seed)
    x = rng.random((n, 2))
    y = ((x[:, 0] > 0.5) ^ (x[:, 1] > 0.5)).astype(np.float64)[:, None]
    return x, y


def binary_xor_like(n=256, seed=0, dtype=torch.float32, device='cpu'):
    x, y = binary_xor_like_arrays(n, seed)
    return XYDataset(torch.tensor(x, dtype=dtype, device=device), torch.tensor(y, dtype=dtype, device=device))
