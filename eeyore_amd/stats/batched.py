"""Diagnostics of every (chain, parameter) series of a stored run in one device pass.

The reference computes its initial-sequence estimator with a Python double loop per chain
(eeyore/stats/inse_mc_cov.py:20-31); here ``ey_inse_univariate`` does it for all C * P series at once, one parameter at a
time (p = 1).  There is no CPU path: the samples must live on the ROCm device (use ``eeyore_amd.stats.inse_mc_cov`` for
the reference's multivariate statistic of one small chain)."""
import ctypes as ct

import torch

from eeyore_amd import _lib as L

_DT = {torch.float32: L.EY_F32, torch.float64: L.EY_F64}


def inse_univariate(samples):
    """samples [n, ...] (e.g. a ChainBuffer's [iterations, C, P]) -> dict of tensors shaped like samples[0]:
    ``sig2`` the initial-sequence estimate of the asymptotic variance of each series (NaN where the reference raises
    'Not enough samples', inse_mc_cov.py:45-46), ``var`` its unbiased sample variance (cov.py:5-15), ``pairs`` the
    number of lag pairs that entered the estimate."""
    if not samples.is_cuda:
        raise RuntimeError("eeyore_amd.stats.batched: the samples must be on the ROCm device (no CPU fallback)")
    if samples.dtype not in _DT:
        raise ValueError(f"unsupported dtype {samples.dtype}")
    if samples.dim() < 1 or samples.shape[0] < 2:
        raise ValueError("at least two iterations are needed")
    x = samples.contiguous()
    n = x.shape[0]
    shape = tuple(x.shape[1:])
    S = x[0].numel()
    sig2 = torch.empty(shape, dtype=torch.float64, device=x.device)
    var = torch.empty(shape, dtype=torch.float64, device=x.device)
    pairs = torch.empty(shape, dtype=torch.int32, device=x.device)
    L.check(L.lib().ey_inse_univariate(L.ptr(x), n, S, _DT[x.dtype], L.ptr(sig2), L.ptr(var), L.ptr(pairs),
                                       ct.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)),
            "ey_inse_univariate")
    return dict(sig2=sig2, var=var, pairs=pairs)


def ess(samples):
    """Effective sample size of each series: multi_ess (eeyore/stats/multi_ess.py:6-14) read for p = 1,
    n * var / sig2."""
    r = inse_univariate(samples)
    return samples.shape[0] * r["var"] / r["sig2"]


def mc_se(samples):
    """eeyore/stats/mc_se.py:4-5 with mc_se_from_cov.py:3-4 for p = 1: sqrt of the asymptotic variance."""
    return torch.sqrt(inse_univariate(samples)["sig2"])
