"""Diagnostics of every (chain, parameter) series of a stored run in one device pass.

The reference computes its initial-sequence estimator with a Python double loop per chain
(eeyore/stats/inse_mc_cov.py:20-31); here ``ey_inse_univariate`` does it for all C * P series at once, one parameter at a
time (p = 1).  There is no CPU path: the samples must live on the ROCm device (use ``eeyore_amd.stats.inse_mc_cov`` for
the reference's multivariate statistic of one small chain)."""
import ctypes as ct

import torch

from eeyore_amd import _lib as L

_DT = {torch.float32: L.EY_F32, torch.float64: L.EY_F64}
MV_WIDTH_MAX = 64   # parameters per chain ey_inse_multivariate takes (MW_P in csrc/ey_stats.hip)


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


# ---------------------------------------------------------------------------------------------- multivariate, many chains
def inse_multivariate(samples, layout="ncp"):
    """The reference's multivariate initial-sequence estimator (eeyore/stats/inse_mc_cov.py:9-83, adjust=False) for
    every chain at once on the device (``ey_inse_multivariate``: one workgroup per chain, p <= 64 parameters).
    ``samples``: [n, C, p] as a chain buffer stores a run (``layout="ncp"``) or [C, n, p] (``layout="cnp"``).
    Returns dict(sig [C,p,p], cov [C,p,p], mean [C,p], pairs [C]): the MC covariance (NaN where the reference raises
    'Not enough samples'), the unbiased sample covariance (cov.py:5-15), the chain means and the lag pairs used."""
    if not samples.is_cuda:
        raise RuntimeError("eeyore_amd.stats.batched: the samples must be on the ROCm device (no CPU fallback)")
    if samples.dtype not in _DT or samples.dim() != 3:
        raise ValueError("samples must be a 3-d float32 / float64 tensor")
    x = samples.contiguous()
    if layout == "ncp":
        n, C, p = x.shape
        sn, sc = C * p, p
    elif layout == "cnp":
        C, n, p = x.shape
        sn, sc = p, n * p
    else:
        raise ValueError("layout must be 'ncp' or 'cnp'")
    kw = dict(dtype=torch.float64, device=x.device)
    sig, cov, mean = torch.empty(C, p, p, **kw), torch.empty(C, p, p, **kw), torch.empty(C, p, **kw)
    pairs = torch.empty(C, dtype=torch.int32, device=x.device)
    L.check(L.lib().ey_inse_multivariate(L.ptr(x), n, C, p, sn, sc, _DT[x.dtype], L.ptr(sig), L.ptr(cov), L.ptr(mean),
                                         L.ptr(pairs), ct.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)),
            "ey_inse_multivariate")
    return dict(sig=sig, cov=cov, mean=mean, pairs=pairs, n=n)


def multi_ess_device(samples, layout="ncp"):
    """multi_ess (eeyore/stats/multi_ess.py:6-14) of every chain: n (det cov / det mc_cov)^(1/p) -> [C]."""
    r = inse_multivariate(samples, layout)
    p = r["sig"].shape[-1]
    return r["n"] * (torch.linalg.det(r["cov"]) / torch.linalg.det(r["sig"])) ** (1.0 / p)


def multi_rhat_from_parts(w_sum, means, n):
    """multi_rhat (eeyore/stats/multi_rhat.py:10-40) from its sufficient parts: ``w_sum`` [p,p] the SUM over chains of
    their MC covariances, ``means`` [m,p] the chain means, n iterations per chain.  This is the form the sharded
    statistic takes: ranks all-reduce w_sum and all-gather means.
    Returns (rhat, imag part of the leading eigenvalue, W, B, is_w_pd, is_b_pd) as the reference does."""
    from .linalg import is_pos_def, nearest_pd
    m = means.shape[0]
    w = w_sum / m
    is_w_pd = is_pos_def(w)
    if not is_w_pd:
        w = nearest_pd(w)
    b = torch.cov(means.mT)   # unbiased covariance of the chain means (eeyore/stats/cov.py:5-15)
    if b.dim() == 0:
        b = b.reshape(1, 1)
    is_b_pd = is_pos_def(b)
    if not is_b_pd:
        b = nearest_pd(b)
    eigvals = torch.linalg.eigvals(torch.matmul(torch.inverse(w), b))
    k = eigvals.real.argmax().item()
    rhat = ((n - 1) / n) + ((m + 1) / m) * eigvals.real[k].item()
    return rhat, eigvals.imag[k].item(), w, b, is_w_pd, is_b_pd


def multi_rhat_device(samples, layout="ncp"):
    """multi_rhat of the chains of one device: per-chain MC covariances and means from one launch."""
    r = inse_multivariate(samples, layout)
    return multi_rhat_from_parts(r["sig"].sum(0), r["mean"], r["n"])


def inse_mc_cov_chains(x, adjust=False):
    """(torch-batched form, any device; the HIP form is ``inse_multivariate``.)  The reference's multivariate initial-sequence estimator (eeyore/stats/inse_mc_cov.py:9-83, adjust=False) for C
    chains at once: x [C, n, p] -> [C, p, p] (``adjust=True``: the eigenvalue correction of :72-81 as well).  Every lag pair is ONE batched product over all chains (the reference's
    torch.ger double loop, :24-31), the positive-definiteness test (:41, a Cholesky attempt on a symmetric matrix,
    eeyore/linalg/is_pos_def.py:3-11) and the determinant test (:62-65) are batched too; chains leave the loop one by
    one through masks.  Chains for which the reference raises 'Not enough samples' (:45-46) come back as NaN.
    For small models (p up to a few dozen); for one parameter at a time over millions of series use inse_univariate."""
    if x.dim() != 3:
        raise ValueError("x must be [chains, iterations, parameters]")
    C, n, p = x.shape
    xc = x - x.mean(1, keepdim=True)                                     # :10
    ub = n // 2                                                          # :14

    def gam(lag):                                                        # :24-31 for every chain
        g = torch.matmul(xc[:, :n - lag].transpose(1, 2), xc[:, lag:]) / n
        if lag == 0:  # the reference's sum of x_i x_i^T is symmetric to the last bit and its positive-definiteness test
            # demands exactly that (is_pos_def.py:4); a BLAS product need not be, so the upper triangle is mirrored
            g = torch.triu(g) + torch.triu(g, 1).transpose(1, 2)
        return g

    def pos_def(m):
        sym = (m == m.transpose(1, 2)).flatten(1).all(1)
        _, info = torch.linalg.cholesky_ex(m)
        return sym & (info == 0)

    sig = torch.zeros(C, p, p, dtype=x.dtype, device=x.device)
    last = torch.zeros(C, dtype=x.dtype, device=x.device)
    state = torch.zeros(C, dtype=torch.int8, device=x.device)            # 0 searching, 1 extending, 2 stopped
    lift = torch.zeros_like(sig) if adjust else None                     # :17-18, the sum of the negative parts
    for m in range(ub):
        if bool((state == 2).all()):
            break
        g0, g1 = gam(2 * m), gam(2 * m + 1)
        G = g0 + g1
        G = (G + G.transpose(1, 2)) / 2                                  # :33-34
        searching, extending = state == 0, state == 1
        cand = torch.where(torch.tensor(m == 0, device=x.device), -g0 + 2 * G, sig + 2 * G)  # :36-39 / :62
        # chains still searching: accept the sum unconditionally, then test positive definiteness (:41-43)
        sig = torch.where(searching[:, None, None], cand, sig)
        found = searching & pos_def(torch.where(searching[:, None, None], sig, torch.eye(p, dtype=x.dtype,
                                                                                       device=x.device).expand(C, p, p)))
        det_new = torch.linalg.det(torch.where(found[:, None, None], sig, cand))
        last = torch.where(found, det_new, last)                         # :48
        # chains extending: keep the candidate only while the determinant strictly grows (:62-70)
        grow = extending & (det_new > last)
        sig = torch.where(grow[:, None, None], cand, sig)
        last = torch.where(grow, det_new, last)
        if adjust and bool(grow.any()):                                  # :72-79: the negative part of this lag pair's Gam
            lam, vec = torch.linalg.eigh(G)
            neg = (vec * lam.clamp(max=0)[:, None, :]) @ vec.transpose(1, 2)
            lift = torch.where(grow[:, None, None], lift - neg, lift)
        state = torch.where(found, torch.ones_like(state), state)
        state = torch.where(extending & ~grow, torch.full_like(state, 2), state)
    if adjust:
        sig = sig + 2 * lift                                             # :81
    sig = torch.where((state == 0)[:, None, None], torch.full_like(sig, float('nan')), sig)
    return sig


def multi_ess_chains(x):
    """multi_ess (eeyore/stats/multi_ess.py:6-14) for C chains at once: x [C, n, p] -> [C]."""
    C, n, p = x.shape
    xc = x - x.mean(1, keepdim=True)
    cov = torch.matmul(xc.transpose(1, 2), xc) / (n - 1)                 # eeyore/stats/cov.py:5-15
    return n * (torch.linalg.det(cov) / torch.linalg.det(inse_mc_cov_chains(x))) ** (1.0 / p)
