"""Post-hoc chain diagnostics behind the reference's function names (eeyore/stats/*.py): sample
covariance, the initial-sequence Monte Carlo covariance (Dai & Jones 2017, the estimator of mcmcse's insec.cpp),
multivariate effective sample size and multivariate R-hat.

Where the work is done:
* samples on the ROCm device go through the HIP kernels (``stats.batched.inse_multivariate`` -> ``ey_inse_multivariate``,
  every chain in one launch) whenever the kernel takes the width;
* anything else -- the reference's own use: one small chain in host memory -- runs the chain-batched torch form
  (``stats.batched.inse_mc_cov_chains``: every lag pair one batched product, the stopping rules as masks), with the single
  chain as a batch of one.  The reference walks every lag with a Python double loop of outer products
  (eeyore/stats/inse_mc_cov.py:20-31, O(n^2 p^2) interpreter steps).
Signatures, argument meaning and error behaviour are the reference's."""
import torch

from . import batched
from .linalg import is_pos_def, nearest_pd  # noqa: F401  (re-exported: eeyore.linalg's names)

_NOT_ENOUGH = 'Not enough samples'   # the reference's message, inse_mc_cov.py:45-46


def _on_device_kernel(x, width, adjust):
    return x.is_cuda and not adjust and width <= batched.MV_WIDTH_MAX and x.dtype in (torch.float32, torch.float64)


def _chain_mc_covs(chains, adjust):
    """Initial-sequence covariance of every chain of ``chains`` [m, n, p] -> [m, p, p] in the chains' dtype; raises the
    reference's RuntimeError if any chain has too few samples for a positive-definite estimate."""
    m, n, p = chains.shape
    if _on_device_kernel(chains, p, adjust):
        res = batched.inse_multivariate(chains, layout="cnp")
        short = res["pairs"] < 0
        sig = res["sig"].to(chains.dtype)
    else:
        sig = batched.inse_mc_cov_chains(chains, adjust=adjust)
        short = torch.isnan(sig).flatten(1).any(1)
    if bool(short.any()):
        raise RuntimeError(_NOT_ENOUGH)
    return sig


# ---------------------------------------------------------------------------------------------- covariance, correlation
def cov(x, rowvar=False):
    """Unbiased sample covariance (eeyore/stats/cov.py:5-15): observations in rows unless ``rowvar``; a vector is one
    variable; more than two dimensions raise ValueError."""
    if x.dim() > 2:
        raise ValueError('x has more than 2 dimensions')
    variables = x.reshape(1, -1) if x.dim() < 2 else x
    if not rowvar and variables.shape[0] != 1:
        variables = variables.mT
    return torch.cov(variables, correction=1)


def cor_from_cov(x):
    """Correlation matrix of a covariance matrix (eeyore/stats/cor_from_cov.py:3-7)."""
    scale = torch.rsqrt(torch.diagonal(x))
    return scale[:, None] * x * scale[None, :]


def cor(x, rowvar=False):
    """Sample correlation matrix (eeyore/stats/cor.py)."""
    return cor_from_cov(cov(x, rowvar))


# ---------------------------------------------------------------------------------------------- Monte Carlo covariance
def inse_mc_cov(x, adjust=False):
    """Initial-sequence estimator of the asymptotic covariance of one chain x [n, p] (eeyore/stats/inse_mc_cov.py:9-83);
    RuntimeError('Not enough samples') as there."""
    return _chain_mc_covs(x.unsqueeze(0), adjust)[0]


def _bad_method(method):
    return ValueError('The method can be inse or iid, {} was given'.format(method))   # the reference's text, mc_cov.py:10


_ESTIMATORS = {   # eeyore/stats/mc_cov.py:4-10
    'inse': lambda x, adjust, rowvar: inse_mc_cov(x, adjust=adjust),
    'iid': lambda x, adjust, rowvar: cov(x, rowvar=rowvar),   # the plain sample covariance
}


def mc_cov(x, method='inse', adjust=False, rowvar=False):
    """Monte Carlo covariance of one chain by the named estimator (eeyore/stats/mc_cov.py:4-10)."""
    if method not in _ESTIMATORS:
        raise _bad_method(method)
    return _ESTIMATORS[method](x, adjust, rowvar)


def mc_se_from_cov(x):
    """Monte Carlo standard errors: root of the diagonal (eeyore/stats/mc_se_from_cov.py:3-4)."""
    return torch.diagonal(x).sqrt()


def _after_mc_cov(finish, cite):
    def stat(x, method='inse', adjust=False, rowvar=False):
        return finish(mc_cov(x, method, adjust, rowvar))
    stat.__doc__ = f"{finish.__name__} of mc_cov(x, ...) ({cite})."
    return stat


mc_se = _after_mc_cov(mc_se_from_cov, "eeyore/stats/mc_se.py:4-5")
mc_cor = _after_mc_cov(cor_from_cov, "eeyore/stats/mc_cor.py")


# ---------------------------------------------------------------------------------------------- ESS and R-hat
def multi_ess(x, mc_cov_mat=None, method='inse', adjust=False):
    """Multivariate effective sample size of one chain x [n, p]: n (det cov / det mc_cov)^(1/p)
    (eeyore/stats/multi_ess.py:6-14), a Python float."""
    n, p = x.shape
    asymptotic = mc_cov(x, method=method, adjust=adjust) if mc_cov_mat is None else mc_cov_mat
    ratio = torch.linalg.det(cov(x)).item() / torch.linalg.det(asymptotic).item()
    return n * ratio ** (1 / p)


def multi_rhat(x, mc_cov_mat=None, method='inse', adjust=False):
    """Multivariate potential scale reduction of the chains x [m, n, p] (eeyore/stats/multi_rhat.py:10-40; Brooks &
    Gelman 1998, lemma 2).  Returns, as the reference, (rhat, imaginary part of the leading eigenvalue, W, B, whether W
    was positive definite, whether B was).  The within-chain matrices of all chains come from one batched pass."""
    m, n, p = x.shape
    if mc_cov_mat is not None:
        within = torch.stack([mc_cov_mat[i] for i in range(m)])
    elif method == 'inse':
        within = _chain_mc_covs(x, adjust)
    elif method in _ESTIMATORS:
        within = torch.stack([mc_cov(x[i], method, adjust) for i in range(m)]).reshape(m, p, p)
    else:
        raise _bad_method(method)
    return batched.multi_rhat_from_parts(within.sum(0), x.mean(1), n)


# ---------------------------------------------------------------------------------------------- running means
def running_mean(x, dim=0):
    """Mean of the first k entries along ``dim`` for every k (eeyore/stats/running_mean.py:3-10)."""
    counts = torch.arange(1, x.shape[dim] + 1, device=x.device, dtype=x.dtype if x.is_floating_point() else None)
    return x.cumsum(dim) / counts.reshape([-1 if d == dim % x.dim() else 1 for d in range(x.dim())])


def recursive_mean(lastmean, n, x, offset=0):
    """The mean after the n-th value from the mean before it (eeyore/stats/recursive_mean.py)."""
    seen = n - offset
    return ((seen - 1) * lastmean + x) / seen
