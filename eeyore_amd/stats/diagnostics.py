"""Post-hoc chain diagnostics with the reference's definitions: sample covariance, the
initial-sequence (INSE) Monte Carlo covariance of Dai & Jones as implemented in mcmcse's insec.cpp, multivariate ESS
and multivariate R-hat.  The reference walks every lag with a Python double loop of outer products
(eeyore/stats/inse_mc_cov.py:20-33); here each lag's autocovariance is one matmul, on whatever device holds the chain.
"""
import torch


def is_pos_def(x):
    """eeyore/linalg/is_pos_def.py:3-11."""
    if torch.equal(x, x.t()):
        try:
            torch.linalg.cholesky(x)
            return True
        except RuntimeError:
            return False
    return False


def nearest_pd(A):
    """Nearest positive-definite matrix (Higham 1988, as eeyore/linalg/nearest_pd.py:9-42; the reference's loop uses
    the removed torch.eig, torch.linalg.eigvalsh stands in)."""
    B = (A + A.T) / 2
    _, s, Vh = torch.linalg.svd(B)
    H = Vh.T @ torch.diag(s) @ Vh
    A3 = (B + H) / 2
    A3 = (A3 + A3.T) / 2
    if is_pos_def(A3):
        return A3
    spacing = torch.finfo(A.dtype).eps * torch.norm(A).item()
    eye = torch.eye(A.shape[0], dtype=A.dtype, device=A.device)
    k = 1
    while not is_pos_def(A3):
        mineig = torch.linalg.eigvalsh(A3).min().item()
        A3 = A3 + eye * (-mineig * k**2 + spacing)
        k += 1
    return A3


def cov(x, rowvar=False):
    """eeyore/stats/cov.py:5-15: unbiased sample covariance; rows are observations unless rowvar."""
    if x.dim() > 2:
        raise ValueError('x has more than 2 dimensions')
    if x.dim() < 2:
        x = x.view(1, -1)
    if not rowvar and x.size(0) != 1:
        x = x.t()
    x_ctr = x - torch.mean(x, dim=1, keepdim=True)
    return x_ctr.matmul(x_ctr.t()).squeeze() / (x.size(1) - 1)


def cor_from_cov(x):
    """eeyore/stats/cor_from_cov.py:3-7."""
    d = 1 / torch.diag(x).sqrt()
    return x * d[None, :] * d[:, None]


def cor(x, rowvar=False):
    return cor_from_cov(cov(x, rowvar=rowvar))


def _gam(x_ctr, lag, n):
    """(1/n) sum_i x_ctr[i] (outer) x_ctr[i+lag]  --  inse_mc_cov.py:24-29 as one matmul.
    At lag 0 the reference's sum of outer products x_i x_i^T is symmetric to the last bit, and its positive-definiteness
    test demands exactly that (is_pos_def.py:4, torch.equal(x, x.t())); a BLAS product need not be, so the upper
    triangle is mirrored."""
    g = x_ctr[:x_ctr.shape[0] - lag].t() @ x_ctr[lag:] / n
    if lag == 0:
        g = torch.triu(g) + torch.triu(g, 1).t()
    return g


def inse_mc_cov(x, adjust=False):
    """eeyore/stats/inse_mc_cov.py:9-83."""
    x_ctr = x - x.mean(0)
    n, p = x.shape
    ub = n // 2
    sn = ub
    if adjust:
        Gamadj = torch.zeros([p, p], dtype=x.dtype, device=x.device)
    Sig = None
    for m in range(ub):
        gam0, gam1 = _gam(x_ctr, 2 * m, n), _gam(x_ctr, 2 * m + 1, n)
        Gam = gam0 + gam1
        Gam = (Gam + Gam.t()) / 2
        Sig = (-gam0 + 2 * Gam) if m == 0 else (Sig + 2 * Gam)
        if is_pos_def(Sig):
            sn = m
            break
    if sn > (ub - 1):
        raise RuntimeError('Not enough samples')
    last_dtm = torch.det(Sig).item()
    for m in range(sn + 1, ub):
        gam0, gam1 = _gam(x_ctr, 2 * m, n), _gam(x_ctr, 2 * m + 1, n)
        Gam = gam0 + gam1
        Gam = (Gam + Gam.t()) / 2
        Sig1 = Sig + 2 * Gam
        current_dtm = torch.det(Sig1).item()
        if current_dtm <= last_dtm:
            break
        Sig = Sig1.clone()
        last_dtm = current_dtm
        if adjust:
            eigenvals, eigenvecs = torch.linalg.eigh(Gam)  # the reference's torch.symeig (:76) no longer exists
            eigenvals = torch.clamp(eigenvals, max=0)
            Gamadj = Gamadj - eigenvecs @ torch.diag(eigenvals) @ eigenvecs.t()
    if adjust:
        Sig = Sig + 2 * Gamadj
    return Sig


def mc_cov(x, method='inse', adjust=False, rowvar=False):
    """eeyore/stats/mc_cov.py:4-10."""
    if method == 'inse':
        return inse_mc_cov(x, adjust=adjust)
    elif method == 'iid':
        return cov(x, rowvar=rowvar)
    raise ValueError('The method can be inse or iid, {} was given'.format(method))


def mc_se_from_cov(x):
    return torch.diag(x).sqrt()


def mc_se(x, method='inse', adjust=False, rowvar=False):
    return mc_se_from_cov(mc_cov(x, method=method, adjust=adjust, rowvar=rowvar))


def mc_cor(x, method='inse', adjust=False, rowvar=False):
    return cor_from_cov(mc_cov(x, method=method, adjust=adjust, rowvar=rowvar))


def multi_ess(x, mc_cov_mat=None, method='inse', adjust=False):
    """eeyore/stats/multi_ess.py:6-14: n (det(cov) / det(mc_cov))^(1/p)."""
    num_iters, num_pars = x.shape
    cov_mat_det = torch.det(cov(x, rowvar=False)).item()
    mc_cov_mat_det = torch.det(
        mc_cov(x, method=method, adjust=adjust, rowvar=False) if mc_cov_mat is None else mc_cov_mat
    ).item()
    return num_iters * ((cov_mat_det / mc_cov_mat_det) ** (1/num_pars))


def multi_rhat(x, mc_cov_mat=None, method='inse', adjust=False):
    """eeyore/stats/multi_rhat.py:10-40.  x [num_chains, num_iters, num_pars].
    Returns (rhat, imag part of the leading eigenvalue, W, B, is_w_pd, is_b_pd)."""
    num_chains, num_iters, num_pars = x.shape
    w = torch.zeros([num_pars, num_pars], dtype=x.dtype, device=x.device)
    for i in range(num_chains):
        w = w + (mc_cov(x[i], method=method, adjust=adjust, rowvar=False) if mc_cov_mat is None else mc_cov_mat[i])
    w = w / num_chains
    is_w_pd = is_pos_def(w)
    if not is_w_pd:
        w = nearest_pd(w)
    b = cov(x.mean(1), rowvar=False)
    is_b_pd = is_pos_def(b)
    if not is_b_pd:
        b = nearest_pd(b)
    eigvals = torch.linalg.eigvals(torch.matmul(torch.inverse(w), b))
    k = eigvals.real.argmax().item()
    rhat = eigvals.real[k].item()
    rhat = ((num_iters - 1) / num_iters) + ((num_chains + 1) / num_chains) * rhat
    return rhat, eigvals.imag[k].item(), w, b, is_w_pd, is_b_pd


def running_mean(x, dim=0):
    """eeyore/stats/running_mean.py:3-10."""
    n = x.size(dim)
    shape = [1] * x.dim()
    shape[dim] = -1
    return torch.cumsum(x, dim=dim) / torch.arange(1, n + 1, device=x.device).view(shape)


def recursive_mean(lastmean, n, x, offset=0):
    k = n - offset
    return ((k - 1) * lastmean + x) / k
