"""Two small matrix helpers the diagnostics need (the reference keeps them in eeyore/linalg/): a positive-definiteness
test and the nearest positive-definite matrix.  Host-side glue on whatever device holds the matrix; not on the hot path."""
import torch


def is_pos_def(a):
    """True when ``a`` equals its transpose bit for bit and a Cholesky factorisation succeeds -- the reference's
    criterion (eeyore/linalg/is_pos_def.py:3-11), without the exception: ``cholesky_ex`` reports the failing minor."""
    if a.dim() != 2 or a.shape[0] != a.shape[1] or not torch.equal(a, a.mT):
        return False
    return int(torch.linalg.cholesky_ex(a).info) == 0


def nearest_pd(a):
    """The nearest positive-definite matrix in the Frobenius norm (Higham 1988; eeyore/linalg/nearest_pd.py:9-42 gets the
    symmetric polar factor from an SVD, here it comes from the eigen-decomposition of the symmetric part, V |L| V^T, the
    same matrix).  If rounding leaves the result semi-definite its diagonal is lifted by growing multiples of the most
    negative eigenvalue until the Cholesky test passes."""
    sym = 0.5 * (a + a.mT)
    lam, vec = torch.linalg.eigh(sym)
    polar = (vec * lam.abs()) @ vec.mT
    out = 0.5 * (sym + polar)
    out = 0.5 * (out + out.mT)
    if is_pos_def(out):
        return out
    ident = torch.eye(a.shape[0], dtype=a.dtype, device=a.device)
    gap = torch.finfo(a.dtype).eps * float(torch.linalg.matrix_norm(a))
    rounds = 0
    while not is_pos_def(out):
        rounds += 1
        lowest = float(torch.linalg.eigvalsh(out)[0])
        out = out + (gap - lowest * rounds * rounds) * ident
    return out
