"""numpy restatement of the reference's initial-sequence estimator for ONE series (p = 1).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): the checker of ``ey_inse_univariate``.  Pinned by
``tests/golden/g8_univariate_stats.npz`` (the reference's ``inse_mc_cov`` and ``cov`` run on single columns of its
own ``examples/stats/chain0[1-4].csv``; ``tests/test_oracle_golden.py``).
"""
import numpy as np


def inse_univariate(x):
    """eeyore/stats/inse_mc_cov.py:9-83 for a series x [n] (the [n, 1] case).  Returns (sig2, pairs_used); raises
    RuntimeError('Not enough samples') as :45-46 does.  ``adjust`` is not a parameter: an accepted lag pair has
    Gam > 0 (:64), so the eigenvalue clamp of :74-80 adds exactly zero when p = 1."""
    x = np.asarray(x)
    n = x.shape[0]
    xc = x - x.mean(0)                                   # :10
    ub = int(np.floor(n / 2))                            # :14
    sn = ub                                              # :15

    def gam(lag):                                        # :24-31, the torch.ger loop for scalars
        return (xc[:n - lag] * xc[lag:]).sum() / n

    sig = None
    for m in range(ub):                                  # :20
        g0, g1 = gam(2 * m), gam(2 * m + 1)
        G = g0 + g1                                      # :33-34
        sig = (-g0 + 2 * G) if m == 0 else (sig + 2 * G)  # :36-39
        if sig > 0:                                      # is_pos_def of a 1x1 matrix, eeyore/linalg/is_pos_def.py:3-11
            sn = m                                       # :42
            break
    if sn > ub - 1:                                      # :45-46
        raise RuntimeError('Not enough samples')
    last = sig                                           # :48 (det of a 1x1 matrix)
    used = sn + 1
    for m in range(sn + 1, ub):                          # :50
        G = gam(2 * m) + gam(2 * m + 1)
        sig1 = sig + 2 * G                               # :62
        if sig1 <= last:                                 # :64-65
            break
        sig, last, used = sig1, sig1, m + 1              # :67-70
    return sig, used


def sample_var(x):
    """eeyore/stats/cov.py:5-15 for one series: the unbiased sample variance."""
    x = np.asarray(x)
    xc = x - x.mean()
    return (xc * xc).sum() / (x.shape[0] - 1)


def ess_univariate(x):
    """eeyore/stats/multi_ess.py:6-14 read for p = 1: n * var / sig2."""
    return x.shape[0] * sample_var(x) / inse_univariate(x)[0]
