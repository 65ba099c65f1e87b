def binary_cross_entropy(x, y, reduction='mean'):
    """Naive BCE on probabilities (eeyore/stats/loss.py:1-11): NaN once a probability is exactly 0 or 1."""
    loss = -(x.log() * y + (1 - x).log() * (1 - y))
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'sum':
        return loss.sum()
    raise ValueError
