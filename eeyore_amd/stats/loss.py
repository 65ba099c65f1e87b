import torch


def binary_cross_entropy(x, y, reduction='mean'):
    """Binary cross entropy on probabilities with the plain logarithms of the reference (eeyore/stats/loss.py:1-11): a
    probability of exactly 0 or 1 gives -inf * 0 = NaN rather than a clamped value, which the samplers then reject."""
    per_element = torch.log(x) * y + torch.log(1 - x) * (1 - y)
    reducers = {'mean': torch.mean, 'sum': torch.sum}
    if reduction not in reducers:
        raise ValueError
    return -reducers[reduction](per_element)
