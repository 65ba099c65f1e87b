from .loss import binary_cross_entropy
