from .loss import binary_cross_entropy
from .linalg import is_pos_def, nearest_pd
from .diagnostics import (cov, cor, cor_from_cov, inse_mc_cov, mc_cov, mc_se, mc_se_from_cov, mc_cor, multi_ess,
                          multi_rhat, running_mean, recursive_mean)
from . import batched
