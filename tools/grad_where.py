#!/usr/bin/env python3
"""Which entries of the headline model's gradient differ from the f64 oracle (a diagnostic for the spill builds):
    EEYORE_AMD_LIB=eeyore_amd/lib/libeeyore_amd_spill_ey_mfma32.so python tools/grad_where.py [rows] [exact]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.datasets import synthetic  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402
from oracle.c_oracle import COracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda", 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
xs, ys = xs[:N], ys[:N]
sigma = float(np.sqrt(3.0))
plan = Plan([4, 32, 32, 3], [1, 1, 1], [1, 1, 0], 1, torch.float32, dev)
if len(sys.argv) > 2:
    plan.f32_products = sys.argv[2]
plan.set_data(torch.tensor(xs, dtype=torch.float32, device=dev), torch.tensor(ys, dtype=torch.float32, device=dev))
plan.set_prior(torch.zeros(plan.P), torch.full((plan.P,), sigma))
co = COracle([4, 32, 32, 3], [1, 1, 0], 1, xs.astype(np.float32).astype(np.float64), ys, 0.0, sigma, dtype=np.float64, nthreads=4)
C = 8
th = (0.5 * plan.philox_normal(C, seed=3, it=0)).contiguous()
t, g = plan.log_target_grad(th)
names = [("W0", 0, 128), ("b0", 128, 160), ("W1", 160, 1184), ("b1", 1184, 1216), ("W2", 1216, 1312), ("b2", 1312, 1315)]
for c in range(C):
    tt, gg, _, _ = co.log_target_grad(th[c].cpu().numpy().astype(np.float64))
    gv = g[c].cpu().numpy()
    bad = np.flatnonzero(np.abs(gv - gg) > 1e-5 * np.abs(gg) + 2e-6 * max(1.0, np.abs(gg).max()))
    where = {n: [int(i - a) for i in bad if a <= i < b] for n, a, b in names}
    print(f"chain {c}: target {t[c].item():.4f} (oracle {tt:.4f}); wrong entries {len(bad)}: " +
          "; ".join(f"{n}{v}" for n, v in where.items() if v))
