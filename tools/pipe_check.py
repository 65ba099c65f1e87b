#!/usr/bin/env python3
"""The pipelined form of the fused f32 kernel (variant bit 3) against the two-waves-per-SIMD form on identical draws:
one HMC draw and a run of several from the same state, seed and iteration numbers.  Prints maximum differences."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.datasets import synthetic  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

dev = torch.device("cuda", 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
N = int(os.environ.get("PC_ROWS", str(len(xs))))
xs, ys = xs[:N], ys[:N]
plan = Plan([4, 32, 32, 3], [1, 1, 1], [1, 1, 0], 1, torch.float32, dev)
plan.set_data(torch.tensor(xs, dtype=torch.float32, device=dev), torch.tensor(ys, dtype=torch.float32, device=dev))
plan.set_prior(torch.zeros(plan.P), torch.full((plan.P,), float(np.sqrt(3.0))))
C = int(os.environ.get("PC_CHAINS", "1024"))
step = float(os.environ.get("PC_STEP", "0.011"))
L = int(os.environ.get("PC_L", "20"))
theta0 = 0.1 * plan.philox_normal(C, seed=0, it=0)
t0, g0 = plan.log_target_grad(theta0)
res = {}
for v in (0, 8):
    plan.set_variant(v)
    th, t, g = theta0.clone(), t0.clone(), g0.clone()
    out = dict(accepted=plan.empty(C, dtype=torch.uint8), rate=plan.empty(C), h_cur=plan.empty(C), h_prop=plan.empty(C))
    plan.hmc_step(th, t, g, step, L, seed=1, it=1, out=out)
    torch.cuda.synchronize()
    one = (th.clone(), t.clone(), g.clone(), out["accepted"].clone(), out["h_prop"].clone())
    for i in range(5):
        plan.hmc_step(th, t, g, step, L, seed=1, it=2 + i, out=out)
    torch.cuda.synchronize()
    res[v] = (one, (th.clone(), t.clone(), g.clone(), out["accepted"].clone(), out["h_prop"].clone()))
plan.set_variant(0)
for name, idx in (("one draw", 0), ("six draws", 1)):
    a, b = res[0][idx], res[8][idx]
    print(f"{name}: rows {N} chains {C} acceptance {a[3].float().mean().item():.3f} / {b[3].float().mean().item():.3f}  "
          f"decisions differ {int((a[3] != b[3]).sum().item())}  max|dtheta| {(a[0] - b[0]).abs().max().item():.3e}  "
          f"max|dtarget| {(a[1] - b[1]).abs().max().item():.3e}  max|dgrad| {(a[2] - b[2]).abs().max().item():.3e}  "
          f"max|dHprop| {(a[4] - b[4]).abs().max().item():.3e}  nan {int(torch.isnan(b[0]).sum().item())}")
