#!/usr/bin/env python3
"""Acceptance of HMC (L=20) on the bench workload as a function of the step size, after a burn-in."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.datasets import synthetic
from eeyore_amd.plan import Plan
dev = torch.device("cuda", 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
plan = Plan([4, 32, 32, 3], [1, 1, 1], [1, 1, 0], 1, torch.float32, dev)
plan.set_data(torch.tensor(xs, dtype=torch.float32, device=dev), torch.tensor(ys, dtype=torch.float32, device=dev))
plan.set_prior(torch.zeros(plan.P), torch.full((plan.P,), float(np.sqrt(3.0))))
C = 4096
for eps in [float(v) for v in sys.argv[1:]] or [0.011, 0.02, 0.03, 0.04, 0.05]:
    theta = 0.1 * plan.philox_normal(C, seed=0, it=0)
    target, grad = plan.log_target_grad(theta)
    accs = []
    for it in range(1, 301):
        out = plan.hmc_step(theta, target, grad, eps, 20, seed=7, it=it)
        if it > 200:
            accs.append(out["accepted"].float().mean().item())
    print(f"eps {eps}: acceptance after 200 burn-in iterations {np.mean(accs):.3f}; mean log-target {target.mean().item():.1f}")
