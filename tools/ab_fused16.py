#!/usr/bin/env python3
"""A/B of library builds on fused16 shapes (EEYORE_AMD_LIB selects the build): leapfrog-steps/s x chains, HMC L = 20."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.datasets import synthetic
from eeyore_amd.plan import Plan
dev = torch.device("cuda", 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
out = []
for dims, acts, dt in (([4, 32, 32, 3], [1, 1, 0], torch.float64), ([4, 16, 16, 3], [1, 1, 0], torch.float32),
                       ([4, 32, 32, 3], [2, 2, 0], torch.float32), ([4, 64, 64, 3], [1, 1, 0], torch.float32)):
    pl = Plan(dims, [1, 1, 1], acts, 1, dt, dev)
    pl.set_data(torch.tensor(xs, dtype=dt, device=dev), torch.tensor(ys, dtype=dt, device=dev))
    pl.set_prior(torch.zeros(pl.P), torch.full((pl.P,), 1.7))
    C = 4096
    th = 0.1 * pl.philox_normal(C, seed=0, it=0)
    t, g = pl.log_target_grad(th)
    for i in range(3):
        pl.hmc_step(th, t, g, 0.02, 20, seed=3, it=1 + i)
    torch.cuda.synchronize()
    best = 0
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(6):
            pl.hmc_step(th, t, g, 0.02, 20, seed=3, it=10 + 6 * rep + i)
        torch.cuda.synchronize()
        best = max(best, C * 20 * 6 / (time.perf_counter() - t0))
    out.append(f"{best:.3e}")
print(os.environ.get("EEYORE_AMD_LIB", "default"), " ".join(out))
