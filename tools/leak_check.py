#!/usr/bin/env python3
"""Plans of every kernel family created, used and dropped 60 times: free device memory must not move.  usage: python tools/leak_check.py"""
import sys, gc, numpy as np, torch
sys.path.insert(0, '.')
from eeyore_amd.plan import Plan
dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
def once(dims, dt, N, C):
    nl = len(dims) - 1
    x = torch.tensor(rng.standard_normal((N, dims[0])), dtype=dt, device=dev)
    y = torch.tensor(np.eye(dims[-1])[rng.integers(0, dims[-1], N)], dtype=dt, device=dev)
    pl = Plan(dims, [1] * nl, [1] * (nl - 1) + [0], 1, dt, dev)
    pl.set_data(x, y); pl.set_prior(torch.zeros(pl.P), torch.ones(pl.P))
    th = 0.1 * pl.philox_normal(C, seed=0, it=0)
    t, g = pl.log_target_grad(th)
    pl.hmc_step(th, t, g, 0.01, 3, seed=1, it=1)
    pl.mala_step(th, t, g, 1e-4, seed=1, it=2)
    torch.cuda.synchronize()
    return pl.kernel
free0 = None
for rep in range(60):
    ks = [once([4, 32, 32, 3], torch.float32, 150, 256), once([4, 16, 3], torch.float64, 100, 64),
          once([10, 100, 10], torch.float32, 256, 128), once([2, 3, 2, 1], torch.float64, 50, 32),
          once([784, 128, 10], torch.float32, 64, 16)]
    gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if rep == 2: free0 = free
    if rep % 15 == 0 or rep == 59: print(rep, ks, f"free {free / 2**30:.3f} GiB")
print("leak since rep 2:", (free0 - free) / 2**20, "MiB")
