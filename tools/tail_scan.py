#!/usr/bin/env python3
"""Evaluation time of the layerwise path on MLP(10-100-10) against the number of data rows (2048 chains): under a kernel
trace this separates the fused last-layer kernel's per-chain overhead (~10 us per round of workgroups) from its cost per
16-row pass (~2.8 us).  usage: rocprofv3 --kernel-trace --stats -- python3 tools/tail_scan.py"""
import sys, numpy as np, torch, time
sys.path.insert(0, '.')
from eeyore_amd.plan import Plan
dev = torch.device('cuda', 0)
dims, C = [10, 100, 10], 2048
for N in (16, 64, 256, 1024):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((N, dims[0])).astype(np.float32)
    y = np.eye(dims[-1], dtype=np.float32)[rng.integers(0, dims[-1], N)]
    pl = Plan(dims, [1, 1], [1, 0], 1, torch.float32, dev)
    pl.set_data(torch.tensor(x, device=dev), torch.tensor(y, device=dev))
    pl.set_prior(torch.zeros(pl.P), torch.ones(pl.P))
    th = 0.1 * pl.philox_normal(C, seed=0, it=0)
    for i in range(3): t, g = pl.log_target_grad(th)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): t, g = pl.log_target_grad(th)
    torch.cuda.synchronize()
    print(N, pl.kernel, f"{(time.perf_counter() - t0) / 20 * 1e6:.1f} us per evaluation of {C} chains")
