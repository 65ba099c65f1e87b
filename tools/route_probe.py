#!/usr/bin/env python3
"""Generic (one wave per chain, theta in LDS) against the layerwise batched-GEMM path on models that fit LDS but that
no fused kernel covers: where should the dispatch send them?  (The answer is ey_api.hip::prefer_large.)
usage: [EY_F64=1] route_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

dev = torch.device("cuda", 0)
CASES = [([20, 100, 100, 5], 512, 1024), ([10, 100, 10], 256, 2048), ([30, 80, 5], 1000, 1024),
         ([8, 40, 40, 40, 3], 300, 2048), ([50, 128, 10], 2000, 512), ([4, 70, 3], 150, 4096)]
CASES += [([10, 100, 10], 256, c) for c in (1, 4, 16, 64, 256)]      # how many chains before the GEMM grids pay off
CASES += [([30, 80, 5], 1000, c) for c in (1, 4, 16, 64)]
CASES += [([6, 24, 4], n, 1024) for n in (100, 400, 1600)]            # how much work per evaluation
CASES += [([12, 48, 6], n, 1024) for n in (100, 400)]
DT = torch.float64 if os.environ.get("EY_F64") else torch.float32
NP = np.float64 if os.environ.get("EY_F64") else np.float32
if os.environ.get("EY_F64"):
    CASES = [([10, 100, 10], 256, 512), ([12, 48, 6], 400, 1024), ([4, 70, 3], 150, 2048), ([6, 24, 4], 400, 1024),
             ([10, 100, 10], 256, 1)]
for dims, N, C in CASES:
    rng = np.random.default_rng(0)
    x = rng.standard_normal((N, dims[0])).astype(NP)
    y = np.eye(dims[-1], dtype=NP)[rng.integers(0, dims[-1], N)]
    K = len(dims) - 1
    res = {}
    for name, variant, flags in (("generic", 0, L.EY_FORCE_GENERIC), ("bgemm", 16, 0)):
        L.lib().ey_debug_set_variant(variant)
        pl = Plan(dims, [1] * K, [1] * (K - 1) + [0], 1, DT, dev)
        pl.set_data(torch.tensor(x, device=dev), torch.tensor(y, device=dev))
        pl.set_prior(torch.zeros(pl.P), torch.ones(pl.P))
        th = 0.1 * pl.philox_normal(C, seed=0, it=0)
        t, g = pl.log_target_grad(th)
        kern = pl.kernel
        for _ in range(2):
            pl.hmc_step(th, t, g, 0.005, 10, seed=1, it=1, flags=flags)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for i in range(n):
            pl.hmc_step(th, t, g, 0.005, 10, seed=1, it=2 + i, flags=flags)
        torch.cuda.synchronize()
        res[name] = (C * 10 * n / (time.perf_counter() - t0), kern)
    L.lib().ey_debug_set_variant(0)
    w = sum(dims[i] * dims[i + 1] for i in range(K))
    print(f"MLP({'-'.join(map(str, dims))}) {str(DT)[6:]} sum d_l d_l+1 = {w} N={N} C={C} P={pl.P}: generic (EY_FORCE_GENERIC; "
          f"the layerwise path where the model does not fit LDS) {res['generic'][0]:.3e}  bgemm {res['bgemm'][0]:.3e} "
          f"leapfrog-steps/s x chains -> x{res['bgemm'][0] / res['generic'][0]:.1f}; default route: {res['generic'][1]}")
