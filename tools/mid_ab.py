#!/usr/bin/env python3
"""Value + gradient evaluations of mid-size models: the fused workgroup kernel (ey_mid.hip) against the layerwise path
(variant bit 13 selects the fused kernel), same process, interleaved rounds.  usage: python tools/mid_ab.py [dims:N:C ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.plan import Plan  # noqa: E402

specs = sys.argv[1:] or ["20,100,100,5:512:1024", "10,100,10:256:2048", "64,100,100,10:512:1024", "30,64,64,4:512:2048", "16,128,128,8:1024:512"]
dev = torch.device("cuda", 0)
for spec in specs:
    ds, N, C = spec.split(":")
    dims, N, C = [int(v) for v in ds.split(",")], int(N), int(C)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((N, dims[0])).astype(np.float32)
    y = np.eye(dims[-1], dtype=np.float32)[rng.integers(0, dims[-1], N)]
    K = len(dims) - 1
    pl = Plan(dims, [1] * K, [1] * (K - 1) + [0], 1, torch.float32, dev)
    pl.set_data(torch.tensor(x, device=dev), torch.tensor(y, device=dev))
    pl.set_prior(torch.zeros(pl.P), torch.ones(pl.P))
    th = 0.1 * pl.philox_normal(C, seed=0, it=0)
    prods = [dims[i] * dims[i + 1] for i in range(K)]
    flops = 2 * N * (2 * sum(prods) + sum(prods[1:])) * C
    times = {0: [], 8192: []}  # 0 = layerwise (default), 8192 = fused
    for rnd in range(4):
        for v in (8192, 0):
            pl.set_variant(v)
            pl.log_target_grad(th)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                pl.log_target_grad(th)
            b.record()
            torch.cuda.synchronize()
            times[v].append(a.elapsed_time(b) / 5)
    pl.set_variant(0)
    lw, md = np.median(times[0]), np.median(times[8192])
    print(f"MLP({'-'.join(map(str, dims))}) N={N} chains={C}: layerwise {lw * 1e3:8.1f} us ({flops / lw / 1e9:6.1f} TFLOP/s, {flops / lw / 1e9 / 157.3:.3f})   "
          f"fused {md * 1e3:8.1f} us ({flops / md / 1e9:6.1f} TFLOP/s, {flops / md / 1e9 / 157.3:.3f})   ratio {lw / md:.2f}")
