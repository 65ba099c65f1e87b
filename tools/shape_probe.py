#!/usr/bin/env python3
"""HMC draws of a few shapes on every kernel family that serves them: default routing, the layerwise path forced (variant 16),
the fused mid-size kernel (16 + 8192) and the generic kernels (EY_FORCE_GENERIC).  usage: tools/shape_probe.py [dims:N:C ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

specs = sys.argv[1:] or ["16,32,32,32,3:150:4096", "64,32,32,10:150:4096", "16,64,64,64,3:256:2048"]
dev = torch.device("cuda", 0)
for spec in specs:
    ds, N, C = spec.split(":")
    dims, N, C = [int(v) for v in ds.split(",")], int(N), int(C)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((N, dims[0])).astype(np.float32)
    y = np.eye(dims[-1], dtype=np.float32)[rng.integers(0, dims[-1], N)]
    K = len(dims) - 1
    prods = [dims[i] * dims[i + 1] for i in range(K)]
    Lf = 10
    for name, dv, var, flags, Cc in (("default", 0, 0, 0, C), ("layerwise", 16, 16, 0, C), ("layerwise only", 16, 16 + 16384, 0, C), ("fused mid", 16, 16 + 8192, 0, C),
                                     ("generic", 0, 0, L.EY_FORCE_GENERIC, min(C, 1024))):
        L.lib().ey_debug_set_variant(dv)
        try:
            pl = Plan(dims, [1] * K, [1] * (K - 1) + [0], 1, torch.float32, dev)
        finally:
            L.lib().ey_debug_set_variant(0)
        pl.set_data(torch.tensor(x, device=dev), torch.tensor(y, device=dev))
        pl.set_prior(torch.zeros(pl.P), torch.ones(pl.P))
        pl.set_variant(var)
        th = 0.1 * pl.philox_normal(Cc, seed=0, it=0)
        t, g = pl.log_target_grad(th)
        try:
            pl.hmc_step(th, t, g, 0.005, Lf, seed=1, it=1, flags=flags)
        except Exception as e:  # noqa: BLE001
            print(f"MLP({'-'.join(map(str, dims))}) {name}: {str(e)[:80]}")
            continue
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(3):
            pl.hmc_step(th, t, g, 0.005, Lf, seed=1, it=2 + i, flags=flags)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        rate = Cc * Lf / (ms * 1e-3)
        fl = (2 * N * (2 * sum(prods) + sum(prods[1:])) + 6 * pl.P) * rate
        print(f"MLP({'-'.join(map(str, dims))}) N={N} {name:10s} kernel {pl.kernel:8s} chains {Cc}: {ms:8.3f} ms per draw (L={Lf}) -> {rate:.3e} leapfrog-steps/s x chains, "
              f"{fl / 1e12:6.2f} TFLOP/s")
