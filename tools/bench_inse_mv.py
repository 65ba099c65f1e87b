#!/usr/bin/env python3
"""Time ey_inse_multivariate (the reference's multivariate initial-sequence estimator for every chain at once) on stored
runs of small models: [n, C, p] f32 AR(1) chains with autocorrelation rho.  python tools/bench_inse_mv.py [rho]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.stats import batched  # noqa: E402

rho = float(sys.argv[1]) if len(sys.argv) > 1 else 0.7
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for n, C, p, what in ((1000, 256, 9, "MLP(2-2-1)"), (1000, 256, 20, "config 2's MLP(2-3-2-1)"), (1000, 4096, 20, ""),
                      (1000, 1024, 27, "MLP(4-3-3)"), (1000, 1024, 64, ""), (200, 4096, 64, "")):
    x = torch.empty(n, C, p, dtype=torch.float32, device=dev)
    x[0].normal_()
    for i in range(1, n):
        torch.randn(C, p, out=x[i], device=dev)
        x[i].add_(x[i - 1], alpha=rho)
    torch.cuda.synchronize()
    batched.inse_multivariate(x[:, :2].contiguous())  # warm
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = batched.inse_multivariate(x)
    b.record()
    torch.cuda.synchronize()
    pairs = r["pairs"].double()
    ess = batched.multi_ess_device(x)
    print(f"n {n} x {C} chains x p {p:2d} {what:24s}: {a.elapsed_time(b):8.2f} ms; lag pairs mean {pairs.mean().item():.1f} max "
          f"{int(pairs.max().item())}, chains without enough samples {(pairs < 0).sum().item()}; multi-ESS mean "
          f"{ess[~ess.isnan()].mean().item():.1f} (one-parameter theory {n * (1 - rho) / (1 + rho):.1f})")
