#!/usr/bin/env python3
"""Time ey_inse_univariate on a stored run the size of BASELINE config 3/4 per GPU: n iterations x 4096 chains x 1315
parameters (f32), AR(1) series with autocorrelation rho.   python tools/bench_inse.py [n] [chains] [rho]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.stats import batched  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rho = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
P = 1315
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.empty(n, C, P, dtype=torch.float32, device=dev)
x[0].normal_()
for i in range(1, n):
    torch.randn(C, P, out=x[i], device=dev)
    x[i].add_(x[i - 1], alpha=rho)
torch.cuda.synchronize()
batched.inse_univariate(x[:, :64])  # warm
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
r = batched.inse_univariate(x)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b)
pairs = r["pairs"].double()
S = C * P
macs = (pairs.clamp(min=0) + 1).sum().item() * 2 * n  # every series also computes the pair on which it stops
ess = n * r["var"] / r["sig2"]
print(f"n {n} x {C} chains x {P} parameters = {S:.3e} series, {x.numel() * 4 / 1e9:.1f} GB: {ms:.1f} ms "
      f"= {S / ms * 1e3:.3e} series/s, {x.numel() * 4 / ms / 1e6:.0f} GB/s of samples read, "
      f"{2 * macs / ms / 1e9:.1f} TFLOP/s (f64 multiply-adds from LDS); lag pairs mean {pairs.mean().item():.1f} "
      f"max {int(pairs.max().item())}; ESS mean {ess.mean().item():.1f} (theory {n * (1 - rho) / (1 + rho):.1f})")
