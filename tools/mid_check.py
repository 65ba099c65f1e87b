#!/usr/bin/env python3
"""The fused mid-size kernel (ey_mid.hip) (variant bit 13) against the layerwise path and the f64 oracle: value and gradient by
parameter block.   usage: python tools/mid_check.py d0,h1[,h2],dK [rows] [lik 0|1] [acts e.g. 1,1,0] [bias e.g. 1,1,1]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.plan import Plan  # noqa: E402
from oracle.c_oracle import COracle  # noqa: E402

dims = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "20,100,100,5").split(",")]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 70
lik = int(sys.argv[3]) if len(sys.argv) > 3 else 1
nl = len(dims) - 1
acts = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1] * (nl - 1) + [0 if lik == 1 else 1]
bias = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [1] * nl
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
x = rng.standard_normal((N, dims[0])).astype(np.float32)
if lik == 1:
    y = np.eye(dims[-1], dtype=np.float32)[rng.integers(0, dims[-1], N)]
else:
    y = (rng.random((N, dims[-1])) < 0.5).astype(np.float32)
from eeyore_amd import _lib as L
L.lib().ey_debug_set_variant(16)
pl = Plan(dims, bias, acts, lik, torch.float32, dev)
L.lib().ey_debug_set_variant(0)
pl.set_data(torch.tensor(x, device=dev), torch.tensor(y, device=dev))
pl.set_prior(torch.zeros(pl.P), torch.full((pl.P,), 2.0))
co = COracle(dims, acts, lik, x.astype(np.float64), y, 0.0, 2.0, dtype=np.float64, nthreads=4, bias=bias) if "bias" in COracle.__init__.__code__.co_varnames else COracle(dims, acts, lik, x.astype(np.float64), y, 0.0, 2.0, dtype=np.float64, nthreads=4)
C = 6
th = (0.3 * pl.philox_normal(C, seed=3, it=0)).contiguous()
res = {}
REF, FUSED = int(os.environ.get('MID_REF', '0')), int(os.environ.get('MID_FUSED', '8192'))
for v in (REF, FUSED):
    pl.set_variant(16 | v)
    t, g = pl.log_target_grad(th)
    torch.cuda.synchronize()
    res[v] = (t.cpu().numpy(), g.cpu().numpy())
pl.set_variant(16)
blocks, at = [], 0
for l in range(nl):
    blocks.append((f"W{l}", at, at + dims[l] * dims[l + 1])); at += dims[l] * dims[l + 1]
    if bias[l]:
        blocks.append((f"b{l}", at, at + dims[l + 1])); at += dims[l + 1]
print(f"kernel {pl.kernel}  dims {dims} rows {N} P {pl.P}")
for c in range(C):
    tt, gg, _, _ = co.log_target_grad(th[c].cpu().numpy().astype(np.float64))
    line = f"chain {c}: target oracle {tt:.4f} layerwise {res[REF][0][c]:.4f} mid {res[FUSED][0][c]:.4f} |"
    for n, a, b in blocks:
        sc = max(1e-6, np.abs(gg[a:b]).max())
        line += f" {n}: {np.abs(res[REF][1][c][a:b] - gg[a:b]).max() / sc:.1e}/{np.abs(res[FUSED][1][c][a:b] - gg[a:b]).max() / sc:.1e}"
    print(line)
