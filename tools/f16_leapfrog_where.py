#!/usr/bin/env python3
"""Diagnostic: WHICH parameters of a fused16 leapfrog / gradient differ from the oracle (indices per chain).
   EEYORE_AMD_LIB=... python tools/f16_leapfrog_where.py 16,24,12 2,0 1 f32 20"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.c_oracle import COracle
from eeyore_amd.plan import Plan
DEV = "cuda:0"
dims = [int(v) for v in sys.argv[1].split(",")]; acts = [int(v) for v in sys.argv[2].split(",")]
lik, tag, N = int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
npdt, dt = (np.float64, torch.float64) if tag == "f64" else (np.float32, torch.float32)
rng = np.random.default_rng(sum(dims) + N)
x = rng.standard_normal((N, dims[0]))
y = np.eye(dims[-1])[rng.integers(0, dims[-1], N)] if lik == 1 else (rng.random((N, dims[-1])) < 0.5).astype(np.float64)
P = sum((dims[l] + 1) * dims[l + 1] for l in range(len(dims) - 1))
mu, sigma = 0.1 * rng.standard_normal(P), 0.5 + rng.random(P)
t_ = lambda a: torch.tensor(np.asarray(a), dtype=dt, device=DEV).contiguous()
pl = Plan(dims, [1] * (len(dims) - 1), acts, lik, dt, DEV); pl.f32_products = os.environ.get("F32_PRODUCTS", "exact")
pl.set_data(t_(x), t_(y)); pl.set_prior(torch.tensor(mu), torch.tensor(sigma))
co = COracle(dims, acts, lik, x, y, mu, sigma, dtype=np.float64, nthreads=4)
C = 6
th0 = (0.3 * rng.standard_normal((C, P))).astype(npdt); p0 = rng.standard_normal((C, P)).astype(npdt)
t, g = pl.log_target_grad(t_(th0))
for c in range(C):
    to, go, _, _ = co.log_target_grad(th0[c].astype(np.float64))
    bad = np.nonzero(np.abs(g[c].cpu().numpy() - go) > 2e-2 * max(1, np.abs(go).max()))[0]
    print(f"gradient chain {c}: target err {abs(t[c].item() - to):.2e}, wrong indices {bad.tolist()[:24]}")
for Ls in (1, 4):
    th, p = t_(th0).clone(), t_(p0).clone()
    tl, gl = pl.leapfrog(th, p, 0.02, Ls)
    for c in range(C):
        tho, po_, to_, go_ = co.leapfrog(th0[c].astype(np.float64), p0[c].astype(np.float64), 0.02, Ls)
        bad = np.nonzero(np.abs(th[c].cpu().numpy() - tho) > 2e-3)[0]
        badp = np.nonzero(np.abs(p[c].cpu().numpy() - po_) > 2e-2)[0]
        badg = np.nonzero(np.abs(gl[c].cpu().numpy() - go_) > 2e-2 * max(1, np.abs(go_).max()))[0]
        if len(badp) and c < 2:
            i = badp[0]
            print(f"   index {i}: theta0 {th0[c, i]:.6f} -> {th[c, i].item():.6f} (oracle {tho[i]:.6f}); momentum {p0[c, i]:.6f} -> {p[c, i].item():.6f} "
                  f"(oracle {po_[i]:.6f}); gradient at the start {g[c, i].item():.6f}, at the end {gl[c, i].item():.6f} (oracle {go_[i]:.6f})")
        print(f"leapfrog L={Ls} chain {c}: wrong theta {bad.tolist()[:12]} momentum {badp.tolist()[:12]} end gradient {badg.tolist()[:12]} target err {abs(tl[c].item() - to_):.2e}")
print("layout:", " ".join(f"W{l}[{sum((dims[j]+1)*dims[j+1] for j in range(l))}..) b{l}[{sum((dims[j]+1)*dims[j+1] for j in range(l)) + dims[l]*dims[l+1]}..)" for l in range(len(dims)-1)))
