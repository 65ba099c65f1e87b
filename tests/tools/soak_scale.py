#!/usr/bin/env python3
"""Large row counts (up to 50 000) and chain counts (up to 100 000) on every kernel family: value and gradient against the f64
C oracle, one HMC draw finite.  usage: python tests/tools/soak_scale.py (under tests/: it uses the oracle as its checker)"""
import os, sys, numpy as np, torch, time
sys.path.insert(0, '.')
from oracle.c_oracle import COracle
from eeyore_amd.plan import Plan
DEV = torch.device('cuda', 0)
rng = np.random.default_rng(1)
for dims, acts, N, C, dtn in (([4, 16, 3], [1, 0], 20000, 8, 'f32'), ([4, 16, 3], [1, 0], 20000, 8, 'f64'),
                              ([10, 100, 10], [1, 0], 30000, 6, 'f32'), ([10, 100, 10], [2, 0], 9000, 6, 'f64'),
                              ([4, 3, 3], [1, 0], 50000, 4, 'f64'), ([4, 32, 32, 3], [1, 1, 0], 5000, 8, 'f32'),
                              ([784, 128, 10], [1, 0], 3000, 5, 'f32'), ([4, 16, 3], [1, 0], 1, 100000, 'f32'),
                              ([10, 100, 10], [1, 0], 1, 50000, 'f32')):
    npdt, dt = (np.float64, torch.float64) if dtn == 'f64' else (np.float32, torch.float32)
    nl = len(dims) - 1
    x = rng.standard_normal((N, dims[0])); y = np.eye(dims[-1])[rng.integers(0, dims[-1], N)]
    P = sum((dims[l] + 1) * dims[l + 1] for l in range(nl))
    pl = Plan(dims, [1] * nl, acts, 1, dt, DEV)
    t_ = lambda a: torch.tensor(np.asarray(a), dtype=dt, device=DEV).contiguous()
    pl.set_data(t_(x), t_(y)); pl.set_prior(torch.zeros(P), torch.ones(P))
    co = COracle(dims, acts, 1, x, y, np.zeros(P), np.ones(P), dtype=np.float64, nthreads=8)
    th0 = 0.05 * rng.standard_normal((C, P))
    t0 = time.perf_counter()
    t, g = pl.log_target_grad(t_(th0)); torch.cuda.synchronize()
    dt_eval = time.perf_counter() - t0
    errs = []
    for c in (0, C - 1):
        to, go, _, _ = co.log_target_grad(th0[c])
        errs.append((abs(t[c].item() - to) / max(1.0, abs(to)), np.abs(g[c].cpu().numpy() - go).max() / max(1.0, np.abs(go).max())))
    out = pl.hmc_step(t_(th0).clone(), t.clone(), g.clone(), 1e-4, 3, seed=1, it=1)
    torch.cuda.synchronize()
    tol = 1e-9 if dtn == 'f64' else 3e-4
    ok = all(e[0] < tol and e[1] < tol * 10 for e in errs) and torch.isfinite(out['h_prop']).all().item()
    print(dims, dtn, 'N', N, 'C', C, pl.kernel, 'rel err value/grad', [f"{e[0]:.1e}/{e[1]:.1e}" for e in errs], 'OK' if ok else 'FAIL')
