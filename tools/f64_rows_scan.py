#!/usr/bin/env python3
"""The f64 headline model on k_fused16 over the number of rows (16-row tiles): time per leapfrog step = per-evaluation
overhead + tiles x time per tile (linear fit), 4096 chains, HMC L = 20, five iterations per launch."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from eeyore_amd.plan import Plan
dev = torch.device('cuda', 0)
dims, tdt = [4, 32, 32, 3], (torch.float32 if len(sys.argv) > 1 and sys.argv[1] == 'f32' else torch.float64)
res = []
for N in (16, 80, 160, 320, 640):
    rng = np.random.default_rng(1)
    xd = rng.standard_normal((N, 4)); yd = np.eye(3)[rng.integers(0, 3, N)]
    pl = Plan(dims, [1, 1, 1], [1, 1, 0], 1, tdt, dev)
    if tdt == torch.float32: pl.set_variant(16)  # keep the f32 run on fused16 too
    pl.set_data(torch.tensor(xd, dtype=tdt, device=dev), torch.tensor(yd, dtype=tdt, device=dev))
    pl.set_prior(torch.zeros(pl.P), torch.full((pl.P,), float(np.sqrt(3.0))))
    C = 4096
    th = 0.05 * pl.philox_normal(C, seed=0, it=0)
    t, g = pl.log_target_grad(th)
    pl.hmc_run(th, t, g, 0.005, 20, 5, seed=3, it=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        pl.hmc_run(th, t, g, 0.005, 20, 5, seed=3, it=100 + 5 * i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (3 * 5 * 20)   # seconds per leapfrog step of 4096 chains
    per_wave_us = dt * 1e6 / (C / 1024.0)            # 1024 waves resident (one per SIMD): a wave's time per evaluation
    res.append((N // 16, per_wave_us))
    print(f"N {N:4d} ({N // 16:2d} tiles) kernel {pl.kernel}: {per_wave_us:8.2f} us per evaluation and wave")
t = np.array([r[0] for r in res], float); y = np.array([r[1] for r in res])
A = np.vstack([t, np.ones_like(t)]).T
slope, icpt = np.linalg.lstsq(A, y, rcond=None)[0]
print(f"fit: {slope:.3f} us per 16-row tile ({slope * 2400:.0f} cycles at 2.4 GHz) + {icpt:.2f} us per evaluation ({icpt * 2400:.0f} cycles)")
