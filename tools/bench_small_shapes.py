#!/usr/bin/env python3
"""The small fused16 shapes next to the headline model (f32, hidden widths <= 16, one or two hidden layers): HMC L = 20, 4096
chains, N = 150, five iterations per launch.   [EEYORE_AMD_LIB=tools/abl/lib_x.so] python tools/bench_small_shapes.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from eeyore_amd.datasets import synthetic
from eeyore_amd.plan import Plan
dev = torch.device('cuda', 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
import os
SHAPES = ([4, 16, 16, 3], [4, 12, 10, 3], [4, 16, 3], [4, 8, 8, 3]) if not os.environ.get('EY_SHAPES32') else ([4, 20, 20, 3], [4, 32, 3], [4, 24, 32, 3], [4, 32, 32, 3])
for dims in SHAPES:
    K = len(dims) - 1
    tdt = torch.float64 if os.environ.get('EY_F64') else torch.float32
    pl = Plan(dims, [1] * K, [1] * (K - 1) + [0], 1, tdt, dev)
    if tdt == torch.float32: pl.f32_products = 'exact'   # (the 4-32-32 model itself then runs on fused16 as well)
    pl.set_data(torch.tensor(xs, dtype=tdt, device=dev), torch.tensor(ys, dtype=tdt, device=dev))
    pl.set_prior(torch.zeros(pl.P), torch.full((pl.P,), float(np.sqrt(3.0))))
    C = 4096
    th = 0.1 * pl.philox_normal(C, seed=0, it=0)
    t, g = pl.log_target_grad(th)
    pl.hmc_run(th, t, g, 0.02, 20, 5, seed=3, it=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(6):
        pl.hmc_run(th, t, g, 0.02, 20, 5, seed=3, it=100 + 5 * i)
    torch.cuda.synchronize()
    r = C * 20 * 30 / (time.perf_counter() - t0)
    prods = [dims[i] * dims[i + 1] for i in range(K)]
    fl = 2 * 150 * (2 * sum(prods) + sum(prods[1:])) + 6 * pl.P
    print(f"MLP({'-'.join(map(str, dims))}) {str(tdt)[6:]} kernel {pl.kernel}: {r:.3e} leapfrog-steps/s x chains = {fl * r / 1e12:.1f} TFLOP/s  checksum {float(th.double().sum()):.6f}")
