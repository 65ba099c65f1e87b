#!/usr/bin/env python3
"""A/B of fused16 builds: the f64 headline model, MLP(4-64-64-3) f32, MLP(4-16-16-3) f32 (HMC L = 20, 4096 chains, N = 150,
five iterations per launch) and the MALA check of the padded H = 16 shape.  EEYORE_AMD_LIB=... python tools/bench_f16_ab.py"""
import sys, time, os, subprocess, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.datasets import synthetic
from eeyore_amd.plan import Plan
dev = torch.device('cuda', 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
out = []
for dims, tdt in (([4, 32, 32, 3], torch.float64), ([4, 64, 64, 3], torch.float32), ([4, 16, 16, 3], torch.float32), ([4, 20, 20, 3], torch.float32)):
    pl = Plan(dims, [1, 1, 1], [1, 1, 0], 1, tdt, dev)
    if tdt == torch.float32: pl.f32_products = 'exact'
    pl.set_data(torch.tensor(xs, dtype=tdt, device=dev), torch.tensor(ys, dtype=tdt, device=dev))
    pl.set_prior(torch.zeros(pl.P), torch.full((pl.P,), float(np.sqrt(3.0))))
    C = 4096
    th = 0.1 * pl.philox_normal(C, seed=0, it=0)
    t, g = pl.log_target_grad(th)
    pl.hmc_run(th, t, g, 0.02, 20, 5, seed=3, it=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4): pl.hmc_run(th, t, g, 0.02, 20, 5, seed=3, it=100 + 5 * i)
    torch.cuda.synchronize()
    r = C * 20 * 20 / (time.perf_counter() - t0)
    prods = [dims[i] * dims[i + 1] for i in range(3)]
    fl = 2 * 150 * (2 * sum(prods) + sum(prods[1:])) + 6 * pl.P
    out.append(f"{'-'.join(map(str, dims))} {str(tdt)[6:]} {fl * r / 1e12:.1f}")
chk = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "f16_check.py"), "4,10,7,3", "1,1,0", "1", "f32", "150"],
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(" | ".join(out), "| padded MALA check:", chk.stdout.strip().split("\n")[-1][:40])
