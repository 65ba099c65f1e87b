#!/usr/bin/env python3
"""k_fused16 on the headline model in f64 (the reference's default dtype) and on MLP(4-64-64-3) in f32: 4096 chains,
N = 150, HMC L = 20, launches of five iterations (the workload tools/profile_fused16.sh profiles)."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from eeyore_amd.datasets import synthetic
from eeyore_amd.plan import Plan
dev = torch.device('cuda', 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
CASES = [([4, 32, 32, 3], torch.float64), ([4, 64, 64, 3], torch.float32)]
if len(sys.argv) > 1 and sys.argv[1] == '--wide':  # ten-class heads (the padded forms' whole delta2 tile)
    CASES = [([8, 32, 32, 10], torch.float32), ([8, 32, 32, 10], torch.float64), ([8, 64, 64, 10], torch.float32)]
for dims, tdt in CASES:
    pl = Plan(dims, [1, 1, 1], [1, 1, 0], 1, tdt, dev)
    rng = np.random.default_rng(1)
    xd = rng.standard_normal((150, dims[0])) if dims[0] != 4 else xs
    yd = np.eye(dims[-1])[rng.integers(0, dims[-1], 150)] if dims[-1] != 3 else ys
    pl.set_data(torch.tensor(xd, dtype=tdt, device=dev), torch.tensor(yd, dtype=tdt, device=dev))
    pl.set_prior(torch.zeros(pl.P), torch.full((pl.P,), float(np.sqrt(3.0))))
    C = 4096
    th = 0.1 * pl.philox_normal(C, seed=0, it=0)
    t, g = pl.log_target_grad(th)
    pl.hmc_run(th, t, g, 0.02, 20, 5, seed=3, it=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4):
        pl.hmc_run(th, t, g, 0.02, 20, 5, seed=3, it=100 + 5 * i)
    torch.cuda.synchronize()
    r = C * 20 * 20 / (time.perf_counter() - t0)
    prods = [dims[i] * dims[i + 1] for i in range(3)]
    fl = 2 * 150 * (2 * sum(prods) + sum(prods[1:])) + 6 * pl.P
    print(f"MLP({'-'.join(map(str, dims))}) {str(tdt)[6:]} kernel {pl.kernel}: {r:.3e} leapfrog-steps/s x chains = {fl * r / 1e12:.1f} TFLOP/s")
