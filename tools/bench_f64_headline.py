#!/usr/bin/env python3
"""The headline model in f64 (the reference's default dtype) on k_fused16: HMC L = 20, 4096 chains, MLP(4-32-32-3), N = 150,
five iterations per launch; prints TFLOP/s and a checksum of the final state (A/B of builds: EEYORE_AMD_LIB=... python
tools/bench_f64_headline.py [dims])."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd.datasets import synthetic
from eeyore_amd.plan import Plan
dev = torch.device('cuda', 0)
dims = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4, 32, 32, 3]
tdt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
xs, ys = synthetic.iris_shaped_arrays(seed=0)
pl = Plan(dims, [1] * (len(dims) - 1), [1] * (len(dims) - 2) + [0], 1, tdt, dev)
pl.set_data(torch.tensor(xs, dtype=tdt, device=dev), torch.tensor(ys, dtype=tdt, device=dev))
if tdt == torch.float32: pl.f32_products = "exact"
pl.set_prior(torch.zeros(pl.P), torch.full((pl.P,), float(np.sqrt(3.0))))
C = 4096
th = 0.1 * pl.philox_normal(C, seed=0, it=0)
t, g = pl.log_target_grad(th)
pl.hmc_run(th, t, g, 0.02, 20, 5, seed=3, it=1)
torch.cuda.synchronize()
best = 0.0
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(4): pl.hmc_run(th, t, g, 0.02, 20, 5, seed=3, it=100 + 20 * rep + 5 * i)
    torch.cuda.synchronize()
    best = max(best, C * 20 * 20 / (time.perf_counter() - t0))
prods = [dims[i] * dims[i + 1] for i in range(len(dims) - 1)]
fl = 2 * 150 * (2 * sum(prods) + sum(prods[1:])) + 6 * pl.P
print(f"{os.environ.get('EEYORE_AMD_LIB', 'shipped'):40s} MLP({'-'.join(map(str, dims))}) {str(tdt)[6:]} kernel {pl.kernel}: {best:.3e} leapfrog-steps/s x chains = "
      f"{fl * best / 1e12:.2f} TFLOP/s; checksum theta {th.double().sum().item():.15e} target {t.double().sum().item():.15e}")
