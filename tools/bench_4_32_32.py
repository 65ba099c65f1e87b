# rates of the 4-32-32 models on both kernels (mfma32 bf16x3 / fused16 exact)
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from eeyore_amd.datasets import synthetic
from eeyore_amd.plan import Plan
dev = torch.device("cuda", 0)
xs, ys = synthetic.iris_shaped_arrays(seed=0)
def f_step(dims, n_rows):
    prods = [dims[i] * dims[i + 1] for i in range(len(dims) - 1)]
    P = sum((dims[i] + 1) * dims[i + 1] for i in range(len(dims) - 1))
    return 2 * n_rows * (2 * sum(prods) + sum(prods[1:])) + 6 * P
for dims, acts, lik in (([4,32,32,3],[1,1,0],1), ([4,32,32,3],[2,2,0],1), ([4,32,32,3],[3,3,0],1), ([4,32,32,1],[1,1,1],0), ([4,32,32,1],[2,2,1],0)):
    y = ys if lik == 1 else ys[:, :1]
    line = f"MLP({'-'.join(map(str,dims))}) acts {acts} {'CE' if lik else 'BCE'}:"
    for prod in ("bf16x3", "exact"):
        pl = Plan(dims, [1,1,1], acts, lik, torch.float32, dev)
        pl.f32_products = prod
        pl.set_data(torch.tensor(xs, dtype=torch.float32, device=dev), torch.tensor(y, dtype=torch.float32, device=dev))
        pl.set_prior(torch.zeros(pl.P), torch.full((pl.P,), float(np.sqrt(3.0))))
        C = 4096
        th = 0.1 * pl.philox_normal(C, seed=0, it=0)
        t, g = pl.log_target_grad(th)
        for i in range(3): pl.hmc_run(th, t, g, 0.02, 20, 5, seed=3, it=1 + 5 * i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4): pl.hmc_run(th, t, g, 0.02, 20, 5, seed=3, it=100 + 5 * i)
        torch.cuda.synchronize()
        r = C * 20 * 20 / (time.perf_counter() - t0)
        fl = f_step(dims, 150) * r / 1e12
        line += f"  {prod} ({pl.kernel}) {r:.3e} steps/s x chains = {fl:.1f} TFLOP/s ({100*fl/157.3:.1f} %)"
    print(line)
