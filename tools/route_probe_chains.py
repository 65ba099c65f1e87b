"""Generic kernel (forced) against the layerwise path on mid-size models over chain counts and dtypes (the second part of
profiles/r02_route_probe.txt): the crossover that ey_api.hip::prefer_large encodes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L
from eeyore_amd.plan import Plan
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for dt in (torch.float64, torch.float32):
    for dims in ([20, 12, 1], [4, 20, 10], [30, 10, 2], [6, 24, 4], [10, 30, 10], [4, 70, 3], [20, 30, 5], [12, 48, 6]):
        lik = 0 if dims[-1] == 1 else 1
        N = 512
        x = rng.uniform(0, 1, (N, dims[0]))
        y = np.eye(dims[-1])[rng.integers(0, dims[-1], N)] if lik == 1 else (rng.uniform(0, 1, (N, 1)) > 0.5).astype(np.float64)
        line = f"{str(dt)[6:]:8s} {str(dims):14s} sum {sum(a * b for a, b in zip(dims[:-1], dims[1:])):5d}"
        for C in (64, 1024, 16384):
            r = []
            for variant, flags in ((0, L.EY_FORCE_GENERIC), (16, 0)):
                L.lib().ey_debug_set_variant(variant)
                plan = Plan(dims, [1] * (len(dims) - 1), [1] * (len(dims) - 2) + [1 if lik == 0 else 0], lik, dt, dev)
                plan.set_data(torch.tensor(x, dtype=dt, device=dev), torch.tensor(y, dtype=dt, device=dev))
                plan.set_prior(torch.zeros(plan.P), torch.full((plan.P,), 3.0))
                th = 0.3 * plan.philox_normal(C, seed=1, it=0)
                t, g = plan.log_target_grad(th)
                for _ in range(2): plan.hmc_step(th, t, g, 0.005, 5, seed=1, it=1, flags=flags)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(4): plan.hmc_step(th, t, g, 0.005, 5, seed=1, it=2 + i, flags=flags)
                torch.cuda.synchronize()
                r.append((time.perf_counter() - t0) / 4 * 1e3)
            L.lib().ey_debug_set_variant(0)
            line += f" | C={C}: generic {r[0]:8.2f} ms bgemm {r[1]:8.2f} ms x{r[0] / r[1]:.2f}"
        print(line, flush=True)
