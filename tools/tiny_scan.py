"""The register-resident evaluation of tiny models against the LDS tile loop of the generic kernels (ey_generic.hip):
value + gradient of C chains at once, over batch rows, chain counts and dtypes.  The numbers behind use_tiny()."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeyore_amd import _lib as L
from eeyore_amd.plan import Plan
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for dt in (torch.float32, torch.float64):
    for dims, acts, lik in (([2, 3, 2, 1], [1, 1, 1], 0), ([4, 3, 3], [1, 0], 1), ([2, 2, 1], [1, 1], 0)):
        for N in (4, 64, 256, 1024):
            x = rng.uniform(0, 1, (N, dims[0]))
            if lik == 0:
                y = (rng.uniform(0, 1, (N, 1)) > 0.5).astype(np.float64)
            else:
                y = np.eye(dims[-1])[rng.integers(0, dims[-1], N)]
            plan = Plan(dims, [1] * (len(dims) - 1), acts, lik, dt, dev)
            plan.set_data(torch.tensor(x, dtype=dt, device=dev), torch.tensor(y, dtype=dt, device=dev))
            plan.set_prior(torch.zeros(plan.P), torch.full((plan.P,), 3.0))
            line = f"{str(dt)[6:]:8s} {str(dims):14s} N={N:5d}"
            for C in (256, 16384):
                th = 0.5 * plan.philox_normal(C, seed=1, it=0)
                res = []
                for variant in (512, 256):  # bit 9: the register-resident evaluation whenever the model qualifies, bit 8: never
                    plan.set_variant(variant)
                    for _ in range(3): plan.log_target_grad(th)
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(20): plan.log_target_grad(th)
                    b.record(); torch.cuda.synchronize()
                    res.append(a.elapsed_time(b) / 20 * 1e3)
                plan.set_variant(0)
                line += f" | C={C}: tiny {res[0]:8.1f} us  lds-loop {res[1]:8.1f} us  x{res[1] / res[0]:.2f}"
            print(line, flush=True)
