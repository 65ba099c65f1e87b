"""Diagnostic: every mode of the fused16 family on one model against the C oracle, error per mode (no early abort).
Used by tools/f16_bisect.sh with EEYORE_AMD_LIB pointing at a diagnostic build of the library.

  python tools/f16_check.py 13,29,4 1,1 0 f64 7      (dims, activations, likelihood, dtype, rows)
Prints one line per mode and a final PASS / FAIL line; exit code 0 / 1."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.c_oracle import COracle  # noqa: E402
from eeyore_amd.plan import Plan  # noqa: E402

DEV = "cuda:0"


def main():
    dims = [int(v) for v in sys.argv[1].split(",")]
    acts = [int(v) for v in sys.argv[2].split(",")]
    lik, tag, N = int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    npdt, dt = (np.float64, torch.float64) if tag == "f64" else (np.float32, torch.float32)
    tol = 1e-9 if tag == "f64" else 2e-3
    rng = np.random.default_rng(sum(dims) + N)
    x = rng.standard_normal((N, dims[0]))
    y = np.eye(dims[-1])[rng.integers(0, dims[-1], N)] if lik == 1 else (rng.random((N, dims[-1])) < 0.5).astype(np.float64)
    P = sum((dims[l] + 1) * dims[l + 1] for l in range(len(dims) - 1))
    mu, sigma = 0.1 * rng.standard_normal(P), 0.5 + rng.random(P)
    t_ = lambda a: torch.tensor(np.asarray(a), dtype=dt, device=DEV).contiguous()
    pl = Plan(dims, [1] * (len(dims) - 1), acts, lik, dt, DEV)
    pl.f32_products = "exact"
    pl.set_data(t_(x), t_(y))
    pl.set_prior(torch.tensor(mu), torch.tensor(sigma))
    co = COracle(dims, acts, lik, x, y, mu, sigma, dtype=np.float64, nthreads=4)
    C = 11
    th0 = ((0.3 if lik == 1 else 0.15) * rng.standard_normal((C, P))).astype(npdt)
    bad = []

    def report(name, err, scale=1.0):
        err = float(err)
        ok = err <= tol * scale
        print(f"  {name:34s} max err {err:.3e}  {'ok' if ok else 'WRONG'}")
        if not ok:
            bad.append(name)

    t, g = pl.log_target_grad(t_(th0))
    lk, pr = pl.log_target(t_(th0))
    ref = [co.log_target_grad(th0[c].astype(np.float64)) for c in range(C)]
    to = np.array([r[0] for r in ref]); go = np.stack([r[1] for r in ref])
    lo = np.array([r[2] for r in ref]); po = np.array([r[3] for r in ref])
    report("grad call: target", np.abs(t.cpu().numpy() - to).max(), 10)
    ge = np.abs(g.cpu().numpy() - go)
    report("grad call: gradient", ge.max(), 10)
    if ge.max() > tol * 10:
        w = np.argwhere(ge > tol * 10)
        print("    wrong gradient elements (chain, index):", w[:12].tolist(), "of", len(w))
    report("value call: lik", np.abs(lk.cpu().numpy() - lo).max(), 10)
    report("value call: prior", np.abs(pr.cpu().numpy() - po).max(), 10)
    if os.environ.get("F16_CHECK_QUICK"):
        print(("FAIL " + "; ".join(bad)) if bad else "PASS", f"kernel={pl.kernel}")
        return 1 if bad else 0
    p0 = rng.standard_normal((C, P)).astype(npdt); u = rng.random(C).astype(npdt)
    f8 = lambda a_: np.asarray(a_, dtype=np.float64).copy()
    eps, Ls = (0.02, 4) if lik == 1 else (0.01, 3)
    th, p = t_(th0).clone(), t_(p0).clone()
    tl, gl = pl.leapfrog(th, p, eps, Ls)
    e_th = e_p = e_t = 0.0
    for c in range(C):
        tho, po_, to_, go_ = co.leapfrog(f8(th0[c]), f8(p0[c]), eps, Ls)
        e_th = max(e_th, np.abs(th[c].cpu().numpy() - tho).max())
        e_p = max(e_p, np.abs(p[c].cpu().numpy() - po_).max())
        e_t = max(e_t, abs(tl[c].item() - to_))
    report("leapfrog: theta", e_th, 10); report("leapfrog: momentum", e_p, 50); report("leapfrog: target", e_t, 20)
    tv0 = f8(t.cpu().numpy()); g0 = f8(g.cpu().numpy())
    for flags in (0, 1):
        th, tv, gg = t_(th0).clone(), t.clone(), g.clone()
        out = pl.hmc_step(th, tv, gg, eps, 5, p0=t_(p0), u=t_(u), flags=flags)
        tho, tvo, go_ = f8(th0), to.copy(), go.copy()
        acc, hc, hp = co.hmc_draw(tho, tvo, go_, f8(p0), f8(u), eps, 5)
        report(f"hmc (flags {flags}): h_cur", np.abs(out["h_cur"].cpu().numpy() - hc).max(), 100)
        report(f"hmc (flags {flags}): h_prop", np.abs(out["h_prop"].cpu().numpy() - hp).max(), 100)
        same = out["accepted"].cpu().numpy() == acc
        report(f"hmc (flags {flags}): theta", np.abs(th.cpu().numpy()[same] - tho[same]).max() if same.any() else 0, 10)
    th, tv, gg = t_(th0).clone(), t.clone(), g.clone()
    out = pl.mala_step(th, tv, gg, 0.004, z=t_(p0), u=t_(u))
    acc, lr = co.mala_draw(f8(th0), to.copy(), go.copy(), f8(p0), f8(u), 0.004)
    report("mala: log_rate", (np.abs(out["log_rate"].cpu().numpy() - lr) / np.maximum(1, np.abs(lr))).max())
    th, tv = t_(th0).clone(), t.clone()
    out = pl.mh_step(th, tv, torch.full((P,), 0.02, dtype=dt), z=t_(p0), u=t_(u))
    acc, lr = co.mh_draw(f8(th0), to.copy(), f8(p0), f8(u), 0.02)
    report("mh: log_rate", (np.abs(out["log_rate"].cpu().numpy() - lr) / np.maximum(1, np.abs(lr))).max())
    print(("FAIL " + "; ".join(bad)) if bad else "PASS", f"kernel={pl.kernel}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
