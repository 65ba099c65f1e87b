"""The fused 16x16x4 matrix-core kernels (ey_fused16.hip: d0-H-H-dK, H in {16, 32, 64}, f32 and f64, CE or BCE, sigmoid /
tanh / relu) through the C ABI against the C oracle, and against the generic kernels on the same inputs.
f64 within 1e-10 relative (the reference's default dtype, eeyore/models/model.py:7), f32 within the stated 2e-4."""
import os

import numpy as np
import pytest
import torch

from oracle.c_oracle import COracle
from tests.helpers import groups, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # dims, hidden/out activations, likelihood, dtype, N
    ([4, 16, 16, 3], [1, 1, 0], 1, "f32", 150),
    ([4, 16, 16, 3], [1, 1, 0], 1, "f64", 77),
    ([4, 32, 32, 3], [1, 1, 0], 1, "f64", 150),   # the headline model in the reference's default dtype
    ([4, 32, 32, 3], [2, 2, 0], 1, "f32", 150),   # tanh: served by ey_mfma32.hip too (bf16x3 form), see _expand
    ([4, 32, 32, 3], [3, 3, 0], 1, "f32", 77),    # relu, likewise
    ([4, 32, 32, 1], [1, 1, 1], 0, "f32", 150),   # the BCE head on 4-32-32, likewise
    ([4, 32, 32, 1], [2, 2, 1], 0, "f32", 45),
    ([4, 32, 32, 1], [3, 3, 1], 0, "f32", 90),
    ([8, 64, 64, 4], [2, 1, 0], 1, "f32", 33),    # widest: d0 = 8 (two k-steps), dK = 4, three waves per CU
    ([2, 16, 16, 1], [1, 1, 1], 0, "f64", 4),     # BCE-sum on a sigmoid output, as the reference's own tests
    ([5, 32, 32, 2], [3, 2, 1], 0, "f64", 49),    # relu, tanh, two BCE outputs
    ([3, 64, 64, 2], [1, 2, 1], 0, "f32", 130),
    ([1, 16, 16, 2], [1, 1, 0], 1, "f32", 1),     # one input, one row
    # hidden widths off the tile grid and unequal: zero-padded to the next of 16 / 32 / 64 (sigmoid(0) = 0.5 in a padded
    # unit must reach nothing)
    ([4, 20, 20, 3], [1, 1, 0], 1, "f64", 60),
    ([4, 10, 7, 3], [1, 1, 0], 1, "f32", 150),
    ([5, 50, 30, 2], [1, 2, 0], 1, "f32", 77),
    ([3, 7, 12, 1], [2, 1, 1], 0, "f64", 40),     # 7 x 12 = 84 >= 256 / 8: still worth the 16 x 16 grid
    ([6, 33, 64, 4], [1, 3, 0], 1, "f32", 45),
    # five to sixteen outputs under CE-sum (ten-class nets): the whole delta2 tile, four k-steps of dH1
    ([8, 32, 32, 10], [1, 1, 0], 1, "f32", 150),
    ([8, 32, 32, 10], [2, 1, 0], 1, "f64", 77),
    ([4, 16, 16, 5], [1, 1, 0], 1, "f64", 33),
    ([6, 64, 48, 16], [3, 1, 0], 1, "f32", 90),
    ([5, 20, 7], [1, 0], 1, "f64", 49),
    ([16, 24, 12], [2, 0], 1, "f32", 20),
    # one hidden layer: the kernel's middle layer is skipped
    ([4, 16, 3], [1, 0], 1, "f64", 150),
    ([8, 32, 2], [2, 0], 1, "f32", 77),
    ([4, 50, 1], [3, 1], 0, "f32", 40),
    ([3, 9, 4], [1, 0], 1, "f64", 33),
]


def _random_cases(n, seed=2024):
    """Shapes the fused kernels take, drawn at random: hidden widths on and off the 16 / 32 / 64 tile grid (within the
    dispatcher's padding limit), d0 <= 16, dK <= 4 (BCE) or <= 16 (CE), every hidden activation, CE or BCE, f32 or f64 (f64: widths <= 32)."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        tag = "f64" if rng.random() < 0.5 else "f32"
        hmax = 32 if tag == "f64" else 64
        h1, h2 = int(rng.integers(3, hmax + 1)), int(rng.integers(3, hmax + 1))
        hp = 16 if max(h1, h2) <= 16 else (32 if max(h1, h2) <= 32 else 64)
        if 8 * h1 * h2 < hp * hp:
            continue
        lik = int(rng.integers(0, 2))
        dK = int(rng.integers(1, 5)) if lik == 0 else int(rng.choice([2, 3, 4, 4, 5, 7, 10, 16]))
        if rng.random() < 0.35:   # one hidden layer (the kernel's middle layer is skipped)
            if 8 * h1 * h1 < (16 if h1 <= 16 else (32 if h1 <= 32 else 64)) ** 2:
                continue
            dims, acts = [int(rng.integers(1, 17)), h1, dK], [int(rng.integers(1, 4)), 1 if lik == 0 else 0]
        else:
            dims = [int(rng.integers(1, 17)), h1, h2, dK]
            acts = [int(rng.integers(1, 4)), int(rng.integers(1, 4)), 1 if lik == 0 else 0]
        out.append((dims, acts, lik, tag, int(rng.choice([1, 7, 16, 33, 90, 150]))))
    return out


CASES += _random_cases(int(os.environ.get("EY_FUZZ_SEEDS", "48")))


def _mfma32_too(dims, acts, lik, tag):
    """The 4-32-32 models with one hidden activation that the fused f32 trajectory kernel also serves, in its bf16x3 form
    (ey_mfma32_kind == 2): such a plan runs on it by default and on these kernels with EY_PRODUCTS_EXACT."""
    if tag != "f32" or dims[:3] != [4, 32, 32] or len(dims) != 4 or acts[0] != acts[1]:
        return False
    return (lik == 1 and dims[3] == 3 and acts[2] == 0 and acts[0] in (2, 3)) or (lik == 0 and dims[3] == 1 and acts[2] == 1)


def _expand(cases):
    out = []
    for case in cases:
        if _mfma32_too(*case[:4]):
            out += [case + ("exact", "fused16"), case + ("bf16x3", "mfma32")]
        else:
            out.append(case + (None, "fused16"))
    return out


def _t(a, dt):
    return torch.tensor(np.asarray(a), dtype=dt, device=DEV).contiguous()


def _setup(dims, acts, lik, tag, N, seed=0, products=None):
    from eeyore_amd.plan import Plan
    npdt, dt = (np.float64, torch.float64) if tag == "f64" else (np.float32, torch.float32)
    rng = np.random.default_rng(sum(dims) + N + seed)
    x = rng.standard_normal((N, dims[0]))
    y = np.eye(dims[-1])[rng.integers(0, dims[-1], N)] if lik == 1 else (rng.random((N, dims[-1])) < 0.5).astype(np.float64)
    P = sum((dims[l] + 1) * dims[l + 1] for l in range(len(dims) - 1))
    mu, sigma = 0.1 * rng.standard_normal(P), 0.5 + rng.random(P)
    pl = Plan(dims, [1] * (len(dims) - 1), acts, lik, dt, DEV)
    if products is not None:
        pl.f32_products = products
    pl.set_data(_t(x, dt), _t(y, dt))
    pl.set_prior(torch.tensor(mu), torch.tensor(sigma))
    co = COracle(dims, acts, lik, x, y, mu, sigma, dtype=npdt, nthreads=4)
    return pl, co, rng, npdt, dt, P


@pytest.mark.parametrize("dims,acts,lik,tag,N,products,kernel", _expand(CASES))
def test_fused16_value_gradient_and_draws_vs_oracle(dims, acts, lik, tag, N, products, kernel):
    from eeyore_amd import _lib as L
    pl, co, rng, npdt, dt, P = _setup(dims, acts, lik, tag, N, products=products)
    assert pl.kernel == kernel and pl.P == P
    C = 11
    scale = 0.3 if lik == 1 else 0.15
    th0 = (scale * rng.standard_normal((C, P))).astype(npdt)
    tol = 1e-10 if tag == "f64" else 2e-4
    t, g = pl.log_target_grad(_t(th0, dt))
    lk, pr = pl.log_target(_t(th0, dt))
    temps = torch.linspace(0.2, 1.0, C, dtype=dt, device=DEV)
    tt, gt = pl.log_target_grad(_t(th0, dt), temp=temps)
    for c in range(C):
        to, go, lo, po = co.log_target_grad(th0[c])
        gs = max(1.0, float(np.abs(go).max()))
        np.testing.assert_allclose(t[c].item(), to, rtol=tol, atol=tol * 10)
        np.testing.assert_allclose(g[c].cpu().numpy(), go, rtol=tol * 10, atol=tol * gs)
        np.testing.assert_allclose([lk[c].item(), pr[c].item()], [lo, po], rtol=tol, atol=tol * 10)
        np.testing.assert_allclose(tt[c].item(), temps[c].item() * to, rtol=tol * 2, atol=tol * 10)
        np.testing.assert_allclose(gt[c].cpu().numpy(), temps[c].item() * go, rtol=tol * 10, atol=tol * gs)
    # the generic kernels on the same inputs
    t2 = pl.hmc_step(_t(th0, dt), t.clone(), g.clone(), 1e-9, 1, p0=torch.zeros(C, P, dtype=dt, device=DEV),
                     u=torch.zeros(C, dtype=dt, device=DEV), flags=L.EY_FORCE_GENERIC | L.EY_RECOMPUTE_INITIAL_GRAD)
    assert t2["accepted"].shape == (C,)
    # ---- HMC.leapfrog: L steps, L + 1 evaluations, momentum negated
    p0 = rng.standard_normal((C, P)).astype(npdt); u = rng.random(C).astype(npdt)
    th, p = _t(th0, dt).clone(), _t(p0, dt).clone()
    eps, Ls = (0.02, 4) if lik == 1 else (0.01, 3)
    tl, gl = pl.leapfrog(th, p, eps, Ls)
    for c in range(0, C, 3):
        tho, po_, to, go = co.leapfrog(th0[c], p0[c], eps, Ls)
        np.testing.assert_allclose(th[c].cpu().numpy(), tho, rtol=tol * 10, atol=tol)
        np.testing.assert_allclose(p[c].cpu().numpy(), po_, rtol=tol * 50, atol=tol * 50)
        np.testing.assert_allclose(tl[c].item(), to, rtol=tol * 5, atol=tol * 20)
    # ---- one HMC draw (cached gradient and the reference's recomputed one), MALA, MH on recorded randomness
    tv0 = t.cpu().numpy().astype(npdt); g0 = g.cpu().numpy().astype(npdt)
    for flags in (0, L.EY_RECOMPUTE_INITIAL_GRAD):
        th, tv, gg = _t(th0, dt).clone(), t.clone(), g.clone()
        out = pl.hmc_step(th, tv, gg, eps, 5, p0=_t(p0, dt), u=_t(u, dt), flags=flags)
        tho, tvo, go = th0.copy(), tv0.copy(), g0.copy()
        acc, hc, hp = co.hmc_draw(tho, tvo, go, p0, u, eps, 5)
        rate = np.minimum(np.exp(np.minimum(hc - hp, 0)), 1)
        decided = np.abs(u - rate) > (1e-8 if tag == "f64" else 5e-3)
        np.testing.assert_array_equal(out["accepted"].cpu().numpy()[decided], acc[decided])
        np.testing.assert_allclose(out["h_prop"].cpu().numpy(), hp, rtol=tol * 10, atol=tol * 100)
        np.testing.assert_allclose(out["h_cur"].cpu().numpy(), hc, rtol=tol * 10, atol=tol * 100)
        same = out["accepted"].cpu().numpy() == acc
        np.testing.assert_allclose(th.cpu().numpy()[same], tho[same], rtol=tol * 10, atol=tol)
        np.testing.assert_allclose(tv.cpu().numpy()[same], tvo[same], rtol=tol * 5, atol=tol * 20)
    th, tv, gg = _t(th0, dt).clone(), t.clone(), g.clone()
    out = pl.mala_step(th, tv, gg, 0.004, z=_t(p0, dt), u=_t(u, dt))
    f8 = lambda a_: np.asarray(a_, dtype=np.float64).copy()
    co64 = COracle(dims, acts, lik, co.x.astype(np.float64), co.y.astype(np.float64), co.mu.astype(np.float64),
                   co.sigma.astype(np.float64), dtype=np.float64, nthreads=4)
    acc, lr = co64.mala_draw(f8(th0), f8(tv0), f8(g0), f8(p0), f8(u), 0.004)
    ltol = (1e-9 if tag == "f64" else 2e-3) * np.maximum(1.0, np.abs(lr))
    assert (np.abs(out["log_rate"].cpu().numpy() - lr) <= ltol).all()
    decided = np.abs(np.log(f8(u)) - lr) > ltol
    np.testing.assert_array_equal(out["accepted"].cpu().numpy()[decided], acc[decided])
    th, tv = _t(th0, dt).clone(), t.clone()
    out = pl.mh_step(th, tv, torch.full((P,), 0.02, dtype=dt), z=_t(p0, dt), u=_t(u, dt))
    acc, lr = co64.mh_draw(f8(th0), f8(tv0), f8(p0), f8(u), 0.02)
    ltol = (1e-9 if tag == "f64" else 2e-3) * np.maximum(1.0, np.abs(lr))
    assert (np.abs(out["log_rate"].cpu().numpy() - lr) <= ltol).all()
    decided = np.abs(np.log(f8(u)) - lr) > ltol
    np.testing.assert_array_equal(out["accepted"].cpu().numpy()[decided], acc[decided])


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_fused16_philox_and_run_blocks(tag):
    """In-kernel Philox == the same streams passed in, bit for bit; ey_hmc_run / ey_mala_run / ey_mh_run of n iterations ==
    n single steps, records included; more chains than resident waves; the generic kernel agrees on accept flags."""
    pl, co, rng, npdt, dt, P = _setup([4, 32, 32, 3], [2, 1, 0], 1, tag, 60)
    C = 700 if tag == "f32" else 300
    th0 = 0.2 * pl.philox_normal(C, seed=9, it=0)
    t0, g0 = pl.log_target_grad(th0)
    a = [th0.clone(), t0.clone(), g0.clone()]
    b = [th0.clone(), t0.clone(), g0.clone()]
    oa = pl.hmc_step(*a, 0.03, 6, seed=21, it=5, chain_offset=1000)
    ob = pl.hmc_step(*b, 0.03, 6, p0=pl.philox_normal(C, 21, 5, 1000), u=pl.philox_uniform(C, 21, 5, 1000))
    for k in ("accepted", "rate", "h_cur", "h_prop"):
        assert torch.equal(oa[k], ob[k]), k
    for x_, y_ in zip(a, b):
        assert torch.equal(x_, y_)
    assert 0 < oa["accepted"].float().mean() < 1
    n = 4
    for kind in ("hmc", "mala", "mh"):
        a = [th0.clone(), t0.clone(), g0.clone()]
        b = [th0.clone(), t0.clone(), g0.clone()]
        smp, tgt, acr = pl.empty(n, C, P), pl.empty(n, C), pl.empty(n, C, dtype=torch.uint8)
        cnt = torch.zeros(C, dtype=torch.int32, device=DEV)
        scale = torch.full((P,), 0.01, dtype=dt)
        if kind == "hmc":
            pl.hmc_run(*a, 0.03, 6, n, seed=3, it=10, samples=smp, targets=tgt, accepted_rec=acr, accept_count=cnt)
        elif kind == "mala":
            pl.mala_run(*a, 0.002, n, seed=3, it=10, samples=smp, targets=tgt, accepted_rec=acr, accept_count=cnt)
        else:
            pl.mh_run(a[0], a[1], scale, n, seed=3, it=10, samples=smp, targets=tgt, accepted_rec=acr, accept_count=cnt)
        total = torch.zeros(C, dtype=torch.int32, device=DEV)
        for i in range(n):
            if kind == "hmc":
                o = pl.hmc_step(*b, 0.03, 6, seed=3, it=10 + i)
            elif kind == "mala":
                o = pl.mala_step(*b, 0.002, seed=3, it=10 + i)
            else:
                o = pl.mh_step(b[0], b[1], scale, seed=3, it=10 + i)
            assert torch.equal(smp[i], b[0]) and torch.equal(tgt[i], b[1]) and torch.equal(acr[i], o["accepted"]), (kind, i)
            total += o["accepted"].to(torch.int32)
        assert torch.equal(cnt, total) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        if kind != "mh":
            assert torch.equal(a[2], b[2])


def test_fused16_serves_the_reference_models_in_f64():
    """G2 / G3 of the reference for MLP(4-32-32-3) in f64 -- its default dtype -- now run on the f64 matrix cores."""
    from eeyore_amd.plan import Plan
    n = 0
    for name, rec in groups(load("g2_grads.npz")).items():
        if not name.startswith("f64/mlp432323"):
            continue
        pl = Plan(rec["dims"].tolist(), [1, 1, 1], rec["acts"].tolist(), int(rec["lik"]), torch.float64, DEV)
        pl.set_data(_t(rec["x"], torch.float64), _t(rec["y"], torch.float64))
        pl.set_prior(torch.tensor(rec["prior_mu"]), torch.tensor(rec["prior_sigma"]))
        assert pl.kernel == "fused16"
        temp = None if np.isnan(rec["temperature"]) else float(rec["temperature"])
        t, g = pl.log_target_grad(_t(rec["theta"], torch.float64), temp=temp)
        np.testing.assert_allclose(t.cpu().numpy(), rec["log_target"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(g.cpu().numpy(), rec["grad"], rtol=1e-10, atol=1e-11)
        n += 1
    assert n >= 2
    m = 0
    for name, rec in groups(load("g3_leapfrog.npz")).items():
        if not name.startswith("f64/mlp432323"):
            continue
        pl = Plan(rec["dims"].tolist(), [1, 1, 1], rec["acts"].tolist(), int(rec["lik"]), torch.float64, DEV)
        pl.set_data(_t(rec["x"], torch.float64), _t(rec["y"], torch.float64))
        pl.set_prior(torch.tensor(rec["prior_mu"]), torch.tensor(rec["prior_sigma"]))
        th, p = _t(rec["theta0"], torch.float64)[None].clone(), _t(rec["p0"], torch.float64)[None].clone()
        t, g = pl.leapfrog(th, p, float(rec["step"]), int(rec["L"]))
        np.testing.assert_allclose(th[0].cpu().numpy(), rec["thetaL"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(p[0].cpu().numpy(), rec["pL"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(t.item(), rec["target"], rtol=1e-9)
        m += 1
    assert m >= 2


@pytest.mark.parametrize("dims,bias,acts,lik,tag,N", [
    ([4, 32, 32, 3], [0, 1, 0], [1, 1, 0], 1, "f32", 150),   # the headline widths without two of the biases
    ([4, 16, 16, 3], [0, 0, 0], [1, 2, 0], 1, "f64", 77),    # no bias anywhere (the reference's tests build such nets)
    ([6, 20, 24, 2], [1, 0, 1], [2, 1, 0], 1, "f64", 40),    # off the tile grid as well
    ([4, 32, 32, 1], [0, 0, 1], [3, 1, 1], 0, "f32", 90),    # BCE head
    ([5, 20, 3], [1, 0], [1, 0], 1, "f64", 33),              # one hidden layer
    ([3, 64, 64, 4], [0, 1, 0], [1, 1, 0], 1, "f32", 45),
])
def test_fused16_layers_without_bias(dims, bias, acts, lik, tag, N):
    """mlp.py:36-43 builds a layer with or without a bias (`bias=[...]`): the fused kernels hold the missing biases as
    padding slots (theta 0, gradient masked, no momentum).  Value, gradient, leapfrog and an HMC draw against the C oracle
    built with the same flags; P counts no bias of such a layer."""
    from eeyore_amd.plan import Plan
    npdt, dt = (np.float64, torch.float64) if tag == "f64" else (np.float32, torch.float32)
    rng = np.random.default_rng(sum(dims) + N + 17)
    x = rng.standard_normal((N, dims[0]))
    y = np.eye(dims[-1])[rng.integers(0, dims[-1], N)] if lik == 1 else (rng.random((N, dims[-1])) < 0.5).astype(np.float64)
    P = sum((dims[l] + bias[l]) * dims[l + 1] for l in range(len(dims) - 1))
    mu, sigma = 0.1 * rng.standard_normal(P), 0.5 + rng.random(P)
    pl = Plan(dims, bias, acts, lik, dt, DEV)
    pl.f32_products = "exact"
    pl.set_data(_t(x, dt), _t(y, dt))
    pl.set_prior(torch.tensor(mu), torch.tensor(sigma))
    assert pl.kernel == "fused16" and pl.P == P
    co = COracle(dims, acts, lik, x, y, mu, sigma, dtype=npdt, bias=bias, nthreads=4)
    C = 9
    th0 = (0.3 * rng.standard_normal((C, P))).astype(npdt)
    tol = 1e-10 if tag == "f64" else 2e-4
    t, g = pl.log_target_grad(_t(th0, dt))
    for c in range(C):
        to, go, _, _ = co.log_target_grad(th0[c])
        gs = max(1.0, float(np.abs(go).max()))
        np.testing.assert_allclose(t[c].item(), to, rtol=tol, atol=tol * 10)
        np.testing.assert_allclose(g[c].cpu().numpy(), go, rtol=tol * 10, atol=tol * gs)
    p0 = rng.standard_normal((C, P)).astype(npdt); u = rng.random(C).astype(npdt)
    th, p = _t(th0, dt).clone(), _t(p0, dt).clone()
    tl, _ = pl.leapfrog(th, p, 0.01, 4)
    for c in range(0, C, 2):
        tho, po_, to, _ = co.leapfrog(th0[c], p0[c], 0.01, 4)
        np.testing.assert_allclose(th[c].cpu().numpy(), tho, rtol=tol * 10, atol=tol)
        np.testing.assert_allclose(p[c].cpu().numpy(), po_, rtol=tol * 50, atol=tol * 50)
        np.testing.assert_allclose(tl[c].item(), to, rtol=tol * 5, atol=tol * 20)
    th, tv, gg = _t(th0, dt).clone(), t.clone(), g.clone()
    out = pl.hmc_step(th, tv, gg, 0.01, 5, p0=_t(p0, dt), u=_t(u, dt))
    tho, tvo, go = th0.copy(), t.cpu().numpy().astype(npdt), g.cpu().numpy().astype(npdt)
    acc, hc, hp = co.hmc_draw(tho, tvo, go, p0, u, 0.01, 5)
    rate = np.minimum(np.exp(np.minimum(hc - hp, 0)), 1)
    decided = np.abs(u - rate) > (1e-8 if tag == "f64" else 5e-3)
    np.testing.assert_array_equal(out["accepted"].cpu().numpy()[decided], acc[decided])
    np.testing.assert_allclose(out["h_prop"].cpu().numpy(), hp, rtol=tol * 10, atol=tol * 100)
    same = out["accepted"].cpu().numpy() == acc
    np.testing.assert_allclose(th.cpu().numpy()[same], tho[same], rtol=tol * 10, atol=tol)
    # in-kernel Philox momentum: a block of iterations == consecutive steps (no momentum leaks into a padding slot)
    a = [th.clone(), tv.clone(), gg.clone()]
    b = [th.clone(), tv.clone(), gg.clone()]
    pl.hmc_run(a[0], a[1], a[2], 0.01, 3, 2, seed=4, it=7)
    for i in range(2):
        pl.hmc_step(b[0], b[1], b[2], 0.01, 3, seed=4, it=7 + i)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_fused16_saturated_bce_is_nan_and_rejected(tag):
    """eeyore/stats/loss.py:2 on the fused kernels: an output sigmoid that rounds to exactly 1 or 0 makes log(1 - h) (1 - y)
    or log(h) y the product -inf * 0 = NaN (or -inf), and a NaN / -inf Hamiltonian is rejected (hmc.py:148).  In f64 that
    takes logits beyond +-709 (exp overflows); the lean reciprocal of the f64 path must return exactly 0 there."""
    from eeyore_amd.plan import Plan
    dt = torch.float64 if tag == "f64" else torch.float32
    dims = [4, 16, 16, 1]
    rng = np.random.default_rng(2)
    N = 24
    x = np.abs(rng.standard_normal((N, 4))) + 0.5
    y = (np.arange(N) % 2).astype(np.float64)[:, None]
    pl = Plan(dims, [1, 1, 1], [3, 3, 1], 0, dt, DEV)   # relu hidden layers: the logit grows with the weights
    pl.set_data(_t(x, dt), _t(y, dt))
    pl.set_prior(torch.zeros(pl.P), torch.full((pl.P,), 100.0))
    assert pl.kernel == "fused16"
    th = torch.zeros(3, pl.P, dtype=dt, device=DEV)
    th[0] = 3.0      # every weight 3: logits of order 4 * 16 * 16 * 27 -> the sigmoid is exactly 1, rows with y = 0 are NaN
    th[1] = 0.01     # a tame state
    th[2] = 3.0
    th[2, -17:] = -3.0  # the last layer negative: the sigmoid is exactly 0, rows with y = 1 are -inf, rows with y = 0 NaN
    lik, _ = pl.log_target(th)
    assert not torch.isfinite(lik[0]) and torch.isfinite(lik[1]) and not torch.isfinite(lik[2])
    t, g = pl.log_target_grad(th)
    th2 = th.clone()
    out = pl.hmc_step(th2, t, g, 1e-3, 3, seed=1, it=0)
    assert out["accepted"][0].item() == 0 and out["accepted"][2].item() == 0
    assert torch.equal(th2[0], th[0]) and torch.equal(th2[2], th[2])


# the shapes of the family's three translation units, one or two hidden layers, exact and padded widths
PLAIN_SHAPES = [
    ([4, 32, 32, 3], [1, 1, 0], 1, "f64", 150),   # ey_fused16_d32: the headline model in the reference's default dtype
    ([4, 24, 20, 3], [1, 2, 0], 1, "f64", 40),    #   padded
    ([5, 32, 3], [1, 0], 1, "f64", 50),           #   one hidden layer
    ([3, 28, 2], [2, 1], 0, "f64", 33),           #   one hidden layer, padded, BCE
    ([4, 64, 64, 3], [1, 1, 0], 1, "f32", 150),   # ey_fused16_d32: f32 H = 64
    ([6, 50, 40, 3], [2, 1, 0], 1, "f32", 70),
    ([4, 64, 3], [1, 0], 1, "f32", 50),
    ([4, 16, 16, 3], [1, 1, 0], 1, "f32", 150),   # ey_fused16_plain: f32 H = 16
    ([4, 12, 10, 3], [1, 3, 0], 1, "f32", 60),
    ([4, 16, 3], [1, 0], 1, "f32", 40),
    ([4, 20, 20, 3], [1, 1, 0], 1, "f32", 90),    # ey_fused16_plain: f32 H = 32
    ([8, 32, 32, 10], [2, 1, 0], 1, "f32", 64),
    ([4, 32, 2], [1, 1], 0, "f32", 48),
    ([4, 16, 16, 3], [1, 1, 0], 1, "f64", 77),    # ey_fused16_plain: f64 H = 16
    ([4, 12, 3], [2, 0], 1, "f64", 30),
]


# ... and shapes drawn at random (another stream than the suite's above; EY_FUZZ_SEEDS scales it: 48 -> 24 shapes)
PLAIN_SHAPES += _random_cases(max(4, int(os.environ.get("EY_FUZZ_SEEDS", "48")) // 2), seed=77)


@pytest.mark.parametrize("dims,acts,lik,tag,N", PLAIN_SHAPES)
def test_plain_hmc_kernels_vs_oracle_and_vs_the_general_kernel(dims, acts, lik, tag, N):
    """Every parameter under the same prior, no temperature, no tuner: the HMC draw then runs an instantiation with those
    options compiled out (F16_HMC_PLAIN) -- the case `HMC.run` on a plain posterior and the bench issue, and the
    one the suite above (per-parameter priors) never reaches.  The draw on recorded randomness against the C oracle
    (hmc.py:126-156), and against the same draw through the general kernel (a temperature vector of ones)."""
    from eeyore_amd.plan import Plan
    npdt, dt = (np.float64, torch.float64) if tag == "f64" else (np.float32, torch.float32)
    rng = np.random.default_rng(sum(dims) + 3 * N)
    x = rng.standard_normal((N, dims[0]))
    y = np.eye(dims[-1])[rng.integers(0, dims[-1], N)] if lik == 1 else (rng.random((N, dims[-1])) < 0.5).astype(np.float64)
    P = sum((dims[l] + 1) * dims[l + 1] for l in range(len(dims) - 1))
    mu0, s0 = 0.05, 1.3
    pl = Plan(dims, [1] * (len(dims) - 1), acts, lik, dt, DEV)
    if tag == "f32":
        pl.f32_products = "exact"
    pl.set_data(_t(x, dt), _t(y, dt))
    pl.set_prior(torch.full((P,), mu0, dtype=torch.float64), torch.full((P,), s0, dtype=torch.float64))
    assert pl.kernel == "fused16"
    co = COracle(dims, acts, lik, x, y, np.full(P, mu0), np.full(P, s0), dtype=npdt, nthreads=4)
    C = 9
    tol = 1e-10 if tag == "f64" else 2e-4
    th0 = ((0.3 if lik == 1 else 0.15) * rng.standard_normal((C, P))).astype(npdt)
    p0 = rng.standard_normal((C, P)).astype(npdt)
    u = rng.random(C).astype(npdt)
    eps, Ls = (0.02, 5) if lik == 1 else (0.01, 4)
    t, g = pl.log_target_grad(_t(th0, dt))
    tv0, g0 = t.cpu().numpy().astype(npdt), g.cpu().numpy().astype(npdt)
    th_p, t_p, g_p = _t(th0, dt).clone(), t.clone(), g.clone()
    plain = pl.hmc_step(th_p, t_p, g_p, eps, Ls, p0=_t(p0, dt), u=_t(u, dt))
    tho, tvo, go = th0.copy(), tv0.copy(), g0.copy()
    acc, hc, hp = co.hmc_draw(tho, tvo, go, p0, u, eps, Ls)
    rate = np.minimum(np.exp(np.minimum(hc - hp, 0)), 1)
    decided = np.abs(u - rate) > (1e-8 if tag == "f64" else 5e-3)
    got = plain["accepted"].cpu().numpy()
    np.testing.assert_array_equal(got[decided], acc[decided])
    np.testing.assert_allclose(plain["h_prop"].cpu().numpy(), hp, rtol=tol * 10, atol=tol * 100)
    np.testing.assert_allclose(plain["h_cur"].cpu().numpy(), hc, rtol=tol * 10, atol=tol * 100)
    same = got == acc
    np.testing.assert_allclose(th_p.cpu().numpy()[same], tho[same], rtol=tol * 10, atol=tol)
    np.testing.assert_allclose(t_p.cpu().numpy()[same], tvo[same], rtol=tol * 5, atol=tol * 20)
    np.testing.assert_allclose(g_p.cpu().numpy()[same], go[same], rtol=tol * 50, atol=tol * 50 * max(1.0, np.abs(go).max()))
    # the general kernel on the same draw: a temperature of one multiplies by exactly one
    th_g, t_g, g_g = _t(th0, dt).clone(), t.clone(), g.clone()
    gen = pl.hmc_step(th_g, t_g, g_g, eps, Ls, p0=_t(p0, dt), u=_t(u, dt), temp=torch.ones(C, dtype=dt, device=DEV))
    gtol = 1e-13 if tag == "f64" else 2e-5
    both = decided & (gen["accepted"].cpu().numpy() == got)
    assert both.sum() >= decided.sum() - 1
    np.testing.assert_allclose(gen["h_prop"].cpu().numpy(), plain["h_prop"].cpu().numpy(), rtol=gtol * 10, atol=gtol * 100)
    np.testing.assert_allclose(th_g.cpu().numpy()[both], th_p.cpu().numpy()[both], rtol=gtol * 10, atol=gtol)
    # whole launches of the plain kernel with the device's own randomness: Philox draws, records, several iterations
    th_r, t_r, g_r = _t(th0, dt).clone(), t.clone(), g.clone()
    th_s, t_s, g_s = _t(th0, dt).clone(), t.clone(), g.clone()
    pl.hmc_run(th_r, t_r, g_r, eps, Ls, 3, seed=11, it=40)
    for i in range(3):
        pl.hmc_step(th_s, t_s, g_s, eps, Ls, seed=11, it=40 + i)
    assert torch.equal(th_r, th_s) and torch.equal(t_r, t_s)


def test_heads_the_fused_kernels_do_not_take_are_routed_elsewhere():
    """BCE-sum on more than four sigmoid outputs and CE-sum on more than sixteen logits are outside the fused kernels'
    range: the plan must say so (another family serves it) and the values must still be the oracle's."""
    from eeyore_amd.plan import Plan
    rng = np.random.default_rng(12)
    for dims, acts, lik in (([8, 16, 16, 6], [1, 1, 1], 0), ([8, 32, 20], [2, 0], 1)):
        N = 40
        x = rng.standard_normal((N, dims[0]))
        y = (rng.random((N, dims[-1])) < 0.5).astype(np.float64) if lik == 0 else np.eye(dims[-1])[rng.integers(0, dims[-1], N)]
        P = sum((dims[l] + 1) * dims[l + 1] for l in range(len(dims) - 1))
        for dt, npdt, tol in ((torch.float64, np.float64, 1e-10), (torch.float32, np.float32, 2e-4)):
            pl = Plan(dims, [1] * (len(dims) - 1), acts, lik, dt, DEV)
            pl.set_data(_t(x, dt), _t(y, dt))
            pl.set_prior(torch.zeros(P), torch.ones(P))
            assert pl.kernel != "fused16"
            co = COracle(dims, acts, lik, x, y, 0.0, 1.0, dtype=npdt, nthreads=4)
            th0 = (0.2 * rng.standard_normal((4, P))).astype(npdt)
            t, g = pl.log_target_grad(_t(th0, dt))
            for c in range(4):
                to, go, _, _ = co.log_target_grad(th0[c])
                np.testing.assert_allclose(t[c].item(), to, rtol=tol * 5, atol=tol * 10)
                np.testing.assert_allclose(g[c].cpu().numpy(), go, rtol=tol * 50, atol=tol * 5 * max(1.0, np.abs(go).max()))


@pytest.mark.parametrize("act", [1, 2])
def test_fused16_f64_exp_limits_match_the_library(act):
    """The lean f64 exp of the fused kernels at its limits: inputs of +-1e308 and products that overflow
    to +-inf in the first layer saturate the hidden units exactly as torch / libm do (sigmoid 0 / 1, tanh -1 / 1) -- value
    and gradient finite and equal to the C oracle's, which calls the library's exp.  (A sigmoid unit at -inf is 1e-300 here,
    not 0 -- the lean reciprocal, DESIGN 4.4 -- which only shows against an input of 1e308, so the sigmoid case keeps W0 > 0.)"""
    from eeyore_amd.plan import Plan
    dims, acts, dt = [1, 16, 16, 2], [act, 1, 0], torch.float64
    x = np.array([[1e308], [-1e308 if act == 2 else 3e307], [1e300], [-1e300 if act == 2 else 1e299], [1.0], [-0.5], [710.0], [-746.0]])
    y = np.eye(2)[[0, 1, 1, 0, 0, 1, 0, 1]]
    P = sum((dims[l] + 1) * dims[l + 1] for l in range(3))
    rng = np.random.default_rng(5)
    th0 = 0.3 * rng.standard_normal((4, P))
    th0[:, :16] = (np.sign(th0[:, :16]) if act == 2 else 1.0) * (5.0 + 15.0 * rng.random((4, 16)))   # |w| in 5 .. 20: w x overflows
    pl = Plan(dims, [1, 1, 1], acts, 1, dt, DEV)
    pl.set_data(_t(x, dt), _t(y, dt))
    pl.set_prior(torch.zeros(P), torch.full((P,), 10.0))
    assert pl.kernel == "fused16"
    co = COracle(dims, acts, 1, x, y, np.zeros(P), np.full(P, 10.0), dtype=np.float64, nthreads=2)
    t, g = pl.log_target_grad(_t(th0, dt))
    for c in range(4):
        to, go, _, _ = co.log_target_grad(th0[c])
        assert np.isfinite(to) and np.isfinite(go).all()
        np.testing.assert_allclose(t[c].item(), to, rtol=1e-10)
        np.testing.assert_allclose(g[c].cpu().numpy(), go, rtol=1e-9, atol=1e-10 * max(1.0, np.abs(go).max()))


def test_fused16_built_at_O1_passes_the_oracle_suite():
    """the -O1 build of this family computed wrong results (MLP(13-29-4) BCE in f64 first of all) because the
    compiler scheduled an LDS load into the SrcC registers of a running v_mfma_f64_16x16x4_f64 directly behind it; the
    build now passes every translation unit's assembly through csrc/mfma_load_hazard.py.  `make o1` builds the family at
    -O1 through the same pipeline (161 loads padded) and this test runs the oracle suite above -- every mode, the fixed
    shapes and the random ones, the reproducer among them -- on that library in a child process."""
    import subprocess
    import sys
    from eeyore_amd import _lib as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(L.O1_LIB_PATH):
        subprocess.check_call(["make", "-C", L.CSRC, "-j8", "o1"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, EEYORE_AMD_LIB=L.O1_LIB_PATH, EY_FUZZ_SEEDS="32")
    check = subprocess.run([sys.executable, os.path.join(root, "tools", "f16_check.py"), "13,29,4", "1,1", "0", "f64", "7"],
                           env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert check.returncode == 0 and "PASS" in check.stdout, check.stdout[-2000:]
    suite = subprocess.run([sys.executable, "-m", "pytest", "tests/test_fused16.py", "-q", "-x", "-p", "no:cacheprovider",
                            "-k", "value_gradient_and_draws or philox_and_run_blocks or layers_without_bias or plain_hmc_kernels"],
                           env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert suite.returncode == 0, suite.stdout[-3000:]
    assert " passed" in suite.stdout and "failed" not in suite.stdout

