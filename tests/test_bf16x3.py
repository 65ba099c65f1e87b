"""The two forms of the fused f32 trajectory kernel's 32x32x32 products (EY_OPT_F32_PRODUCTS, include/eeyore_amd.h):
'bf16x3' (three exact bf16 pieces per f32 operand, six piece products on v_mfma_f32_32x32x16_bf16, f32 accumulation; the
default) and 'exact' (v_mfma_f32_32x32x2_f32).  Both must be f32-equivalent: every f32 fixture of the reference at the
tolerances of tests/test_gpu_parity.py, and an error against the f64 oracle that is no larger for bf16x3 than for the exact
form (measured in full by tests/tools/bf3_error_probe.py -> profiles/r03_bf3_error_probe.txt).
Reference semantics: eeyore/models/mlp.py:45-50 (nn.Linear in the model's dtype), eeyore/samplers/hmc.py:100-156."""
import numpy as np
import pytest
import torch

from oracle.c_oracle import COracle
from tests.helpers import groups, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODES = ("bf16x3", "exact")


def _t(a, dt=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dt, device=DEV).contiguous()


def _plan(rec, n_rows=None):
    from eeyore_amd.plan import Plan
    dims = rec["dims"].tolist()
    pl = Plan(dims, [1] * (len(dims) - 1), rec["acts"].tolist(), int(rec["lik"]), torch.float32, DEV)
    x, y = rec["x"], rec["y"]
    if n_rows is not None:
        reps = -(-n_rows // x.shape[0])
        x, y = np.tile(x, (reps, 1))[:n_rows], np.tile(y, (reps, 1))[:n_rows]
    pl.set_data(_t(x), _t(y))
    pl.set_prior(torch.tensor(rec["prior_mu"]), torch.tensor(rec["prior_sigma"]))
    return pl, x, y


def _headline():
    return dict(groups(load("g4_hmc_traces.npz"))["mlp432323_synth"])


def test_option_surface():
    pl, _, _ = _plan(_headline())
    assert pl.kernel == "mfma32" and pl.f32_products == "bf16x3"  # the default of a new plan
    pl.f32_products = "exact"
    assert pl.f32_products == "exact"
    with pytest.raises(ValueError):
        pl.f32_products = "tf32"
    from eeyore_amd import _lib as L
    with pytest.raises(ValueError):
        L.check(L.lib().ey_plan_set_option(pl.handle, 99, 0), "ey_plan_set_option")
    with pytest.raises(ValueError):
        L.check(L.lib().ey_plan_set_option(pl.handle, L.EY_OPT_F32_PRODUCTS, 7), "ey_plan_set_option")
    # a switch of one plan is not a switch of another
    other, _, _ = _plan(_headline())
    assert other.f32_products == "bf16x3"


@pytest.mark.parametrize("mode", MODES)
def test_g2_g3_f32_fixtures_in_both_forms(mode):
    """The reference's own f32 values (G2: upto_grad_log_target, G3: HMC.leapfrog) at test_gpu_parity.py's f32 tolerances."""
    n = 0
    for name, rec in groups(load("g2_grads.npz")).items():
        if not (name.startswith("f32/") and "mlp432323" in name):
            continue
        pl, _, _ = _plan(rec)
        pl.f32_products = mode
        temp = None if ("temperature" not in rec or np.isnan(rec["temperature"])) else float(rec["temperature"])
        t, g = pl.log_target_grad(_t(rec["theta"]), temp=temp)
        np.testing.assert_allclose(t.cpu().numpy(), rec["log_target"], rtol=2e-4, atol=2e-3)
        np.testing.assert_allclose(g.cpu().numpy(), rec["grad"], rtol=2e-4, atol=2e-4)
        n += 1
    for name, rec in groups(load("g3_leapfrog.npz")).items():
        if not (name.startswith("f32/") and "mlp432323" in name):
            continue
        pl, _, _ = _plan(rec)
        pl.f32_products = mode
        th, p = _t(rec["theta0"])[None].clone(), _t(rec["p0"])[None].clone()
        t, g = pl.leapfrog(th, p, float(rec["step"]), int(rec["L"]))
        np.testing.assert_allclose(th[0].cpu().numpy(), rec["thetaL"], rtol=5e-4, atol=5e-4)
        np.testing.assert_allclose(p[0].cpu().numpy(), rec["pL"], rtol=5e-4, atol=5e-3)
        np.testing.assert_allclose(t.item(), rec["target"], rtol=5e-4, atol=5e-3)
        np.testing.assert_allclose(g[0].cpu().numpy(), rec["grad"], rtol=5e-3, atol=5e-3)
        n += 1
    assert n >= 5


def test_error_against_f64_is_no_larger_than_the_exact_form():
    """512 seeded chains at three parameter scales: |kernel - f64 oracle on the same f32 inputs| of the log-target and of
    the gradient (relative to the chain's largest gradient entry).  Bars: bf16x3's maximum <= 1.5 x the exact form's and its
    rms <= 1.25 x (measured: 0.96-1.00 and 0.85-1.0, profiles/r03_bf3_error_probe.txt), both far inside f32 tolerance."""
    rec = _headline()
    pl, x, y = _plan(rec)
    o64 = COracle(rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"]), np.asarray(x, np.float32).astype(np.float64), y,
                  rec["prior_mu"], np.asarray(rec["prior_sigma"], np.float32).astype(np.float64), dtype=np.float64, nthreads=8)
    for scale0 in (0.1, 1.0, 3.0):
        C = 512
        th = (scale0 * pl.philox_normal(C, seed=11, it=0)).contiguous()
        thn = th.cpu().numpy().astype(np.float64)
        tt, gg = np.zeros(C), np.zeros((C, pl.P))
        for c in range(C):
            tt[c], gg[c], _, _ = o64.log_target_grad(thn[c])
        err = {}
        for mode in MODES:
            pl.f32_products = mode
            t, g = pl.log_target_grad(th)
            et = np.abs(t.cpu().numpy().astype(np.float64) - tt) / np.maximum(1.0, np.abs(tt))
            eg = np.abs(g.cpu().numpy().astype(np.float64) - gg) / np.abs(gg).max(1, keepdims=True)
            err[mode] = (et.max(), np.sqrt((et ** 2).mean()), eg.max(), np.sqrt((eg ** 2).mean()))
        b, e = err["bf16x3"], err["exact"]
        print(f"theta ~ {scale0} N(0,1): target max/rms bf16x3 {b[0]:.2e}/{b[1]:.2e} exact {e[0]:.2e}/{e[1]:.2e}; "
              f"gradient max/rms bf16x3 {b[2]:.2e}/{b[3]:.2e} exact {e[2]:.2e}/{e[3]:.2e}")
        assert b[0] <= 1.5 * e[0] + 1e-8 and b[2] <= 1.5 * e[2] + 1e-8, (scale0, err)
        assert b[1] <= 1.25 * e[1] + 1e-9 and b[3] <= 1.25 * e[3] + 1e-9, (scale0, err)
        assert b[0] < 5e-7 and b[2] < 1e-5


@pytest.mark.parametrize("mode", MODES)
def test_every_entry_point_against_the_oracle(mode):
    """HMC / MALA / MH draws with recorded randomness and the run form with in-kernel Philox, 96 chains, ragged rows (N =
    141: a last tile of 13 rows takes the peeled copy), per-chain temperature and step: accept decisions equal the f32 C
    oracle's wherever its margin exceeds the f32 tolerance, states within tolerance; run == consecutive steps bit for bit."""
    rec = _headline()
    pl, x, y = _plan(rec, n_rows=141)
    pl.f32_products = mode
    co = COracle(rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"]), x, y, rec["prior_mu"], rec["prior_sigma"],
                 dtype=np.float32, nthreads=8)
    rng = np.random.default_rng(5)
    C, P, step, Ls = 96, pl.P, 0.02, 7
    th0 = (0.2 * rng.standard_normal((C, P))).astype(np.float32)
    p0 = rng.standard_normal((C, P)).astype(np.float32)
    u = rng.random(C).astype(np.float32)
    th = _t(th0)
    t, g = pl.log_target_grad(th)
    tho, tvo, go = th0.copy(), t.cpu().numpy().copy(), g.cpu().numpy().copy()
    out = pl.hmc_step(th, t, g, step, Ls, p0=_t(p0), u=_t(u))
    acc, hc, hp = co.hmc_draw(tho, tvo, go, p0, u, step, Ls)
    rate = np.minimum(np.exp(np.minimum(hc - hp, 0.0)), 1)
    decided = np.abs(u - rate) > 2e-3 * np.maximum(1.0, np.abs(hc - hp))
    got = out["accepted"].cpu().numpy()
    assert decided.sum() > 0.8 * C and (got[decided] == acc[decided]).all()
    same = got == acc
    np.testing.assert_allclose(th.cpu().numpy()[same], tho[same], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(out["h_prop"].cpu().numpy(), hp, rtol=2e-4, atol=2e-2)
    # MALA and MH from the new state
    t, g = pl.log_target_grad(th)
    for kind in ("mala", "mh"):
        z = rng.standard_normal((C, P)).astype(np.float32)
        uu = rng.random(C).astype(np.float32)
        a = [th.clone(), t.clone(), g.clone()]
        tho, tvo, go = th.cpu().numpy().copy(), t.cpu().numpy().copy(), g.cpu().numpy().copy()
        if kind == "mala":
            o = pl.mala_step(a[0], a[1], a[2], 2e-4, z=_t(z), u=_t(uu))
            _, lr = co.mala_draw(tho, tvo, go, z, uu, 2e-4)
        else:
            o = pl.mh_step(a[0], a[1], torch.full((P,), 4e-3, device=DEV), z=_t(z), u=_t(uu))
            _, lr = co.mh_draw(tho, tvo, z, uu, 4e-3)
        np.testing.assert_allclose(o["log_rate"].cpu().numpy(), lr, rtol=2e-3, atol=2e-2)
    # blocks of iterations per launch == one launch per iteration, bit for bit (same kernel, same Philox streams)
    a = [th.clone(), t.clone(), g.clone()]
    b = [th.clone(), t.clone(), g.clone()]
    pl.hmc_run(a[0], a[1], a[2], step, Ls, 3, seed=9, it=4)
    for i in range(3):
        pl.hmc_step(b[0], b[1], b[2], step, Ls, seed=9, it=4 + i)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


def test_batches_beyond_the_bf16x3_image_take_the_exact_form():
    """The bf16x3 form's larger per-wave LDS region leaves room for 16 row tiles (512 rows); a larger batch is served by the
    exact form whatever the option says -- the same bits -- and a batch that fits again goes back."""
    rec = _headline()
    pl, _, _ = _plan(rec, n_rows=600)
    th = 0.1 * pl.philox_normal(8, seed=3, it=0)
    res = {}
    for mode in MODES:
        pl.f32_products = mode
        res[mode] = pl.log_target_grad(th)
    assert torch.equal(res["bf16x3"][0], res["exact"][0]) and torch.equal(res["bf16x3"][1], res["exact"][1])
    pl2, _, _ = _plan(rec, n_rows=512)
    res = {}
    for mode in MODES:
        pl2.f32_products = mode
        res[mode] = pl2.log_target_grad(th)
    assert not torch.equal(res["bf16x3"][1], res["exact"][1])  # two different summation orders
    np.testing.assert_allclose(res["bf16x3"][1].cpu().numpy(), res["exact"][1].cpu().numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("n_rows", [141, 320, 352])
def test_both_layouts_of_the_bf16x3_form(n_rows):
    """Batches of at most ten row tiles (N <= 320) run the bf16x3 form with H0 and delta1 crossing to the dW1 product as
    bf16 pieces through transposed-read images; larger ones (and plans with variant bit 2 set) keep the f32 round trips and
    the second pair of splits.  The two differ in the order of two piece products of dH0 and in db1's summation order only:
    same log-targets to the last bits, gradients within a few ulp of the largest entry, the same draws; and each against the
    f64 oracle on 24 chains at the row count given (320 = the last batch the images fit, 352 = the first they do not)."""
    rec = _headline()
    pl, x, y = _plan(rec, n_rows=n_rows)
    o64 = COracle(rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"]), np.asarray(x, np.float32).astype(np.float64), y,
                  rec["prior_mu"], np.asarray(rec["prior_sigma"], np.float32).astype(np.float64), dtype=np.float64, nthreads=8)
    C = 24
    th = (0.3 * pl.philox_normal(C, seed=21, it=0)).contiguous()
    thn = th.cpu().numpy().astype(np.float64)
    res = {}
    for variant in (0, 4):
        pl.set_variant(variant)
        t, g = pl.log_target_grad(th)
        a = [th.clone(), t.clone(), g.clone()]
        out = pl.hmc_step(a[0], a[1], a[2], 0.01, 6, seed=5, it=1)
        res[variant] = (t.cpu().numpy(), g.cpu().numpy(), a[0].cpu().numpy(), out["accepted"].cpu().numpy(),
                        out["h_prop"].cpu().numpy())
    pl.set_variant(0)
    for c in range(C):
        tt, gg, _, _ = o64.log_target_grad(thn[c])
        for variant in (0, 4):
            np.testing.assert_allclose(res[variant][0][c], tt, rtol=2e-6, atol=2e-4)
            np.testing.assert_allclose(res[variant][1][c], gg, rtol=1e-5, atol=2e-6 * np.abs(gg).max())
    np.testing.assert_allclose(res[0][0], res[4][0], rtol=1e-6)
    np.testing.assert_allclose(res[0][1], res[4][1], rtol=0, atol=4e-6 * np.abs(res[4][1]).max())
    assert (res[0][3] == res[4][3]).all()
    np.testing.assert_allclose(res[0][2], res[4][2], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(res[0][4], res[4][4], rtol=1e-5, atol=1e-3)
    if n_rows > 320:  # one layout serves such a batch whatever the variant says
        assert np.array_equal(res[0][1], res[4][1]) and np.array_equal(res[0][2], res[4][2])


@pytest.mark.parametrize("n_rows", [1, 8, 24, 25, 32, 33, 56, 57, 64, 96, 129, 160])
def test_row_counts_around_the_tile_edges(n_rows):
    """One row to five tiles, with last tiles of every kind (full, more than 24 rows, at most 24 rows: the peeled copy whose
    fourth 8-row k-group is all padding): value and gradient of both product forms against the f64 oracle, and a fused HMC
    draw with in-kernel momentum that is the same in a block of iterations and in single launches."""
    rec = _headline()
    pl, x, y = _plan(rec, n_rows=n_rows)
    o64 = COracle(rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"]), np.asarray(x, np.float32).astype(np.float64), y,
                  rec["prior_mu"], np.asarray(rec["prior_sigma"], np.float32).astype(np.float64), dtype=np.float64, nthreads=8)
    C = 16
    th = (0.5 * pl.philox_normal(C, seed=31 + n_rows, it=0)).contiguous()
    thn = th.cpu().numpy().astype(np.float64)
    for mode in MODES:
        pl.f32_products = mode
        t, g = pl.log_target_grad(th)
        for c in range(C):
            tt, gg, _, _ = o64.log_target_grad(thn[c])
            np.testing.assert_allclose(t[c].item(), tt, rtol=2e-6, atol=2e-4)
            np.testing.assert_allclose(g[c].cpu().numpy(), gg, rtol=1e-5, atol=2e-6 * max(1.0, np.abs(gg).max()))
        a = [th.clone(), t.clone(), g.clone()]
        b = [th.clone(), t.clone(), g.clone()]
        pl.hmc_run(a[0], a[1], a[2], 0.01, 4, 3, seed=6, it=2)
        for i in range(3):
            pl.hmc_step(b[0], b[1], b[2], 0.01, 4, seed=6, it=2 + i)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        assert torch.isfinite(a[1]).all()


@pytest.mark.parametrize("n_rows", [96, 352])
def test_other_4_32_32_models_in_both_layouts(n_rows):
    """The tanh / CE and sigmoid / BCE 4-32-32 models on the fused f32 kernel (bf16x3 form only): batches of at most ten row
    tiles take the piece images in HMC launches, larger ones the first layout; value, gradient and an HMC draw with
    recorded randomness against the f64 / f32 oracles at both sizes."""
    from eeyore_amd.plan import Plan
    rng = np.random.default_rng(40 + n_rows)
    for dims, acts, lik in (([4, 32, 32, 3], [2, 2, 0], 1), ([4, 32, 32, 1], [1, 1, 1], 0)):
        x = rng.standard_normal((n_rows, 4))
        y = np.eye(3)[rng.integers(0, 3, n_rows)] if lik == 1 else (rng.random((n_rows, 1)) < 0.5).astype(np.float64)
        pl = Plan(dims, [1, 1, 1], acts, lik, torch.float32, DEV)
        pl.set_data(_t(x), _t(y))
        P = pl.P
        pl.set_prior(torch.zeros(P), torch.full((P,), 1.5))
        assert pl.kernel == "mfma32" and pl.f32_products == "bf16x3"
        x32, y32 = x.astype(np.float32), y.astype(np.float32)
        o64 = COracle(dims, acts, lik, x32.astype(np.float64), y, 0.0, 1.5, dtype=np.float64, nthreads=8)
        o32 = COracle(dims, acts, lik, x32, y32, 0.0, 1.5, dtype=np.float32, nthreads=8)
        C = 12
        th0 = (0.25 * rng.standard_normal((C, P))).astype(np.float32)
        t, g = pl.log_target_grad(_t(th0))
        for c in range(C):
            tt, gg, _, _ = o64.log_target_grad(th0[c].astype(np.float64))
            np.testing.assert_allclose(t[c].item(), tt, rtol=5e-6, atol=5e-4)
            np.testing.assert_allclose(g[c].cpu().numpy(), gg, rtol=2e-4, atol=5e-6 * max(1.0, np.abs(gg).max()))
        p0 = rng.standard_normal((C, P)).astype(np.float32); u = rng.random(C).astype(np.float32)
        th, tv, gv = _t(th0).clone(), t.clone(), g.clone()
        out = pl.hmc_step(th, tv, gv, 0.01, 6, p0=_t(p0), u=_t(u))
        tho, tvo, go = th0.copy(), t.cpu().numpy().copy(), g.cpu().numpy().copy()
        acc, hc, hp = o32.hmc_draw(tho, tvo, go, p0, u, 0.01, 6)
        rate = np.minimum(np.exp(np.minimum(hc - hp, 0.0)), 1)
        decided = np.abs(u - rate) > 2e-3 * np.maximum(1.0, np.abs(hc - hp))
        got = out["accepted"].cpu().numpy()
        assert decided.sum() > 0.7 * C and (got[decided] == acc[decided]).all()
        np.testing.assert_allclose(out["h_prop"].cpu().numpy(), hp, rtol=3e-4, atol=3e-2)


@pytest.mark.parametrize("n_rows", [33, 64, 150, 200, 256])
def test_pipelined_tile_loop_equals_the_two_wave_form(n_rows):
    """Variant bit 3: the HMC draw of the headline model with the pipelined tile loop (one wave per SIMD, two row tiles in
    flight, eval_pipe / k_mfma32p; 2 .. 8 row tiles) against the shipped form (two waves per SIMD, one tile at a time).  The
    arithmetic of a tile is the same operation for operation, so positions, gradients and accept decisions agree bit for
    bit and the log-targets to an ulp (the row sums of the log-likelihood contract differently); each also against the f64
    oracle; a launch of several iterations with records equals single launches."""
    rec = _headline()
    pl, x, y = _plan(rec, n_rows=n_rows)
    o64 = COracle(rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"]), np.asarray(x, np.float32).astype(np.float64), y,
                  rec["prior_mu"], np.asarray(rec["prior_sigma"], np.float32).astype(np.float64), dtype=np.float64, nthreads=8)
    C, L, step = 40, 7, 0.02
    th0 = (0.2 * pl.philox_normal(C, seed=77 + n_rows, it=0)).contiguous()
    t0, g0 = pl.log_target_grad(th0)
    res = {}
    for variant in (0, 8):
        pl.set_variant(variant)
        a = [th0.clone(), t0.clone(), g0.clone()]
        out = pl.hmc_step(a[0], a[1], a[2], step, L, seed=9, it=1)
        one = [v.clone() for v in a] + [out["accepted"].clone(), out["h_prop"].clone()]
        samples, targets, acc = pl.empty(4, C, pl.P), pl.empty(4, C), pl.empty(4, C, dtype=torch.uint8)
        pl.hmc_run(a[0], a[1], a[2], step, L, 4, seed=9, it=2, samples=samples, targets=targets, accepted_rec=acc)
        b = [v.clone() for v in one[:3]]
        for i in range(4):
            pl.hmc_step(b[0], b[1], b[2], step, L, seed=9, it=2 + i)
            assert torch.equal(samples[i], b[0]) and torch.equal(targets[i], b[1])
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
        res[variant] = one + [a[0].clone(), a[1].clone()]
    pl.set_variant(0)
    v0, v8 = res[0], res[8]
    assert torch.equal(v0[0], v8[0]) and torch.equal(v0[2], v8[2]) and torch.equal(v0[3], v8[3]) and torch.equal(v0[5], v8[5])
    np.testing.assert_allclose(v8[1].cpu().numpy(), v0[1].cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(v8[4].cpu().numpy(), v0[4].cpu().numpy(), rtol=1e-6, atol=1e-3)
    assert 0 < int(v8[3].sum().item())
    # the leapfrog's end point through the oracle: the proposal of an accepted chain
    thn, tn, gn = th0.cpu().numpy().astype(np.float64), t0.cpu().numpy(), g0.cpu().numpy()
    for c in np.flatnonzero(v8[3].cpu().numpy())[:6]:
        tt, gg, _, _ = o64.log_target_grad(v8[0][c].cpu().numpy().astype(np.float64))
        np.testing.assert_allclose(v8[1][c].item(), tt, rtol=2e-6, atol=2e-4)
        np.testing.assert_allclose(v8[2][c].cpu().numpy(), gg, rtol=1e-5, atol=2e-6 * max(1.0, np.abs(gg).max()))
