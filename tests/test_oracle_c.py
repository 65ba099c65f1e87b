"""The C oracle (oracle/mlp_oracle.c) against the golden vectors captured from the reference."""
import numpy as np

from oracle.c_oracle import COracle
from tests.helpers import groups, load


def _co(rec, dtype=np.float64, **kw):
    t = None
    if "temperature" in rec and not np.isnan(rec["temperature"]):
        t = float(rec["temperature"])
    return COracle(rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"]), rec["x"], rec["y"], rec["prior_mu"],
                   rec["prior_sigma"], dtype=dtype, temperature=t, **kw)


def test_c_g1_kats():
    for name, rec in groups(load("g1_kats.npz")).items():
        co = _co(rec)
        t, g, lik, prior = co.log_target_grad(rec["theta"])
        np.testing.assert_allclose(t, rec["log_target"], rtol=1e-13)
        np.testing.assert_allclose(lik, rec["log_lik"], rtol=1e-13)
        np.testing.assert_allclose(prior, rec["log_prior"], rtol=1e-13)
        np.testing.assert_allclose(g, rec["grad"], rtol=1e-11, atol=1e-13)


def test_c_g2_grads():
    for name, rec in groups(load("g2_grads.npz")).items():
        f64 = name.startswith("f64")
        co = _co(rec, np.float64 if f64 else np.float32)
        for i in range(rec["theta"].shape[0]):
            t, g, _, _ = co.log_target_grad(rec["theta"][i])
            np.testing.assert_allclose(t, rec["log_target"][i], rtol=1e-10 if f64 else 2e-4, atol=0 if f64 else 2e-3)
            np.testing.assert_allclose(g, rec["grad"][i], rtol=1e-9 if f64 else 2e-4, atol=1e-12 if f64 else 2e-4)


def test_c_g3_leapfrog():
    for name, rec in groups(load("g3_leapfrog.npz")).items():
        if not name.startswith("f64"):
            continue
        co = _co(rec)
        th, p, t, g = co.leapfrog(rec["theta0"], rec["p0"], float(rec["step"]), int(rec["L"]))
        np.testing.assert_allclose(th, rec["thetaL"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(p, rec["pL"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(t, rec["target"], rtol=1e-10)


def _replay(rec, kind):
    co = _co(rec)
    th = rec["theta0"][None].copy()
    tv = np.array([rec["init_target"]], dtype=np.float64)
    g = rec["init_grad"][None].copy()
    for it in range(rec["z"].shape[0]):
        z = rec["z"][it][None].copy()
        u = np.array([rec["u"][it]])
        if kind == "hmc":
            acc, _, _ = co.hmc_draw(th, tv, g, z, u, float(rec["step"]), int(rec["L"]))
        elif kind == "mala":
            acc, _ = co.mala_draw(th, tv, g, z, u, float(rec["par"]))
        else:
            acc, _ = co.mh_draw(th, tv, z, u, float(rec["par"]))
        assert int(acc[0]) == int(rec["accepted"][it]), it
        np.testing.assert_allclose(th[0], rec["sample"][it], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(tv[0], rec["target_val"][it], rtol=1e-9)


def test_c_g4_hmc_traces():
    for name, rec in groups(load("g4_hmc_traces.npz")).items():
        _replay(rec, "hmc")


def test_c_g5_mala_mh_traces():
    for name, rec in groups(load("g5_mala_mh_traces.npz")).items():
        _replay(rec, "mala" if name.startswith("mala") else "mh")


def test_c_multichain_threads_match_serial():
    rec = groups(load("g4_hmc_traces.npz"))["mlp2321"]
    rng = np.random.default_rng(0)
    C, P = 16, rec["theta0"].shape[0]
    th0 = 0.3 * rng.standard_normal((C, P))
    p0 = rng.standard_normal((C, P))
    u = rng.random(C)
    outs = []
    for nt in (1, 4):
        co = _co(rec, nthreads=nt)
        th = th0.copy()
        tv = np.zeros(C)
        g = np.zeros((C, P))
        for c in range(C):
            tv[c], g[c], _, _ = co.log_target_grad(th[c])
        acc, hc, hp = co.hmc_draw(th, tv, g, p0.copy(), u, 0.5, 6)
        outs.append((th, tv, acc, hc, hp))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
    assert 0 < outs[0][2].sum() < C
