"""The numpy oracle against every golden vector captured from the reference (tests/golden/make_golden.py)
and against the known-answer values of the reference's own tests."""
import numpy as np
import pytest

from oracle import mlp_oracle as orc
from tests.helpers import groups, load, spec_from

KAT = {  # SURVEY.md section 4, computed with the reference
    "mlp221/log_lik": -16.08587869723768,
    "mlp221/log_prior": -106.60224679884205,
    "mlp221_s100/log_target": -65.81269034997256,
    "mlp2321/log_lik": -7.39132872690876,
    "mlp4323/log_lik": -187.79398747047628,
    "mlp433/log_lik": -176.25449918882558,
}
KAT_GRAD_221 = [-0.31125019417302047, -0.31040841228821037, 0.0002700236585229453, 0.00015006352409646542,
                -0.36974887883749585, -0.000346705861808311, -1.909804588374762, -1.9979385440124289,
                -1.9991671747169164]


def test_g1_kats_match_survey_values():
    z = load("g1_kats.npz")
    for k, v in KAT.items():
        assert float(z[k]) == pytest.approx(v, rel=0, abs=1e-12)
    np.testing.assert_allclose(z["mlp221_s100/grad"], KAT_GRAD_221, rtol=0, atol=1e-13)


def test_g1_oracle():
    for name, rec in groups(load("g1_kats.npz")).items():
        spec = spec_from(rec)
        th = rec["theta"]
        assert orc.log_lik(spec, th, rec["x"], rec["y"]) == pytest.approx(float(rec["log_lik"]), rel=1e-13)
        assert orc.log_prior(spec, th) == pytest.approx(float(rec["log_prior"]), rel=1e-13)
        t, g = orc.upto_grad_log_target(spec, th, rec["x"], rec["y"])
        assert t == pytest.approx(float(rec["log_target"]), rel=1e-13)
        np.testing.assert_allclose(g, rec["grad"], rtol=1e-11, atol=1e-13)


def test_parameter_layout_mlp2321():
    # tests/test_binary_classif_mlp2321_log_lik.py:50-64 of the reference: W then b per layer
    spec = orc.Spec([2, 3, 2, 1], [1, 1, 1], 0)
    assert spec.w_off == [0, 9, 17] and spec.b_off == [6, 15, 19] and spec.P == 20


@pytest.mark.parametrize("tag,rtol,atol", [("f64", 1e-10, 1e-12), ("f32", 2e-4, 2e-4)])
def test_g2_grads(tag, rtol, atol):
    dt = np.float64 if tag == "f64" else np.float32
    n = 0
    for name, rec in groups(load("g2_grads.npz")).items():
        if not name.startswith(tag):
            continue
        spec = spec_from(rec, dt)
        for i in range(rec["theta"].shape[0]):
            th = rec["theta"][i].astype(dt)
            t, g = orc.upto_grad_log_target(spec, th, rec["x"].astype(dt), rec["y"].astype(dt))
            assert t.dtype == dt and g.dtype == dt
            np.testing.assert_allclose(t, rec["log_target"][i], rtol=rtol, atol=atol * 10)
            np.testing.assert_allclose(g, rec["grad"][i], rtol=rtol, atol=atol)
            if tag == "f64":
                ll = orc.log_lik(spec, th, rec["x"], rec["y"])
                lp = orc.log_prior(spec, th)
                np.testing.assert_allclose(ll, rec["log_lik"][i], rtol=1e-12)
                np.testing.assert_allclose(lp, rec["log_prior"][i], rtol=1e-12)
            n += 1
    assert n > 10


def test_g3_leapfrog_f64():
    n = 0
    for name, rec in groups(load("g3_leapfrog.npz")).items():
        if not name.startswith("f64"):
            continue
        spec = spec_from(rec)
        th, p, t, g = orc.leapfrog(spec, rec["theta0"], rec["p0"], rec["x"], rec["y"], float(rec["step"]), int(rec["L"]))
        np.testing.assert_allclose(th, rec["thetaL"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(p, rec["pL"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(t, rec["target"], rtol=1e-10)
        np.testing.assert_allclose(g, rec["grad"], rtol=1e-8, atol=1e-10)
        n += 1
    assert n == 18


def _replay(rec, kind):
    spec = spec_from(rec)
    x, y = rec["x"], rec["y"]
    cur = dict(sample=rec["theta0"].copy(), target_val=rec["init_target"][()], grad_val=rec["init_grad"].copy())
    iters = rec["z"].shape[0]
    margins = []
    for it in range(iters):
        if kind == "hmc":
            cur, info = orc.hmc_draw(spec, cur, rec["z"][it], rec["u"][it], x, y, float(rec["step"]), int(rec["L"]))
            margins.append(abs(float(rec["u"][it]) - float(info["rate"])))
        elif kind == "mala":
            cur, info = orc.mala_draw(spec, cur, rec["z"][it], rec["u"][it], x, y, float(rec["par"]))
            margins.append(abs(np.log(float(rec["u"][it])) - float(info["log_rate"])))
        else:
            cur, info = orc.mh_draw(spec, cur, rec["z"][it], rec["u"][it], x, y, float(rec["par"]))
            margins.append(abs(np.log(float(rec["u"][it])) - float(info["log_rate"])))
        assert info["accepted"] == int(rec["accepted"][it]), (it, margins[-1])
        np.testing.assert_allclose(cur["sample"], rec["sample"][it], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(cur["target_val"], rec["target_val"][it], rtol=1e-9)
    return min(margins)


def test_g4_hmc_traces_accept_bit_exact():
    for name, rec in groups(load("g4_hmc_traces.npz")).items():
        m = _replay(rec, "hmc")
        assert m > 1e-9, (name, m)  # no decision sits inside rounding distance of the threshold


def test_g5_mala_mh_traces_accept_bit_exact():
    for name, rec in groups(load("g5_mala_mh_traces.npz")).items():
        m = _replay(rec, "mala" if name.startswith("mala") else "mh")
        assert m > 1e-9, (name, m)


def test_g6_power_posterior():
    z = load("g6_power_posterior.npz")
    K = len(z["ladder"])
    np.testing.assert_allclose(orc.pt_ladder(K), z["ladder"], rtol=1e-15)
    for i in range(K):
        np.testing.assert_allclose(orc.pt_categorical_probs(i, K, float(z["b"])), z[f"cat_probs/{i}"], rtol=1e-12)
    spec = orc.Spec(z["dims"].tolist(), z["acts"].tolist(), int(z["lik"]), mu=z["prior_mu"], sigma=z["prior_sigma"])
    ell = np.array([orc.log_target(spec, z["samples"][i], z["x"], z["y"]) for i in range(K)])
    np.testing.assert_allclose(ell, z["ell"], rtol=1e-12)
    for (i, j), lr, lq in zip(z["pairs"], z["log_rate"], z["log_q"]):
        got = orc.pt_swap_log_rate(lq[0], lq[1], ell[i], ell[j], z["ladder"][i], z["ladder"][j])
        np.testing.assert_allclose(got, lr, rtol=1e-9, atol=1e-10)


def test_g8_univariate_inse_oracle_matches_reference():
    """oracle/diagnostics_oracle.py against the reference's inse_mc_cov / cov run on single columns (G8)."""
    from oracle import diagnostics_oracle as do
    z = load("g8_univariate_stats.npz")
    x = z["chains"]
    for i in range(x.shape[0]):
        for j in range(x.shape[2]):
            sig, used = do.inse_univariate(x[i, :, j])
            np.testing.assert_allclose(sig, z["inse"][i, j], rtol=1e-11)
            np.testing.assert_allclose(do.sample_var(x[i, :, j]), z["var"][i, j], rtol=1e-12)
            assert used >= 1
            sig, _ = do.inse_univariate(x[i, :200, j])
            np.testing.assert_allclose(sig, z["inse_first200"][i, j], rtol=1e-11)
            np.testing.assert_allclose(do.sample_var(x[i, :200, j]), z["var_first200"][i, j], rtol=1e-12)
    with np.testing.assert_raises(RuntimeError):
        do.inse_univariate(np.ones(10))  # a constant series never gives a positive Sig


def test_g7_is_the_reference_published_check():
    """G7 must be the reference's own example (examples/stats/multi_rhat.py:14, multi_ess.py: genfromtxt with no
    skipped row, 1000 x 3 per chain): SURVEY.md section 4 quotes these two values from running it."""
    z = load("g7_stats.npz")
    assert z["chains"].shape == (4, 1000, 3)
    assert float(z["multi_rhat"]) == 1.0134832973360262
    assert float(z["multi_ess"][0]) == 564.6937234344964
    z8 = load("g8_univariate_stats.npz")
    assert z8["chains"].shape == (4, 1000, 3) and np.array_equal(z8["chains"], z["chains"])


# ------------------------------------------------------------------------------------------------ G10: LogisticRegression
def _lr_spec(z, bias, dt=np.float64):
    D = z["x"].shape[1]
    P = D + int(bias)
    return orc.Spec([D, 1], [1], 0, bias=[int(bias)], mu=np.zeros(P, dt), sigma=np.full(P, float(z["prior_sigma"]), dt))


@pytest.mark.parametrize("tag,rtol,atol", [("f64", 1e-12, 1e-13), ("f32", 2e-5, 2e-5)])
def test_g10_logistic_regression_values_and_gradients(tag, rtol, atol):
    """The reference's LogisticRegression (eeyore/models/logistic_regression.py:8-37) is the one-layer case of the MLP
    oracle: sigmoid(x w + b), BCE-sum, Normal prior."""
    z = load("g10_logistic_regression.npz")
    dt = np.float64 if tag == "f64" else np.float32
    for bias in (1, 0):
        spec = _lr_spec(z, bias, dt)
        key = f"{tag}/bias{bias}"
        for i in range(z[f"{key}/theta"].shape[0]):
            th = z[f"{key}/theta"][i].astype(dt)
            x, y = z["x"].astype(dt), z["y"].astype(dt)
            np.testing.assert_allclose(orc.log_lik(spec, th, x, y), z[f"{key}/log_lik"][i], rtol=rtol, atol=atol * 100)
            np.testing.assert_allclose(orc.log_prior(spec, th), z[f"{key}/log_prior"][i], rtol=rtol, atol=atol * 100)
            t, g = orc.upto_grad_log_target(spec, th, x, y)
            np.testing.assert_allclose(t, z[f"{key}/log_target"][i], rtol=rtol, atol=atol * 100)
            np.testing.assert_allclose(g, z[f"{key}/grad"][i], rtol=rtol * 10, atol=atol * 10)


def test_g10_logistic_regression_mh_trace():
    z = load("g10_logistic_regression.npz")
    rec = {k[3:]: z[k] for k in z.files if k.startswith("mh/")}
    spec = _lr_spec(z, 1)
    cur = dict(sample=rec["theta0"].copy(), target_val=rec["init_target"][()])
    margins = []
    for it in range(rec["z"].shape[0]):
        cur, info = orc.mh_draw(spec, cur, rec["z"][it], rec["u"][it], z["x"], z["y"], float(rec["par"]))
        margins.append(abs(np.log(float(rec["u"][it])) - float(info["log_rate"])))
        assert info["accepted"] == int(rec["accepted"][it]), (it, margins[-1])
        np.testing.assert_allclose(cur["sample"], rec["sample"][it], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(cur["target_val"], rec["target_val"][it], rtol=1e-10)
    assert min(margins) > 1e-9 and 0 < rec["accepted"].sum() < len(rec["accepted"])
