"""The torch-CPU restatement of the reference's op sequence (oracle/torch_autograd_path.py, the "reference-faithful"
CPU baseline of bench.py) against the reference's golden vectors: gradients (G2), HMC.leapfrog (G3), HMC.draw traces
with bit-exact accept decisions (G4)."""
import numpy as np
import torch

from oracle.torch_autograd_path import TorchReferencePath
from tests.helpers import groups, load


def _path(rec, dtype=torch.float64, temperature=None):
    return TorchReferencePath(rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"]), rec["x"], rec["y"],
                              rec["prior_mu"], rec["prior_sigma"], dtype=dtype, temperature=temperature)


def test_gradients_match_g2():
    n = 0
    for name, rec in groups(load("g2_grads.npz")).items():
        if not name.startswith("f64") or "mlp432323_iris" in name:
            continue
        t = None if np.isnan(rec["temperature"]) else float(rec["temperature"])
        tp = _path(rec, temperature=t)
        for i in range(min(2, rec["theta"].shape[0])):
            v, g = tp.upto_grad_log_target(torch.tensor(rec["theta"][i]))
            np.testing.assert_allclose(v.item(), rec["log_target"][i], rtol=1e-12)
            np.testing.assert_allclose(g.detach().numpy(), rec["grad"][i], rtol=1e-10, atol=1e-12)
        n += 1
    assert n >= 10


def test_leapfrog_matches_g3():
    n = 0
    for name, rec in groups(load("g3_leapfrog.npz")).items():
        if not name.startswith("f64") or "iris" in name:
            continue
        tp = _path(rec)
        th, p, t, g = tp.leapfrog(torch.tensor(rec["theta0"]), torch.tensor(rec["p0"]), float(rec["step"]), int(rec["L"]))
        np.testing.assert_allclose(th.detach().numpy(), rec["thetaL"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(p.detach().numpy(), rec["pL"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(t.item(), rec["target"], rtol=1e-12)
        n += 1
    assert n >= 5


def test_hmc_draws_match_g4():
    for name in ("cfg1_big_step", "mlp2321"):
        rec = groups(load("g4_hmc_traces.npz"))[name]
        tp = _path(rec)
        cur = tp.start(torch.tensor(rec["theta0"]))
        np.testing.assert_allclose(cur["target_val"].item(), rec["init_target"], rtol=1e-13)
        for it in range(rec["z"].shape[0]):
            cur = tp.hmc_draw(cur, float(rec["step"]), int(rec["L"]), p0=torch.tensor(rec["z"][it]),
                              u=torch.tensor([rec["u"][it]]))
            assert cur["accepted"] == int(rec["accepted"][it]), (name, it)
            np.testing.assert_allclose(cur["sample"].numpy(), rec["sample"][it], rtol=1e-9, atol=1e-11)
