"""Chain diagnostics against the values the reference computes on its own examples/stats/chain0[1-4].csv (G7)."""
import numpy as np
import torch

import eeyore_amd.stats as st
from eeyore_amd.chains import ChainList, ChainLists
from tests.helpers import load


def test_cov_inse_ess_rhat_match_reference():
    z = load("g7_stats.npz")
    x = torch.tensor(z["chains"])
    for i in range(4):
        np.testing.assert_allclose(st.cov(x[i]).numpy(), z["cov"][i], rtol=1e-12)
        np.testing.assert_allclose(st.inse_mc_cov(x[i]).numpy(), z["inse_mc_cov"][i], rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(st.multi_ess(x[i]), z["multi_ess"][i], rtol=1e-8)
    rhat, imag, W, B, wpd, bpd = st.multi_rhat(x)
    np.testing.assert_allclose(rhat, z["multi_rhat"], rtol=1e-9)
    np.testing.assert_allclose(W.numpy(), z["W"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(B.numpy(), z["B"], rtol=1e-10)
    assert wpd and bpd and imag == 0


def test_chain_list_summaries():
    z = load("g7_stats.npz")
    x = torch.tensor(z["chains"])
    chains = []
    for c in range(4):
        ch = ChainList()
        for i in range(x.shape[1]):
            ch.update(dict(sample=x[c, i], target_val=torch.tensor(0.0), accepted=1))
        chains.append(ch)
    np.testing.assert_allclose(chains[0].multi_ess(), z["multi_ess"][0], rtol=1e-8)
    np.testing.assert_allclose(chains[1].mc_cov().numpy(), z["inse_mc_cov"][1], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(chains[1].mc_se().numpy(), np.sqrt(np.diag(z["inse_mc_cov"][1])), rtol=1e-9)
    cl = ChainLists.from_chain_list(chains)
    s = cl.summary(keys=['multi_ess', 'multi_rhat', 'mean', 'acceptance'])
    np.testing.assert_allclose(s['multi_rhat'], z["multi_rhat"], rtol=1e-9)
    np.testing.assert_allclose(s['multi_ess'], z["multi_ess"].mean(), rtol=1e-8)
    assert s['acceptance'] == 1 and s['mean'].shape == (3,)


def test_running_mean_and_nearest_pd():
    x = torch.arange(1.0, 6.0)
    assert torch.equal(st.running_mean(x), torch.tensor([1.0, 1.5, 2.0, 2.5, 3.0]))
    a = torch.tensor([[1.0, 2.0], [2.0, 1.0]], dtype=torch.float64)  # indefinite
    p = st.nearest_pd(a)
    assert st.is_pos_def(p) and not st.is_pos_def(a)


def test_batched_multivariate_inse_and_ess_over_chains():
    """stats.batched.inse_mc_cov_chains / multi_ess_chains: all chains per lag in one batched product, against the
    reference's numbers (G7) and the per-chain function on chains that stop at different lags."""
    from eeyore_amd.stats import batched
    z = load("g7_stats.npz")
    x = torch.tensor(z["chains"])
    np.testing.assert_allclose(batched.inse_mc_cov_chains(x).numpy(), z["inse_mc_cov"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(batched.multi_ess_chains(x).numpy(), z["multi_ess"], rtol=1e-8)
    rng = np.random.default_rng(0)
    y = torch.tensor(rng.standard_normal((7, 240, 4)).cumsum(1) * 0.1 + rng.standard_normal((7, 240, 4)))
    y[3] = 1.5  # a constant chain: never positive definite -> the reference raises 'Not enough samples'
    got = batched.inse_mc_cov_chains(y)
    for i in range(7):
        if i == 3:
            assert torch.isnan(got[i]).all()
            continue
        np.testing.assert_allclose(got[i].numpy(), st.inse_mc_cov(y[i]).numpy(), rtol=1e-10, atol=1e-13)


def test_adjusted_inse_adds_a_positive_semidefinite_lift():
    """inse_mc_cov(adjust=True) (inse_mc_cov.py:72-81; the reference's own code calls the removed torch.symeig, so there is
    no captured value): the adjusted estimate is the plain one plus twice the summed negative parts of the accepted lag
    pairs' Gam, i.e. plus a positive semi-definite matrix; methods and error text as the reference."""
    import pytest
    z = load("g7_stats.npz")
    x = torch.tensor(z["chains"])
    for i in range(4):
        plain, adj = st.inse_mc_cov(x[i]), st.inse_mc_cov(x[i], adjust=True)
        lift = adj - plain
        assert torch.equal(lift, lift.T) or torch.allclose(lift, lift.T, atol=1e-15)
        assert torch.linalg.eigvalsh((lift + lift.T) / 2).min().item() > -1e-12
        np.testing.assert_allclose(st.mc_se(x[i], adjust=True).numpy(), np.sqrt(np.diag(adj.numpy())), rtol=1e-12)
    np.testing.assert_allclose(st.mc_cov(x[0], method='iid').numpy(), z["cov"][0], rtol=1e-12)
    with pytest.raises(ValueError, match='The method can be inse or iid, nope was given'):
        st.mc_cov(x[0], method='nope')
    with pytest.raises(ValueError):
        st.multi_rhat(x, method='nope')
    with pytest.raises(RuntimeError, match='Not enough samples'):
        st.inse_mc_cov(torch.ones(20, 3, dtype=torch.float64))
    r_iid = st.multi_rhat(x, method='iid')[0]
    assert np.isfinite(r_iid)
