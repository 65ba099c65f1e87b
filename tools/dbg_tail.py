import sys, numpy as np, torch
sys.path.insert(0, '.')
from eeyore_amd.plan import Plan
from eeyore_amd import _lib as L
DEV = torch.device('cuda', 0)
for dims, acts, N in (([20, 128, 10], [1, 0], 96), ([20, 64, 10], [1, 0], 96), ([20, 128, 3], [1, 0], 50), ([20, 32, 10], [2, 0], 33), ([784, 128, 10], [1, 0], 96), ([784, 128, 10], [1, 0], 96)):
    rng = np.random.default_rng(0)
    x = rng.random((N, dims[0])) * (rng.random((N, dims[0])) < 0.19); y = np.eye(dims[-1])[np.arange(N) % dims[-1]]
    L.lib().ey_debug_set_variant(16)
    pl = Plan(dims, [1, 1], acts, 1, torch.float32, DEV)
    pl.set_data(torch.tensor(x, dtype=torch.float32, device=DEV), torch.tensor(y, dtype=torch.float32, device=DEV))
    pl.set_prior(torch.zeros(pl.P), torch.ones(pl.P))
    th = torch.tensor(0.05 * rng.standard_normal((3, pl.P)), dtype=torch.float32, device=DEV)
    t1, g1 = pl.log_target_grad(th)
    L.lib().ey_debug_set_variant(16 | 64)
    t2, g2 = pl.log_target_grad(th)
    L.lib().ey_debug_set_variant(0)
    d = (g1 - g2).abs()
    xt = torch.tensor(x, dtype=torch.float64, device=DEV); lab = torch.tensor(np.arange(N) % dims[-1], device=DEV)
    thd = th.double().requires_grad_(True)
    W0 = thd[:, :dims[0]*dims[1]].reshape(3, dims[1], dims[0]); b0 = thd[:, dims[0]*dims[1]:dims[0]*dims[1]+dims[1]]
    o1 = dims[0]*dims[1]+dims[1]
    W1 = thd[:, o1:o1+dims[1]*dims[2]].reshape(3, dims[2], dims[1]); b1 = thd[:, o1+dims[1]*dims[2]:]
    h = torch.einsum('nd,chd->cnh', xt, W0) + b0[:, None, :]
    h = torch.sigmoid(h) if acts[0] == 1 else torch.tanh(h)
    z = torch.einsum('cnh,ckh->cnk', h, W1) + b1[:, None, :]
    ll = -torch.nn.functional.cross_entropy(z.reshape(-1, dims[2]), lab.repeat(3), reduction='sum')
    lp = -0.5 * (thd ** 2).sum()
    (ll + lp).backward()
    gref = thd.grad
    print('   vs f64 autograd: tail', (g1 - gref).abs().max().item(), 'no-tail', (g2 - gref).abs().max().item())
    print(dims, pl.kernel, 't', (t1 - t2).abs().max().item(), 'g max', d.max().item())
    K = len(dims) - 1
    off = 0
    for l in range(K):
        nw = dims[l] * dims[l + 1]
        print('   layer', l, 'dW', d[:, off:off + nw].max().item(), 'db', d[:, off + nw:off + nw + dims[l + 1]].max().item())
        if l == K - 1:
            dd = d[0, off:off + nw].reshape(dims[l + 1], dims[l])
            print('   per j', dd.max(1).values.cpu().numpy().round(4))
            print('   per i (first 32)', dd.max(0).values.cpu().numpy().round(3)[:32])
        off += nw + dims[l + 1]
