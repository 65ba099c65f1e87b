"""The N>1 path on CPU: world_size-2 gloo processes shard the chains and combine chain statistics with one
all-reduce; the result must equal the single-process statistic on all chains."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eeyore_amd.distributed import ChainStats, shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _chains(C=10, iters=40, P=7):
    rng = np.random.default_rng(0)
    base = rng.standard_normal((1, 1, P))
    offs = 0.3 * rng.standard_normal((C, 1, P))
    x = base + offs + rng.standard_normal((C, iters, P)).cumsum(1) * 0.05
    acc = (rng.random((C, iters)) < 0.7)
    return torch.tensor(x), torch.tensor(acc)


def _summary(x, acc):
    st = ChainStats(x.shape[0], x.shape[2], "cpu")
    for i in range(x.shape[1]):
        st.update(x[:, i], acc[:, i])
    return st.summary()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, acc = _chains()
    off, cnt = shard(x.shape[0], rank, world)
    s = _summary(x[off:off + cnt], acc[off:off + cnt])
    from eeyore_amd.distributed import reduce_ess
    e = reduce_ess(_fake_ess()[off:off + cnt])
    s.update({"ess_" + k: v for k, v in e.items()})
    if rank == 0:
        q.put({k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in s.items()})
    dist.barrier()
    dist.destroy_process_group()


def _fake_ess(C=10, P=7):
    e = torch.tensor(np.random.default_rng(4).uniform(5.0, 300.0, (C, P)))
    e[3, 2] = float("nan")  # a series without enough samples
    return e


def test_shard_is_a_partition():
    for C in (1, 7, 8, 4096, 32768):
        for world in (1, 2, 3, 8):
            parts = [shard(C, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(p[1] for p in parts) == C
            for a, b in zip(parts, parts[1:]):
                assert a[0] + a[1] == b[0]
            assert max(p[1] for p in parts) - min(p[1] for p in parts) <= 1


def test_rhat_formula_matches_reference_form():
    x, acc = _chains()
    s = _summary(x, acc)
    m, n = x.shape[0], x.shape[1]
    means = x.mean(1)
    W = x.var(1, unbiased=True).mean(0)
    B = means.var(0, unbiased=True)
    want = (n - 1) / n + (m + 1) / m * (B / W)  # eeyore/stats/multi_rhat.py:38 on the diagonal
    np.testing.assert_allclose(s["rhat"].numpy(), want.numpy(), rtol=1e-10)
    assert abs(s["acceptance"] - acc.double().mean().item()) < 1e-12


def test_world_size_2_gloo_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x, acc = _chains()
    want = _summary(x, acc)
    np.testing.assert_allclose(got["rhat"], want["rhat"].numpy(), rtol=1e-10)
    np.testing.assert_allclose(got["mean"], want["mean"].numpy(), rtol=1e-12)
    assert got["num_chains"] == x.shape[0] and got["num_samples"] == x.shape[1]
    assert abs(got["acceptance"] - want["acceptance"]) < 1e-12
    from eeyore_amd.distributed import reduce_ess
    e = reduce_ess(_fake_ess())
    for k in ("min", "mean", "total"):
        np.testing.assert_allclose(got["ess_" + k], e[k].numpy(), rtol=1e-12)
    assert got["ess_num_chains"] == 10 and got["ess_not_enough"] == e["not_enough"] == 1
    full = _fake_ess()
    np.testing.assert_allclose(e["min"].numpy()[2], np.nanmin(full.numpy()[:, 2]))
    np.testing.assert_allclose(e["mean"].numpy()[2], np.nanmean(full.numpy()[:, 2]))


# ------------------------------------------------------------------------------------------------ tempering exchange
def _torch_decide(ell_i, ell_j, t_i, t_j, u, dlogq=None):
    lr = (t_i - t_j) * (ell_j - ell_i)
    return (torch.log(u) < lr).to(torch.uint8), lr


def _pt_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eeyore_amd.distributed import TemperingExchange
    R = 50
    temps = [(i / world) ** 4 for i in range(1, world + 1)]
    ex = TemperingExchange(temps, R, rank, world, "cpu", seed=3, decide=_torch_decide)
    hist = []
    for it in range(6):
        g = torch.Generator().manual_seed(100 * it + rank)
        ell = -50.0 + 10.0 * torch.randn(R, generator=g, dtype=torch.float64)
        ex.exchange(ell)
        hist.append(ex.labels.clone())
    q.put((rank, torch.stack(hist).numpy(), ex.num_swaps))
    dist.barrier()
    dist.destroy_process_group()


def test_tempering_exchange_keeps_a_permutation_and_agrees_across_ranks():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pt_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r, hist, swaps = q.get(timeout=120)
        got[r] = (hist, swaps)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    labels = np.stack([got[r][0] for r in range(world)])  # [rank, attempt, replica]
    # at every attempt, the ranks' labels of each replica are a permutation of the ladder positions
    assert (np.sort(labels, axis=0) == np.arange(world)[:, None, None]).all()
    assert len({got[r][1] for r in range(world)}) == 1 and got[0][1] > 0  # same swap count everywhere, some accepted
    # positions move by at most one ladder step per attempt
    assert np.abs(np.diff(labels, axis=1)).max() <= 1


def test_tempering_exchange_single_process_matches_rule():
    from eeyore_amd.distributed import TemperingExchange
    ex = TemperingExchange([1.0], 4, 0, 1, "cpu", decide=_torch_decide)
    assert int(ex.exchange(torch.zeros(4, dtype=torch.float64))) == 0 and ex.labels.tolist() == [0, 0, 0, 0]
    assert ex.temperature_vector(torch.float32).tolist() == [1.0] * 4


def test_tempering_exchange_uniforms_are_the_library_philox_stream():
    """The shared accept variates of a sweep are ey_philox_uniform's stream keyed by (seed, pair id, attempt): checked
    against the numpy Philox twin (itself pinned by Random123's vectors), bit for bit."""
    from eeyore_amd.distributed import TemperingExchange
    from oracle import philox_oracle as po
    ex = TemperingExchange([0.1, 0.4, 1.0], 5, 0, 3, "cpu", seed=77, decide=_torch_decide)
    for attempt in (0, 1, 9):
        u = ex._uniform(15, "cpu", attempt)
        assert np.array_equal(u.numpy(), po.uniform(15, 77, attempt, 0, np.float64))


# ------------------------------------------------------------------------------------------------ sharded multi_rhat
def _rhat_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import eeyore_amd.stats as st
    from eeyore_amd.distributed import multi_rhat_from_local_parts
    from tests.helpers import load
    z = load("g7_stats.npz")
    x = torch.tensor(z["chains"])                     # [4, 1000, 3]
    mine = x[:3] if rank == 0 else x[3:]              # uneven shards: 3 chains and 1
    w_sum = sum(st.inse_mc_cov(c) for c in mine)      # (on a GPU rank ey_inse_multivariate gives these in one launch)
    r = multi_rhat_from_local_parts(w_sum, mine.mean(1), x.shape[1])
    q.put((rank, r[0], r[2].numpy(), r[3].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_multi_rhat_equals_the_reference_value():
    """W all-reduced, chain means all-gathered: two ranks with 3 + 1 of the reference's example chains give
    the reference's published multi_rhat on both ranks."""
    from tests.helpers import load
    z = load("g7_stats.npz")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rhat_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, rhat, W, B in got:
        np.testing.assert_allclose(rhat, float(z["multi_rhat"]), rtol=1e-10)
        np.testing.assert_allclose(W, z["W"], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(B, z["B"], rtol=1e-10)


def _rhat_short_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eeyore_amd.distributed import multi_rhat_from_local_parts
    p = 3
    w = torch.eye(p, dtype=torch.float64)
    means = torch.zeros(2 if rank == 0 else 1, p, dtype=torch.float64)
    short = torch.tensor(1 if rank == 1 else 0)       # one chain of rank 1 had 'Not enough samples'
    try:
        multi_rhat_from_local_parts(w, means, 100, not_enough_local=short)
        q.put((rank, "returned"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_multi_rhat_raises_on_every_rank_when_one_chain_is_too_short():
    """The reference raises 'Not enough samples' from inside its loop over chains (inse_mc_cov.py:45-46 via
    multi_rhat.py:19); sharded, the chain may live on another rank: every rank must raise, none may return NaN."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rhat_short_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all("Not enough samples (1 of 3 chains)" in got[r] for r in range(world)), got


# ------------------------------------------------------------------------------------------- bench.py's own launcher
def _run_bench(extra, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, env=env, capture_output=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    return r.returncode, [json.loads(ln) for ln in lines], r.stderr.decode()


def test_bench_launches_its_own_ranks_on_cpu():
    """`python bench.py --gpus 2` with no launcher around it starts two rank processes itself (gloo here), takes the
    MAX of their timed regions and prints exactly ONE JSON line with n_gpus = 2; --gpus 1 stays a single process.
    (--dry-run: the plumbing without the device; the measuring form of the same command runs in the -m gpu suite.)"""
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"],
                                {"EEYORE_DIST_BACKEND": "gloo"})
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 20 and lines[0]["warmup"] == 5
    assert lines[0]["dry_run"] is True and lines[0]["value"] is None  # never mistaken for a measurement
    assert lines[0]["ms_per_step"] * 20 >= 20.0 - 1e-6  # the slower rank (sleeps 20 ms) sets the time: MAX over ranks
    rc, lines, err = _run_bench(["--gpus", "1", "--dry-run"])
    assert rc == 0 and len(lines) == 1 and lines[0]["n_gpus"] == 1, err


def test_bench_launcher_reports_a_failed_rank():
    """A rank that dies makes the launcher exit non-zero (here: a world size the children refuse)."""
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run", "--chains-per-gpu", "not-a-number"])
    assert rc != 0


# ----------------------------------------------------------------------------------- eight ranks, uneven shards (gloo)
# BASELINE configs[3] and [4] run on eight GPUs, and no 8-GPU node is available to this suite: these tests take the
# whole multi-rank path -- chain statistics, sharded multi_rhat, the tempering exchange, bench.py's own launcher --
# through eight gloo ranks holding DIFFERENT numbers of chains, so that the first RCCL run is not also the first 8-rank run.
_EIGHT_COUNTS = [3, 1, 2, 4, 1, 3, 2, 1]  # chains per rank: 17 in all


def _ar_chains(C=17, n=400, p=3):
    rng = np.random.default_rng(7)
    x = np.zeros((C, n, p))
    e = rng.standard_normal((C, n, p))
    x[:, 0] = e[:, 0]
    for i in range(1, n):
        x[:, i] = 0.6 * x[:, i - 1] + e[:, i]
    return torch.tensor(x + 0.05 * rng.standard_normal((C, 1, p)))


def _eight_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import eeyore_amd.stats as st
    from eeyore_amd.distributed import TemperingExchange, multi_rhat_from_local_parts
    x = _ar_chains()
    off = sum(_EIGHT_COUNTS[:rank])
    mine = x[off:off + _EIGHT_COUNTS[rank]]
    # (a) running-moment R-hat summary: one all-reduce of [3, P]
    acc = torch.ones(mine.shape[0], dtype=torch.bool)
    s = ChainStats(mine.shape[0], mine.shape[2], "cpu")
    for i in range(mine.shape[1]):
        s.update(mine[:, i], acc)
    summ = s.summary()
    # (b) the sharded multi_rhat: all-reduce of the sum of MC covariances, all-gather of the (padded) chain means
    w_sum = sum(st.inse_mc_cov(c) for c in mine)
    r = multi_rhat_from_local_parts(w_sum, mine.mean(1), x.shape[1])
    # (c) the tempering exchange: eight ladder positions, one per rank
    temps = [(i / world) ** 4 for i in range(1, world + 1)]
    ex = TemperingExchange(temps, 20, rank, world, "cpu", seed=5, decide=_torch_decide)
    hist = []
    for it in range(8):
        g = torch.Generator().manual_seed(1000 * it + rank)
        ex.exchange(-50.0 + 10.0 * torch.randn(20, generator=g, dtype=torch.float64))
        hist.append(ex.labels.clone())
    q.put((rank, summ["rhat"].numpy(), summ["num_chains"], float(r[0]), r[2].numpy(), torch.stack(hist).numpy(), ex.num_swaps))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_with_uneven_shards_agree_with_one_process():
    import eeyore_amd.stats as st
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=300)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x = _ar_chains()
    one = _summary(x, torch.ones(x.shape[0], x.shape[1], dtype=torch.bool))
    ref = st.multi_rhat(x)
    for r in range(world):
        rhat, m, mrhat, W, labels, swaps = got[r]
        assert m == 17
        np.testing.assert_allclose(rhat, one["rhat"].numpy(), rtol=1e-11)
        np.testing.assert_allclose(mrhat, float(ref[0]), rtol=1e-10)
        np.testing.assert_allclose(W, ref[2].numpy(), rtol=1e-10, atol=1e-14)
    labels = np.stack([got[r][4] for r in range(world)])  # [rank, attempt, replica]
    assert (np.sort(labels, axis=0) == np.arange(world)[:, None, None]).all()  # a permutation of the ladder at every attempt
    assert len({got[r][5] for r in range(world)}) == 1 and got[0][5] > 0
    assert np.abs(np.diff(labels, axis=1)).max() <= 1


def test_bench_dry_run_with_eight_ranks():
    """`python bench.py --gpus 8 --dry-run` (gloo): the launcher, the rendezvous, the MAX over eight ranks and the two
    statistics collectives of the measuring run (R-hat summary, ESS gather) over uneven shards; one JSON line."""
    rc, lines, err = _run_bench(["--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-run"],
                                {"EEYORE_DIST_BACKEND": "gloo"}, timeout=600)
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["dry_run"] is True and lines[0]["value"] is None
    col = lines[0]["config"]["collectives"]
    assert col["rhat_num_chains"] == col["ess_num_chains"] == col["expected_num_chains"] == sum(3 + r % 3 for r in range(8))
    assert np.isfinite(col["rhat_max"])
    assert lines[0]["ms_per_step"] * 20 >= 80.0 - 1e-6  # the slowest of the eight ranks (sleeps 80 ms) sets the time
