"""Chains shard embarrassingly across GPUs: one process per GPU, no collective inside a step.

RCCL (torch.distributed backend "nccl" on ROCm; "gloo" for the CPU tests) is used only to combine chain
statistics: per-rank partial sums over chains of a few [P]-vectors are all-reduced, so the payload is O(P)
floats however many chains a rank holds (xGMI ring all-reduce is one-link bound -- keep it small).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world_size, local_rank).  A single process needs no group."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # EEYORE_DIST_BACKEND=gloo: rehearse the multi-rank flow on a box with fewer GPUs than ranks
            backend = os.environ.get("EEYORE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            local = local % torch.cuda.device_count()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard(num_chains, rank, world):
    """Chains [offset, offset+count) of rank `rank`: contiguous, sizes differing by at most one."""
    base, rem = divmod(num_chains, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


class ChainStats:
    """Running per-chain moments on the device and their cross-rank combination.

    ``update(theta)`` adds one saved iteration of every local chain ([C, P]); ``summary()`` returns the
    potential scale reduction per parameter in the form the reference uses for its multivariate statistic
    (eeyore/stats/multi_rhat.py:38: (n-1)/n + (m+1)/m * lambda), restricted to the diagonal:
    W = mean over chains of the within-chain variance, B = variance of the chain means, lambda = B / W.
    Only sum_c mean_c, sum_c mean_c^2 and sum_c var_c ([3, P] doubles) and the chain count cross the wire."""

    def __init__(self, num_chains, num_params, device):
        self.n = 0
        self.s1 = torch.zeros(num_chains, num_params, dtype=torch.float64, device=device)
        self.s2 = torch.zeros(num_chains, num_params, dtype=torch.float64, device=device)
        self.acc = torch.zeros(num_chains, dtype=torch.float64, device=device)

    def attach(self, plan):
        """Let the plan's step kernels accumulate into this object (Plan.attach_moments): no separate pass over
        [C, P] per iteration, and ``update`` must then not be called for those steps."""
        plan.attach_moments(self.s1, self.s2, self.acc, on_step=self._count_step)

    def _count_step(self):
        self.n += 1

    def update(self, theta, accepted=None):
        if theta.is_cuda and theta.dtype in (torch.float32, torch.float64) and theta.is_contiguous():
            # one streaming HIP pass (ey_stats_update) instead of several elementwise torch kernels
            import ctypes as ct
            from . import _lib as L
            C, P = theta.shape
            L.check(L.lib().ey_stats_update(
                L.ptr(theta), L.ptr(accepted), C, P, 0 if theta.dtype == torch.float32 else 1, L.ptr(self.s1),
                L.ptr(self.s2), L.ptr(self.acc) if accepted is not None else None,
                ct.c_void_p(torch.cuda.current_stream(theta.device).cuda_stream)), "ey_stats_update")
            self.n += 1
            return
        t = theta.to(torch.float64)
        self.s1 += t
        self.s2.addcmul_(t, t)
        if accepted is not None:
            self.acc += accepted.to(torch.float64)
        self.n += 1

    def local_partials(self):
        n = self.n
        mean = self.s1 / n
        var = (self.s2 - n * mean * mean) / (n - 1)  # unbiased, as eeyore/stats/cov.py:15
        part = torch.stack([mean.sum(0), (mean * mean).sum(0), var.sum(0)])
        extra = torch.stack([torch.tensor(float(mean.shape[0]), dtype=torch.float64, device=part.device),
                             self.acc.sum()])  # stays on the device: no host round trip before the all-reduce
        return part, extra

    def summary(self, group=None):
        part, extra = self.local_partials()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(extra, op=dist.ReduceOp.SUM, group=group)
        m = extra[0].item()
        n = self.n
        gmean = part[0] / m
        B = (part[1] - m * gmean * gmean) / (m - 1)
        W = part[2] / m
        rhat = (n - 1) / n + (m + 1) / m * (B / W)
        return dict(rhat=rhat, mean=gmean, W=W, B=B, num_chains=int(m), num_samples=n,
                    acceptance=extra[1].item() / (m * n))


def reduce_ess(ess, group=None):
    """Combine per-rank effective sample sizes ess [C_local, P] (ChainBuffer.ess()) into job-wide figures per
    parameter: the minimum over all chains, the mean over all chains and the total (chains are independent, so their
    effective sizes add).  Series for which the estimator had not enough samples (NaN) are left out and counted.
    Two small all-reduces of [P] (+1) doubles over RCCL; nothing per chain crosses the wire."""
    e = ess.to(torch.float64)
    ok = torch.isfinite(e)
    big = torch.full_like(e, float("inf"))
    emin = torch.where(ok, e, big).amin(0)
    esum = torch.where(ok, e, torch.zeros_like(e)).sum(0)
    cnt = torch.cat([ok.sum(0).to(torch.float64), torch.tensor([float(e.shape[0])], dtype=torch.float64, device=e.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(emin, op=dist.ReduceOp.MIN, group=group)
        packed = torch.cat([esum, cnt])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        esum, cnt = packed[:esum.numel()], packed[esum.numel():]
    used, chains = cnt[:-1], cnt[-1]
    return dict(min=emin, mean=esum / used.clamp(min=1), total=esum, num_chains=int(chains.item()),
                not_enough=int((chains * used.numel() - used.sum()).item()))


def multi_rhat_sharded(samples_local, layout="ncp", group=None):
    """multi_rhat (eeyore/stats/multi_rhat.py:10-40) over the chains of ALL ranks (collective 1): every
    rank reduces its own chains on its device (``ey_inse_multivariate``: per-chain MC covariance and mean), then the
    sum of the MC covariances is all-reduced ([p,p] doubles) and the chain means are all-gathered ([C_local,p] doubles
    per rank; ranks may hold different numbers of chains).  Every rank returns the same
    (rhat, imag, W, B, is_w_pd, is_b_pd).  A chain for which the reference's estimator raises 'Not enough samples'
    (inse_mc_cov.py:45-46) makes every rank raise that error, as the reference's loop over chains would
    (multi_rhat.py:19): the flag travels with the chain counts, it costs no collective of its own."""
    from .stats import batched
    r = batched.inse_multivariate(samples_local, layout)
    short = torch.isnan(r["sig"]).flatten(1).any(1).sum()
    return multi_rhat_from_local_parts(torch.nan_to_num(r["sig"], nan=0.0).sum(0), r["mean"], r["n"], group=group,
                                       not_enough_local=short)


def multi_rhat_from_local_parts(w_sum_local, means_local, n, group=None, not_enough_local=None):
    """The collective half of ``multi_rhat_sharded`` (on whatever device / backend the tensors live on): one all-reduce
    of the [p,p] sum, one all-gather of (chain count, chains without enough samples) -- read back once, for every rank
    together -- and one all-gather of the padded chain means."""
    from .stats import batched
    w_sum, means = w_sum_local.clone(), means_local
    dev = means.device
    short = torch.zeros((), dtype=torch.int64, device=dev) if not_enough_local is None else not_enough_local.to(torch.int64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        dist.all_reduce(w_sum, op=dist.ReduceOp.SUM, group=group)
        mine = torch.stack([torch.tensor(means.shape[0], dtype=torch.int64, device=dev), short.reshape(())])
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        info = torch.stack(parts).tolist()  # the one host read-back of the gather
        counts, n_short = [c for c, _ in info], sum(s_ for _, s_ in info)
        padded = torch.zeros(max(counts), means.shape[1], dtype=means.dtype, device=dev)
        padded[:means.shape[0]] = means
        gathered = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(gathered, padded, group=group)
        means = torch.cat([g[:c] for g, c in zip(gathered, counts)])
    else:
        n_short = int(short.item())
    if n_short:
        raise RuntimeError(f'Not enough samples ({n_short} of {means.shape[0]} chains)')
    return batched.multi_rhat_from_parts(w_sum, means, n)


class TemperingExchange:
    """Parallel tempering across ranks by exchanging temperature LABELS, never states (collective 2).

    Layout: ``world`` ranks, each holding R chains; chain r of every rank together form replica r of the ladder, and
    ``labels[r]`` on a rank is the ladder position (0..K-1, K = world) that chain currently samples at.  Initially rank g
    holds position g for every replica.  An exchange attempt between ladder positions k and k+1 of one replica needs
    only the two untempered log-targets ell: log_rate = (t_k - t_{k+1}) (ell_{k+1} - ell_k), the reference's
    between-chain log-rate for adjacent partners (eeyore/samplers/power_posterior_sampler.py:135-141; the proposal terms
    cancel for the deterministic even/odd neighbour schedule), accept iff log u < log_rate (:160).

    One all-gather of ell and labels ([R] floats + [R] ints per rank: 32 KB at R = 4096) over RCCL gives every rank all
    it needs; every rank evaluates the same decisions from a shared counter-based uniform stream and updates only its own
    labels, so no state (407 KB per chain at P = 101 770) ever crosses xGMI.  ``decide`` is the swap-decision operator
    (default: the HIP kernel behind ``ey_pt_swap_decide``)."""

    def __init__(self, temperatures, num_replicas, rank, world, device, seed=0, decide=None, group=None):
        if len(temperatures) != world:
            raise ValueError("one ladder position per rank")
        self.temps = torch.as_tensor(temperatures, dtype=torch.float64)
        self.rank, self.world, self.group, self.device = rank, world, group, device
        self.R = num_replicas
        self.seed = int(seed)
        self.labels = torch.full((num_replicas,), rank, dtype=torch.int64, device=device)
        self.attempts = 0
        self._swaps = torch.zeros((), dtype=torch.int64, device=device)  # accepted exchanges so far, kept on the device
        if decide is None:
            from .plan import pt_swap_decide as decide
        self.decide = decide

    @property
    def num_swaps(self):
        """Accepted exchanges so far over the whole ladder (reading it synchronises; the sweep itself does not)."""
        return int(self._swaps.item())

    def temperature_vector(self, dtype):
        """Per-chain temperatures to pass to the step kernels."""
        return self.temps.to(self.labels.device)[self.labels].to(dtype)

    def _uniform(self, n, device, attempt):
        """u[pair] for pair = k * R + r (ladder pair k, replica r) of this attempt, from the library's Philox4x32-10
        stream keyed by (seed, pair id, attempt) (``ey_philox_uniform``: the accept-variate stream with the pair id in
        the chain slot), so every rank draws the same numbers with no communication.  On the device: one small launch.
        For CPU tensors (gloo rehearsals of the exchange logic) the same blocks come from the host entry point
        ``ey_philox_block``."""
        import ctypes as ct
        from . import _lib as L
        if torch.device(device).type == "cuda":
            out = torch.empty(n, dtype=torch.float64, device=device)
            L.check(L.lib().ey_philox_uniform(L.ptr(out), n, self.seed, attempt, 0, L.EY_F64,
                                              ct.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                    "ey_philox_uniform")
            return out
        key = (ct.c_uint32 * 2)(self.seed & 0xffffffff, (self.seed >> 32) & 0xffffffff)
        words = (ct.c_uint32 * 4)()
        out = torch.empty(n, dtype=torch.float64)
        it_lo, it_hi = attempt & 0xffffffff, (attempt >> 32) & 0xffffffff
        for pair in range(n):  # counter = (block 0, chain_lo, iter_lo, iter_hi << 8 | stream 1 | chain_hi << 20)
            ctr = (ct.c_uint32 * 4)(0, pair & 0xffffffff, it_lo, ((it_hi << 8) | 1 | ((pair >> 32) << 20)) & 0xffffffff)
            L.check(L.lib().ey_philox_block(ctr, key, words), "ey_philox_block")
            out[pair] = float((words[0] << 21) | (words[1] >> 11)) * 2.0 ** -53
        return out

    def _gather(self, t):
        if self.world == 1:
            return t[None]
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t.contiguous(), group=self.group)
        return torch.stack(out)

    def exchange(self, ell_local):
        """One even/odd neighbour sweep.  ``ell_local`` [R]: untempered log-targets of this rank's chains.  Returns the
        number of accepted exchanges over the whole ladder as a 0-d tensor (no host synchronisation here);
        ``self.labels`` is updated in place.  All pairs of the sweep go through ONE swap-decision call."""
        ell = self._gather(ell_local.detach().to(torch.float64))          # [world, R] by rank
        lab = self._gather(self.labels)                                   # [world, R]
        new_lab, accepted = self._sweep(ell, lab)
        self.labels = new_lab[self.rank].clone()
        return accepted

    def _sweep(self, ell, lab):
        """The sweep on the gathered arrays: ``ell`` [world, R] log-targets and ``lab`` [world, R] ladder positions by
        holder -> (new positions [world, R], accepted exchanges).  The same on every rank."""
        dev = ell.device
        # rank_of[k, r]: which rank holds ladder position k of replica r; ell_at[k, r] its log-target
        rank_of = torch.empty_like(lab)
        ar = torch.arange(self.R, device=dev)
        rank_of[lab, ar[None].expand_as(lab)] = torch.arange(self.world, device=dev)[:, None].expand_as(lab)
        ell_at = ell[rank_of, ar[None].expand_as(rank_of)]
        attempt = self.attempts
        self.attempts += 1
        ks = torch.arange(attempt % 2, self.world - 1, 2, device=dev)   # the lower positions of this sweep's pairs
        if ks.numel() == 0:
            return lab, torch.zeros((), dtype=torch.int64, device=dev)
        u = self._uniform(self.R * self.world, dev, attempt).view(self.world, self.R)
        temps = self.temps.to(dev)
        n = ks.numel()
        swap, _ = self.decide(ell_at[ks].reshape(-1).contiguous(), ell_at[ks + 1].reshape(-1).contiguous(),
                              temps[ks][:, None].expand(n, self.R).reshape(-1).contiguous(),
                              temps[ks + 1][:, None].expand(n, self.R).reshape(-1).contiguous(),
                              u[ks].reshape(-1).contiguous())
        m = swap.bool().view(n, self.R)
        # the two holders of an accepted pair trade ladder positions
        lo_rank, hi_rank = rank_of[ks], rank_of[ks + 1]                      # [n, R]
        cols = ar[None].expand(n, self.R)
        new_lab = lab.clone()
        new_lab[lo_rank[m], cols[m]] = (ks[:, None] + 1).expand(n, self.R)[m]
        new_lab[hi_rank[m], cols[m]] = ks[:, None].expand(n, self.R)[m]
        accepted = m.sum()
        self._swaps = self._swaps + accepted.to(self._swaps.device)
        return new_lab, accepted


class LocalTemperingLadder(TemperingExchange):
    """The same ladder held by ONE process: K positions x R replicas as one chain batch of K R chains on one GPU (row
    k R + r starts at position k of replica r), exchanged by the same sweep -- same pair ids, same Philox accept variates,
    same decisions as K ranks would take (tests/test_tempering_gpu.py holds the two against each other).  BASELINE config
    5's algorithm on a single device: ``labels`` is [K, R], ``temperature_vector`` and ``exchange`` take / give [K R]."""

    def __init__(self, temperatures, num_replicas, device, seed=0, decide=None):
        K = len(temperatures)
        super().__init__(temperatures, num_replicas, 0, K, device, seed=seed, decide=decide)
        self.labels = torch.arange(K, dtype=torch.int64, device=device)[:, None].expand(K, num_replicas).contiguous()

    def temperature_vector(self, dtype):
        return self.temps.to(self.labels.device)[self.labels].to(dtype).reshape(-1)

    def exchange(self, ell_all):
        """``ell_all`` [K R] (or [K, R]): untempered log-targets of every chain, by holder."""
        ell = ell_all.detach().to(torch.float64).reshape(self.world, self.R)
        self.labels, accepted = self._sweep(ell, self.labels)
        return accepted
