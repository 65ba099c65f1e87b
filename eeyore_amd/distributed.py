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
        extra = torch.tensor([float(mean.shape[0]), float(self.acc.sum().item())], dtype=torch.float64,
                             device=part.device)
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
