#!/usr/bin/env python3
"""Headline benchmark: leapfrog-steps/sec x chains, HMC on MLP(4-32-32-3), Iris-shaped synthetic data.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one HMC iteration of every chain of the rank (momentum draw from the in-kernel Philox stream, L = 20
leapfrog steps, accept, state update, and the state of every chain RECORDED as the reference's run loop records it after
burn-in, eeyore/samplers/serial_sampler.py:35-52, eeyore/chains/chain_list.py:64-67), BASELINE.json configs[2]:
4096 chains per GPU, MLP(4-32-32-3; sigmoid, sigmoid, None), CE-sum, prior N(0, sqrt 3), N = 150 rows, fp32.
Steps are issued as `HMC.run` issues them: whole launches of 25 iterations (ey_hmc_run writing samples / targets /
accept flags into the chain buffer); --steps is rounded UP to whole launches and the line says so.
Chains shard over ranks with no collective in the step (weak scaling: 4096 chains per GPU, configs[3] at N = 8);
when N > 1 the per-parameter R-hat and ESS summaries are combined over RCCL once at the end of the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIMS = [4, 32, 32, 3]
N_ROWS = 150
L_STEPS = 20
CHAINS_PER_GPU = 4096
STEP_SIZE = 0.024  # ~70 % acceptance after burn-in on this target (tools/step_sweep.py: 0.02 -> 0.87, 0.03 -> 0.40)
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_FLOPS = 2.5e15  # dense bf16 matrix peak (MI355X_MICROARCH.md; the 2:1-sparsity headline is twice that)
MEASURED_F32_MFMA_TFLOPS = 145.9  # tools/peak_probe.hip on the box (profiles/r01_peak_probe.txt): the sustained clock


def flops_per_leapfrog_step(dims, n_rows):
    """F_step = 2 N (2 sum_l d_l d_{l+1} + sum_{l>=1} d_l d_{l+1}) + 6 P (unpadded, no credit for the
    reference's redundant first evaluation)."""
    prods = [dims[i] * dims[i + 1] for i in range(len(dims) - 1)]
    P = sum((dims[i] + 1) * dims[i + 1] for i in range(len(dims) - 1))
    return 2 * n_rows * (2 * sum(prods) + sum(prods[1:])) + 6 * P


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for name in ("ey_mfma32.hip", "ey_common.h"):
        with open(os.path.join(ROOT, "eeyore_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def cpu_reference_faithful(x, y, sigma, budget_s=8.0):
    """the reference's own op sequence on torch-CPU autograd (oracle/torch_autograd_path.py:
    nn.Linear -> sigmoid -> CrossEntropyLoss(sum) -> Normal.log_prob -> autograd.grad(create_graph=True), L + 1
    evaluations per draw, one chain after the other), timed on this host.  fp32 as the reference's examples run
    (examples/samplers/mlp/iris/mala_cpu_chainlist.py:30).  The reference itself cannot travel to this box."""
    from oracle.torch_autograd_path import TorchReferencePath
    tp = TorchReferencePath(DIMS, [1, 1, 0], 1, x, y, 0.0, sigma, dtype=torch.float32)
    # one thread: the path is dispatch-bound (~1 ms per evaluation whatever the thread count), and
    # torch's default of one thread per hardware thread of a 256-thread host only adds synchronisation to 32-wide products
    threads_before = torch.get_num_threads()
    torch.set_num_threads(1)
    torch.manual_seed(0)
    cur = tp.start(0.1 * torch.randn(tp.P, dtype=torch.float32))
    for _ in range(2):
        cur = tp.hmc_draw(cur, STEP_SIZE, L_STEPS)
    t0 = time.perf_counter()
    iters = 0
    while time.perf_counter() - t0 < budget_s:
        cur = tp.hmc_draw(cur, STEP_SIZE, L_STEPS)
        iters += 1
    t = time.perf_counter() - t0
    torch.set_num_threads(threads_before)
    return {"value": iters * L_STEPS / t, "unit": "leapfrog-steps/sec x chains", "cores": 1,
            "kind": "port", "sample": f"1 chain x {iters} HMC iterations (L={L_STEPS}, L+1 autograd evaluations each), f32, "
                                      f"torch {torch.__version__} CPU autograd op for op as the reference, {t:.1f} s; chains "
                                      f"run serially in the reference, so x chains = the same number"}


def cpu_reference_faithful_config1(budget_s=4.0):
    """the same torch-CPU counterpart for BASELINE configs[0] -- HMC, 1 chain, MLP(2-2-1) sigmoid-sigmoid,
    BCE-sum, XOR 4 x 2, prior N(0, 100), step 0.1, L = 10, f64, theta0 = the reference's test vector
    (eeyore/samplers/hmc.py:126-170 on tests/*mlp221*'s model) -- timed on this host beside the GPU run."""
    from oracle.torch_autograd_path import TorchReferencePath
    x = np.array([[0., 0.], [0., 1.], [1., 0.], [1., 1.]])
    y = np.array([[0.], [1.], [1.], [0.]])
    tp = TorchReferencePath([2, 2, 1], [1, 1], 0, x, y, 0.0, 100.0, dtype=torch.float64)
    threads_before = torch.get_num_threads()
    torch.set_num_threads(1)
    torch.manual_seed(0)
    cur = tp.start(torch.tensor([1.1, -2.9, -0.4, 0.8, 4.3, 9.2, 4.44, -3.4, 7.2], dtype=torch.float64))
    for _ in range(5):
        cur = tp.hmc_draw(cur, 0.1, 10)
    t0 = time.perf_counter()
    iters = 0
    while time.perf_counter() - t0 < budget_s:
        cur = tp.hmc_draw(cur, 0.1, 10)
        iters += 1
    t = time.perf_counter() - t0
    torch.set_num_threads(threads_before)
    return {"value": iters * 10 / t, "unit": "leapfrog-steps/sec x chains", "cores": 1, "kind": "port",
            "sample": f"BASELINE configs[0]: 1 chain x {iters} HMC iterations (L=10, L+1 autograd evaluations each), MLP(2-2-1) on XOR, "
                      f"f64, torch {torch.__version__} CPU autograd op for op as the reference, {t:.1f} s"}


def gpu_config1(dev, iters=2000):
    """BASELINE configs[0] on the device through the C ABI (the generic kernels' register-resident form): one chain,
    MLP(2-2-1), XOR, f64, L = 10, launches of 100 draws (ey_hmc_run), beside the CPU line above."""
    from eeyore_amd.plan import Plan
    x = torch.tensor([[0., 0.], [0., 1.], [1., 0.], [1., 1.]], dtype=torch.float64, device=dev)
    y = torch.tensor([[0.], [1.], [1.], [0.]], dtype=torch.float64, device=dev)
    pl = Plan([2, 2, 1], [1, 1], [1, 1], 0, torch.float64, dev)
    pl.set_data(x, y)
    pl.set_prior(torch.zeros(pl.P), torch.full((pl.P,), 100.0))
    th = torch.tensor([[1.1, -2.9, -0.4, 0.8, 4.3, 9.2, 4.44, -3.4, 7.2]], dtype=torch.float64, device=dev)
    t, g = pl.log_target_grad(th)
    pl.hmc_run(th, t, g, 0.1, 10, 100, seed=3, it=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(iters // 100):
        pl.hmc_run(th, t, g, 0.1, 10, 100, seed=3, it=101 + 100 * b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": (iters // 100) * 100 * 10 / dt, "unit": "leapfrog-steps/sec x chains", "kernel": pl.kernel,
            "sample": f"1 chain x {(iters // 100) * 100} HMC iterations in launches of 100 (L=10), f64, {dt * 1e3:.1f} ms"}


def usable_cpus():
    """Hardware threads this process may run on: the scheduler's affinity mask, cut down by a cgroup CPU quota where one
    is set (a container may see every thread of the host and still be throttled to a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), None))):
        try:
            with open(path) as f:
                quota, period = parse(f.read())
            if period is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().strip()
            if quota not in ("max", "-1") and float(quota) > 0:
                n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return n


def cpu_baseline(x, y, sigma, cores, budget_s=9.0):
    """The C oracle (oracle/mlp_oracle.c, a port of the reference's algorithm) timed on `cores` threads of this host
    (one chain per thread at a time, OpenMP over chains) on a bounded sample of the same workload.  The oracle is the
    checker, never the product."""
    from oracle.c_oracle import COracle
    host_cpus = os.cpu_count() or 1
    co = COracle(DIMS, [1, 1, 0], 1, x, y, 0.0, sigma, dtype=np.float32, nthreads=cores)
    rng = np.random.default_rng(0)
    P = co.P

    def run(C, iters):
        th = (0.1 * rng.standard_normal((C, P))).astype(np.float32)
        tv = np.zeros(C, np.float32)
        g = np.zeros((C, P), np.float32)
        for c in range(C):
            tv[c], g[c], _, _ = co.log_target_grad(th[c])
        t0 = time.perf_counter()
        for _ in range(iters):
            co.hmc_draw(th, tv, g, rng.standard_normal((C, P)).astype(np.float32), rng.random(C).astype(np.float32),
                        STEP_SIZE, L_STEPS)
        return time.perf_counter() - t0

    t_probe = run(2 * cores, 1)
    # scale the sample to ~budget_s of wall time: more chains first (keeps every core busy), then iterations
    per_chain_iter = t_probe / 2.0  # every thread took two chains through one iteration
    total = max(1.0, budget_s / max(per_chain_iter, 1e-4))  # chain-iterations per thread that fit the budget
    per_thread = int(max(2, min(16, total // 100)))  # few chains per thread: their start-up evaluations are serial
    iters = int(max(1, min(100, total // per_thread)))
    C0 = cores * per_thread
    t = run(C0, iters)
    return {"value": C0 * iters * L_STEPS / t, "unit": "leapfrog-steps/sec x chains", "cores": cores, "kind": "port",
            "host_cpus": host_cpus,
            "sample": f"{C0} chains x {iters} HMC iterations (L={L_STEPS}, L+1 gradient evaluations each as "
                      f"hmc.py:104), f32, C oracle with OpenMP over chains ({cores} of the host's {host_cpus} hardware "
                      f"threads), {t:.1f} s"}


def through_sampler_run(xs, ys, sigma, theta0, dev, n_iters, block, seed, chain_offset, world, gloo):
    """The same workload through the plugin surface a reference script uses: eeyore_amd.samplers.HMC(model, theta0 [C, P],
    ...).run(num_epochs, num_burnin_epochs=0) recording every iteration into its ChainBuffer [iters, C, P]
    (eeyore/samplers/serial_sampler.py:35-52, eeyore/chains/chain_list.py:64-67).  One untimed run of a block first (the
    buffer is allocated then), the same barrier / synchronize bracket and MAX over ranks as the headline.  -> seconds."""
    from torch.distributions import Normal
    from torch.utils.data import DataLoader
    from eeyore_amd.chains import ChainBuffer
    from eeyore_amd.constants import loss_functions
    from eeyore_amd.datasets import XYDataset
    from eeyore_amd.models import mlp
    from eeyore_amd.samplers import HMC
    data = XYDataset(torch.tensor(xs, dtype=torch.float32, device=dev), torch.tensor(ys, dtype=torch.float32, device=dev))
    loader = DataLoader(data, batch_size=len(data), shuffle=False)
    model = mlp.MLP(loss=loss_functions['multiclass_classification'],
                    hparams=mlp.Hyperparameters(dims=DIMS, bias=3 * [True], activations=[torch.sigmoid, torch.sigmoid, None]),
                    dtype=torch.float32, device=dev)
    P = model.num_params()
    model.prior = Normal(torch.zeros(P, device=dev), torch.full((P,), sigma, device=dev))
    sampler = HMC(model, theta0=theta0, dataloader=loader, step=STEP_SIZE, num_steps=L_STEPS, seed=seed,
                  chain_offset=chain_offset, chain=ChainBuffer(capacity=n_iters))
    sampler.fused_block = block
    sampler.run(num_epochs=block, num_burnin_epochs=0)   # untimed: allocates the chain buffer, warms the path
    sampler.counter.reset()
    sampler.get_chain().rewind()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    sampler.run(num_epochs=n_iters, num_burnin_epochs=0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    chain = sampler.get_chain()
    assert len(chain) == n_iters and chain.get_samples().shape == (n_iters, theta0.shape[0], P)
    return elapsed, float(chain.acceptance_rate().mean().item())


def self_launch(n_ranks, argv):
    """`bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process -- which has not
    touched the GPU: no HIP call, no torch.cuda call before this point -- starts N fresh rank processes with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, as torch.distributed.run would, waits for them, relays rank 0's one
    JSON line and returns non-zero if any rank failed.  (Children, not exec: a rank must start from a clean process.)"""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.strip()]
    for ln in lines:  # the one JSON line on stdout; anything else a library printed there (gloo's banner) on stderr
        print(ln, flush=True, file=sys.stdout if ln.lstrip().startswith("{") else sys.stderr)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def dry_run(args, rank, world):
    """The multi-rank skeleton of main() without the device: barrier, timed region, MAX over ranks, one line from rank 0."""
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    # the collectives the measuring run issues behind its timed region, on host tensors: the R-hat summary (one all-reduce of
    # the [3, P] partial sums, distributed.ChainStats.summary) and the ESS gather (distributed.reduce_ess), over UNEVEN
    # shards of seeded chains (rank r holds 3 + r % 3 of them), so that the first multi-GPU run is not their first N-rank run
    from eeyore_amd.distributed import ChainStats, reduce_ess
    rng = np.random.default_rng(100 + rank)
    c_local, p_dim, n_it = 3 + rank % 3, 11, 30
    x = torch.tensor(rng.standard_normal((c_local, n_it, p_dim)).cumsum(1) * 0.1)
    st = ChainStats(c_local, p_dim, "cpu")
    for i in range(n_it):
        st.update(x[:, i], torch.tensor(rng.random(c_local) < 0.7))
    summ = st.summary()
    ess = reduce_ess(torch.tensor(rng.uniform(5.0, 300.0, (c_local, p_dim))))
    if rank == 0:
        print(json.dumps({"metric": "leapfrog-steps/sec x chains, HMC MLP(4-32-32-3)", "value": None, "dry_run": True,
                          "unit": "leapfrog-steps/sec x chains", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
                          "higher_is_better": True, "scaling": "weak",
                          "config": {"collectives": {"rhat_num_chains": summ["num_chains"], "rhat_max": float(summ["rhat"].max().item()),
                                                     "ess_num_chains": ess["num_chains"],
                                                     "expected_num_chains": sum(3 + r % 3 for r in range(world))}}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def config5_share(dev, chains=4096, iters=3, burnin=20, step=0.024):
    """BASELINE configs[4]'s share of ONE GPU, beside the headline (rank 0, N = 1 only; not the metric): HMC L = 20 on
    MLP(784-128-10), MNIST-shaped synthetic data (N = 1024 rows, ~19 % non-zero, 10 balanced classes), `chains` chains at
    one temperature, the layerwise path (tools/bench_config5.py is the stand-alone form with the tempering ladder).  Freed
    again before it returns."""
    import numpy as np
    import torch
    from eeyore_amd.plan import Plan
    N, L = 1024, 20
    rng = np.random.default_rng(0)
    x = (rng.random((N, 784)) * (rng.random((N, 784)) < 0.19)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[np.arange(N) % 10]
    pl = Plan([784, 128, 10], [1, 1], [1, 0], 1, torch.float32, dev)
    pl.set_data(torch.tensor(x, device=dev), torch.tensor(y, device=dev))
    pl.set_prior(torch.zeros(pl.P), torch.ones(pl.P))
    th = 0.05 * pl.philox_normal(chains, seed=0, it=0)
    t, g = pl.log_target_grad(th)
    for b in range(burnin):
        pl.hmc_step(th, t, g, step, L, seed=1, it=1 + b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    accs = []
    e0.record()
    for it in range(iters):
        accs.append(pl.hmc_step(th, t, g, step, L, seed=1, it=1000 + it)["accepted"])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    f_step = 2 * N * (2 * (784 * 128 + 128 * 10) + 128 * 10) + 6 * pl.P
    tflops = f_step * chains * L / (ms * 1e-3) / 1e12
    out = {"workload": f"HMC L=20, {chains} chains, MLP(784-128-10) sigmoid-linear, CE-sum, prior N(0,1), MNIST-shaped synthetic "
                       f"N={N} (BASELINE configs[4], one GPU's temperature)", "kernel": pl.kernel, "ms_per_hmc_iteration": ms,
           "leapfrog_steps_per_sec_x_chains": chains * L / (ms * 1e-3), "flops_per_leapfrog_step_per_chain": f_step,
           "tflops": tflops, "frac_of_f32_mfma_peak": tflops / PEAK_F32_MFMA_TFLOPS, "f32_products": pl.f32_products,
           "step_size": step, "burnin_iterations": burnin, "timed_iterations": iters,
           "acceptance": round(float(torch.stack(accs).float().mean().item()), 4)}
    del pl, th, t, g
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--chains-per-gpu", type=int, default=CHAINS_PER_GPU)
    ap.add_argument("--iters-per-launch", type=int, default=25,
                    help="HMC iterations (bench steps) per kernel launch: ey_hmc_run, as HMC.run issues them; 1 = ey_hmc_step")
    ap.add_argument("--min-launches", type=int, default=5,
                    help="the timed region holds at least this many launches (of --iters-per-launch iterations each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-record", action="store_true",
                    help="do not record the chains in the timed region (round 3's figure: the kernel without the sample store)")
    ap.add_argument("--no-sampler-run", action="store_true", help="skip the same workload timed through samplers.HMC.run")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the secondary measurement of BASELINE configs[4]'s per-GPU share (N = 1 only, ~10 s)")
    ap.add_argument("--prewarm-seconds", type=float, default=1.0,
                    help="untimed launches before the timed region, on top of --warmup, until the clocks have settled")
    ap.add_argument("--force-generic", action="store_true", help="time the generic VALU kernel instead of the MFMA one")
    ap.add_argument("--no-compare", action="store_true",
                    help="do not also time the other form of the kernel's products (profiling runs: one kernel only)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / collectives only (no GPU needed, prints value null): the CPU-side test of "
                         "the multi-rank plumbing; never a measurement")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    from eeyore_amd import _lib as L
    from eeyore_amd.datasets import synthetic
    from eeyore_amd.distributed import ChainStats, init_from_env, reduce_ess
    from eeyore_amd.plan import Plan

    rank, world, local = init_from_env()
    local = local % max(1, torch.cuda.device_count())  # only differs in a gloo rehearsal on a smaller box
    gloo = world > 1 and dist.get_backend() == "gloo"
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, rank, world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    C = args.chains_per_gpu
    chain_offset = rank * C

    xs, ys = synthetic.iris_shaped_arrays(seed=0)
    sigma = float(np.sqrt(3.0))
    plan = Plan(DIMS, [1, 1, 1], [1, 1, 0], 1, torch.float32, dev)
    plan.set_data(torch.tensor(xs, dtype=torch.float32, device=dev), torch.tensor(ys, dtype=torch.float32, device=dev))
    plan.set_prior(torch.zeros(plan.P), torch.full((plan.P,), sigma))
    P = plan.P

    # theta0 = 0.1 N(0,1), seed 0, keyed by the global chain id so results do not depend on the GPU count
    theta = 0.1 * plan.philox_normal(C, seed=0, it=0, chain_offset=chain_offset)
    target, grad = plan.log_target_grad(theta)
    out = dict(accepted=plan.empty(C, dtype=torch.uint8), rate=plan.empty(C), h_cur=plan.empty(C), h_prop=plan.empty(C))
    stats = ChainStats(C, P, dev)
    flags = L.EY_FORCE_GENERIC if args.force_generic else 0
    seed = 2024

    ipl = max(1, args.iters_per_launch)
    if args.force_generic:
        ipl = 1  # the generic family replays attached moments from recorded samples; keep its one-step form

    # --steps rounded up to whole launches: the timed region is what HMC.run issues after burn-in -- and
    # to at least --min-launches of them: one 21 ms launch is a single sample of a quantity that moves by
    # +-10 % from launch to launch; `steps` in the line is what was timed, `config.steps_requested` what was asked for
    n_launches = max((args.steps + ipl - 1) // ipl, 1 if args.force_generic else max(1, args.min_launches))
    steps_timed = n_launches * ipl
    # the chain buffer of the timed region, [iterations, C, P] + targets + accept flags, as ChainBuffer holds a run
    # (200 steps x 4096 x 1315 floats = 4.3 GB), and one launch's worth for the untimed launches around it
    record = not args.no_record and not args.force_generic
    n_rec = steps_timed if record else 0
    rec = dict(s=plan.empty(max(n_rec, ipl), C, P), t=plan.empty(max(n_rec, ipl), C),
               a=plan.empty(max(n_rec, ipl), C, dtype=torch.uint8)) if record else None

    launch_events = None  # a list while the timed region runs: one event in front of every launch (and one behind the last)

    def steps(it, n, rec_at=None):
        """n bench steps = n HMC iterations of every chain, starting at iteration number `it`: whole launches of `ipl`
        iterations (what HMC.run does after burn-in), then the remainder.  Every launch records the chains' states,
        log-targets and accept flags: the timed region into consecutive blocks of the chain buffer starting at `rec_at`,
        the untimed launches into its first block."""
        done = 0
        while done < n:
            k = min(ipl, n - done)
            if launch_events is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                launch_events.append(ev)
            kw = {}
            if record:
                r0 = 0 if rec_at is None else rec_at + done
                kw = dict(samples=rec["s"][r0:r0 + k], targets=rec["t"][r0:r0 + k], accepted_rec=rec["a"][r0:r0 + k])
            if k == 1 and not record:
                plan.hmc_step(theta, target, grad, STEP_SIZE, L_STEPS, seed=seed, it=it + done,
                              chain_offset=chain_offset, flags=flags, out=out)
            else:
                plan.hmc_run(theta, target, grad, STEP_SIZE, L_STEPS, k, seed=seed, it=it + done,
                             chain_offset=chain_offset, flags=flags, out=out, **kw)
            done += k

    # the running chain moments behind the R-hat summary are accumulated by the step kernel itself (attached moments)
    stats.attach(plan)
    it = 1
    steps(it, args.warmup); it += args.warmup
    stats.summary()  # warms the torch elementwise kernels (and the RCCL communicator) of the summary below
    warm_stats, stats = stats, ChainStats(C, P, dev)  # the timed region's accumulators, allocated and zeroed now
    # Clock settling, independent of --warmup: the device drops its clocks whenever it idles for a few milliseconds
    # (the allocations and host work above) and ramps them over the first hundreds of milliseconds of load, so a 20-step
    # run (23 ms) would be timed on the ramp.  Untimed launches, the same as the timed ones, queued without a gap; the
    # timed region follows the last of them directly (one synchronize, one barrier, no other host work in between).
    # With several ranks the SAME number of settling launches is queued on every rank from a common start, so that
    # they reach the barrier in front of the timed region together (a rank that waited there for a slower one would
    # start its timed steps on an idle device's clock).
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_pre = time.perf_counter()
    for _ in range(4):
        steps(it, ipl); it += ipl
    torch.cuda.synchronize()
    batch_s = max(time.perf_counter() - t_pre, 1e-4)
    n_batches = max(0, int(np.ceil(args.prewarm_seconds / batch_s)) - 1)
    if world > 1:
        nb = torch.tensor([n_batches], dtype=torch.int64, device="cpu" if gloo else dev)
        dist.all_reduce(nb, op=dist.ReduceOp.MAX)
        n_batches = int(nb.item())
    for _ in range(4 * n_batches):
        steps(it, ipl); it += ipl
    torch.cuda.synchronize()
    stats.attach(plan)  # host-side only: no device work between the warm launches and the timed ones
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    launch_events = []
    steps(it, steps_timed, rec_at=0); it += steps_timed
    ev_end = torch.cuda.Event(enable_timing=True)
    ev_end.record()
    timed_events, launch_events = launch_events + [ev_end], None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # the gather of chain statistics (R-hat summary; one RCCL all-reduce of [3, P] partial sums when world > 1) happens
    # once per run, not per step: timed on its own and reported in config
    t1 = time.perf_counter()
    summ = stats.summary() if steps_timed > 1 else None
    torch.cuda.synchronize()
    summary_ms = 1e3 * (time.perf_counter() - t1)
    # ... and the effective sample sizes (configs[3]: "gather of R-hat/ESS"): every (chain, parameter) series of the
    # recorded buffer through ey_inse_univariate on this rank, then min / mean / total per parameter over all ranks
    # (two small all-reduces, distributed.reduce_ess); also outside the step clock, timed on its own
    ess = None
    if record and steps_timed >= 20:  # (the driver's --steps 20 becomes one launch of 25: a short series, but the gather runs)
        from eeyore_amd.stats import batched
        t2 = time.perf_counter()
        ess = reduce_ess(batched.ess(rec["s"][:steps_timed]))
        torch.cuda.synchronize()
        ess_ms = 1e3 * (time.perf_counter() - t2)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # dominant kernel, timed per launch with HIP events on the launch stream (torch's current stream)
    def launch_ms(n_ev=10):
        """One launch of ipl iterations of the dominant kernel ALONE: with the moments detached a recording launch does the
        same work (the moments of a recorded launch are not in the kernel: they come from ey_stats_update_run behind it,
        which the timed region above includes and this bracket must not)."""
        nonlocal it
        plan.detach_moments()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
        for a, b in evs:
            a.record(); steps(it, ipl); b.record(); it += ipl
        torch.cuda.synchronize()
        stats.attach(plan)
        return float(np.mean([a.elapsed_time(b) for a, b in evs]))

    kern_ms = launch_ms()
    # the same launches with the other form of the kernel's 32x32x32 products (EY_OPT_F32_PRODUCTS), for the record
    products = plan.f32_products if plan.kernel == "mfma32" else None
    other = None
    if products is not None and not args.force_generic and not args.no_compare:
        plan.f32_products = "exact" if products == "bf16x3" else "bf16x3"
        steps(it, ipl); it += ipl
        other = (plan.f32_products, launch_ms())
        plan.f32_products = products

    # the same workload through samplers.HMC.run -> ChainBuffer (the path a reference script takes), same start, own clock
    via_sampler = None
    if record and not args.no_sampler_run:
        theta0 = 0.1 * plan.philox_normal(C, seed=0, it=0, chain_offset=chain_offset)
        s_elapsed, s_acc = through_sampler_run(xs, ys, sigma, theta0, dev, steps_timed, ipl, seed, chain_offset, world, gloo)
        via_sampler = (s_elapsed, s_acc)

    launch_gaps = [a.elapsed_time(b) for a, b in zip(timed_events[:-1], timed_events[1:])]
    if rank == 0:
        f_step = flops_per_leapfrog_step(DIMS, N_ROWS)
        total_chains = C * world
        value = total_chains * L_STEPS * steps_timed / elapsed
        achieved_tflops = f_step * L_STEPS * C * ipl / (kern_ms * 1e-3) / 1e12
        line = {
            "metric": "leapfrog-steps/sec x chains, HMC MLP(4-32-32-3)",
            "value": value,
            "unit": "leapfrog-steps/sec x chains",
            "n_gpus": world,
            "steps": steps_timed,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps_timed,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (bf16x3 products, f32 accumulate)" if products == "bf16x3" and not args.force_generic else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"HMC L={L_STEPS}, {C} chains/GPU ({total_chains} total), MLP(4-32-32-3) sigmoid-sigmoid-"
                            f"linear, CE-sum, prior N(0,sqrt3), Iris-shaped synthetic N={N_ROWS} (BASELINE configs[2]"
                            f"{'; configs[3] sharding' if world > 1 else ''})",
                "chains_per_gpu": C, "num_steps": L_STEPS, "step_size": STEP_SIZE, "rng": "in-kernel Philox4x32-10",
                "kernel": "generic" if args.force_generic else plan.kernel,
                "f32_products": None if args.force_generic else products,
                "gradient_evaluations_per_iteration": L_STEPS, "iterations_per_launch": ipl,
                "steps_requested": args.steps,
                "steps_note": (f"--steps {args.steps} rounded up to {steps_timed} = {steps_timed // ipl} whole launch(es) of {ipl} "
                               f"iterations, as HMC.run issues them (at least --min-launches {args.min_launches}); `steps` and "
                               f"`ms_per_step` are those of the timed region") if steps_timed != args.steps else "whole launches",
                "launches_timed": steps_timed // ipl,
                "launch_ms_min": round(min(launch_gaps), 4), "launch_ms_median": round(float(np.median(launch_gaps)), 4),
                "launch_ms_max": round(max(launch_gaps), 4),
                "launch_ms_note": "start-to-start of consecutive launches inside the timed region (HIP events on the launch stream)",
                "recorded_in_timed_region": ("samples [steps, C, P], targets [steps, C], accepted [steps, C] (the chain buffer "
                                             "HMC.run fills after burn-in)") if record else None,
                "stats_summary_ms": round(summary_ms, 3),
                "value_including_stats_summary": total_chains * L_STEPS * steps_timed / (elapsed + 1e-3 * summary_ms),
                "acceptance": None if summ is None else round(summ["acceptance"], 4),
                "rhat_max": None if summ is None else float(summ["rhat"].max().item()),
            },
            "roofline": {
                "bound": "mfma", "achieved": achieved_tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tflops / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                "kernel_ms": kern_ms, "leapfrog_steps_per_launch": C * L_STEPS * ipl,
                "flops_per_leapfrog_step_per_chain": f_step,
                "peak_measured": MEASURED_F32_MFMA_TFLOPS, "frac_of_measured": achieved_tflops / MEASURED_F32_MFMA_TFLOPS,
                # what the matrix pipes sustain alone at the socket's power limit (tools/mfma_power_probe.hip, operands whose bits
                # change; profiles/r05_mfma_power_probe.txt, r05_clock_power.txt: this kernel runs at 1 360 W and 2.36 GHz)
                "sustained_at_power_limit": {"f32_mfma_tflops": 139.5, "bf16_mfma_tflops": [1731.7, 1814.6]},
            },
        }
        if ess is not None:
            line["config"]["ess"] = {"summary_ms": round(ess_ms, 3), "min_over_chains_and_parameters": float(ess["min"].min().item()),
                                     "mean": float(ess["mean"].mean().item()), "total_per_parameter_mean": float(ess["total"].mean().item()),
                                     "num_chains": ess["num_chains"], "series_without_enough_samples": ess["not_enough"],
                                     "from": f"{steps_timed} recorded iterations of every chain (ey_inse_univariate), combined "
                                             f"over ranks by distributed.reduce_ess"}
        if world == 1 and not args.no_config5 and not args.force_generic and not args.no_compare:
            try:  # a secondary figure: never in the way of the headline line
                line["config"]["secondary"] = {"config5_share_one_gpu": config5_share(dev)}
            except Exception as e:  # noqa: BLE001
                line["config"]["secondary"] = {"config5_share_one_gpu": {"error": repr(e)[:200]}}
        if via_sampler is not None:
            s_value = total_chains * L_STEPS * steps_timed / via_sampler[0]
            line["config"]["through_sampler_run"] = {
                "value": s_value, "unit": "leapfrog-steps/sec x chains", "seconds": via_sampler[0], "iterations": steps_timed,
                "ratio_to_headline": s_value / value, "acceptance": round(via_sampler[1], 4),
                "path": "eeyore_amd.samplers.HMC(model, theta0 [C, P], ...).run(num_epochs, 0) -> ChainBuffer [iters, C, P]"}
        if other is not None:
            o_tflops = f_step * L_STEPS * C * ipl / (other[1] * 1e-3) / 1e12
            line["config"]["kernels"] = {
                products: {"kernel_ms": kern_ms, "tflops": achieved_tflops, "frac": achieved_tflops / PEAK_F32_MFMA_TFLOPS,
                           "leapfrog_steps_per_sec_x_chains": C * L_STEPS * ipl / (kern_ms * 1e-3)},
                other[0]: {"kernel_ms": other[1], "tflops": o_tflops, "frac": o_tflops / PEAK_F32_MFMA_TFLOPS,
                           "leapfrog_steps_per_sec_x_chains": C * L_STEPS * ipl / (other[1] * 1e-3)},
            }
        if products == "bf16x3" and not args.force_generic:
            # The f32-equivalent fraction above prices the ALGORITHMIC f32 flops against the f32 matrix peak.  Beside it, how
            # much of the kernel's time each matrix pipe is issuing (per SIMD, at the nominal 2.4 GHz): per 32-row tile 36
            # v_mfma_f32_32x32x16_bf16 (32 cycles each), 2 v_mfma_f32_32x32x2_f32 (64) and 60 v_mfma_f32_4x4x1_16b_f32 (8.4)
            tiles = (N_ROWS + 31) // 32
            simd_cycles = 1024 * 2.4e9 * kern_ms * 1e-3
            per = C * L_STEPS * ipl * tiles
            line["roofline"]["f32_equivalent"] = True
            line["roofline"]["frac_is"] = ("f32-equivalent: algorithmic f32 flops / f32 MFMA peak, measured on the default kernel, "
                                           "whose three 32x32x32 products per tile issue on the bf16 pipe (exact 3-piece split of "
                                           "each f32 operand, f32 accumulate); frac_exact_f32_products is the like-for-like figure")
            if other is not None:
                line["roofline"]["frac_exact_f32_products"] = o_tflops / PEAK_F32_MFMA_TFLOPS
            # what the bf16 pipe itself is asked for: 36 MFMAs of 2 x 32 x 32 x 16 flop per tile against the dense bf16 peak
            line["roofline"]["bf16_pipe_flops_frac"] = per * 36 * 32768 / (kern_ms * 1e-3) / PEAK_BF16_MFMA_FLOPS
            line["roofline"]["bf16_mfma_issue_frac"] = per * 36 * 32 / simd_cycles
            line["roofline"]["f32_mfma_issue_frac"] = per * (2 * 64 + 60 * 8.4) / simd_cycles

        # HBM traffic of the dominant kernel from the committed PMC passes (tools/pmc_passes.sh; separate --pmc runs,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950); bench.py cannot profile itself
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc) and not args.force_generic and C == CHAINS_PER_GPU:
            with open(pmc) as f:
                pm = json.load(f)
            # only counters collected on THIS kernel source count (tools/pmc_passes.sh records its hash)
            if pm.get("kernel", "").startswith("k_mfma32") and pm.get("kernel_source_sha256") == kernel_source_hash():
                # per launch of `ipl` iterations (the PMC run's dispatches held pm["iterations_per_launch"] each)
                line["roofline"]["traffic"] = ((2.0 * pm["FETCH_SIZE_KB"] + pm["WRITE_SIZE_KB"]) * 1024.0 * ipl
                                               / pm.get("iterations_per_launch", 1))
                line["roofline"]["traffic_source"] = pm.get("source", "profiles/pmc_latest.json")
                if "behind_every_launch" in pm:  # the moments pass over the records (ey_stats_update_run), per launch
                    b = pm["behind_every_launch"]
                    line["roofline"]["traffic_of_the_moments_pass_behind_each_launch"] = (2.0 * b["FETCH_SIZE_KB"] + b["WRITE_SIZE_KB"]) * 1024.0
                line["roofline"]["traffic_algorithmic"] = {
                    "survey_8d_bytes_per_launch": (2 * P + P) * 4 * C * ipl,
                    "note": "SURVEY 8(d): (2 P + P_store) * 4 / L bytes per leapfrog step per chain = theta and gradient in, the "
                            "recorded sample out, per iteration; an accepted draw also writes theta and the gradient back"}
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is an N = 1 figure (the other ranks would wait for it)
            # BASELINE.md section 3 item 3, "generous CPU": one chain per hardware thread on ALL threads this process may
            # use; beside it the 16 threads that are a one-GPU box's share of the host, and the reference-faithful path
            # (a box may show all 256 threads of its host and schedule this job on a share of them -- measured on the pool:
            # 256 threads 2.3e4, 16 threads 7.4e4 -- so both are timed and the FASTER one is the baseline, the other kept)
            ncpu = usable_cpus()
            base = cpu_baseline(xs, ys, sigma, ncpu)
            if ncpu > 16:
                share = cpu_baseline(xs, ys, sigma, 16, budget_s=5.0)
                if share["value"] > base["value"]:
                    base, share = share, base
                base["other_thread_count"] = {k: share[k] for k in ("value", "cores", "sample")}
            line["cpu_baseline"] = base
            line["cpu_baseline"]["reference_faithful"] = cpu_reference_faithful(xs, ys, sigma, budget_s=6.0)
            try:  # the same for BASELINE configs[0], CPU and device side by side (secondary figures)
                line["cpu_baseline"]["reference_faithful_config1"] = cpu_reference_faithful_config1()
                line["cpu_baseline"]["reference_faithful_config1"]["gpu_same_config"] = gpu_config1(dev)
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"]["reference_faithful_config1"] = {"error": repr(e)[:200]}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
