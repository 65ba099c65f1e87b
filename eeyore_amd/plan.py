"""Plan: the Python handle of a C-ABI ``ey_plan`` (include/eeyore_amd.h) operating on torch device tensors.

PyTorch is plumbing here (device memory, streams); all arithmetic of the hot path happens in the HIP library.
A plan fixes the model (dims/bias/activations/likelihood/dtype); data and prior are attached to it.
"""
import ctypes as ct

import torch

from . import _lib as L

_DT = {torch.float32: L.EY_F32, torch.float64: L.EY_F64}


def _stream(device):
    return ct.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Plan:
    def __init__(self, dims, bias, acts, likelihood, dtype, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(
                f"eeyore_amd: the hot path runs on an MI355X through the HIP library; device '{device}' is not a ROCm "
                "device and there is no CPU fallback")
        if dtype not in _DT:
            raise ValueError(f"unsupported dtype {dtype}")
        self.dtype = dtype
        self.dims = [int(d) for d in dims]
        n = len(self.dims) - 1
        self.handle = ct.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        L.check(L.lib().ey_plan_create(ct.byref(self.handle), n, (ct.c_int * (n + 1))(*self.dims),
                                       (ct.c_int * n)(*[int(b) for b in bias]), (ct.c_int * n)(*[int(a) for a in acts]),
                                       int(likelihood), _DT[dtype], idx), "ey_plan_create")
        P = ct.c_int64()
        L.check(L.lib().ey_plan_num_params(self.handle, ct.byref(P)), "ey_plan_num_params")
        self.P = P.value
        self._data_key, self._data_ref = None, (None, None)
        self._prior_key = None
        self._moments = None
        self._da_refs = None  # tensors an attached dual averaging points into: alive for as long as it is attached

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                L.lib().ey_plan_destroy(self.handle)
                self.handle = ct.c_void_p()
        except Exception:
            pass

    @property
    def kernel(self):
        return L.lib().ey_plan_kernel(self.handle).decode()

    @property
    def f32_products(self):
        """'bf16x3' or 'exact': how the fused f32 trajectory kernel forms its 32x32x32 products (EY_OPT_F32_PRODUCTS in
        include/eeyore_amd.h); only plans that kernel serves are affected."""
        v = ct.c_int()
        L.check(L.lib().ey_plan_get_option(self.handle, L.EY_OPT_F32_PRODUCTS, ct.byref(v)), "ey_plan_get_option")
        return "exact" if v.value == L.EY_PRODUCTS_EXACT else "bf16x3"

    @f32_products.setter
    def f32_products(self, mode):
        if mode not in ("bf16x3", "exact"):
            raise ValueError("f32_products must be 'bf16x3' or 'exact'")
        L.check(L.lib().ey_plan_set_option(self.handle, L.EY_OPT_F32_PRODUCTS,
                                           L.EY_PRODUCTS_EXACT if mode == "exact" else L.EY_PRODUCTS_BF16X3),
                "ey_plan_set_option")

    @property
    def row_waves(self):
        """'off', 'on' or 'auto': several waves per chain for tiny models on batches of two row tiles or more
        (EY_OPT_ROW_WAVES in include/eeyore_amd.h: a latency option that changes the order of the gradient sums).  'off' is
        the default: a chain's bits then do not depend on how many chains share its launch; 'auto' opts in to the waves
        whenever the launch would leave the chip idle."""
        v = ct.c_int()
        L.check(L.lib().ey_plan_get_option(self.handle, L.EY_OPT_ROW_WAVES, ct.byref(v)), "ey_plan_get_option")
        return ("off", "on", "auto")[v.value]

    @row_waves.setter
    def row_waves(self, mode):
        if mode not in ("off", "on", "auto"):
            raise ValueError("row_waves must be 'off', 'on' or 'auto'")
        L.check(L.lib().ey_plan_set_option(self.handle, L.EY_OPT_ROW_WAVES, ("off", "on", "auto").index(mode)),
                "ey_plan_set_option")

    def set_variant(self, variant):
        """Diagnostic switches of THIS plan (ey_plan_set_variant); returns the previous value."""
        return L.lib().ey_plan_set_variant(self.handle, int(variant))

    # ------------------------------------------------------------------ data / prior
    def _prep(self, t, shape=None):
        t = torch.as_tensor(t)
        t = t.to(device=self.device, dtype=self.dtype).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def set_data(self, x, y):
        """Attach the batch (asynchronous on the current stream: device-to-device copies and two small kernels).
        The upload is skipped only when the SAME tensor objects, unmodified (``_version``), are passed again: the plan
        keeps references to them, so their storage cannot be recycled for another batch while the key is live (an
        address-based key matches a freed temporary's successor and would silently keep the old data)."""
        key = (x._version, y._version)
        if self._data_key is not None and self._data_ref[0] is x and self._data_ref[1] is y and key == self._data_key:
            return
        xd = self._prep(x)
        yd = self._prep(y)
        if xd.dim() != 2 or xd.shape[1] != self.dims[0]:
            raise ValueError(f"x must be [N, {self.dims[0]}], got {tuple(xd.shape)}")
        yd = yd.reshape(xd.shape[0], -1)
        if yd.shape[1] != self.dims[-1]:
            raise ValueError(f"y must be [N, {self.dims[-1]}], got {tuple(yd.shape)}")
        L.check(L.lib().ey_plan_set_data(self.handle, L.ptr(xd), L.ptr(yd), xd.shape[0], _stream(self.device)),
                "ey_plan_set_data")
        self._data_key, self._data_ref = key, (x, y)
        self.N = xd.shape[0]

    def set_prior(self, mu, sigma):
        mu = self._prep(torch.broadcast_to(torch.as_tensor(mu), (self.P,)), (self.P,))
        sigma = self._prep(torch.broadcast_to(torch.as_tensor(sigma), (self.P,)), (self.P,))
        L.check(L.lib().ey_plan_set_prior(self.handle, L.ptr(mu), L.ptr(sigma), _stream(self.device)),
                "ey_plan_set_prior")

    # ------------------------------------------------------------------ helpers
    def _theta(self, theta):
        if theta.device != self.device or theta.dtype != self.dtype or not theta.is_contiguous():
            raise ValueError("theta must be a contiguous tensor of the plan's dtype on the plan's device")
        if theta.dim() != 2 or theta.shape[1] != self.P:
            raise ValueError(f"theta must be [C, {self.P}]")
        return theta.shape[0]

    def _opt(self, t, C):
        if t is None:
            return None
        t = torch.as_tensor(t, dtype=self.dtype, device=self.device)
        if t.dim() == 0:
            t = t.expand(C)
        return t.contiguous()

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dtype, device=self.device)

    # ------------------------------------------------------------------ compute
    def log_target(self, theta, temp=None, prior_only=False):
        C = self._theta(theta)
        lik, prior = (None if prior_only else self.empty(C)), self.empty(C)
        temp = self._opt(temp, C)
        L.check(L.lib().ey_log_target(self.handle, L.ptr(theta), L.ptr(temp), C, L.ptr(lik), L.ptr(prior),
                                      _stream(self.device)), "ey_log_target")
        return lik, prior

    def log_lik_rows(self, theta, temp=None):
        """[C, N]: the log-likelihood term of every data row under every chain's parameters (ey_log_lik_rows)."""
        C = self._theta(theta)
        rows = self.empty(C, self.N)
        temp = self._opt(temp, C)
        L.check(L.lib().ey_log_lik_rows(self.handle, L.ptr(theta), L.ptr(temp), C, L.ptr(rows), _stream(self.device)),
                "ey_log_lik_rows")
        return rows

    def log_target_grad(self, theta, temp=None):
        C = self._theta(theta)
        target, grad = self.empty(C), self.empty(C, self.P)
        temp = self._opt(temp, C)
        L.check(L.lib().ey_log_target_grad(self.handle, L.ptr(theta), L.ptr(temp), C, L.ptr(target), L.ptr(grad),
                                           _stream(self.device)), "ey_log_target_grad")
        return target, grad

    # ------------------------------------------------------------------ attached dual averaging
    def attach_da(self, state, step_vec, table, n, d, log_eub=None, final_avg=True):
        """ey_plan_attach_da; the plan keeps the three tensors alive until ``detach_da`` (the library holds raw device
        pointers into them and writes the step and the state after every adapting iteration)."""
        L.check(L.lib().ey_plan_attach_da(self.handle, L.ptr(state), L.ptr(step_vec), L.ptr(table), int(n),
                                          int(step_vec.shape[0]), float(d),
                                          float('nan') if log_eub is None else float(log_eub), int(bool(final_avg))),
                "ey_plan_attach_da")
        self._da_refs = (state, step_vec, table)

    def detach_da(self):
        """Drop whatever dual averaging is attached (a no-op when none is): safe to call before every run."""
        L.check(L.lib().ey_plan_attach_da(self.handle, None, None, None, 0, 0, 0.5, float('nan'), 0),
                "ey_plan_attach_da")
        self._da_refs = None

    # ------------------------------------------------------------------ attached running moments
    def attach_moments(self, s1, s2, acc, on_step=None):
        """From now on every hmc_step / mala_step / mh_step also adds the state each chain is left in to the double
        accumulators s1, s2 [C, P] and its accept flag to acc [C] (ey_plan_attach_moments): inside the fused kernel where
        there is one.  ``on_step`` is called after each such step (e.g. to count iterations)."""
        for t in (s1, s2, acc):
            if t.dtype != torch.float64 or not t.is_contiguous() or t.device != self.device:
                raise ValueError("moment accumulators must be contiguous float64 tensors on the plan's device")
        C = s1.shape[0]
        if tuple(s1.shape) != (C, self.P) or tuple(s2.shape) != (C, self.P) or tuple(acc.shape) != (C,):
            raise ValueError(f"expected s1, s2 [C, {self.P}] and acc [C]")
        L.check(L.lib().ey_plan_attach_moments(self.handle, L.ptr(s1), L.ptr(s2), L.ptr(acc), C),
                "ey_plan_attach_moments")
        self._moments = (s1, s2, acc, on_step)  # keeps the tensors alive

    def detach_moments(self):
        L.check(L.lib().ey_plan_attach_moments(self.handle, None, None, None, 0), "ey_plan_attach_moments")
        self._moments = None

    def _stepped(self, times=1):
        if self._moments is not None and self._moments[3] is not None:
            for _ in range(times):
                self._moments[3]()

    def hmc_step(self, theta, target, grad, step, num_steps, p0=None, u=None, step_vec=None, temp=None, seed=0, it=0,
                 chain_offset=0, flags=0, out=None):
        C = self._theta(theta)
        if out is None:
            out = dict(accepted=self.empty(C, dtype=torch.uint8), rate=self.empty(C), h_cur=self.empty(C),
                       h_prop=self.empty(C))
        temp, step_vec, u = self._opt(temp, C), self._opt(step_vec, C), self._opt(u, C)
        L.check(L.lib().ey_hmc_step(self.handle, L.ptr(theta), L.ptr(target), L.ptr(grad), L.ptr(p0), L.ptr(u),
                                    float(step), L.ptr(step_vec), int(num_steps), L.ptr(temp), C, int(seed), int(it),
                                    int(chain_offset), int(flags), L.ptr(out["accepted"]), L.ptr(out["rate"]),
                                    L.ptr(out["h_cur"]), L.ptr(out["h_prop"]), _stream(self.device)), "ey_hmc_step")
        self._stepped()
        return out

    def hmc_run(self, theta, target, grad, step, num_steps, n_iters, step_vec=None, temp=None, seed=0, it=0,
                chain_offset=0, flags=0, samples=None, targets=None, accepted_rec=None, accept_count=None, out=None):
        """``n_iters`` HMC iterations (it, it + 1, ...) of every chain in ONE launch (ey_hmc_run): the in-kernel Philox
        streams only, bit-identical to ``n_iters`` calls of ``hmc_step`` without ``p0`` / ``u``.  Optional records of the
        state after each iteration: ``samples`` [n_iters, C, P], ``targets`` [n_iters, C], ``accepted_rec`` [n_iters, C]
        uint8 (contiguous views, e.g. slices of a ChainBuffer's storage); ``accept_count`` [C] int32 is incremented."""
        C = self._theta(theta)
        if out is None:
            out = dict(accepted=self.empty(C, dtype=torch.uint8))
        temp, step_vec = self._opt(temp, C), self._opt(step_vec, C)
        n_iters = int(n_iters)
        self._records(n_iters, C, samples, targets, accepted_rec, accept_count)
        L.check(L.lib().ey_hmc_run(self.handle, L.ptr(theta), L.ptr(target), L.ptr(grad), float(step), L.ptr(step_vec),
                                   int(num_steps), L.ptr(temp), C, int(seed), int(it), int(chain_offset), int(flags),
                                   n_iters, L.ptr(samples), L.ptr(targets), L.ptr(accepted_rec), L.ptr(accept_count),
                                   L.ptr(out["accepted"]), _stream(self.device)), "ey_hmc_run")
        self._stepped(n_iters)
        return out

    def _records(self, n_iters, C, samples, targets, accepted_rec, accept_count):
        for name, t, shape, dt in (("samples", samples, (n_iters, C, self.P), self.dtype),
                                   ("targets", targets, (n_iters, C), self.dtype),
                                   ("accepted_rec", accepted_rec, (n_iters, C), torch.uint8),
                                   ("accept_count", accept_count, (C,), torch.int32)):
            if t is not None and (tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous()
                                  or t.device != self.device):
                raise ValueError(f"{name} must be a contiguous {dt} tensor of shape {shape} on the plan's device")

    def mala_run(self, theta, target, grad, step, n_iters, step_vec=None, temp=None, seed=0, it=0, chain_offset=0,
                 flags=0, samples=None, targets=None, accepted_rec=None, accept_count=None, out=None):
        """``n_iters`` MALA iterations of every chain in one launch (ey_mala_run); records as in ``hmc_run``."""
        C = self._theta(theta)
        if out is None:
            out = dict(accepted=self.empty(C, dtype=torch.uint8))
        temp, step_vec = self._opt(temp, C), self._opt(step_vec, C)
        n_iters = int(n_iters)
        self._records(n_iters, C, samples, targets, accepted_rec, accept_count)
        L.check(L.lib().ey_mala_run(self.handle, L.ptr(theta), L.ptr(target), L.ptr(grad), float(step), L.ptr(step_vec),
                                    L.ptr(temp), C, int(seed), int(it), int(chain_offset), int(flags), n_iters,
                                    L.ptr(samples), L.ptr(targets), L.ptr(accepted_rec), L.ptr(accept_count),
                                    L.ptr(out["accepted"]), _stream(self.device)), "ey_mala_run")
        self._stepped(n_iters)
        return out

    def mh_run(self, theta, target, scale, n_iters, temp=None, seed=0, it=0, chain_offset=0, flags=0, samples=None,
               targets=None, accepted_rec=None, accept_count=None, out=None):
        """``n_iters`` random-walk MH iterations of every chain in one launch (ey_mh_run); records as in ``hmc_run``."""
        C = self._theta(theta)
        if out is None:
            out = dict(accepted=self.empty(C, dtype=torch.uint8))
        scale = self._prep(torch.broadcast_to(torch.as_tensor(scale, dtype=self.dtype, device=self.device), (self.P,)))
        temp = self._opt(temp, C)
        n_iters = int(n_iters)
        self._records(n_iters, C, samples, targets, accepted_rec, accept_count)
        L.check(L.lib().ey_mh_run(self.handle, L.ptr(theta), L.ptr(target), L.ptr(scale), L.ptr(temp), C, int(seed),
                                  int(it), int(chain_offset), int(flags), n_iters, L.ptr(samples), L.ptr(targets),
                                  L.ptr(accepted_rec), L.ptr(accept_count), L.ptr(out["accepted"]),
                                  _stream(self.device)), "ey_mh_run")
        self._stepped(n_iters)
        return out

    def leapfrog(self, theta, p, step, num_steps, step_vec=None, temp=None):
        """HMC.leapfrog (hmc.py:100-124) in place on theta [C,P], p [C,P]; returns (target [C], grad [C,P])."""
        C = self._theta(theta)
        self._theta(p)
        target, grad = self.empty(C), self.empty(C, self.P)
        temp, step_vec = self._opt(temp, C), self._opt(step_vec, C)
        L.check(L.lib().ey_hmc_leapfrog(self.handle, L.ptr(theta), L.ptr(p), float(step), L.ptr(step_vec),
                                        int(num_steps), L.ptr(temp), C, L.ptr(target), L.ptr(grad),
                                        _stream(self.device)), "ey_hmc_leapfrog")
        return target, grad

    def mala_step(self, theta, target, grad, step, z=None, u=None, step_vec=None, temp=None, seed=0, it=0,
                  chain_offset=0, flags=0, out=None):
        C = self._theta(theta)
        if out is None:
            out = dict(accepted=self.empty(C, dtype=torch.uint8), log_rate=self.empty(C))
        temp, step_vec, u = self._opt(temp, C), self._opt(step_vec, C), self._opt(u, C)
        L.check(L.lib().ey_mala_step(self.handle, L.ptr(theta), L.ptr(target), L.ptr(grad), L.ptr(z), L.ptr(u),
                                     float(step), L.ptr(step_vec), L.ptr(temp), C, int(seed), int(it),
                                     int(chain_offset), int(flags), L.ptr(out["accepted"]), L.ptr(out["log_rate"]),
                                     _stream(self.device)), "ey_mala_step")
        self._stepped()
        return out

    def mh_step(self, theta, target, scale, z=None, u=None, temp=None, seed=0, it=0, chain_offset=0, flags=0, out=None):
        C = self._theta(theta)
        if out is None:
            out = dict(accepted=self.empty(C, dtype=torch.uint8), log_rate=self.empty(C))
        scale = self._prep(torch.broadcast_to(torch.as_tensor(scale, dtype=self.dtype, device=self.device), (self.P,)))
        temp, u = self._opt(temp, C), self._opt(u, C)
        L.check(L.lib().ey_mh_step(self.handle, L.ptr(theta), L.ptr(target), L.ptr(z), L.ptr(u), L.ptr(scale),
                                   L.ptr(temp), C, int(seed), int(it), int(chain_offset), int(flags),
                                   L.ptr(out["accepted"]), L.ptr(out["log_rate"]), _stream(self.device)), "ey_mh_step")
        self._stepped()
        return out

    def pt_swap_decide(self, ell_i, ell_j, t_i, t_j, u, dlogq=None):
        return pt_swap_decide(ell_i, ell_j, t_i, t_j, u, dlogq=dlogq)

    def philox_normal(self, C, seed, it, chain_offset=0):
        out = self.empty(C, self.P)
        L.check(L.lib().ey_philox_normal(L.ptr(out), C, self.P, int(seed), int(it), int(chain_offset), _DT[self.dtype],
                                         _stream(self.device)), "ey_philox_normal")
        return out

    def philox_uniform(self, C, seed, it, chain_offset=0):
        out = self.empty(C)
        L.check(L.lib().ey_philox_uniform(L.ptr(out), C, int(seed), int(it), int(chain_offset), _DT[self.dtype],
                                          _stream(self.device)), "ey_philox_uniform")
        return out


def pt_swap_decide(ell_i, ell_j, t_i, t_j, u, dlogq=None):
    """PowerPosteriorSampler.between_chain_move decision (power_posterior_sampler.py:135-163) for C pairs."""
    C = ell_i.shape[0]
    dt = ell_i.dtype
    dev = ell_i.device
    args = [a.to(device=dev, dtype=dt).contiguous() if a is not None else None for a in (ell_i, ell_j, t_i, t_j, dlogq, u)]
    swap = torch.empty(C, dtype=torch.uint8, device=dev)
    log_rate = torch.empty(C, dtype=dt, device=dev)
    L.check(L.lib().ey_pt_swap_decide(*[L.ptr(a) for a in args], C, _DT[dt], L.ptr(swap), L.ptr(log_rate), _stream(dev)),
            "ey_pt_swap_decide")
    return swap, log_rate
