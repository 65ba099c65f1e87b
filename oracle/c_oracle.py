"""ctypes binding of the C oracle (oracle/mlp_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as ct
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("EEYORE_ORACLE_LIB", os.path.join(HERE, "_build", "liboracle.so"))  # override: the sanitizer build
OC_MAX_LAYERS = 8


class OcSpec(ct.Structure):
    _fields_ = [("nl", ct.c_int), ("dims", ct.c_int * (OC_MAX_LAYERS + 1)), ("bias", ct.c_int * OC_MAX_LAYERS),
                ("acts", ct.c_int * OC_MAX_LAYERS), ("lik", ct.c_int)]


def build(force=False):
    src = [os.path.join(HERE, f) for f in ("mlp_oracle.c", "mlp_oracle_impl.h")]
    if "EEYORE_ORACLE_LIB" in os.environ:
        return LIB  # a build someone else made (tests/test_sanitizers.py)
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        subprocess.check_call(["make", "-C", HERE, "-B"], stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ct.CDLL(build())
        _lib.oc_num_params.restype = ct.c_int
        _lib.oc_work_size.restype = ct.c_int
        _lib.oc_f64_log_target_grad.restype = ct.c_double
        _lib.oc_f32_log_target_grad.restype = ct.c_float
    return _lib


def make_spec(dims, acts, lik, bias=None):
    s = OcSpec()
    s.nl = len(dims) - 1
    for i, d in enumerate(dims):
        s.dims[i] = int(d)
    for i in range(s.nl):
        s.acts[i] = int(acts[i])
        s.bias[i] = 1 if bias is None else int(bias[i])
    s.lik = int(lik)
    return s


def _p(a):
    return a.ctypes.data_as(ct.c_void_p) if a is not None else None


class COracle:
    """Holds one model/dataset/prior in the chosen dtype and exposes the chain-batched steps."""

    def __init__(self, dims, acts, lik, x, y, mu, sigma, dtype=np.float64, bias=None, temperature=None, nthreads=1):
        self.spec = make_spec(dims, acts, lik, bias)
        self.dt = np.dtype(dtype)
        self.sfx = "f64" if self.dt == np.float64 else "f32"
        self.x = np.ascontiguousarray(x, dtype=self.dt)
        self.y = np.ascontiguousarray(y, dtype=self.dt)
        self.N = self.x.shape[0]
        self.P = lib().oc_num_params(ct.byref(self.spec))
        self.mu = np.ascontiguousarray(np.broadcast_to(mu, (self.P,)), dtype=self.dt)
        self.sigma = np.ascontiguousarray(np.broadcast_to(sigma, (self.P,)), dtype=self.dt)
        self.temp = float("nan") if temperature is None else float(temperature)
        self.nthreads = nthreads
        self.work = np.zeros(lib().oc_work_size(ct.byref(self.spec)), dtype=self.dt)

    def _fn(self, name):
        return getattr(lib(), f"oc_{self.sfx}_{name}")

    def log_target_grad(self, theta, want_grad=True):
        theta = np.ascontiguousarray(theta, dtype=self.dt)
        grad = np.zeros(self.P, dtype=self.dt) if want_grad else None
        lik = np.zeros(1, dtype=self.dt)
        prior = np.zeros(1, dtype=self.dt)
        t = self._fn("log_target_grad")(ct.byref(self.spec), _p(theta), _p(self.x), _p(self.y), ct.c_int(self.N),
                                        _p(self.mu), _p(self.sigma), ct.c_double(self.temp), _p(grad), _p(lik),
                                        _p(prior), _p(self.work))
        return self.dt.type(t), grad, lik[0], prior[0]

    def leapfrog(self, theta0, p0, step, L):
        th = np.array(theta0, dtype=self.dt)
        p = np.array(p0, dtype=self.dt)
        g = np.zeros(self.P, dtype=self.dt)
        t = np.zeros(1, dtype=self.dt)
        self._fn("leapfrog")(ct.byref(self.spec), _p(th), _p(p), _p(self.x), _p(self.y), ct.c_int(self.N), _p(self.mu),
                             _p(self.sigma), ct.c_double(self.temp), ct.c_double(step), ct.c_int(L), _p(t), _p(g),
                             _p(self.work))
        return th, p, t[0], g

    def hmc_draw(self, theta, target, grad, p0, u, step, L):
        """In-place on theta [C,P], target [C], grad [C,P]. Returns (accepted[C] uint8, h_cur, h_prop)."""
        C = theta.shape[0]
        acc = np.zeros(C, dtype=np.uint8)
        hc = np.zeros(C, dtype=self.dt)
        hp = np.zeros(C, dtype=self.dt)
        for a in (theta, target, grad, p0, u):
            assert a.dtype == self.dt and a.flags.c_contiguous
        self._fn("hmc_draw_chains")(ct.byref(self.spec), ct.c_int(C), _p(theta), _p(target), _p(grad), _p(p0), _p(u),
                                    _p(self.x), _p(self.y), ct.c_int(self.N), _p(self.mu), _p(self.sigma),
                                    ct.c_double(self.temp), ct.c_double(step), ct.c_int(L), _p(acc), _p(hc), _p(hp),
                                    ct.c_int(self.nthreads))
        return acc, hc, hp

    def mala_draw(self, theta, target, grad, z, u, step):
        C = theta.shape[0]
        acc = np.zeros(C, dtype=np.uint8)
        lr = np.zeros(C, dtype=self.dt)
        for a in (theta, target, grad, z, u):
            assert a.dtype == self.dt and a.flags.c_contiguous
        self._fn("mala_draw_chains")(ct.byref(self.spec), ct.c_int(C), _p(theta), _p(target), _p(grad), _p(z), _p(u),
                                     _p(self.x), _p(self.y), ct.c_int(self.N), _p(self.mu), _p(self.sigma),
                                     ct.c_double(self.temp), ct.c_double(step), _p(acc), _p(lr), ct.c_int(self.nthreads))
        return acc, lr

    def mh_draw(self, theta, target, z, u, scale):
        C = theta.shape[0]
        acc = np.zeros(C, dtype=np.uint8)
        lr = np.zeros(C, dtype=self.dt)
        scale = np.ascontiguousarray(np.broadcast_to(scale, (self.P,)), dtype=self.dt)
        for a in (theta, target, z, u):
            assert a.dtype == self.dt and a.flags.c_contiguous
        self._fn("mh_draw_chains")(ct.byref(self.spec), ct.c_int(C), _p(theta), _p(target), _p(z), _p(u), _p(scale),
                                   _p(self.x), _p(self.y), ct.c_int(self.N), _p(self.mu), _p(self.sigma),
                                   ct.c_double(self.temp), _p(acc), _p(lr), ct.c_int(self.nthreads))
        return acc, lr
