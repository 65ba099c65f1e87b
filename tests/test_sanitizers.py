"""CPU-side sanitizer runs (GPU sanitizers are not available on this pool).

* the C oracle built with -fsanitize=address,undefined (`make -C oracle asan`) evaluates golden known-answer cases and a
  few multi-chain draws: no out-of-bounds access, no undefined behaviour, same values as the plain build;
* the HIP library's HOST code built the same way (`make -C eeyore_amd/csrc asan`; the device code is compiled, not run)
  goes through every entry point that needs no device: Philox known answers, plan creation failing cleanly, argument
  validation with null plans and bad options, the error-string path.

Each run is a child python with the matching sanitizer runtime preloaded (gcc's for the oracle, clang's for the library);
ASAN aborts the child on the first finding, so a zero exit code is the assertion."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, preload, env_extra):
    env = dict(os.environ, LD_PRELOAD=preload, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=900, cwd=ROOT)
    return r.returncode, r.stdout.decode(), r.stderr.decode()


def _file_name(compiler, name):
    out = subprocess.run([compiler, f"-print-file-name={name}"], capture_output=True).stdout.decode().strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


ORACLE_CODE = r"""
import numpy as np
from oracle import c_oracle
from oracle.c_oracle import COracle
from tests.helpers import groups, load
assert c_oracle.LIB.endswith("liboracle_asan.so")
n = 0
for name, rec in groups(load("g1_kats.npz")).items():
    co = COracle(rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"]), rec["x"], rec["y"], rec["prior_mu"],
                 rec["prior_sigma"], dtype=np.float64)
    t, g, lik, prior = co.log_target_grad(rec["theta"])
    assert abs(t - rec["log_target"]) <= 1e-10 * max(1.0, abs(rec["log_target"])), name
    assert np.allclose(g, rec["grad"], rtol=1e-10, atol=1e-12), name
    n += 1
rec = dict(groups(load("g4_hmc_traces.npz"))["mlp432323_synth"])
for dt in (np.float64, np.float32):
    co = COracle(rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"]), rec["x"], rec["y"], rec["prior_mu"],
                 rec["prior_sigma"], dtype=dt, nthreads=4)
    rng = np.random.default_rng(0)
    C, P = 9, co.P
    th = (0.1 * rng.standard_normal((C, P))).astype(dt)
    tv = np.zeros(C, dt); g = np.zeros((C, P), dt)
    for c in range(C):
        tv[c], g[c], _, _ = co.log_target_grad(th[c])
    co.hmc_draw(th, tv, g, rng.standard_normal((C, P)).astype(dt), rng.random(C).astype(dt), 0.02, 5)
    co.mala_draw(th, tv, g, rng.standard_normal((C, P)).astype(dt), rng.random(C).astype(dt), 1e-4)
    co.mh_draw(th, tv, rng.standard_normal((C, P)).astype(dt), rng.random(C).astype(dt), 1e-3)
    thL, pL, t, gL = co.leapfrog(th[0], rng.standard_normal(P).astype(dt), 0.01, 3)
    assert np.isfinite(thL).all() and np.isfinite(tv).all()
print("oracle under asan+ubsan ok:", n, "known answers")
"""


def test_c_oracle_under_address_and_undefined_behaviour_sanitizers():
    rt = _file_name("gcc", "libasan.so")
    if rt is None:
        pytest.skip("gcc's libasan.so not found")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so")
    ubsan = _file_name("gcc", "libubsan.so")
    rc, out, err = _run(ORACLE_CODE, rt + (":" + ubsan if ubsan else ""), {"EEYORE_ORACLE_LIB": lib})
    assert rc == 0 and "oracle under asan+ubsan ok" in out, (out[-2000:], err[-4000:])


HOST_CODE = r"""
import ctypes as ct
import numpy as np
from eeyore_amd import _lib as L
lib = L.lib()
assert L.LIB_PATH.endswith("libeeyore_amd_asan.so") and lib.ey_version() >= 100
# Philox4x32-10 known answers (Random123's kat_vectors), through the host entry point
from oracle import philox_oracle as po
u32 = ct.c_uint32
for ctr, key, want in po.RANDOM123_KAT:
    out = (u32 * 4)()
    assert lib.ey_philox_block((u32 * 4)(*ctr), (u32 * 2)(*key), out) == 0
    assert tuple(out) == tuple(want), (ctr, list(out))
assert lib.ey_philox_block(None, None, None) == -1 and b"null" in lib.ey_last_error()
# plan creation: bad arguments are refused before any device call; a good one needs a device (fails cleanly here without
# one, succeeds and is destroyed on a GPU box)
h = ct.c_void_p()
I = ct.c_int
assert lib.ey_plan_create(ct.byref(h), 0, (I * 2)(1, 1), None, (I * 1)(0), 1, 0, 0) == -1
assert lib.ey_plan_create(ct.byref(h), 2, (I * 3)(4, 0, 3), None, (I * 2)(1, 0), 1, 0, 0) == -1
assert lib.ey_plan_create(ct.byref(h), 2, (I * 3)(4, 3, 3), None, (I * 2)(1, 9), 1, 0, 0) == -1
assert lib.ey_plan_create(ct.byref(h), 2, (I * 3)(4, 3, 3), None, (I * 2)(1, 0), 5, 0, 0) == -1
assert lib.ey_plan_create(ct.byref(h), 2, (I * 3)(4, 3, 3), None, (I * 2)(1, 0), 1, 7, 0) == -1
assert lib.ey_plan_create(None, 2, (I * 3)(4, 3, 3), None, (I * 2)(1, 0), 1, 0, 0) == -1
rc = lib.ey_plan_create(ct.byref(h), 2, (I * 3)(4, 3, 3), (I * 2)(1, 1), (I * 2)(1, 0), 1, 1, 0)
if rc == 0:
    P = ct.c_int64()
    assert lib.ey_plan_num_params(h, ct.byref(P)) == 0 and P.value == 27
    v = I()
    assert lib.ey_plan_get_option(h, L.EY_OPT_F32_PRODUCTS, ct.byref(v)) == 0 and v.value in (0, 1)
    assert lib.ey_plan_set_option(h, L.EY_OPT_F32_PRODUCTS, 5) == -1 and lib.ey_plan_set_option(h, 77, 0) == -1
    assert lib.ey_plan_set_variant(h, 3) == 0 and lib.ey_plan_set_variant(h, 0) == 3
    assert lib.ey_plan_kernel(h) in (b"generic", b"fused16", b"bgemm", b"mfma32")
    assert lib.ey_hmc_step(h, None, None, None, None, None, 0.1, None, 5, None, 4, 0, 0, 0, 0, None, None, None, None, None) == -4
    assert lib.ey_plan_destroy(h) == 0
else:
    assert rc in (-1, -3), rc   # no such device / HIP runtime error, with a message
    assert len(lib.ey_last_error()) > 0
# null plans and null buffers on every entry point that checks them first
nul = None
assert lib.ey_plan_num_params(nul, nul) == -1
assert lib.ey_plan_kernel(nul) == b"generic"
assert lib.ey_plan_set_option(nul, 1, 0) == -1 and lib.ey_plan_get_option(nul, 1, nul) == -1
assert lib.ey_plan_set_variant(nul, 0) == -1
assert lib.ey_plan_set_data(nul, nul, nul, 4, nul) == -1 and lib.ey_plan_set_prior(nul, nul, nul, nul) == -1
assert lib.ey_log_target(nul, nul, nul, 4, nul, nul, nul) == -1
assert lib.ey_log_target_grad(nul, nul, nul, 4, nul, nul, nul) == -1
assert lib.ey_hmc_leapfrog(nul, nul, nul, 0.1, nul, 5, nul, 4, nul, nul, nul) == -1
assert lib.ey_mala_step(nul, nul, nul, nul, nul, nul, 0.1, nul, nul, 4, 0, 0, 0, 0, nul, nul, nul) == -1
assert lib.ey_mh_step(nul, nul, nul, nul, nul, nul, nul, 4, 0, 0, 0, 0, nul, nul, nul) == -1
assert lib.ey_plan_attach_da(nul, nul, nul, nul, 0, 0, 0.5, 0.0, 0) == -1
assert lib.ey_plan_attach_moments(nul, nul, nul, nul, 0) == -1
assert lib.ey_pt_swap_decide(nul, nul, nul, nul, nul, nul, 4, 0, nul, nul, nul) == -1
assert lib.ey_philox_normal(nul, 4, 4, 0, 0, 0, 0, nul) == -1 and lib.ey_philox_uniform(nul, 4, 0, 0, 0, 5, nul) == -1
assert lib.ey_stats_update(nul, nul, 4, 4, 0, nul, nul, nul, nul) == -1
assert lib.ey_inse_univariate(nul, 10, 4, 0, nul, nul, nul, nul) == -1
assert lib.ey_plan_destroy(nul) == 0
old = lib.ey_debug_set_variant(16)
assert lib.ey_debug_set_variant(old) == 16
print("host ABI under asan+ubsan ok")
"""


def test_library_host_code_under_address_and_undefined_behaviour_sanitizers():
    rt = _file_name("/opt/rocm/bin/hipcc", "libclang_rt.asan-x86_64.so")
    if rt is None:
        pytest.skip("clang's ASAN runtime not found")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "eeyore_amd", "csrc"), "asan", "-j6"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "eeyore_amd", "lib", "libeeyore_amd_asan.so")
    rc, out, err = _run(HOST_CODE, rt, {"EEYORE_AMD_LIB": lib})
    assert rc == 0 and "host ABI under asan+ubsan ok" in out, (out[-2000:], err[-4000:])
