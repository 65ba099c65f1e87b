"""ctypes binding of the C-ABI HIP library (include/eeyore_amd.h).

The library is the product path: there is no CPU or eager fallback.  ``lib()`` raises if the shared
object is missing, and every compute call raises ``RuntimeError`` with ``ey_last_error()`` on failure.
"""
import ctypes as ct
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EEYORE_AMD_LIB", os.path.join(HERE, "lib", "libeeyore_amd.so"))  # override: diagnostics
CSRC = os.path.join(HERE, "csrc")

EY_F32, EY_F64 = 0, 1
EY_ACT_NONE, EY_ACT_SIGMOID, EY_ACT_TANH, EY_ACT_RELU = 0, 1, 2, 3
EY_LIK_BCE_SUM, EY_LIK_CE_SUM = 0, 1
EY_RECOMPUTE_INITIAL_GRAD, EY_FORCE_GENERIC = 1, 2

# every symbol include/eeyore_amd.h declares: (name, restype, argtypes)
_vp, _i, _i64, _u64, _u32, _d = ct.c_void_p, ct.c_int, ct.c_int64, ct.c_uint64, ct.c_uint32, ct.c_double
EY_OPT_F32_PRODUCTS, EY_PRODUCTS_BF16X3, EY_PRODUCTS_EXACT = 1, 0, 1
EY_OPT_ROW_WAVES, EY_ROW_WAVES_OFF, EY_ROW_WAVES_ON, EY_ROW_WAVES_AUTO = 2, 0, 1, 2

SYMBOLS = {
    "ey_version": (_i, []),
    "ey_last_error": (ct.c_char_p, []),
    "ey_plan_create": (_i, [ct.POINTER(_vp), _i, ct.POINTER(_i), ct.POINTER(_i), ct.POINTER(_i), _i, _i, _i]),
    "ey_plan_destroy": (_i, [_vp]),
    "ey_plan_num_params": (_i, [_vp, ct.POINTER(_i64)]),
    "ey_plan_kernel": (ct.c_char_p, [_vp]),
    "ey_plan_set_data": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "ey_plan_set_prior": (_i, [_vp, _vp, _vp, _vp]),
    "ey_log_target": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ey_log_target_grad": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "ey_hmc_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _i, _vp, _i64, _u64, _u64, _u64, _u32, _vp, _vp, _vp,
                         _vp, _vp]),
    "ey_hmc_leapfrog": (_i, [_vp, _vp, _vp, _d, _vp, _i, _vp, _i64, _vp, _vp, _vp]),
    "ey_mala_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _i64, _u64, _u64, _u64, _u32, _vp, _vp, _vp]),
    "ey_mh_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _u64, _u64, _u64, _u32, _vp, _vp, _vp]),
    "ey_pt_swap_decide": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "ey_philox_normal": (_i, [_vp, _i64, _i64, _u64, _u64, _u64, _i, _vp]),
    "ey_philox_uniform": (_i, [_vp, _i64, _u64, _u64, _u64, _i, _vp]),
    "ey_philox_block": (_i, [ct.POINTER(ct.c_uint32), ct.POINTER(ct.c_uint32), ct.POINTER(ct.c_uint32)]),
    "ey_stats_update": (_i, [_vp, _vp, _i64, _i64, _i, _vp, _vp, _vp, _vp]),
    "ey_plan_attach_moments": (_i, [_vp, _vp, _vp, _vp, _i64]),
    "ey_hmc_run": (_i, [_vp, _vp, _vp, _vp, _d, _vp, _i, _vp, _i64, _u64, _u64, _u64, _u32, _i, _vp, _vp, _vp, _vp,
                        _vp, _vp]),
    "ey_mala_run": (_i, [_vp, _vp, _vp, _vp, _d, _vp, _vp, _i64, _u64, _u64, _u64, _u32, _i, _vp, _vp, _vp, _vp, _vp,
                         _vp]),
    "ey_mh_run": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _u64, _u64, _u64, _u32, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ey_log_lik_rows": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "ey_inse_univariate": (_i, [_vp, _i64, _i64, _i, _vp, _vp, _vp, _vp]),
    "ey_plan_attach_da": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _d, _d, _i]),
    "ey_inse_multivariate": (_i, [_vp, _i64, _i64, _i64, _i64, _i64, _i, _vp, _vp, _vp, _vp, _vp]),
    "ey_debug_set_variant": (_i, [_i]),
    "ey_plan_set_variant": (_i, [_vp, _i]),
    "ey_plan_set_option": (_i, [_vp, _i, _i]),
    "ey_plan_get_option": (_i, [_vp, _i, ct.POINTER(ct.c_int)]),
    "ey_debug_bgemm": (_i, [_vp, _vp, _vp, _i, _i, _i] + [_i64] * 9 + [_vp, _i64, _i, _i, _vp]),
}


def build(force=False, verbose=False, diagnostics=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU).  ``diagnostics``: also the fused16
    family once more at -O1 (the build that computed wrong results before the hazard pass), on which
    tests/test_fused16.py runs its oracle suite -- a diagnostic library: a compiler failure that only -O1 shows must not
    fail the build of the production library, so it is not part of the default build (the test builds it when missing)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=out)
    if diagnostics:
        subprocess.check_call(["make", "-C", CSRC, "-j4", "o1"], stdout=out)
    return LIB_PATH


O1_LIB_PATH = os.path.join(HERE, "lib", "libeeyore_amd_f16o1.so")
SPILL_LIB_DIR = os.path.join(HERE, "lib")  # `make spill`: diagnostic builds libeeyore_amd_spill_<unit>.so


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"eeyore_amd: HIP library {LIB_PATH} is missing -- run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C eeyore_amd/csrc`); there is no CPU fallback")
        L = ct.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().ey_last_error().decode()
        if rc == -1:
            raise ValueError(f"eeyore_amd {what}: {msg}")
        raise RuntimeError(f"eeyore_amd {what}: {msg} (status {rc})")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ct.c_void_p(t.data_ptr())
