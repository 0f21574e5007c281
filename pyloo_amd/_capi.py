"""ctypes binding of libpyloo_amd.so (include/pyloo_amd.h) -- the only way the Python front
reaches the GPU.  There is deliberately no CPU fallback: a missing library or GPU raises."""

import ctypes as C
import os

import numpy as np

PLA_F64, PLA_F32 = 0, 1
PLA_HOST, PLA_DEVICE = 0, 1
PLA_PSIS, PLA_SIS, PLA_TIS = 0, 1, 2
METHOD_CODES = {"psis": PLA_PSIS, "sis": PLA_SIS, "tis": PLA_TIS}
AGG_N, AGG_SUM_LOO, AGG_M2_LOO, AGG_SUM_LPPD, AGG_N_HIGH, AGG_N_NONFINITE, AGG_MIN_DIAG, AGG_N_SLOW = range(8)
AGG_COUNT = 8
ABI_VERSION = 4

# every symbol declared in include/pyloo_amd.h
SYMBOLS = (
    "pla_abi_version", "pla_last_error", "pla_device_count", "pla_engine_create", "pla_engine_destroy",
    "pla_tail_count", "pla_psis_loo", "pla_importance_weights", "pla_reduce_pointwise", "pla_waic",
    "pla_psis_loo_rows", "pla_waic_rows", "pla_e_loo", "pla_e_loo_quantiles",
    "pla_engine_set_frozen", "pla_engine_set_timing", "pla_engine_kernel_ms", "pla_engine_first_kernel_ms", "pla_fill_synthetic",
    "pla_engine_last_kernels", "pla_aggregate_pack", "pla_aggregate_merge", "pla_fill_synthetic_chains",
    "pla_env_overrides", "pla_engine_stream_stats",
)


class EngineError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, code, msg):
        super().__init__(f"libpyloo_amd status {code}: {msg}")
        self.code = code


_lib = None


def load_library():
    """dlopen the in-tree library; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    from .build import LIB_PATH  # (imported here so that `python -m pyloo_amd.build` runs the module once)

    lib_path = os.environ.get("PYLOO_AMD_LIB", LIB_PATH)  # profiling builds (tools/ablate.sh)
    if not os.path.exists(lib_path):
        raise RuntimeError(
            f"{lib_path} is missing: the HIP engine has not been built "
            "(run `python -m pyloo_amd.build`).  pyloo_amd has no CPU fallback."
        )
    # PyTorch-ROCm wheels bundle their own HIP runtime (same SONAME libamdhip64.so.7).  Two HIP
    # runtimes in one process cannot both own the GPU, so torch must be loaded FIRST: our
    # library's NEEDED entry then binds to the runtime torch already mapped, and torch tensors,
    # streams and this engine share one HIP context.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for pure NumPy callers
        pass
    lib = C.CDLL(lib_path)
    i64, dbl, vp, ci = C.c_int64, C.c_double, C.c_void_p, C.c_int
    lib.pla_abi_version.restype = ci
    lib.pla_last_error.restype = C.c_char_p
    lib.pla_device_count.argtypes = [C.POINTER(ci)]
    lib.pla_engine_create.argtypes = [ci, C.POINTER(vp)]
    lib.pla_engine_destroy.argtypes = [vp]
    lib.pla_tail_count.argtypes = [i64, dbl, C.POINTER(i64)]
    lib.pla_psis_loo.argtypes = [vp, vp, ci, i64, i64, i64, i64, ci, i64, dbl, dbl, ci, vp, vp, vp, vp, vp]
    lib.pla_importance_weights.argtypes = [vp, vp, ci, i64, i64, i64, i64, ci, i64, ci, vp, vp, vp]
    lib.pla_reduce_pointwise.argtypes = [vp, vp, vp, vp, i64, dbl, ci, vp, vp]
    lib.pla_waic.argtypes = [vp, vp, ci, i64, i64, i64, i64, dbl, ci, vp, vp, vp, vp, vp]
    lib.pla_psis_loo_rows.argtypes = [vp, vp, ci, i64, i64, i64, i64, vp, i64, ci, i64, dbl, dbl, ci, vp, vp, vp, vp, vp]
    lib.pla_waic_rows.argtypes = [vp, vp, ci, i64, i64, i64, i64, vp, i64, dbl, ci, vp, vp, vp, vp, vp]
    lib.pla_e_loo.argtypes = [vp, vp, vp, vp, ci, i64, i64, i64, i64, i64, ci, vp, vp, vp, vp, vp, vp]
    lib.pla_e_loo_quantiles.argtypes = [vp, vp, vp, ci, i64, i64, i64, i64, vp, i64, ci, vp, vp]
    lib.pla_engine_set_frozen.argtypes = [vp, ci]
    lib.pla_engine_set_timing.argtypes = [vp, ci]
    lib.pla_engine_kernel_ms.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64)]
    lib.pla_engine_first_kernel_ms.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64)]
    lib.pla_fill_synthetic.argtypes = [vp, vp, ci, i64, i64, i64, C.c_uint64, dbl, dbl, dbl, dbl, vp]
    lib.pla_fill_synthetic_chains.argtypes = [vp, vp, ci, i64, i64, i64, C.c_uint64, ci, dbl, dbl, dbl, dbl, vp]
    lib.pla_engine_last_kernels.argtypes = [vp, C.c_char_p, ci]
    lib.pla_aggregate_pack.argtypes = [vp, vp, ci, ci, vp, vp]
    lib.pla_aggregate_merge.argtypes = [vp, vp, ci, vp, vp]
    lib.pla_env_overrides.argtypes = [C.c_char_p, ci]
    lib.pla_engine_stream_stats.argtypes = [vp, C.POINTER(i64)]
    for name in SYMBOLS:
        getattr(lib, name)  # AttributeError if the header and the library disagree
        if name != "pla_last_error":
            getattr(lib, name).restype = ci
    if lib.pla_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libpyloo_amd ABI {lib.pla_abi_version()} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(code):
    if code != 0:
        raise EngineError(code, load_library().pla_last_error().decode("utf-8", "replace"))


def env_overrides():
    """Run-time switches of the library that are set in the environment ("" when none is): ``pla_env_overrides``."""
    buf = C.create_string_buffer(1024)
    check(load_library().pla_env_overrides(buf, 1024))
    return buf.value.decode("utf-8", "replace")


def device_count():
    n = C.c_int(0)
    rc = load_library().pla_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def tail_count(n_draws, reff):
    out = C.c_int64(0)
    check(load_library().pla_tail_count(int(n_draws), float(reff), C.byref(out)))
    return out.value


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.float64:
        return PLA_F64
    if dt == np.float32:
        return PLA_F32
    raise TypeError(f"unsupported dtype {dt}")
