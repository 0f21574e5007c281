"""Python handle on one ``pla_engine`` (one GPU) and the array plumbing around the C ABI.

Inputs may be NumPy arrays (host memory: the library stages them through the device) or
torch CUDA tensors (device memory: zero copies, work is enqueued on torch's current stream
and results come back as CUDA tensors).  torch is used only for device memory and streams.
"""

import ctypes as C
import threading

import numpy as np

from . import _capi
from ._capi import AGG_COUNT, METHOD_CODES, PLA_DEVICE, PLA_HOST, check, dtype_code, load_library

_engines = {}
_lock = threading.Lock()


def _is_torch_tensor(a):
    return type(a).__module__.split(".")[0] == "torch" and hasattr(a, "data_ptr")


class Engine:
    """One engine per (process, device).  Use :func:`get_engine`."""

    def __init__(self, device=0):
        lib = load_library()
        n = _capi.device_count()
        if n <= 0:
            raise RuntimeError(
                "pyloo_amd needs an AMD GPU (MI355X / gfx950): no HIP device is visible. "
                "There is no CPU fallback."
            )
        h = C.c_void_p()
        check(lib.pla_engine_create(int(device), C.byref(h)))
        self._lib = lib
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pla_engine_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        import torch

        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _as_2d_host(a, allow_obs_fastest=False):
        a = np.asarray(a)
        if a.dtype not in (np.float64, np.float32):
            a = a.astype(np.float64)
        if a.ndim != 2:
            raise ValueError("expected a 2-D (n_obs, n_draws) array")
        if allow_obs_fastest and Engine._host_obs_fastest(a):
            return a  # a (chain, draw, *obs) buffer viewed as (obs, sample): the library takes it as it is
        if a.shape[1] > 1 and a.strides[1] != a.itemsize:
            a = np.ascontiguousarray(a)
        if a.shape[0] > 1 and (a.strides[0] % a.itemsize != 0 or a.strides[0] < 0):
            a = np.ascontiguousarray(a)
        return a

    @staticmethod
    def _host_obs_fastest(a):
        n, s = a.shape
        return n > 1 and s > 1 and a.strides[0] == a.itemsize and a.strides[1] % a.itemsize == 0 and a.strides[1] >= n * a.itemsize

    @staticmethod
    def _host_strides(a):
        """(stride_obs, stride_draw) in elements of a host matrix prepared by :meth:`_as_2d_host`."""
        n, s = a.shape
        if Engine._host_obs_fastest(a):
            return 1, a.strides[1] // a.itemsize
        return (a.strides[0] // a.itemsize if n > 1 else s), 1

    @staticmethod
    def _draws_fastest(t):
        """ArviZ keeps log-likelihoods as (chain, draw, *obs): the (obs, sample) view of such a buffer has the
        observations fastest.  The entry points of the library take that layout as it is (a tiled transpose kernel
        feeds the row kernels block by block); this copy is only for layouts the library would walk with strides
        (if the copy does not fit, the strided general kernel takes them)."""
        import torch

        if t.dim() == 2 and t.shape[1] > 1 and t.stride(1) != 1:
            try:
                return t.contiguous()
            except torch.OutOfMemoryError:  # pragma: no cover - needs a nearly full device
                return t
        return t

    @staticmethod
    def _library_layout(t, rows=None):
        """Layouts the library reads fast without a copy: draws fastest, or observations fastest (stride 1 along
        the observations, no row selection); anything else goes through :meth:`_draws_fastest`."""
        if t.dim() == 2 and t.shape[0] > 1 and t.shape[1] > 1 and t.stride(0) == 1 and t.stride(1) >= t.shape[0] and rows is None:
            return t
        return Engine._draws_fastest(t)

    # ------------------------------------------------------------------ LOO pass
    def psis_loo(self, ll, tail_count=0, method="psis", scale_value=1.0, good_k=0.7, pointwise=True, aggregate=True,
                 rows=None):
        """Fused pass over an (n_obs, n_draws) log-likelihood matrix (``pla_psis_loo``).

        Returns ``dict(diag, loo_i, lppd_i, agg)`` -- NumPy arrays for NumPy input, CUDA tensors
        for CUDA-tensor input (``agg`` included; nothing is synchronised in that case).
        ``rows``: optional observation indices; the pass then runs over those rows only
        (``pla_psis_loo_rows``) and the outputs have one entry per index.
        """
        mcode = METHOD_CODES[method]
        if _is_torch_tensor(ll):
            return self._psis_loo_device(ll, tail_count, mcode, scale_value, good_k, pointwise, aggregate, rows)
        a = self._as_2d_host(ll, allow_obs_fastest=rows is None)
        n, s = a.shape
        so, sd = self._host_strides(a)
        if rows is not None:
            idx = self._host_rows(rows, n)
            m = idx.size
            diag, loo_i, lppd_i = (np.empty(m), np.empty(m), np.empty(m)) if pointwise else (None, None, None)
            agg = np.zeros(AGG_COUNT) if aggregate else None
            p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)  # noqa: E731
            check(self._lib.pla_psis_loo_rows(self._h, a.ctypes.data_as(C.c_void_p), dtype_code(a.dtype), n, s, so, 1,
                                              p(idx), m, mcode, int(tail_count), float(scale_value), float(good_k),
                                              PLA_HOST, None, p(diag), p(loo_i), p(lppd_i), p(agg)))
            return {"diag": diag, "loo_i": loo_i, "lppd_i": lppd_i, "agg": agg}
        diag = np.empty(n) if pointwise else None
        loo_i = np.empty(n) if pointwise else None
        lppd_i = np.empty(n) if pointwise else None
        agg = np.zeros(AGG_COUNT) if aggregate else None
        p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)  # noqa: E731
        check(self._lib.pla_psis_loo(self._h, a.ctypes.data_as(C.c_void_p), dtype_code(a.dtype), n, s, so, sd,
                                     mcode, int(tail_count), float(scale_value), float(good_k), PLA_HOST,
                                     None, p(diag), p(loo_i), p(lppd_i), p(agg)))
        return {"diag": diag, "loo_i": loo_i, "lppd_i": lppd_i, "agg": agg}

    @staticmethod
    def _host_rows(rows, n_obs):
        """Index list as contiguous int64, range-checked like NumPy indexing would (no negative wrap-around:
        loo_subsample.py:266-271 rejects indices outside [0, n))."""
        idx = np.ascontiguousarray(np.asarray(rows).reshape(-1), dtype=np.int64)
        if idx.size and (idx.min() < 0 or idx.max() >= n_obs):
            raise IndexError(f"row indices must lie in [0, {n_obs}), got range [{idx.min()}, {idx.max()}]")
        return idx

    def _device_rows(self, rows, n_obs, device):
        import torch

        if _is_torch_tensor(rows):  # already on the device: the kernels clamp, the caller vouches for the range
            return rows.to(device=device, dtype=torch.int64).contiguous().reshape(-1)
        return torch.from_numpy(self._host_rows(rows, n_obs)).to(device)

    def _psis_loo_device(self, t, tail_count, mcode, scale_value, good_k, pointwise, aggregate, rows=None):
        import torch

        if t.dim() != 2 or not t.is_cuda:
            raise ValueError("expected a 2-D CUDA tensor")
        if t.dtype not in (torch.float64, torch.float32):
            raise TypeError(f"unsupported dtype {t.dtype}")
        t = self._library_layout(t, rows)
        n, s = t.shape
        dev = t.device
        idx = None if rows is None else self._device_rows(rows, n, dev)
        m = n if idx is None else idx.numel()
        diag = torch.empty(m, dtype=torch.float64, device=dev) if (pointwise or aggregate) else None
        loo_i = torch.empty(m, dtype=torch.float64, device=dev) if (pointwise or aggregate) else None
        lppd_i = torch.empty(m, dtype=torch.float64, device=dev) if (pointwise or aggregate) else None
        agg = torch.empty(AGG_COUNT, dtype=torch.float64, device=dev) if aggregate else None  # (every slot is written)
        p = lambda x: None if x is None else C.c_void_p(x.data_ptr())  # noqa: E731
        code = _capi.PLA_F64 if t.dtype == torch.float64 else _capi.PLA_F32
        if idx is not None:
            check(self._lib.pla_psis_loo_rows(self._h, C.c_void_p(t.data_ptr()), code, n, s, t.stride(0), t.stride(1),
                                              p(idx), m, mcode, int(tail_count), float(scale_value), float(good_k),
                                              PLA_DEVICE, self._stream(), p(diag), p(loo_i), p(lppd_i), p(agg)))
            return {"diag": diag, "loo_i": loo_i, "lppd_i": lppd_i, "agg": agg}
        check(self._lib.pla_psis_loo(self._h, C.c_void_p(t.data_ptr()), code, n, s, t.stride(0), t.stride(1),
                                     mcode, int(tail_count), float(scale_value), float(good_k), PLA_DEVICE,
                                     self._stream(), p(diag), p(loo_i), p(lppd_i), p(agg)))
        return {"diag": diag, "loo_i": loo_i, "lppd_i": lppd_i, "agg": agg}

    # ------------------------------------------------------------------ weights pass
    def importance_weights(self, logw, tail_count=0, method="psis"):
        """(n_obs, n_draws) log ratios -> (lw, diag) (``pla_importance_weights``)."""
        mcode = METHOD_CODES[method]
        if _is_torch_tensor(logw):
            import torch

            t = self._library_layout(logw)
            n, s = t.shape
            lw = torch.empty((n, s), dtype=t.dtype, device=t.device)
            diag = torch.empty(n, dtype=torch.float64, device=t.device)
            code = _capi.PLA_F64 if t.dtype == torch.float64 else _capi.PLA_F32
            check(self._lib.pla_importance_weights(self._h, C.c_void_p(t.data_ptr()), code, n, s, t.stride(0),
                                                   t.stride(1), mcode, int(tail_count), PLA_DEVICE, self._stream(),
                                                   C.c_void_p(lw.data_ptr()), C.c_void_p(diag.data_ptr())))
            return lw, diag
        a = self._as_2d_host(logw)
        n, s = a.shape
        so = a.strides[0] // a.itemsize if n > 1 else s
        lw = np.empty((n, s), dtype=a.dtype)
        diag = np.empty(n)
        check(self._lib.pla_importance_weights(self._h, a.ctypes.data_as(C.c_void_p), dtype_code(a.dtype), n, s, so, 1,
                                               mcode, int(tail_count), PLA_HOST, None,
                                               lw.ctypes.data_as(C.c_void_p), diag.ctypes.data_as(C.c_void_p)))
        return lw, diag

    # ------------------------------------------------------------------ WAIC pass
    def waic(self, ll, scale_value=1.0, pointwise=True, aggregate=True, rows=None):
        """(n_obs, n_draws) log-likelihood -> ``dict(lppd_i, var_i, waic_i, agg)`` (``pla_waic``; the
        slots of ``agg`` are documented in include/pyloo_amd.h).  ``rows``: optional observation indices
        (``pla_waic_rows``), one output entry per index."""
        if _is_torch_tensor(ll):
            import torch

            t = ll
            if t.dim() != 2 or not t.is_cuda:
                raise ValueError("expected a 2-D CUDA tensor")
            if t.dtype not in (torch.float64, torch.float32):
                raise TypeError(f"unsupported dtype {t.dtype}")
            t = self._library_layout(t, rows)
            n, s = t.shape
            idx = None if rows is None else self._device_rows(rows, n, t.device)
            m = n if idx is None else idx.numel()
            mk = lambda: torch.empty(m, dtype=torch.float64, device=t.device)  # noqa: E731
            lppd_i, var_i, waic_i = (mk(), mk(), mk()) if (pointwise or aggregate) else (None, None, None)
            agg = torch.zeros(AGG_COUNT, dtype=torch.float64, device=t.device) if aggregate else None
            p = lambda x: None if x is None else C.c_void_p(x.data_ptr())  # noqa: E731
            code = _capi.PLA_F64 if t.dtype == torch.float64 else _capi.PLA_F32
            if idx is not None:
                check(self._lib.pla_waic_rows(self._h, C.c_void_p(t.data_ptr()), code, n, s, t.stride(0), t.stride(1),
                                              p(idx), m, float(scale_value), PLA_DEVICE, self._stream(), p(lppd_i),
                                              p(var_i), p(waic_i), p(agg)))
                return {"lppd_i": lppd_i, "var_i": var_i, "waic_i": waic_i, "agg": agg}
            check(self._lib.pla_waic(self._h, C.c_void_p(t.data_ptr()), code, n, s, t.stride(0), t.stride(1),
                                     float(scale_value), PLA_DEVICE, self._stream(), p(lppd_i), p(var_i), p(waic_i), p(agg)))
            return {"lppd_i": lppd_i, "var_i": var_i, "waic_i": waic_i, "agg": agg}
        a = self._as_2d_host(ll, allow_obs_fastest=rows is None)
        n, s = a.shape
        so, sd = self._host_strides(a)
        idx = None if rows is None else self._host_rows(rows, n)
        m = n if idx is None else idx.size
        lppd_i, var_i, waic_i = (np.empty(m), np.empty(m), np.empty(m)) if pointwise else (None, None, None)
        agg = np.zeros(AGG_COUNT) if aggregate else None
        p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)  # noqa: E731
        if idx is not None:
            check(self._lib.pla_waic_rows(self._h, a.ctypes.data_as(C.c_void_p), dtype_code(a.dtype), n, s, so, 1, p(idx), m,
                                          float(scale_value), PLA_HOST, None, p(lppd_i), p(var_i), p(waic_i), p(agg)))
            return {"lppd_i": lppd_i, "var_i": var_i, "waic_i": waic_i, "agg": agg}
        check(self._lib.pla_waic(self._h, a.ctypes.data_as(C.c_void_p), dtype_code(a.dtype), n, s, so, sd,
                                 float(scale_value), PLA_HOST, None, p(lppd_i), p(var_i), p(waic_i), p(agg)))
        return {"lppd_i": lppd_i, "var_i": var_i, "waic_i": waic_i, "agg": agg}

    # ------------------------------------------------------------------ weighted expectations
    def e_loo(self, x, log_weights, log_ratios=None, tail_len=20):
        """(n_obs, n_draws) draws ``x`` + log-weights (+ raw log ratios) -> ``dict(mean, var, k_mean, k_var, k_none)``
        (``pla_e_loo``: e_loo.py:214-236 per observation).  NumPy in -> NumPy out; CUDA tensors in -> CUDA tensors out."""
        if _is_torch_tensor(x):
            import torch

            mats = [x, log_weights] + ([log_ratios] if log_ratios is not None else [])
            if any((not _is_torch_tensor(m)) or m.shape != x.shape or m.dim() != 2 for m in mats):
                raise ValueError("x, log_weights and log_ratios must be 2-D CUDA tensors of one shape")
            dt = torch.float64 if any(m.dtype == torch.float64 for m in mats) else torch.float32
            mats = [m.to(dt) for m in mats]
            if any(m.stride() != mats[0].stride() for m in mats) or mats[0].stride(1) <= 0:
                mats = [m.contiguous() for m in mats]
            t = mats[0]
            n, s = t.shape
            out = {k: torch.empty(n, dtype=torch.float64, device=t.device) for k in ("mean", "var", "k_mean", "k_var", "k_none")}
            code = _capi.PLA_F64 if dt == torch.float64 else _capi.PLA_F32
            p = lambda m: C.c_void_p(m.data_ptr())  # noqa: E731
            check(self._lib.pla_e_loo(self._h, p(mats[0]), p(mats[1]), p(mats[2]) if len(mats) > 2 else None, code, n, s,
                                      t.stride(0), t.stride(1), int(tail_len), PLA_DEVICE, self._stream(),
                                      *(p(out[k]) for k in ("mean", "var", "k_mean", "k_var", "k_none"))))
            return out
        mats = [np.asarray(x), np.asarray(log_weights)] + ([np.asarray(log_ratios)] if log_ratios is not None else [])
        if any(m.ndim != 2 or m.shape != mats[0].shape for m in mats):
            raise ValueError("x, log_weights and log_ratios must be 2-D arrays of one shape")
        dt = np.float32 if all(m.dtype == np.float32 for m in mats) else np.float64
        mats = [np.ascontiguousarray(m, dtype=dt) for m in mats]
        n, s = mats[0].shape
        out = {k: np.empty(n) for k in ("mean", "var", "k_mean", "k_var", "k_none")}
        p = lambda m: m.ctypes.data_as(C.c_void_p)  # noqa: E731
        check(self._lib.pla_e_loo(self._h, p(mats[0]), p(mats[1]), p(mats[2]) if len(mats) > 2 else None, dtype_code(dt), n, s,
                                  s, 1, int(tail_len), PLA_HOST, None,
                                  *(p(out[k]) for k in ("mean", "var", "k_mean", "k_var", "k_none"))))
        return out

    def e_loo_quantiles(self, x, log_weights, probs):
        """(n_obs, n_draws) draws + log-weights, quantile levels -> (n_obs, n_probs) weighted quantiles (``pla_e_loo_quantiles``:
        e_loo.py:468-515, 534-554)."""
        pr = np.ascontiguousarray(np.atleast_1d(np.asarray(probs, dtype=np.float64)))
        if _is_torch_tensor(x):
            import torch

            mats = [x, log_weights]
            if any((not _is_torch_tensor(m)) or m.shape != x.shape or m.dim() != 2 for m in mats):
                raise ValueError("x and log_weights must be 2-D CUDA tensors of one shape")
            dt = torch.float64 if any(m.dtype == torch.float64 for m in mats) else torch.float32
            mats = [m.to(dt) for m in mats]
            if mats[0].stride() != mats[1].stride() or mats[0].stride(1) <= 0:
                mats = [m.contiguous() for m in mats]
            t = mats[0]
            n, s = t.shape
            out = torch.empty((n, pr.size), dtype=torch.float64, device=t.device)
            code = _capi.PLA_F64 if dt == torch.float64 else _capi.PLA_F32
            check(self._lib.pla_e_loo_quantiles(self._h, C.c_void_p(mats[0].data_ptr()), C.c_void_p(mats[1].data_ptr()), code, n, s,
                                                t.stride(0), t.stride(1), pr.ctypes.data_as(C.c_void_p), pr.size, PLA_DEVICE,
                                                self._stream(), C.c_void_p(out.data_ptr())))
            return out
        mats = [np.asarray(x), np.asarray(log_weights)]
        if any(m.ndim != 2 or m.shape != mats[0].shape for m in mats):
            raise ValueError("x and log_weights must be 2-D arrays of one shape")
        dt = np.float32 if all(m.dtype == np.float32 for m in mats) else np.float64
        mats = [np.ascontiguousarray(m, dtype=dt) for m in mats]
        n, s = mats[0].shape
        out = np.empty((n, pr.size))
        check(self._lib.pla_e_loo_quantiles(self._h, mats[0].ctypes.data_as(C.c_void_p), mats[1].ctypes.data_as(C.c_void_p),
                                            dtype_code(dt), n, s, s, 1, pr.ctypes.data_as(C.c_void_p), pr.size, PLA_HOST, None,
                                            out.ctypes.data_as(C.c_void_p)))
        return out

    # ------------------------------------------------------------------ reductions
    def reduce_pointwise(self, diag, loo_i, lppd_i, good_k):
        if _is_torch_tensor(loo_i):
            import torch

            agg = torch.zeros(AGG_COUNT, dtype=torch.float64, device=loo_i.device)
            check(self._lib.pla_reduce_pointwise(self._h, C.c_void_p(diag.data_ptr()), C.c_void_p(loo_i.data_ptr()),
                                                 C.c_void_p(lppd_i.data_ptr()), loo_i.numel(), float(good_k),
                                                 PLA_DEVICE, self._stream(), C.c_void_p(agg.data_ptr())))
            return agg
        d, l, p = (np.ascontiguousarray(x, dtype=np.float64) for x in (diag, loo_i, lppd_i))
        agg = np.zeros(AGG_COUNT)
        check(self._lib.pla_reduce_pointwise(self._h, d.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p),
                                             p.ctypes.data_as(C.c_void_p), l.size, float(good_k), PLA_HOST, None,
                                             agg.ctypes.data_as(C.c_void_p)))
        return agg

    # ------------------------------------------------------------------ bench helpers
    def fill_synthetic(self, t, seed, row0=0, k_lo=0.05, k_hi=0.60, heavy_lo=0.0, heavy_hi=0.0):
        import torch

        code = _capi.PLA_F64 if t.dtype == torch.float64 else _capi.PLA_F32
        n, s = t.shape
        assert t.is_contiguous()
        check(self._lib.pla_fill_synthetic(self._h, C.c_void_p(t.data_ptr()), code, n, s, int(row0), int(seed),
                                           k_lo, k_hi, heavy_lo, heavy_hi, self._stream()))

    def fill_synthetic_chains(self, t, seed, row0=0, chains=4, rho=0.9, offset_sd=0.3, k_lo=0.05, k_hi=0.60):
        """Rows as MCMC delivers them: chain-major stack of AR(1) chains with per-chain offsets (``pla_fill_synthetic_chains``)."""
        import torch

        code = _capi.PLA_F64 if t.dtype == torch.float64 else _capi.PLA_F32
        n, s = t.shape
        assert t.is_contiguous()
        check(self._lib.pla_fill_synthetic_chains(self._h, C.c_void_p(t.data_ptr()), code, n, s, int(row0), int(seed), int(chains),
                                                  float(rho), float(offset_sd), k_lo, k_hi, self._stream()))

    def set_frozen(self, on):
        """Frozen: calls that would reallocate engine workspace fail (EngineError -6) instead of invalidating the
        raw pointers a captured HIP graph holds (include/pyloo_amd.h, "HIP graphs")."""
        check(self._lib.pla_engine_set_frozen(self._h, 1 if on else 0))

    def set_timing(self, on):
        check(self._lib.pla_engine_set_timing(self._h, 1 if on else 0))

    def kernel_ms(self):
        ms, k = C.c_double(0), C.c_int64(0)
        check(self._lib.pla_engine_kernel_ms(self._h, C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def first_kernel_ms(self):
        """Accumulated time of the first (dominant) kernel of the two-kernel PSIS-LOO passes since the last call."""
        ms, k = C.c_double(0), C.c_int64(0)
        check(self._lib.pla_engine_first_kernel_ms(self._h, C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def last_kernels(self):
        """Which kernels the last PSIS-LOO / weights call launched (text; for benchmark records)."""
        buf = C.create_string_buffer(512)
        check(self._lib.pla_engine_last_kernels(self._h, buf, 512))
        return buf.value.decode("utf-8", "replace")

    def stream_gave_up(self):
        """Streamed passes since the last call in which the fit kernel stopped waiting for the sweep (``pla_engine_stream_stats``;
        synchronises the device).  0 in a healthy run."""
        n = C.c_int64(0)
        check(self._lib.pla_engine_stream_stats(self._h, C.byref(n)))
        return int(n.value)

    def aggregate_pack(self, agg, rank, world, table):
        """``table`` (world x 8, this engine's device) = zeros except row ``rank`` = ``agg``: one kernel on the current stream."""
        check(self._lib.pla_aggregate_pack(self._h, C.c_void_p(agg.data_ptr()), int(rank), int(world), C.c_void_p(table.data_ptr()),
                                           self._stream()))

    def aggregate_merge(self, table, world, out):
        """``out`` (8) = the per-rank aggregate rows of ``table`` merged (Chan, Golub & LeVeque): one kernel on the current stream."""
        check(self._lib.pla_aggregate_merge(self._h, C.c_void_p(table.data_ptr()), int(world), C.c_void_p(out.data_ptr()), self._stream()))


def get_engine(device=None):
    """Process-wide engine for ``device`` (default: torch's current CUDA device, else 0)."""
    if device is None:
        device = 0
        try:
            import torch

            if torch.cuda.is_available():
                device = torch.cuda.current_device()
        except Exception:
            device = 0
    with _lock:
        eng = _engines.get(device)
        if eng is None:
            eng = _engines[device] = Engine(device)
        return eng
