"""A stand-in for ``pyloo_amd.engine.Engine`` backed by the CPU oracle.

TEST INFRASTRUCTURE ONLY: it lets the CPU test-suite exercise the host-side Python of
``pyloo_amd`` (argument handling, warnings, ``ELPDData`` packing, sharded reduction) without a
GPU.  The product never uses it; GPU tests go through the real engine and the C ABI.
"""

import numpy as np

from oracle import psis_oracle as orc
from pyloo_amd._capi import AGG_COUNT


class OracleEngine:
    device = "cpu-oracle"

    def psis_loo(self, ll, tail_count=0, method="psis", scale_value=1.0, good_k=0.7, pointwise=True, aggregate=True):
        ll = np.asarray(ll, dtype=np.float64)
        n, s = ll.shape
        with np.errstate(all="ignore"):
            if method == "psis":
                lw = np.empty_like(ll)
                diag = np.empty(n)
                for i in range(n):
                    lw[i], diag[i] = orc.psis_row(-ll[i], tail_count)
            else:
                lw, diag = orc.importance_weights(-ll, method)
            loo_i = scale_value * np.array([orc.lse(r) for r in lw + ll], dtype=np.float64).reshape(n)
            lppd_i = np.array([orc.lse(r, b_inv=s) for r in ll], dtype=np.float64).reshape(n)
        agg = np.zeros(AGG_COUNT)
        agg[0] = n
        agg[1] = loo_i.sum()
        agg[2] = np.sum((loo_i - loo_i.mean()) ** 2) if n else 0.0
        agg[3] = lppd_i.sum()
        agg[4] = np.sum(diag > good_k)
        agg[5] = np.sum(~np.isfinite(diag))
        agg[6] = diag.min() if n else np.inf
        return {"diag": diag, "loo_i": loo_i, "lppd_i": lppd_i, "agg": agg}

    def waic(self, ll, scale_value=1.0, pointwise=True, aggregate=True):
        raw = np.asarray(ll, dtype=np.float64)
        w = orc.waic_arrays(raw, scale_value)
        n = raw.shape[0]
        agg = np.zeros(AGG_COUNT)
        agg[0] = n
        agg[1] = w["waic_i"].sum()
        agg[2] = np.sum((w["waic_i"] - w["waic_i"].mean()) ** 2) if n else 0.0
        agg[3] = w["var_i"].sum()
        agg[4] = np.sum(w["var_i"] > 0.4)
        agg[6] = w["var_i"].min() if n else np.inf
        agg[7] = np.sum(~np.isfinite(raw))
        return {"lppd_i": w["lppd_i"], "var_i": w["var_i"], "waic_i": w["waic_i"], "agg": agg}

    def importance_weights(self, logw, tail_count=0, method="psis"):
        logw = np.asarray(logw)
        with np.errstate(all="ignore"):
            if method == "psis":
                lw = np.empty_like(logw)
                diag = np.empty(logw.shape[0])
                for i in range(logw.shape[0]):
                    lw[i], diag[i] = orc.psis_row(logw[i], tail_count)
                return lw, diag
            return orc.importance_weights(logw, method)

    def e_loo(self, x, log_weights, log_ratios=None, tail_len=20):
        x, lw = np.asarray(x, dtype=np.float64), np.asarray(log_weights, dtype=np.float64)
        lr = None if log_ratios is None else np.asarray(log_ratios, dtype=np.float64)
        with np.errstate(all="ignore"):
            r = orc.e_loo_arrays(x, lw, lr)
        return {k: r[k] for k in ("mean", "var", "k_mean", "k_var", "k_none")}

