#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REAL reference.

Run only in the build container (``/root/reference`` does not exist on the GPU
box):  ``python tests/golden/make_golden.py``

How the reference is executed: ``import pyloo`` is not
possible here (xarray / arviz / pymc are not installed), but the hot path's
arithmetic lives in ``pyloo/psis.py``, ``pyloo/utils.py``, ``pyloo/sis.py`` and
``pyloo/tis.py``, which only touch xarray/arviz through import lines and
``isinstance`` checks.  They are loaded *in place* from ``/root/reference`` as
submodules of an empty in-memory package, with two empty in-memory placeholder
modules registered for the absent third-party imports.  No reference source or
bytecode is copied anywhere; only inputs and the numbers the reference's
functions return are written to ``*.npz``.

``loo()`` itself (``loo.py``) needs real xarray, so the aggregate lines
``loo.py:286-342`` are evaluated here with the reference's own primitives
(``make_ufunc``, ``_logsumexp``, ``psislw``) in the order ``loo.py`` applies them.
"""

import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

REF = "/root/reference/pyloo"


def load_reference():
    xr = types.ModuleType("xarray")
    xr.DataArray = type("DataArray", (), {})
    xr.apply_ufunc = None
    az = types.ModuleType("arviz")
    az.InferenceData = type("InferenceData", (), {})
    az.convert_to_inference_data = None
    sys.modules.setdefault("xarray", xr)
    sys.modules.setdefault("arviz", az)
    pkg = types.ModuleType("pyloo")
    pkg.__path__ = []
    sys.modules["pyloo"] = pkg
    mods = {}
    for name in ("utils", "psis", "sis", "tis"):
        spec = importlib.util.spec_from_file_location(f"pyloo.{name}", f"{REF}/{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"pyloo.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    # ndarray inputs: xarray.apply_ufunc(f, *a, kwargs=kw, **_) degenerates to f(*a, **kw)
    passthrough = lambda f, *a, kwargs=None, **_: f(*a, **(kwargs or {}))  # noqa: E731
    for m in mods.values():
        if hasattr(m, "apply_ufunc"):
            m.apply_ufunc = passthrough
    return mods


def run_reference(mods, ll, reff):
    """psislw + the aggregate arithmetic of loo.py:286-342 with reference primitives."""
    psis, utils = mods["psis"], mods["utils"]
    S = ll.shape[-1]
    N = int(np.prod(ll.shape[:-1]))
    rec = []
    real_gpdfit = psis._gpdfit

    def recording_gpdfit(ary):
        k, sigma = real_gpdfit(ary)
        rec.append((len(ary), k, sigma))
        return k, sigma

    cutoff_ind = -int(np.ceil(min(S / 5.0, 3 * (S / reff) ** 0.5))) - 1
    cutoffmin = np.log(np.finfo(float).tiny)
    tail_len = np.zeros(ll.shape[:-1], dtype=np.int64)
    xcut = np.zeros(ll.shape[:-1])
    sig = np.full(ll.shape[:-1], np.nan)
    with np.errstate(all="ignore"):
        lw, khat = psis.psislw(-ll, reff)  # loo.py:286-288 (method psis)
        # per-row intermediates, for debugging only
        psis._gpdfit = recording_gpdfit
        try:
            for idx in np.ndindex(ll.shape[:-1]):
                rec.clear()
                x = (-ll[idx]).copy()
                psis._psislw(x, cutoff_ind, cutoffmin)
                xs = -ll[idx] - np.max(-ll[idx])
                xc = max(np.sort(xs)[cutoff_ind], cutoffmin)
                xcut[idx] = xc
                tail_len[idx] = int(np.sum(xs > xc))
                if rec:
                    sig[idx] = rec[0][2]
        finally:
            psis._gpdfit = real_gpdfit
        lwll = lw + ll  # loo.py:289
        lse = utils.make_ufunc(utils._logsumexp, n_dims=1, ravel=False)
        loo_i = lse(lwll)  # loo.py:319-324, scale "log"
        lppd_i = lse(ll, b_inv=S)  # loo.py:329-337
    out = {
        "lw": lw,
        "khat": np.asarray(khat, dtype=np.float64),
        "loo_i": np.asarray(loo_i, dtype=np.float64),
        "lppd_i": np.asarray(lppd_i, dtype=np.float64),
        "xcutoff": xcut,
        "tail_len": tail_len,
        "sigma": sig,
        "cutoff_ind": np.int64(cutoff_ind),
    }
    # aggregates over the rows whose pointwise values are finite (a NaN row would
    # poison the sums; loo() never sees one because loo.py:218-227 replaces NaN)
    ok = np.isfinite(loo_i) & np.isfinite(lppd_i)
    li = np.asarray(loo_i)[ok]
    n = li.size
    good_k = min(1 - 1 / np.log10(S), 0.7)
    elpd = li.sum()  # loo.py:326
    se = (n * np.var(li)) ** 0.5  # loo.py:327
    lppd = np.sum(np.asarray(lppd_i)[ok])  # loo.py:329
    out.update(
        agg_rows=ok,
        elpd_loo=elpd,
        se=se,
        lppd=lppd,
        p_loo=lppd - elpd / 1,  # loo.py:339
        p_loo_se=np.sqrt(np.sum(np.var(li))),  # loo.py:340
        looic=-2 * elpd,
        looic_se=2 * se,  # loo.py:341-342
        good_k=good_k,
        n_high_k=np.int64(np.sum(np.asarray(khat)[ok] > good_k)),  # loo.py:292-293
    )
    return out


def main():
    mods = load_reference()
    psis, utils, sis, tis = mods["psis"], mods["utils"], mods["sis"], mods["tis"]

    for name, S, reff, dt, edges in cases.CASES:
        dtype = cases.DTYPES[dt]
        if S >= 20000:
            ll = np.concatenate(
                [cases.pareto_rows(S, [0.1, 0.5, 0.9, 1.2]), cases.gauss_rows(S, [1.0]), cases.bounded_row(S)]
            ).astype(dtype)
            labels = ["pareto_k0.1", "pareto_k0.5", "pareto_k0.9", "pareto_k1.2", "gauss_1", "bounded"]
        else:
            ll, labels = cases.standard_matrix(S, dtype, edges)
        save = {"ll": ll, "reff": np.float64(reff), "labels": np.array(labels)}
        if dt == "f32":
            nat = run_reference(mods, ll, reff)  # reference's own mixed precision
            for k in ("lw", "khat", "loo_i", "lppd_i"):
                save["native32_" + k] = nat[k]
            res = run_reference(mods, ll.astype(np.float64), reff)  # parity target
        else:
            res = run_reference(mods, ll, reff)
        save.update(res)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **save)
        print(name, ll.shape, "khat[:4]", res["khat"][:4], "elpd", res["elpd_loo"])

    # ---- known answer of SURVEY.md section 8c (values also printed in BASELINE.md) ----
    ll = cases.known_answer_matrix()
    res = run_reference(mods, ll, 1.0)
    np.savez_compressed(os.path.join(HERE, "known_answer_s4000.npz"), ll=ll, reff=np.float64(1.0), **res)
    print("known answer", res["khat"], res["elpd_loo"], res["se"], res["p_loo"])

    # ---- shapes: 1-D input -> 0-d khat; (d1, d2, S) input ----
    with np.errstate(all="ignore"):
        x1 = -cases.pareto_rows(100, [0.5])[0]
        lw1, k1 = psis.psislw(x1, 0.7)
        x3 = -cases.standard_matrix(100, np.float64, False)[0][:6].reshape(2, 3, 100)
        lw3, k3 = psis.psislw(x3, 0.7)
        small = np.array([1.0, 1.1, 1.2, 1.3])  # test_psis.py:95-99
        lws, ks = psis.psislw(small)
        const = np.ones(100)  # test_psis.py:121-125
        lwc, kc = psis.psislw(const)
    np.savez_compressed(
        os.path.join(HERE, "shapes.npz"),
        x1=x1, lw1=lw1, k1=np.asarray(k1), x3=x3, lw3=lw3, k3=np.asarray(k3),
        small=small, lw_small=lws, k_small=np.asarray(ks), const=const, lw_const=lwc, k_const=np.asarray(kc),
    )

    # ---- unit vectors for the 1-D primitives ----
    unit = {}
    with np.errstate(all="ignore"):
        for i, (n, k) in enumerate([(5, 0.2), (20, 0.5), (135, 0.3), (190, 0.7), (190, 1.2), (425, 0.9), (800, 0.1)]):
            p = (np.arange(n) + 0.5) / n
            ary = np.sort(np.expm1(-k * np.log1p(-p)) / k * (0.5 + 0.1 * i))
            kk, ss = psis._gpdfit(ary)
            unit[f"gpdfit_in_{i}"] = ary
            unit[f"gpdfit_out_{i}"] = np.array([kk, ss])
        gi = []
        for probs in ([0.1, 0.5, 0.9], [0.0, 0.5, 1.0], [-0.1, 0.5, 1.1]):  # test_psis.py:72-92
            for kappa in (-1, -0.5, 0, 0.5, 1):
                for sigma in (0, 1, 2):
                    gi.append(np.concatenate([probs, [kappa, sigma], psis._gpinv(np.array(probs), kappa, sigma)]))
        unit["gpinv_table"] = np.array(gi)
        v = cases.pareto_rows(100, [0.5, 1.2])
        unit["lse_in"] = v
        unit["lse_plain"] = np.array([utils._logsumexp(r) for r in v])
        unit["lse_binv"] = np.array([utils._logsumexp(r, b_inv=100) for r in v])
        unit["lse_b"] = np.array([utils._logsumexp(r, b=0.25) for r in v])
        v32 = v.astype(np.float32)
        unit["lse_f32"] = np.array([utils._logsumexp(r) for r in v32])
        for nm, f in (("sis", lambda r: sis._sislw(r)), ("tis", lambda r: tis._tislw(r, len(r)))):
            lws_, ess_ = [], []
            for r in -v:
                a, b = f(r.copy())
                lws_.append(a)
                ess_.append(b)
            unit[f"{nm}_lw"] = np.array(lws_)
            unit[f"{nm}_ess"] = np.array(ess_)
    np.savez_compressed(os.path.join(HERE, "units.npz"), **unit)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
