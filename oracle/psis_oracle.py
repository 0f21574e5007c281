"""CPU oracle for the PSIS-LOO hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module is a NumPy restatement of the arithmetic of the reference's hot path
(jordandeklerk/pyloo, ``pyloo/psis.py``, ``pyloo/utils.py:305-359``,
``pyloo/loo.py:286-342``, ``pyloo/sis.py:86-106``, ``pyloo/tis.py:91-120``, ``pyloo/e_loo.py:328-559``).  It exists
only so that ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` have something to check and time the HIP engine against.  Nothing under
``pyloo_amd/`` imports it; the product path has no CPU fallback.

Pinning: ``tests/test_oracle_golden.py`` checks every function below against the
fixtures in ``tests/golden/*.npz``, which were produced by the reference's real functions
(``tests/golden/make_golden.py``).  Parity is therefore pinned, not assumed.

Structure: one Python iteration per observation, like the reference's
``make_ufunc`` loop (``utils.py:171-175``) -- this is the "reference-faithful" mode that
BASELINE.md section 4 asks to be timed as the CPU baseline.
"""

import numpy as np

LOG_TINY = float(np.log(np.finfo(np.float64).tiny))  # psis.py:90 / base.py:142
EPS = float(np.finfo(np.float64).eps)


# --------------------------------------------------------------------------- helpers
def tail_count(n_draws, reff=1.0):
    """M = number of draws treated as the tail; ``cutoff_ind = -M - 1`` (base.py:139-141)."""
    return int(np.ceil(min(n_draws / 5.0, 3.0 * (n_draws / reff) ** 0.5)))


def good_k_threshold(n_draws):
    """loo.py:249."""
    return min(1.0 - 1.0 / np.log10(n_draws), 0.7)


def lse(v, b_inv=None):
    """Max-shifted log-sum-exp of a 1-D vector in the vector's own dtype (utils.py:305-359).

    ``b_inv`` divides the sum (``lse(v, b_inv=S)`` is the log of the mean).  NaN in ``v``
    gives NaN; an all ``-inf`` vector gives NaN (``-inf - -inf``), as in the reference.
    """
    v = np.asarray(v)
    if v.dtype.kind in "iu":
        v = v.astype(np.float64)
    ftype = v.dtype.type
    if b_inv is not None and b_inv == 0:
        return np.inf
    top = v.max()
    shifted = v - top
    np.exp(shifted, out=shifted)
    total = ftype(np.log(shifted.sum(dtype=v.dtype)))
    if b_inv is not None:
        top = ftype(top - np.log(b_inv))  # utils.py:353-354 (max is adjusted in the array dtype)
    return ftype(total + top)


def gpd_fit(y):
    """Zhang & Stephens (2009) empirical-Bayes GPD fit on ascending ``y``  (psis.py:163-208).

    Returns ``(k, sigma)`` with the weakly-informative prior on k (10 pseudo draws at 0.5).
    """
    y = np.asarray(y)
    n = y.shape[0]
    m = 30 + int(n**0.5)  # psis.py:184
    j = np.arange(1, m + 1, dtype=np.float64)
    theta = 1.0 - np.sqrt(m / (j - 0.5))  # psis.py:186
    theta = theta / (3 * y[int(n / 4 + 0.5) - 1])  # psis.py:187 (prior_bs = 3, first quartile)
    theta = theta + 1.0 / y[-1]  # psis.py:188
    kj = np.log1p(-theta[:, None] * y).mean(axis=1)  # psis.py:190
    prof = n * (np.log(-(theta / kj)) - kj - 1.0)  # psis.py:191
    w = 1.0 / np.exp(prof - prof[:, None]).sum(axis=1)  # psis.py:192
    keep = w >= 10 * EPS  # psis.py:194-197
    if not keep.all():
        w = w[keep]
        theta = theta[keep]
    w = w / w.sum()  # psis.py:198
    theta_hat = np.sum(theta * w)  # psis.py:201
    k_hat = np.log1p(-theta_hat * y).mean()  # psis.py:203
    sigma = -k_hat / theta_hat  # psis.py:205
    k_hat = (n * k_hat + 10 * 0.5) / (n + 10)  # psis.py:206 (prior_k = 10)
    return k_hat, sigma


def gpd_quantile(p, k, sigma):
    """Inverse CDF of the generalised Pareto distribution (psis.py:211-231)."""
    p = np.asarray(p, dtype=np.float64)
    q = np.full_like(p, np.nan)
    if sigma <= 0:  # psis.py:214-215
        return q
    inside = (p > 0) & (p < 1)
    if abs(k) < EPS:  # psis.py:218-219 / 224-225
        q[inside] = -np.log1p(-p[inside])
    else:
        q[inside] = np.expm1(-k * np.log1p(-p[inside])) / k
    q *= sigma
    if not inside.all():  # psis.py:229-230
        q[p == 0] = 0
        q[p == 1] = np.inf if k >= 0 else -sigma / k
    return q


def psis_row(logw, M, floor=LOG_TINY, details=None):
    """Pareto-smooth one vector of log importance ratios (psis.py:114-160).

    Returns ``(lw, k)``: normalised smoothed log weights (same dtype as ``logw``) and the
    tail-index estimate.  ``logw`` is not modified.  ``M`` is the tail count
    (``cutoff_ind = -M - 1``).
    """
    x = np.array(logw, copy=True)
    x -= x.max()  # psis.py:134
    kth = np.sort(x)[-M - 1]  # psis.py:135-136; NaNs sort last like argsort
    cut = max(kth, floor)  # python max: a NaN first argument wins (psis.py:136)
    e_cut = np.exp(cut)  # psis.py:138
    (where,) = np.nonzero(x > cut)  # psis.py:139 (strict: ties at the cutoff leave the tail)
    n_tail = where.shape[0]
    k = np.inf
    sigma = np.nan
    if n_tail > 4:  # psis.py:142-144
        tail = x[where]
        order = np.argsort(tail)  # psis.py:146
        y = np.exp(tail) - e_cut  # psis.py:147
        k, sigma = gpd_fit(y[order])  # psis.py:148
        if np.isfinite(k):  # psis.py:150
            p = np.arange(0.5, n_tail) / n_tail  # psis.py:153
            smooth = np.log(gpd_quantile(p, k, sigma) + e_cut)  # psis.py:154-155
            x[where[order]] = smooth  # psis.py:156 (rank-order scatter)
            x[x > 0] = 0  # psis.py:157
    x -= lse(x)  # psis.py:158
    if details is not None:
        details.update(xcutoff=cut, tail_len=n_tail, sigma=sigma)
    return x, k


def sis_row(logw):
    """Plain self-normalised importance sampling (sis.py:86-106): (lw, ESS)."""
    x = np.array(logw, copy=True)
    x -= x.max()
    x -= lse(x)
    w = np.exp(x)
    return x, 1.0 / np.sum(w**2)


def tis_row(logw, n_draws=None):
    """Truncated importance sampling, Ionides (2008) (tis.py:91-120): (lw, ESS)."""
    x = np.array(logw, copy=True)
    n_draws = x.shape[0] if n_draws is None else n_draws
    x -= x.max()
    log_z = lse(x) - np.log(n_draws)
    x = np.minimum(x, log_z + 0.5 * np.log(n_draws))
    x -= lse(x)
    w = np.exp(x)
    return x, 1.0 / np.sum(w**2)


# --------------------------------------------------------------------- batched fronts
def _rows(a):
    a = np.asarray(a)
    return a.reshape(-1, a.shape[-1]), a.shape[:-1]


def psislw(log_weights, reff=1.0):
    """(..., S) array -> (lw (..., S), k (...)); k is always float64 (psis.py:78-111)."""
    flat, lead = _rows(log_weights)
    M = tail_count(flat.shape[1], reff)
    lw = np.empty_like(flat)
    k = np.empty(flat.shape[0], dtype=np.float64)
    with np.errstate(all="ignore"):
        for i in range(flat.shape[0]):  # the loop of utils.py:171-175
            lw[i], k[i] = psis_row(flat[i], M)
    return lw.reshape(np.shape(log_weights)), k.reshape(lead)


def importance_weights(log_weights, method="psis", reff=1.0):
    """base.py:29-175 for ndarray input."""
    method = str(getattr(method, "value", method)).lower()
    if method == "psis":
        return psislw(log_weights, reff)
    if method not in ("sis", "tis"):
        raise ValueError(f"Invalid method '{method}'. Must be one of: psis, sis, tis")
    flat, lead = _rows(log_weights)
    lw = np.empty_like(flat)
    d = np.empty(flat.shape[0], dtype=np.float64)
    with np.errstate(all="ignore"):
        for i in range(flat.shape[0]):
            lw[i], d[i] = sis_row(flat[i]) if method == "sis" else tis_row(flat[i], flat.shape[1])
    return lw.reshape(np.shape(log_weights)), d.reshape(lead)


def loo_pointwise(ll, reff=1.0, method="psis"):
    """Per-observation pieces of loo.py:286-337 for an (N, S) log-likelihood matrix.

    Returns dict(khat|ess, loo_i (scale "log"), lppd_i, lw).
    """
    ll = np.asarray(ll)
    N, S = ll.shape
    with np.errstate(all="ignore"):
        lw, diag = importance_weights(-ll, method, reff)  # loo.py:286-288
        lwll = lw + ll  # loo.py:289
        loo_i = np.array([lse(r) for r in lwll], dtype=np.float64)  # loo.py:319-324
        lppd_i = np.array([lse(r, b_inv=S) for r in ll], dtype=np.float64)  # loo.py:329-337
    return {"diag": diag, "loo_i": loo_i, "lppd_i": lppd_i, "lw": lw}


def loo_aggregate(loo_i, lppd_i, khat, n_draws, scale_value=1):
    """loo.py:326-342 + 291-293 from pointwise values (``loo_i`` on the "log" scale)."""
    li = scale_value * np.asarray(loo_i, dtype=np.float64)
    n = li.size
    elpd = li.sum()
    se = (n * np.var(li)) ** 0.5
    lppd = np.sum(lppd_i)
    gk = good_k_threshold(n_draws)
    return {
        "elpd_loo": elpd,
        "se": se,
        "lppd": lppd,
        "p_loo": lppd - elpd / scale_value,
        "p_loo_se": np.sqrt(np.sum(np.var(li))),
        "looic": -2 * elpd,
        "looic_se": 2 * se,
        "good_k": gk,
        "n_high_k": int(np.sum(np.asarray(khat) > gk)),
    }


def loo_arrays(ll, reff=1.0, scale_value=1):
    """Everything ``loo()`` computes from an (N, S) matrix, as plain numbers."""
    pw = loo_pointwise(ll, reff)
    out = loo_aggregate(pw["loo_i"], pw["lppd_i"], pw["diag"], ll.shape[1], scale_value)
    out.update(khat=pw["diag"], loo_i=scale_value * pw["loo_i"], lppd_i=pw["lppd_i"], lw=pw["lw"])
    return out


# ---------------------------------------------------------------------------------------------
# WAIC (waic.py:109-160)
# ---------------------------------------------------------------------------------------------
def waic_sanitize(ll):
    """waic.py:112-135: NaN -> -1e10, +inf -> 1e10, -inf -> -1e10 (returns a copy and the two flags)."""
    ll = np.array(ll, dtype=np.float64, copy=True)
    has_nan, has_inf = bool(np.isnan(ll).any()), bool(np.isinf(ll).any())
    ll[np.isnan(ll)] = -1e10
    ll = np.where(np.isinf(ll), np.where(ll > 0, 1e10, -1e10), ll)
    return ll, has_nan, has_inf


def waic_arrays(ll, scale_value=1):
    """waic.py:137-161 on an (n_obs, n_draws) matrix: per-observation lppd_i (``_logsumexp`` with
    ``b_inv = n_samples``, one call per observation as ``wrap_xarray_ufunc`` does), the population
    variance over draws (xarray ``.var`` = ``np.var``, ddof 0), waic_i and the summaries."""
    ll, has_nan, has_inf = waic_sanitize(ll)
    n, s = ll.shape
    lppd_i = np.array([lse(r, b_inv=s) for r in ll], dtype=np.float64).reshape(n)   # waic.py:137-143
    vars_lpd = ll.var(axis=-1)                                                       # waic.py:145
    waic_i = scale_value * (lppd_i - vars_lpd)                                       # waic.py:158
    return {
        "lppd_i": lppd_i,
        "var_i": vars_lpd,
        "waic_i": waic_i,
        "elpd_waic": np.sum(waic_i),                                                 # waic.py:160
        "se": (n * np.var(waic_i)) ** 0.5,                                           # waic.py:159
        "p_waic": np.sum(vars_lpd),                                                  # waic.py:161
        "warning": bool(np.any(vars_lpd > 0.4)),                                     # waic.py:147
        "has_nan": has_nan,
        "has_inf": has_inf,
    }


# ---------------------------------------------------------------------------------------------
# Weighted expectations (e_loo.py) -- SURVEY section 8 f4
# ---------------------------------------------------------------------------------------------
def k_hat_row(h, log_ratios, tail_len=20):
    """Function-specific Pareto k of one observation (e_loo.py:328-390), followed line by line -- INCLUDING the order in
    which the reference hands the tails to ``_gpdfit``: ``sorted_r - cutoff`` etc. are DESCENDING with a last element of
    exactly 0, where ``_gpdfit`` expects ascending values; ``1 / ary[-1]`` is then +-inf, every grid weight NaN, none
    survives ``w >= 10 eps`` and the fit degenerates to ``k = (n * 0 + 5) / (n + 10)`` (1/6 for a 20-draw tail).  That is what
    the reference returns, so that is what is restated (and what the fixtures of make_golden_e_loo.py hold)."""
    lr = np.asarray(log_ratios)
    with np.errstate(all="ignore"):
        r = np.exp(lr - np.max(lr))                                   # e_loo.py:350
        top = -np.sort(-r)[:tail_len]                                 # 351
        if len(top) < 5 or np.allclose(top, top[0]):                  # 353-354
            k_r = np.inf
        else:
            k_r, _ = gpd_fit(top - top[-1])                           # 356-357
        if (h is None or np.allclose(h, h[0]) or len(np.unique(h)) == 2 or np.any(np.isnan(h))
                or np.any(np.isinf(h))):                              # 359-366
            return k_r
        hr = h * r                                                    # 368
        left = np.sort(hr)[:tail_len]                                 # 370
        right = -np.sort(-hr)[:tail_len]                              # 371
        if len(left) < 5 or np.allclose(left, left[0]):               # 373-377
            k_left = -np.inf
        else:
            k_left, _ = gpd_fit(-(left - left[-1]))
        if len(right) < 5 or np.allclose(right, right[0]):            # 379-383
            k_right = -np.inf
        else:
            k_right, _ = gpd_fit(right - right[-1])
        k_hr = max(k_left, k_right)                                   # 385 (Python max: NaN-order dependent, kept)
        if np.isnan(k_hr) and np.isnan(k_r):                          # 387-388
            return np.nan
        return max(k_hr, k_r)                                         # 390


def weighted_variance_row(x, w):
    """e_loo.py:518-531 (``w`` normalised)."""
    if np.allclose(x, x[0]):
        return 0.0
    wss = np.sum(w**2)
    if np.isclose(wss, 1.0):
        return 0.0
    mean = np.sum(w * x)
    mean_sq = np.sum(w * x**2)
    return max((mean_sq - mean**2) / (1 - wss), 0.0)


def weighted_quantile_row(x, w, prob, stable=False):
    """e_loo.py:534-554.  ``stable=True`` orders equal draws by index (the reference's ``np.argsort`` is unstable, so which
    of a group of equal draws comes first -- the only one that interpolates from the value below -- is arbitrary there; the
    engine takes the lowest index)."""
    if np.allclose(w, w[0]):
        return np.quantile(x, prob)
    order = np.argsort(x, kind="stable") if stable else np.argsort(x)
    xs, ws = x[order], w[order]
    ww = np.cumsum(ws) / np.sum(ws)
    ids = np.where(ww >= prob)[0]
    if len(ids) == 0:
        return xs[-1]
    wi = ids[0]
    if wi == 0:
        return xs[0]
    w1, x1 = ww[wi - 1], xs[wi - 1]
    return x1 + (xs[wi] - x1) * (prob - w1) / (ww[wi] - w1)


def pareto_min_ss(k):
    """e_loo.py:393-398."""
    return 10 ** (1 / (1 - max(0, k))) if k < 1 else float("inf")


def pareto_khat_threshold(n_samples):
    """e_loo.py:401-403."""
    return 1 - 1 / np.log10(n_samples)


def pareto_convergence_rate(k, n_samples):
    """e_loo.py:406-427."""
    if k < 0:
        return 1.0
    if k > 1:
        return 0.0
    if k == 0.5:
        return 1 - 1 / np.log(n_samples)
    if 0 < k < 1:
        n = n_samples
        return max(0, (2 * (k - 1) * n ** (2 * k + 1) + (1 - 2 * k) * n ** (2 * k) + n**2) / ((n - 1) * (n - n ** (2 * k))))
    return 1.0


def e_loo_arrays(x, log_weights, log_ratios=None, probs=None):
    """``e_loo`` (e_loo.py:56-264) for (N, S) arrays, every ``type`` at once: weighted mean (430-437), variance (440-459),
    sd (462-465), quantiles for ``probs`` (468-515), and the k of each type (226-236: h = x for the mean, x**2 for variance
    and sd, None for quantiles; ``log_ratios`` defaults to the log-weights, 223-224)."""
    x = np.asarray(x)
    lw = np.asarray(log_weights)
    lr = lw if log_ratios is None else np.asarray(log_ratios)
    n, s = x.shape
    with np.errstate(all="ignore"):
        nlw = np.stack([row - lse(row) for row in lw])                # 557-559, one _logsumexp per row
        w = np.exp(nlw)
        out = {
            "mean": (w * x).sum(axis=-1),
            "var": np.array([weighted_variance_row(x[i], w[i]) for i in range(n)], dtype=np.float64),
            "k_mean": np.array([k_hat_row(x[i], lr[i]) for i in range(n)], dtype=np.float64),
            "k_var": np.array([k_hat_row(x[i] ** 2, lr[i]) for i in range(n)], dtype=np.float64),
            "k_none": np.array([k_hat_row(None, lr[i]) for i in range(n)], dtype=np.float64),
        }
        out["sd"] = np.sqrt(out["var"])
        if probs is not None:
            out["quant"] = np.array([[weighted_quantile_row(x[i], w[i], p) for p in np.atleast_1d(probs)] for i in range(n)],
                                    dtype=np.float64)
    return out



# ---- loo_predictive_metric / loo_score: the closed forms around psislw + e_loo ----------------------------------------
def predictive_metric(y, yhat, metric):
    """{"estimate", "se"} of one metric (pyloo loo_predictive_metric.py:234-356)."""
    y, yhat = np.asarray(y, dtype=np.float64), np.asarray(yhat, dtype=np.float64)
    n = len(y)
    if metric in ("mae", "mse", "rmse"):
        e = np.abs(y - yhat) if metric == "mae" else (y - yhat) ** 2      # 247-250, 268-271
        est, se = np.mean(e), np.std(e, ddof=1) / np.sqrt(n)
        if metric == "rmse":                                               # 291-298: first-order Taylor expansion
            return {"estimate": np.sqrt(est), "se": np.sqrt(se**2 / est / 4)}
        return {"estimate": est, "se": se}
    hit = (yhat > 0.5).astype(int) == y                                    # 319-320, 348
    if metric == "acc":
        est = np.mean(hit.astype(int))
        return {"estimate": est, "se": np.sqrt(est * (1 - est) / n)}       # 321-326
    neg = y == 0                                                           # 349-356
    tn, tp = np.mean(hit[neg]), np.mean(hit[~neg])
    return {"estimate": (tp + tn) / 2, "se": np.sqrt((tp * (1 - tp) + tn * (1 - tn)) / 4 / n)}


def loo_predictive_metric_arrays(x, ll, y, metric="mae", reff=1.0):
    """loo_predictive_metric.py:208-231 on (n_obs, n_draws) arrays: psislw(-ll), weighted mean of x, the metric."""
    lw, _ = psislw(-ll, reff)
    pred = e_loo_arrays(x, lw, -ll)["mean"]
    return predictive_metric(y, pred, metric)


def crps(exx, exy, scale=False):
    """loo_score.py:326-346."""
    return -exy / exx - 0.5 * np.log(exx) if scale else 0.5 * exx - exy


def loo_score_arrays(x, x2, y, ll, reff=1.0, permutations=1, scale=False):
    """Pointwise LOO-CRPS / SCRPS on (n_obs, n_draws) arrays (loo_score.py:219-239, 277-323); draws the pairings from
    NumPy's global generator like the reference (305)."""
    s = x.shape[-1]
    exx = 0.0
    for _ in range(permutations):
        shuffle = np.random.permutation(s)
        joint = -ll - ll[:, shuffle]
        lw, _ = psislw(joint, reff)
        exx = exx + e_loo_arrays(np.abs(x - x2[:, shuffle]), lw, joint)["mean"]
    exx = exx / permutations
    lw, k = psislw(-ll, reff)
    exy = e_loo_arrays(np.abs(x - np.asarray(y).reshape(-1, 1)), lw, -ll)["mean"]
    return crps(exx, exy, scale), k


# ---------------------------------------------------------------------------------------------
# Vectorised NumPy backend (the second CPU line of the bench).  Same arithmetic as
# ``loo_pointwise`` with the per-observation Python loop replaced by whole-matrix NumPy calls; rows
# whose tail is not exactly M long (ties at the cutoff, the log(DBL_MIN) floor, non-finite entries) fall
# back to the row-by-row code.  Not the reference's code path; with cache-sized chunks it runs at about the
# speed of the loop (each row's NumPy calls are already vectorised over S), which is why the loop is the baseline.
# ---------------------------------------------------------------------------------------------
def loo_pointwise_vectorised(ll, reff=1.0, chunk=64):
    ll = np.asarray(ll, dtype=np.float64)
    N, S = ll.shape
    M = tail_count(S, reff)
    khat = np.empty(N)
    loo_i = np.empty(N)
    lppd_i = np.empty(N)
    logS = np.log(S)
    for lo in range(0, N, chunk):
        blk = ll[lo:lo + chunk]
        n = blk.shape[0]
        x = -blk
        x = x - x.max(axis=1, keepdims=True)                      # psis.py:134
        order = np.argsort(x, axis=1)                              # psis.py:135
        xs = np.take_along_axis(x, order, axis=1)
        cut = xs[:, -M - 1]
        tail = xs[:, -M:]                                          # ascending (psis.py:146)
        plain = np.isfinite(x).all(axis=1) & (cut > LOG_TINY) & (tail[:, 0] > cut)   # exactly M strictly above the cutoff
        e_cut = np.exp(cut)[:, None]
        with np.errstate(all="ignore"):
            y = np.exp(tail) - e_cut                               # psis.py:147
            # ---- _gpdfit for every row at once (psis.py:163-208) ----
            m_est = 30 + int(M**0.5)
            j = np.arange(1, m_est + 1, dtype=np.float64)
            theta = 1.0 - np.sqrt(m_est / (j - 0.5))
            theta = theta[None, :] / (3 * y[:, int(M / 4 + 0.5) - 1])[:, None] + (1.0 / y[:, -1])[:, None]
            kj = np.log1p(-theta[:, :, None] * y[:, None, :]).mean(axis=2)
            prof = M * (np.log(-(theta / kj)) - kj - 1.0)
            w = 1.0 / np.exp(prof[:, None, :] - prof[:, :, None]).sum(axis=2)
            w = np.where(w >= 10 * EPS, w, 0.0)
            w = w / w.sum(axis=1, keepdims=True)
            theta_hat = (theta * w).sum(axis=1)
            k_raw = np.log1p(-theta_hat[:, None] * y).mean(axis=1)
            sigma = -k_raw / theta_hat
            k = (M * k_raw + 5.0) / (M + 10.0)
            plain &= np.isfinite(k) & (sigma > 0) & (np.abs(k) >= EPS)
            # ---- _gpinv + scatter + normalise (psis.py:150-158) ----
            p = np.arange(0.5, M) / M
            q = sigma[:, None] * np.expm1(-k[:, None] * np.log1p(-p)[None, :]) / k[:, None]
            smooth = np.minimum(np.log(q + e_cut), 0.0)
            lw = x.copy()
            np.put_along_axis(lw, order[:, -M:], smooth, axis=1)
            top = lw.max(axis=1, keepdims=True)
            lw -= np.log(np.exp(lw - top).sum(axis=1, keepdims=True)) + top
            t = lw + blk                                           # loo.py:289
            tt = t.max(axis=1, keepdims=True)
            li = (np.log(np.exp(t - tt).sum(axis=1, keepdims=True)) + tt)[:, 0]
            bt = blk.max(axis=1, keepdims=True)
            lp = (np.log(np.exp(blk - bt).sum(axis=1, keepdims=True)) + bt)[:, 0] - logS
        khat[lo:lo + n], loo_i[lo:lo + n], lppd_i[lo:lo + n] = k, li, lp
        for r in np.nonzero(~plain)[0]:                            # the rows the batch formulas do not cover
            one = loo_pointwise(blk[r:r + 1], reff)
            khat[lo + r], loo_i[lo + r], lppd_i[lo + r] = one["diag"][0], one["loo_i"][0], one["lppd_i"][0]
    return {"diag": khat, "loo_i": loo_i, "lppd_i": lppd_i}
