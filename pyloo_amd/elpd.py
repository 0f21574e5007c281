"""``ELPDData``: the result object of ``loo()`` -- a ``pandas.Series`` with a report printer.

Mirrors the LOO part of the reference container (pyloo/elpd.py:100-498): same index keys,
same properties, same printed report (README.md:76-84 of the reference), including the Pareto-k
table with bins ``(-inf, good_k], (good_k, 1], (1, inf)`` (elpd.py:300-330).  The k-fold,
LOGO, sub-sampling and non-factorised report variants are out of scope.
"""

from copy import copy as _copy
from copy import deepcopy as _deepcopy

import numpy as np
import pandas as pd

_REPORT = """
Computed from {n_samples} posterior samples and {n_points} observations log-likelihood matrix.

         Estimate       SE
elpd_loo   {elpd:<8.2f}    {se:<.2f}
p_loo       {p_loo:<8.2f}    {p_loo_se:<.2f}
looic      {looic:<8.2f}    {looic_se:<.2f}"""

_WAIC_REPORT = """
Computed from {n_samples} posterior samples and {n_points} observations log-likelihood matrix.

          Estimate       SE
elpd_waic   {elpd:<8.2f}    {se:<.2f}
p_waic       {p_waic:<8.2f}        -"""

_K_TABLE = """
------

Pareto k diagnostic values:
                         Count   Pct.
(-Inf, {gk:.2f}]   (good)      {c0:d}   {p0:.1f}%
   ({gk:.2f}, 1]   (bad)         {c1:d}    {p1:.1f}%
   (1, Inf)   (very bad)    {c2:d}    {p2:.1f}%"""

_ALL_GOOD = "\n\nAll Pareto k estimates are good (k < {gk:.1f}).\nSee help('pareto-k-diagnostic') for details."
_SOME_HIGH = (
    "\n\nSome Pareto k diagnostic values are high (k > {gk:.1f}), indicating that the importance"
    " sampling approximation is unreliable. Consider using moment matching or exact LOO for more"
    " accurate estimates. Use pointwise=True to see detailed diagnostics."
)
_WARNED = "\n\nThere has been a warning during the calculation. Please check the results."


class ELPDData(pd.Series):
    """Expected-log-pointwise-predictive-density results with a friendly ``print``."""

    _metadata = ["_method"]

    @property
    def _constructor(self):  # keep the subclass through pandas operations
        return ELPDData

    def __str__(self):
        kind = str(self.index[0]).split("_")[-1]
        if kind == "waic":
            # The reference's printer accepts the kind (elpd.py:125) but then reads loo-only keys; this
            # report follows its standard layout with the WAIC rows.
            text = _WAIC_REPORT.format(n_samples=self.n_samples, n_points=self.n_data_points, elpd=self["elpd_waic"],
                                       se=self["se"], p_waic=self["p_waic"])
            return text + (_WARNED if self.warning else "")
        if kind != "loo":
            raise ValueError("Invalid ELPDData object")
        tail = ""
        if "pareto_k" in self and self.get("good_k", None) is not None:
            gk = self["good_k"]
            kv = np.asarray(getattr(self["pareto_k"], "values", self["pareto_k"]), dtype=float).ravel()
            counts = np.histogram(kv, bins=np.asarray([-np.inf, gk, 1, np.inf]))[0]
            if counts[1] == 0 and counts[2] == 0:
                tail = _ALL_GOOD.format(gk=gk)
            else:
                pct = counts / counts.sum() * 100
                tail = _K_TABLE.format(gk=gk, c0=int(counts[0]), c1=int(counts[1]), c2=int(counts[2]),
                                       p0=pct[0], p1=pct[1], p2=pct[2])
        elif self.method == "psis":
            tail = (_SOME_HIGH if self.warning else _ALL_GOOD).format(gk=0.7)
        text = _REPORT.format(
            n_samples=self.n_samples, n_points=self.n_data_points, elpd=self["elpd_loo"], se=self["se"],
            p_loo=self["p_loo"], p_loo_se=self["p_loo_se"], looic=self["looic"], looic_se=self["looic_se"],
        )
        if self.warning:
            text += _WARNED
        return text + tail

    def __repr__(self):
        return self.__str__()

    def copy(self, deep=True):
        out = pd.Series.copy(self)
        for key in out.keys():
            out[key] = _deepcopy(out[key]) if deep else _copy(out[key])
        return ELPDData(out)

    @property
    def n_samples(self):
        return self["n_samples"]

    @property
    def n_data_points(self):
        return self["n_data_points"]

    @property
    def warning(self):
        return self["warning"]

    @property
    def method(self):
        return getattr(self, "_method", "psis")

    @method.setter
    def method(self, value):
        object.__setattr__(self, "_method", value)
