"""pyloo_amd -- MI355X-native PSIS-LOO engine behind pyloo's ``loo`` / ``psislw`` /
``compute_importance_weights`` API.  The compute path is hand-written HIP (gfx950) reached
through the C ABI in ``include/pyloo_amd.h``; there is no CPU fallback."""

from .base import ISMethod, compute_importance_weights
from .e_loo import ExpectationResult, compute_pareto_k, e_loo, k_hat
from .elpd import ELPDData
from .loo import loo, loo_from_matrix
from .loo_i import loo_i
from .loo_predictive_metric import loo_predictive_metric, predictive_metric_from_matrix
from .loo_score import LooScoreResult, loo_score, score_from_matrix
from .loo_subsample import loo_subsample, loo_subsample_from_matrix
from .psis import psislw
from .rcparams import rcParams
from .waic import waic, waic_from_matrix

__all__ = ["ISMethod", "ELPDData", "ExpectationResult", "compute_importance_weights", "compute_pareto_k", "e_loo", "k_hat", "loo", "loo_from_matrix", "loo_i", "loo_predictive_metric", "predictive_metric_from_matrix", "loo_score", "score_from_matrix", "LooScoreResult", "loo_subsample",
           "loo_subsample_from_matrix", "psislw", "rcParams", "waic",
           "waic_from_matrix"]
__version__ = "0.1.0"
