"""``psislw`` -- Pareto smoothed importance sampling (pyloo/psis.py:25-111) on the HIP engine."""

from .base import ISMethod, compute_importance_weights

__all__ = ["psislw"]


def psislw(log_weights, reff=1.0):
    """Pareto smoothed importance sampling (PSIS).

    Parameters and return values as in psis.py:25-111: ``log_weights`` is ``(..., S)`` (or a
    DataArray with a ``__sample__`` dimension), the result is ``(lw_out, kss)`` -- smoothed,
    truncated and normalised log weights plus the Pareto shape estimate per observation.
    """
    return compute_importance_weights(log_weights, method=ISMethod.PSIS, reff=reff)
