"""Oracle (TEST INFRASTRUCTURE): split-conformal calibration in numpy.

The reference imports these functions from ``Neural_PDE.UQ.inductive_cp``
(``Marginal/NS_Residuals_CP.py:58``, ``Joint/Burgers_Residuals_CP.py:56-59``), an un-vendored
submodule (``.gitmodules:1-3``) that is empty in the reference snapshot.  Pin status per function:

* ``modulation_func``, ``ncf_metric_joint``, ``filter_sims_joint`` / ``emp_cov_joint``: **PINNED** to
  outputs of the reference's own in-tree statements -
  ``Tests/test_advection_inv_sampling_marginal.py:428`` (modulation), ``:430-431``
  (``conf_metric_joint``), ``:464-465`` (prediction sets + joint coverage) and
  ``Joint/Burgers_Residuals_CP.py:298-300`` (``filter_sims_joint``), compiled from those files and
  executed by ``tests/golden/make_golden.py`` -> ``tests/golden/conformal_ref.npz``, ``filters.npz``.
* ``filter_sims_within_bounds``: PINNED (``Active_Learning/Advection_AL_Marginal.py:169-198``).
* ``calibrate`` (and ``emp_cov``, its marginal counterpart): **PARITY UNPINNED, permanently** - no
  source and no restatement exists anywhere in the reference tree.  The form below is the standard
  split-CP quantile inferred from the call sites (``Marginal/Wave_Residuals_CP.py:253,288-290``,
  ``Joint/Burgers_Residuals_CP.py:257,283``); it is tested against numpy only.
"""
from __future__ import annotations

import numpy as np


def quantile_level(n, alpha):
    """q = ceil((n+1)(1-alpha))/n, float64, as numpy evaluates it."""
    return np.ceil((n + 1) * (1 - alpha)) / n


def kth_index(n, alpha):
    """0-based sorted index numpy's method='higher' picks for that level:
    ceil(q*(n-1)) in float64 (numpy/lib/_function_base_impl.py, 'higher')."""
    q = quantile_level(n, alpha)
    if not (0.0 <= q <= 1.0):
        raise ValueError("Quantiles must be in the range [0, 1]")
    return int(np.ceil((n - 1) * q))


def calibrate(scores, n, alpha):
    """q-hat along axis 0; shape ``scores.shape[1:]`` (scalar for 1-D scores)."""
    return np.quantile(scores, quantile_level(n, alpha), axis=0, method="higher")


def modulation_func(a, b):
    """Per-cell population std of (a-b) over the calibration axis (ddof=0)."""
    return np.std(a - b, axis=0)


def ncf_metric_joint(a, b, modulation):
    """Per-sample sup-norm of the modulated error."""
    return np.max(np.abs(a - b) / modulation, axis=tuple(range(1, a.ndim)))


def emp_cov(pred_sets, y):
    return ((y >= pred_sets[0]) & (y <= pred_sets[1])).mean()


def filter_sims_joint(pred_sets, y):
    """Joint/Burgers_Residuals_CP.py:298-300 (in the reference itself)."""
    axes = tuple(np.arange(1, len(y.shape)))
    return (y >= pred_sets[0]).all(axis=axes) & (y <= pred_sets[1]).all(axis=axes)


def filter_sims_within_bounds(lower_bound, upper_bound, samples, threshold, within=False):
    """Active_Learning/Advection_AL_Marginal.py:169-198 (in the reference itself; pinned by
    tests/golden/filters.npz): per-sample fraction of cells inside [lo,hi] (within) or on/outside
    the bounds (not within), compared with ``threshold``."""
    lo, hi, y = np.array(lower_bound), np.array(upper_bound), np.array(samples)
    hit = ((y >= lo) & (y <= hi)) if within else ((y <= lo) | (y >= hi))
    return hit.mean(axis=tuple(range(1, y.ndim))) >= threshold


def emp_cov_joint(pred_sets, y):
    return filter_sims_joint(pred_sets, y).mean()


ALPHA_LEVELS = np.arange(0.05, 0.95 + 0.1, 0.1)   # Marginal/Wave_Residuals_CP.py:284
