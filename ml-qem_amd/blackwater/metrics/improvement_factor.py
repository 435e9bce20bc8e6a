"""Improvement factor of a mitigation method (arXiv:2210.07194), as the reference reports it
(blackwater/metrics/improvement_factor.py:47-114):

    IF = sqrt(n_shots * sum (noisy - ideal)^2) / sqrt(n_mitigation_shots * sum (mitigated - ideal)^2)

summed over every trial of every problem.  A scalar host-side metric; plus the RMSE / mean-L2 /
MAE summaries the reference's notebooks print (docs/tutorials/__ml_models.py:230-253, h17_compare_over_steps cell [14]).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from math import sqrt
from typing import Any, List, Optional, Sequence, Tuple, Union

import numpy as np

from ..exception import BlackwaterException


@dataclass
class Trial:
    """One (noisy, mitigated) expectation-value pair."""

    noisy: float
    mitigated: float


@dataclass
class Problem:
    """Trials of one circuit/observable pair and its exact expectation value."""

    trials: List[Trial]
    ideal_exp_value: float
    circuit: Optional[Any] = None
    observable: Optional[Any] = None


def _as_problems(problems) -> List[Problem]:
    if len(problems) == 0:
        raise BlackwaterException("Problem list should not be empty.")
    if isinstance(problems[0], Problem):
        return list(problems)
    return [Problem([Trial(n, m) for n, m in trials], ideal) for ideal, trials in problems]


def improvement_factor(problems: Union[List[Problem], List[Tuple[float, List[Tuple[float, float]]]]], n_shots: int,
                       n_mitigation_shots: int) -> float:
    """``problems``: ``Problem`` objects or ``(ideal, [(noisy, mitigated), ...])`` tuples."""
    probs = _as_problems(problems)
    before = sum((t.noisy - p.ideal_exp_value) ** 2 for p in probs for t in p.trials)
    after = sum((t.mitigated - p.ideal_exp_value) ** 2 for p in probs for t in p.trials)
    return sqrt(n_shots * before) / sqrt(n_mitigation_shots * after)


def error_summary(ideal: Sequence, predicted: Sequence) -> dict:
    """RMSE (over all components), mean L2 distance per circuit and MAE -- the three numbers the reference's notebooks
    and BASELINE.md quote -- for [num_circuits, k] arrays."""
    a, b = np.atleast_2d(np.asarray(ideal, dtype=np.float64)), np.atleast_2d(np.asarray(predicted, dtype=np.float64))
    d = a - b
    return {"rmse": float(np.sqrt(np.mean(d ** 2))), "mean_l2": float(np.mean(np.linalg.norm(d, axis=1))),
            "mae": float(np.mean(np.abs(d)))}


def mitigation_report(ideal: Sequence, noisy: Sequence, mitigated: Sequence) -> dict:
    """The numbers the reference's evaluation cell prints after training (docs/tutorials/__ml_models.py:207-253):
    ``RMSE_noisy_q`` / ``RMSE_mitigated_q`` for every measured qubit q, their all-qubit counterparts
    (sqrt of the mean of the per-qubit mean squared distances), plus mean-L2 and MAE before/after for comparison with
    BASELINE.md's tables.  Inputs: [num_circuits, k] arrays (a trailing singleton axis of the dataset's [.,1,k] is fine)."""
    def as2d(v):
        v = np.asarray(v, dtype=np.float64)
        return v.reshape(v.shape[0], -1)

    a, n, m = as2d(ideal), as2d(noisy), as2d(mitigated)
    if not (a.shape == n.shape == m.shape):
        raise ValueError(f"shape mismatch: ideal {a.shape}, noisy {n.shape}, mitigated {m.shape}")
    sq_n, sq_m = (a - n) ** 2, (a - m) ** 2
    out = {}
    for q in range(a.shape[1]):
        out[f"RMSE_noisy_{q}"] = float(np.sqrt(sq_n[:, q].mean()))
        out[f"RMSE_mitigated_{q}"] = float(np.sqrt(sq_m[:, q].mean()))
    out["RMSE_noisy"] = float(np.sqrt(np.mean(sq_n.mean(axis=0))))
    out["RMSE_mitigated"] = float(np.sqrt(np.mean(sq_m.mean(axis=0))))
    before, after = error_summary(a, n), error_summary(a, m)
    out.update({"L2_noisy": before["mean_l2"], "L2_mitigated": after["mean_l2"], "MAE_noisy": before["mae"],
                "MAE_mitigated": after["mae"]})
    return out
