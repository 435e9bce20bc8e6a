"""improvement_factor against the reference's own known answers (tests/metrics/test_improvement_factor.py)."""
import numpy as np
import pytest

from blackwater.exception import BlackwaterException
from blackwater.metrics.improvement_factor import Problem, Trial, error_summary, improvement_factor


def test_reference_known_answers():
    assert improvement_factor([Problem([Trial(1.0, 2.0)], 0.0)], 1, 1) == 0.5
    two = [Problem([Trial(3.0, 4.0), Trial(1.0, 2.0)], 2.0), Problem([Trial(3.0, 4.0)], 2.0)]
    assert improvement_factor(two, n_shots=3, n_mitigation_shots=2) == 0.75
    assert improvement_factor([(0.0, [(1.0, 2.0)])], 1, 1) == 0.5
    assert improvement_factor([(2.0, [(3.0, 4.0), (1.0, 2.0)]), (2.0, [(3.0, 4.0)])], 3, 2) == 0.75
    with pytest.raises(BlackwaterException):
        improvement_factor([], 1, 1)


def test_error_summary_matches_golden_g3(g1):
    s = error_summary(g1["ideal"], g1["noisy"])
    assert round(s["mean_l2"], 6) == 0.027510  # h17_compare_over_steps.ipynb:513, L2_noisy step 0
    assert s["mae"] < s["rmse"]


def test_mitigation_report_matches_reference_formulas():
    """RMSE_noisy_q / RMSE_mitigated_q and their all-qubit versions as printed by the reference's evaluation cell
    (docs/tutorials/__ml_models.py:240-247), recomputed here the long way."""
    from blackwater.metrics.improvement_factor import mitigation_report

    rng = np.random.default_rng(0)
    ideal, noisy, mit = rng.normal(size=(50, 1, 4)), rng.normal(size=(50, 1, 4)), rng.normal(size=(50, 1, 4))
    rep = mitigation_report(ideal, noisy, mit)
    per_q = []
    for q in range(4):
        d = [np.square(i[0][q] - m[0][q]) for i, m in zip(ideal, mit)]
        per_q.append(np.mean(d))
        assert rep[f"RMSE_mitigated_{q}"] == pytest.approx(np.sqrt(np.mean(d)))
        assert rep[f"RMSE_noisy_{q}"] == pytest.approx(np.sqrt(np.mean([np.square(i[0][q] - n[0][q]) for i, n in zip(ideal, noisy)])))
    assert rep["RMSE_mitigated"] == pytest.approx(np.sqrt(np.mean(per_q)))
    assert rep["L2_noisy"] == pytest.approx(np.mean([np.linalg.norm(i[0] - n[0]) for i, n in zip(ideal, noisy)]))
    with pytest.raises(ValueError):
        mitigation_report(ideal, noisy[:10], mit)
