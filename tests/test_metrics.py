"""improvement_factor against the reference's own known answers (tests/metrics/test_improvement_factor.py)."""
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
