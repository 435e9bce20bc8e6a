"""Error type of the plugin boundary.

The reference raises ``BlackwaterException`` (blackwater/exception.py:1-5) when an estimator wrapper is handed an
observable that is not a Pauli sum (library/ngem/estimator.py:52-55, library/learning/estimator.py:225-228) and when the
improvement-factor metric gets an empty problem list; the same sites raise it here.
"""


class BlackwaterException(Exception):
    """Contract violation at the estimator / metrics boundary."""

    def __init__(self, message: str = ""):
        super().__init__(message)
        self.message = message
