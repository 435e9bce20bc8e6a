"""Error type of the plugin boundary (reference: blackwater/exception.py:1-5)."""


class BlackwaterException(Exception):
    """Raised for contract violations at the estimator boundary (e.g. unsupported observable types)."""
