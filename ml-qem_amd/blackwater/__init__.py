"""MI355X-native drop-in for the ml-qem (``blackwater``) expectation-value-regressor hot path.

Public surface mirrors the reference package: ``blackwater.data.utils``, ``blackwater.data.generators.exp_val``,
``blackwater.data.loaders.exp_val``, ``blackwater.library.ngem.estimator``, ``blackwater.library.learning.estimator``.
The per-batch arithmetic lives in ``csrc/`` (HIP, gfx950) behind the C ABI declared in ``include/mlqem_hip.h``.
"""
