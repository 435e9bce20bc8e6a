"""The few qiskit primitives types the estimator wrappers touch, resolved lazily.

When qiskit is installed its own ``EstimatorResult`` / ``JobV1`` / ``transpile`` are used, so the wrappers plug into
``qiskit.primitives.BaseEstimator.run`` exactly like the reference's.  Without qiskit (this build's test boxes) light
stand-ins with the same attributes take their place; nothing here is imported from qiskit at module scope.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Sequence

import numpy as np

from ..data.circuit import Circuit


@dataclass
class EstimatorResultLite:
    values: np.ndarray
    metadata: List[Dict[str, Any]] = field(default_factory=list)


def make_estimator_result(values, metadata):
    try:
        from qiskit.primitives import EstimatorResult  # type: ignore

        return EstimatorResult(values, metadata)
    except Exception:  # qiskit absent (or too new/old for this signature)
        return EstimatorResultLite(values, metadata)


def job_base():
    try:
        from qiskit.providers import JobV1  # type: ignore

        return JobV1
    except Exception:
        return object


def transpile_and_bind(circuit: Any, backend: Any, params: Sequence[float], transpile_options: Dict[str, Any],
                       do_transpile: bool = True, keep_text: bool = False):
    """``transpile(circuit, backend, **options).bind_parameters(params)`` for qiskit circuits; circuits given as
    QASM text or as this package's ``Circuit`` are already in the backend basis and fully bound.  ``keep_text``: a
    consumer that scans OpenQASM text natively (the batched paths) gets the text back as it came -- OpenQASM 2 has no free
    parameters, and turning 2e4 statements into Python objects costs ~0.4 s per circuit."""
    if hasattr(circuit, "data") and hasattr(circuit, "qubits") and not isinstance(circuit, Circuit):
        if do_transpile:
            from qiskit import transpile  # type: ignore

            circuit = transpile(circuit, backend, **transpile_options)
        binder = getattr(circuit, "bind_parameters", None) or getattr(circuit, "assign_parameters")
        return binder(params)
    if keep_text and isinstance(circuit, str) and len(params) == 0:
        return circuit
    return Circuit.from_any(circuit).bind_parameters(params)


def model_device(model):
    import torch

    if isinstance(model, torch.nn.Module):
        for p in model.parameters():
            return p.device
    return None
