"""Duck-typed stand-ins for the two qiskit objects the data layer reads: a V1 backend's calibration
snapshot and a Pauli-sum observable.  They exist so the encoders run where qiskit is not installed; when a
real qiskit ``BackendV1`` / ``SparsePauliOp`` is passed instead, the same accessors are used on it.
"""
from __future__ import annotations

import json
from typing import Any, Dict, Iterable, List, Sequence, Tuple

_PREFIX_POW10 = {"f": -15, "p": -12, "n": -9, "u": -6, "µ": -6, "m": -3, "k": 3, "M": 6, "G": 9, "T": 12}


def _apply_unit_prefix(value: float, unit: str) -> float:
    """Scales ``value`` given in ``unit`` (e.g. 'us') to the SI base unit, dividing for negative powers so
    59.69864328663569 us -> 5.9698643286635694e-05 exactly as the reference datasets hold it
    (what ``BackendProperties.qubit_property`` does inside blackwater/data/utils.py:163-168)."""
    if not unit or len(unit) == 1 or unit[0] not in _PREFIX_POW10:
        return value
    p = _PREFIX_POW10[unit[0]]
    return value / (10 ** -p) if p < 0 else value * (10 ** p)


class _Nduv:
    def __init__(self, d: Dict[str, Any]):
        self.name, self.unit, self.value = d["name"], d.get("unit", ""), d["value"]


class _GateProps:
    def __init__(self, d: Dict[str, Any]):
        self.gate = d["gate"]
        self.qubits = list(d["qubits"])
        self.parameters = [_Nduv(p) for p in d["parameters"]]
        self.name = d.get("name", self.gate)


class BackendPropertiesLite:
    """Subset of qiskit ``BackendProperties`` built from its ``to_dict()`` form."""

    def __init__(self, props: Dict[str, Any]):
        self._raw = props
        self.backend_name = props.get("backend_name", "backend")
        self.qubits = [[_Nduv(p) for p in q] for q in props["qubits"]]
        self.gates = [_GateProps(g) for g in props["gates"]]

    def qubit_property(self, qubit: int) -> Dict[str, Tuple[float, Any]]:
        return {p.name: (_apply_unit_prefix(p.value, p.unit), None) for p in self.qubits[qubit]}

    def to_dict(self) -> Dict[str, Any]:
        return self._raw


class StaticBackend:
    """A V1-style backend that only knows its name and calibration snapshot."""

    def __init__(self, name: str, properties: Dict[str, Any] | BackendPropertiesLite):
        self._name = name
        self._props = properties if isinstance(properties, BackendPropertiesLite) else BackendPropertiesLite(properties)

    def name(self) -> str:
        return self._name

    def properties(self) -> BackendPropertiesLite:
        return self._props

    @staticmethod
    def from_json(path: str, name: str | None = None) -> "StaticBackend":
        with open(path, "r") as fh:
            raw = json.load(fh)
        return StaticBackend(name or "fake_" + raw.get("backend_name", "backend").replace("ibmq_", ""), raw)


class PauliObservable:
    """Sum of Pauli strings with real/complex coefficients (the slice of ``SparsePauliOp`` the path uses)."""

    def __init__(self, terms: Iterable[Tuple[str, complex]] | str):
        if isinstance(terms, str):
            terms = [(terms, 1.0)]
        self._terms: List[Tuple[str, complex]] = [(str(l), c) for l, c in terms]

    def to_list(self) -> List[Tuple[str, complex]]:
        return list(self._terms)

    # SparsePauliOp-like views used by TorchLearningModelProcessor (learning/estimator.py:170-181)
    @property
    def paulis(self) -> List[str]:
        return [l for l, _ in self._terms]

    @property
    def coeffs(self) -> List[complex]:
        return [c for _, c in self._terms]

    def __iter__(self):
        for l, c in self._terms:
            yield PauliObservable([(l, c)])

    def __len__(self):
        return len(self._terms)

    def __repr__(self):
        return f"PauliObservable({self._terms!r})"


def pauli_terms(op: Any) -> List[Tuple[str, complex]]:
    """(label, coeff) pairs of a Pauli-sum observable given as SparsePauliOp / PauliSumOp / PauliObservable /
    a plain label / a list of pairs."""
    if isinstance(op, str):
        return [(op, 1.0)]
    if isinstance(op, (list, tuple)):
        return [(str(l), c) for l, c in op]
    prim = getattr(op, "primitive", None)  # opflow PauliSumOp wraps a SparsePauliOp
    if prim is not None and hasattr(prim, "to_list"):
        scale = getattr(op, "coeff", 1.0)
        return [(str(l), c * scale) for l, c in prim.to_list()]
    if hasattr(op, "to_list"):
        return [(str(l), c) for l, c in op.to_list()]
    raise TypeError(f"not a Pauli-sum observable: {type(op).__name__}")


def is_pauli_observable(op: Any) -> bool:
    """The isinstance(obs, (PauliSumOp, SparsePauliOp)) gate of the estimators, duck-typed."""
    if isinstance(op, PauliObservable):
        return True
    return type(op).__name__ in ("PauliSumOp", "SparsePauliOp")
