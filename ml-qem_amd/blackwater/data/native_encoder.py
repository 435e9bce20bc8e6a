"""ctypes front-end of the native (C++) circuit encoder ``mlqem_encode_qasm`` -- the fast path behind
``circuit_to_graph_data_json`` for the arrays the models actually consume (x, op->op edges, edge_attr, depth)."""
from __future__ import annotations

import ctypes
from typing import Any, Dict, Tuple

import numpy as np

from ..native import _lib


class _Props(ctypes.Structure):
    _fields_ = [("num_qubits", ctypes.c_int), ("t1", ctypes.POINTER(ctypes.c_double)),
                ("t2", ctypes.POINTER(ctypes.c_double)), ("readout", ctypes.POINTER(ctypes.c_double)),
                ("num_gate_types", ctypes.c_int), ("gate_names", ctypes.POINTER(ctypes.c_char_p)),
                ("num_gate_props", ctypes.c_int), ("gate_keys", ctypes.POINTER(ctypes.c_char_p)),
                ("gate_error", ctypes.POINTER(ctypes.c_double)), ("gate_length", ctypes.POINTER(ctypes.c_double))]


class NativeEncoder:
    """Holds the C view of one ``properties`` dict (as returned by ``get_backend_properties_v1``)."""

    def __init__(self, properties: Dict[str, Any]):
        qp = properties["qubits_props"]
        n = len(qp)
        get = lambda i: qp[i] if i in qp else qp[str(i)]
        dbl = lambda vals: np.ascontiguousarray(vals, dtype=np.float64)
        self._t1, self._t2 = dbl([get(i)["t1"] for i in range(n)]), dbl([get(i)["t2"] for i in range(n)])
        self._ro = dbl([get(i)["readout_error"] for i in range(n)])
        names = [g.encode() for g in properties["gates_set"]]
        keys = list(properties["gate_props"].keys())
        self._ge = dbl([properties["gate_props"][k].get("gate_error", 0.0) for k in keys])
        self._gl = dbl([properties["gate_props"][k].get("gate_length", 0.0) for k in keys])
        self._names = (ctypes.c_char_p * len(names))(*names)
        self._keys = (ctypes.c_char_p * max(len(keys), 1))(*[k.encode() for k in keys])
        dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        self._props = _Props(n, dp(self._t1), dp(self._t2), dp(self._ro), len(names), self._names, len(keys),
                             self._keys, dp(self._ge), dp(self._gl))
        self._lib = _lib.load()

    def encode(self, qasm: str, use_gate_features: bool = True, use_qubit_features: bool = True):
        """-> (x [N,F] f64, edge_index [2,E] int64, edge_attr [E,3] f64, depth)."""
        lib, text = self._lib, qasm.encode()
        n, e = ctypes.c_int64(0), ctypes.c_int64(0)
        f, d = ctypes.c_int(0), ctypes.c_int(0)
        args = (text, ctypes.byref(self._props), int(use_qubit_features), int(use_gate_features), ctypes.byref(n),
                ctypes.byref(e), ctypes.byref(f), ctypes.byref(d))
        code = lib.mlqem_encode_qasm(*args, None, None, None, None)
        if code != 0:
            self._raise(code)
        x = np.empty((n.value, f.value), dtype=np.float64)
        src, dst = np.empty(e.value, dtype=np.int32), np.empty(e.value, dtype=np.int32)
        attr = np.empty((e.value, 3), dtype=np.float64)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        code = lib.mlqem_encode_qasm(*args, vp(x), vp(src), vp(dst), vp(attr))
        if code != 0:
            self._raise(code)
        return x, np.stack([src, dst]).astype(np.int64), attr, d.value

    def _raise(self, code):
        msg = self._lib.mlqem_encode_last_error().decode()
        if "not in the backend's gates_set" in msg:
            raise KeyError(msg)
        raise Exception(msg or f"mlqem_encode_qasm failed with code {code}")


def circuit_features(qasm: str, gate_names, bin_edges) -> Tuple[np.ndarray, np.ndarray]:
    """(gate counts [len(gate_names)], rotation-angle histogram [len(bin_edges) - 1]) of one OpenQASM-2 circuit through
    ``mlqem_circuit_features_qasm`` -- the per-circuit part of ``encode_data`` / ``encode_data_v2_ecr``."""
    lib = _lib.load()
    names = [g.encode() for g in gate_names]
    c_names = (ctypes.c_char_p * max(len(names), 1))(*names)
    edges = np.ascontiguousarray(bin_edges, dtype=np.float64)
    counts = np.zeros(len(names), dtype=np.int64)
    hist = np.zeros(max(len(edges) - 1, 0), dtype=np.int64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    code = lib.mlqem_circuit_features_qasm(qasm.encode(), c_names, len(names), vp(edges), len(edges), vp(counts), vp(hist))
    if code != 0:
        raise Exception(lib.mlqem_encode_last_error().decode() or f"mlqem_circuit_features_qasm failed with code {code}")
    return counts, hist
