"""Circuit-level feature rows for the MLP regressors (host side).

Reference: docs/tutorials/mlp.py:124-252 (== blackwater/library/learning/mlp.py:111-203):
``encode_data`` -> ``[8 backend means x100 | gate counts x0.01 | rz/rx/ry angle histogram x0.01 | noisy | basis]``
and ``encode_data_v2_ecr`` (no backend block, gate list ``[two_q, sx, x, id, rz]``, 160 bins).

Two reference behaviours are load-bearing for the checkpoints and are kept:
the backend means select gate records by SUBSTRING of the record key ('x' also picks 'cx_*' and 'sx_*'),
and qubit 0 is left out of the T1/T2/readout means (its dict key ``0`` is falsy in the reference's filter).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ...data.circuit import Circuit

_ROTATIONS = ("rx", "ry", "rz")


def count_gates_by_rotation_angle(circuit: Any, bin_size: float) -> List[int]:
    """Histogram of single-qubit rx/ry/rz angles over [-2pi, 2pi] in bins of ``bin_size``."""
    circ = Circuit.from_any(circuit)
    angles = [float(op.params[0]) for op in circ.ops if op.name in _ROTATIONS and len(op.qubits) == 1]
    edges = np.arange(-2 * np.pi, 2 * np.pi + bin_size, bin_size)
    counts, _ = np.histogram(angles, bins=edges)
    return [int(c) for c in counts]


def recursive_dict_loop(my_dict, parent_key=None, out=None, target_key1=None, target_key2=None):
    """Collects ``val`` of every leaf ``target_key2`` whose enclosing dict's key is truthy and contains
    ``target_key1`` as a substring; ``0.`` when nothing matched."""
    found = [] if out is None else out
    # depth-first in insertion order (order only matters to float summation order of the mean)
    def walk(d, parent):
        for key, val in d.items():
            if isinstance(val, dict):
                walk(val, key)
            elif parent and target_key1 in str(parent) and key == target_key2:
                found.append(val)
    walk(my_dict, parent_key)
    return found or 0.


def backend_summary_vector(properties: Dict[str, Any]) -> torch.Tensor:
    """The 8 backend means (cx, id, sx, x, rz gate errors; readout, T1, T2), times 100 (float64)."""
    selectors = [("cx", "gate_error"), ("id", "gate_error"), ("sx", "gate_error"), ("x", "gate_error"),
                 ("rz", "gate_error"), ("", "readout_error"), ("", "t1"), ("", "t2")]
    vec = [np.mean(recursive_dict_loop(properties, out=[], target_key1=a, target_key2=b)) for a, b in selectors]
    return torch.tensor(np.asarray(vec, dtype=np.float64)) * 100


def _unwrap_single(noisy_exp_vals):
    if isinstance(noisy_exp_vals[0], list) and len(noisy_exp_vals[0]) == 1:
        return [v[0] for v in noisy_exp_vals]
    return noisy_exp_vals


def _fill_rows(X, circuits, gates_set, offset, bin_size, n_bins, n_vals, noisy_exp_vals, meas_bases, native=False):
    c0, c1 = offset, offset + len(gates_set)
    a1 = c1 + n_bins
    v1 = a1 + n_vals
    edges = np.arange(-2 * np.pi, 2 * np.pi + bin_size, bin_size)
    if native:
        # the C++ op scan shared with the graph encoder (mlqem_circuit_features_qasm): same integers, no Python parse; the rows
        # of a whole run() are filled by three array assignments instead of three tensor constructions per circuit
        from ...data.native_encoder import circuit_features_batch

        for circuit in circuits:
            if not isinstance(circuit, str):
                raise TypeError("native=True takes OpenQASM-2 text (the 'circuit' field of the reference's datasets)")
        counts, hists = circuit_features_batch(list(circuits), gates_set, edges)      # host threads over the circuits
        for i in range(len(circuits)):
            if n_vals > 1:
                assert len(noisy_exp_vals[i]) == n_vals
            elif n_vals == 1:
                assert isinstance(noisy_exp_vals[i], float)
        if len(circuits):
            # integer counts times a python float promote to float32, as in the reference: float32(count) * float32(0.01).  Filled through
            # numpy on the tensor's own memory: the same three assignments as torch CPU ops open an OpenMP region each, which on a host
            # whose CPU share is below its core count costs tens of milliseconds per region (0.15 s of a 0.21 s process_batch of 1024)
            xn = X.numpy()
            xn[:, c0:c1] = counts.astype(np.float32) * np.float32(0.01)
            xn[:, c1:a1] = hists.astype(np.float32) * np.float32(0.01)
            xn[:, a1:v1] = np.asarray(noisy_exp_vals, dtype=np.float32).reshape(len(circuits), -1)
    for i, circuit in enumerate(circuits if not native else []):
        circ = Circuit.from_any(circuit)
        tally = circ.count_ops()
        counts, hist = [tally.get(g, 0) for g in gates_set], count_gates_by_rotation_angle(circ, bin_size)
        # integer tensors times a python float promote to float32, as in the reference
        X[i, c0:c1] = torch.tensor(counts) * 0.01
        X[i, c1:a1] = torch.tensor(hist) * 0.01
        if n_vals > 1:
            assert len(noisy_exp_vals[i]) == n_vals
        elif n_vals == 1:
            assert isinstance(noisy_exp_vals[i], float)
        X[i, a1:v1] = torch.tensor(noisy_exp_vals[i])
    if meas_bases != [[]]:
        assert len(meas_bases) == len(circuits)
        if len(circuits):
            X.numpy()[:, v1:] = np.asarray(meas_bases, dtype=np.float32)


def encode_data(circuits, properties, ideal_exp_vals, noisy_exp_vals, num_qubits, meas_bases=None, native=False):
    """Rows ``[8 | len(gates_set) | 40 | num_qubits | len(basis)]`` (58 wide for FakeLima, 4 observables)."""
    noisy_exp_vals = _unwrap_single(noisy_exp_vals)
    gates_set = sorted(properties["gates_set"])
    if meas_bases is None:
        meas_bases = [[]]
    vec = backend_summary_vector(properties)
    bin_size = 0.1 * np.pi
    n_bins = int(np.ceil(4 * np.pi / bin_size))
    X = torch.from_numpy(np.zeros((len(circuits), len(vec) + len(gates_set) + n_bins + num_qubits + len(meas_bases[0])), dtype=np.float32))
    X.numpy()[:, : len(vec)] = np.asarray(vec, dtype=np.float32)[None, :]
    _fill_rows(X, circuits, gates_set, len(vec), bin_size, n_bins, num_qubits, noisy_exp_vals, meas_bases, native)
    return X, torch.tensor(ideal_exp_vals, dtype=torch.float32)


def encode_data_v2_ecr(circuits, ideal_exp_vals, noisy_exp_vals, obs_size, meas_bases=None, two_q_gate="ecr",
                       native=False):
    """Rows ``[5 gate counts | 160 angle bins | obs_size | len(basis)]`` (the demo feature set)."""
    noisy_exp_vals = _unwrap_single(noisy_exp_vals)
    if meas_bases is None:
        meas_bases = [[]]
    gates_set = [two_q_gate, "sx", "x", "id", "rz"]
    bin_size = 0.025 * np.pi
    n_bins = int(np.ceil(4 * np.pi / bin_size))
    X = torch.from_numpy(np.zeros((len(circuits), len(gates_set) + n_bins + obs_size + len(meas_bases[0])), dtype=np.float32))
    _fill_rows(X, circuits, gates_set, 0, bin_size, n_bins, obs_size, noisy_exp_vals, meas_bases, native)
    return X, torch.tensor(ideal_exp_vals, dtype=torch.float32)
