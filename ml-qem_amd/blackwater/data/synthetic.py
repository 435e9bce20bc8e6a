"""Synthetic TFIM-Trotter circuit corpora for benchmarking (no qiskit, no simulator).

The circuit STRUCTURE is the reference's Trotterised transverse-field Ising layer
(docs/tutorials/h13_ising_data_gen.ipynb:247; 100-qubit variant h24_ising_data_gen_zne_hardware_100q.ipynb cell [4]):
``rx(2 h dt)`` on every qubit | barrier | ``cx-rz(-2 J dt)-cx`` on even bonds | barrier | same on odd bonds | barrier,
repeated ``steps`` times, then ``measure_all``.  It is lowered with one fixed rule per gate to the basis
{rz, sx, x, ecr|cx} so that op counts land near what the reference's transpiled circuits hold (100 qubits:
~2.0k ops and 198 two-qubit gates per step; measured there: 1,584 / 8,999 / 20,711 ops at 1 / 6 / 10 steps).
Labels are synthetic: they only feed the loss and the MAE plumbing.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np

from .backends import StaticBackend
from .circuit import Circuit, CircuitOp
from .utils import circuit_to_graph_data_json, get_backend_properties_v1


def tfim_circuit(nq: int, steps: int, J: float, h: float = 0.66 * math.pi, dt: float = 0.5,
                 two_q: str = "ecr") -> Circuit:
    ops: List[CircuitOp] = []
    theta, phi = 2 * h * dt, -2 * J * dt
    all_q = tuple(range(nq))

    def rx(q):
        ops.extend([CircuitOp("rz", (q,), (), (math.pi / 2,)), CircuitOp("sx", (q,)),
                    CircuitOp("rz", (q,), (), (theta + math.pi,)), CircuitOp("sx", (q,)),
                    CircuitOp("rz", (q,), (), (5 * math.pi / 2,))])

    def cx(a, b):
        if two_q == "cx":
            ops.append(CircuitOp("cx", (a, b)))
            return
        ops.extend([CircuitOp("rz", (a,), (), (-math.pi / 2,)), CircuitOp("rz", (b,), (), (-math.pi,)),
                    CircuitOp("sx", (b,)), CircuitOp("rz", (b,), (), (-math.pi,)), CircuitOp(two_q, (a, b)),
                    CircuitOp("x", (a,)), CircuitOp("sx", (b,))])

    def bonds(first_qubits):
        for q0 in first_qubits:
            cx(q0, q0 + 1)
        for q0 in first_qubits:
            ops.append(CircuitOp("rz", (q0 + 1,), (), (phi,)))
        for q0 in first_qubits:
            cx(q0, q0 + 1)

    for _ in range(steps):
        for q in all_q:
            rx(q)
        ops.append(CircuitOp("barrier", all_q))
        bonds(range(0, nq - 1, 2))
        ops.append(CircuitOp("barrier", all_q))
        bonds(range(1, nq - 2, 2))
        ops.append(CircuitOp("barrier", all_q))
    ops.append(CircuitOp("barrier", all_q))
    ops.extend(CircuitOp("measure", (q,), (q,)) for q in all_q)
    return Circuit(nq, nq, ops)


def synthetic_backend(nq: int, two_q: str = "ecr", seed: int = 0) -> StaticBackend:
    """FakeLima-like calibration table stretched to ``nq`` qubits on a line (seeded)."""
    rng = np.random.default_rng(seed)
    nduv = lambda name, unit, value: {"name": name, "unit": unit, "value": float(value)}
    qubits = [[nduv("T1", "us", rng.uniform(20, 120)), nduv("T2", "us", rng.uniform(20, 120)),
               nduv("readout_error", "", rng.uniform(0.01, 0.06))] for _ in range(nq)]
    log_u = lambda: math.exp(rng.uniform(math.log(1e-4), math.log(2e-2)))
    gates = []
    for q in range(nq):
        for g, length in (("id", 35.5), ("rz", 0.0), ("sx", 35.5), ("x", 35.5)):
            gates.append({"gate": g, "qubits": [q], "name": f"{g}{q}",
                          "parameters": [nduv("gate_error", "", 0.0 if g == "rz" else log_u()),
                                         nduv("gate_length", "ns", length)]})
        gates.append({"gate": "reset", "qubits": [q], "name": f"reset{q}",
                      "parameters": [nduv("gate_length", "ns", 5351.1)]})
    for q in range(nq - 1):
        for a, b in ((q, q + 1), (q + 1, q)):
            gates.append({"gate": two_q, "qubits": [a, b], "name": f"{two_q}{a}_{b}",
                          "parameters": [nduv("gate_error", "", log_u()), nduv("gate_length", "ns", 660.0)]})
    return StaticBackend(f"synthetic_{nq}q", {"backend_name": f"synthetic_{nq}q", "qubits": qubits, "gates": gates})


def tfim_corpus(nq: int, steps_list, n_J: int, seed: int = 42, two_q: str = "ecr", exp_value_size: int = 1,
                add_self_loops: bool = True) -> Dict[str, list]:
    """Encoded graphs for every (steps, J) pair: ``steps_list`` x ``n_J`` circuits, J ~ U(0, 0.66 pi) with
    ``np.random.seed(seed)`` as in the reference's ``get_Js`` (h24 notebook cell [7]).

    The graph of a TFIM circuit depends on J only through the rz angles of the bond rotations (feature
    column 0), so each step count is encoded once with the real encoder and the J-dependent column is rewritten.
    """
    props = get_backend_properties_v1(synthetic_backend(nq, two_q))
    rs = np.random.RandomState(seed)
    Js = rs.uniform(0, 0.66 * math.pi, size=n_J)
    label_rng = np.random.default_rng(seed + 1)
    xs, eis, ys, noisy, depth, obs = [], [], [], [], [], []
    for steps in steps_list:
        circ = tfim_circuit(nq, steps, J=1.0, two_q=two_q)  # phi = -2*J*dt = -1.0: marks the J-dependent nodes
        graph = circuit_to_graph_data_json(circ, props, use_gate_features=True, use_qubit_features=True)
        x0 = np.asarray(graph["nodes"]["DAGOpNode"], dtype=np.float32)
        ei = np.asarray(graph["edges"]["DAGOpNode_wire_DAGOpNode"]["edge_index"], dtype=np.int64)
        if add_self_loops:  # the training path's dataset transform (loaders/exp_val.py:33)
            loops = np.arange(x0.shape[0], dtype=np.int64)
            ei = np.concatenate([ei, np.stack([loops, loops])], axis=1)
        bond_nodes = np.array([k for k, op in enumerate(circ.ops) if op.name == "rz" and op.params[0] == -1.0],
                              dtype=np.int64)
        n2q = sum(1 for op in circ.ops if op.name == two_q)
        d = circ.depth()
        for J in Js:
            x = x0.copy()
            x[bond_nodes, 0] = np.float32(-2 * J * 0.5)
            ideal = label_rng.uniform(-1, 1, size=exp_value_size)
            xs.append(x)
            eis.append(ei)
            ys.append(ideal)
            noisy.append(ideal * math.exp(-2e-4 * n2q) + label_rng.normal(0, 0.01, size=exp_value_size))
            depth.append([float(d)])
            o = np.zeros((1, 4 * nq + 1), dtype=np.float32)
            o[0, 0] = 1.0
            o[0, 1::4] = 1.0                                   # identity everywhere ...
            qz = int(label_rng.integers(0, nq))
            o[0, 1 + 4 * qz], o[0, 2 + 4 * qz] = 0.0, 1.0      # ... except one Z
            obs.append(o)
    return {"x": xs, "edge_index": eis, "y": np.asarray(ys, np.float32), "noisy": np.asarray(noisy, np.float32),
            "depth": np.asarray(depth, np.float32), "observable": np.asarray(obs, np.float32)}
