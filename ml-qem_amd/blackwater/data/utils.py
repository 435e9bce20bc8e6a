"""Circuit -> graph encoders and backend/observable flatteners (host side of the hot path).

Mirrors the call surface of the reference's blackwater/data/utils.py for the functions on the path:
``get_backend_properties_v1`` (:139-175), ``circuit_to_graph_data_json`` (:198-389),
``encode_pauli_sum_op`` (:447-474) and the legacy ``circuit_to_pyg_data`` (:52-123).  The Aer-backed label
helpers of that file need a quantum simulator and are out of scope.

The reference converts the circuit to a qiskit DAG and lists ``dag.edges()``.  No DAG library is used
here: because a DAG built by appending ops has, per qubit wire, one chain in -> op -> ... -> op -> out,
the node list and the edge list (in rustworkx's listing order: nodes by index, each node's out-edges most
recently inserted first) are reproduced by replaying the appends on per-node adjacency lists.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

from .backends import pauli_terms
from .circuit import Circuit

ADDITIONAL_GATE_TYPES = ["barrier", "measure"]

# default gate vocabulary of the legacy homogeneous encoder (26 names, utils.py:19-49)
available_gate_names = (
    "id u1 u2 u3 x y z h s sdg t tdg rx ry rz "
    "cx cy cz ch crz cu1 cu3 swap rzz "
    "ccx cswap"
).split()


def gate_to_index(gate: Any) -> str:
    """'cx' on qubits [0, 1] -> 'cx_0_1' (utils.py:126-136)."""
    return "_".join([str(gate.gate)] + [str(q) for q in gate.qubits])


def get_backend_properties_v1(backend: Any, gates_order: Optional[Sequence[str]] = None) -> Dict[str, Any]:
    """Flattens a V1 backend's calibration data (reference: utils.py:139-175).

    ``gates_set`` is sorted here; the reference builds it from a Python ``set`` so its order changes with
    PYTHONHASHSEED (SURVEY.md section 5 "Determinism").  Pass ``gates_order`` to reproduce the column order a
    given checkpoint was trained with.
    """
    props = backend.properties()
    n = len(props.qubits)

    qubits_props = {}
    for q in range(n):
        qp = props.qubit_property(q)
        qubits_props[q] = {
            "index": q,
            "t1": qp.get("T1", (0, 0))[0],
            "t2": qp.get("T2", (0, 0))[0],
            "readout_error": qp.get("readout_error", (0, 0))[0],
        }

    gate_props = {}
    for gate in props.gates:
        key = gate_to_index(gate)
        rec = {"index": key, "gate_error": 0.0, "gate_length": 0.0}
        rec.update({p.name: p.value for p in gate.parameters})
        gate_props[key] = rec

    names = {g.gate for g in props.gates}
    if gates_order is not None:
        if set(gates_order) != names:
            raise ValueError(f"gates_order {list(gates_order)} does not match backend gates {sorted(names)}")
        gates_set = list(gates_order)
    else:
        gates_set = sorted(names)

    name = backend.name() if callable(getattr(backend, "name", None)) else getattr(backend, "name", "backend")
    return {"name": name, "gates_set": gates_set, "num_qubits": n, "qubits_props": qubits_props,
            "gate_props": gate_props}


def _qubit_props(properties: Dict[str, Any], index: int) -> Dict[str, Any]:
    table = properties["qubits_props"]
    return table[index] if index in table else table[str(index)]  # str keys after a JSON round trip


def circuit_to_graph_data_json(
    circuit: Any,
    properties: Dict[str, Any],
    use_gate_features: bool = False,
    use_qubit_features: bool = False,
) -> Dict[str, Dict[str, Any]]:
    """Encodes a circuit as the reference's heterogeneous graph dict (reference: utils.py:198-389).

    Node per instruction in program order with feature
    ``[p0,p1,p2] | one-hot(gates_set+['barrier','measure']) | [t1 x3, t2 x3, readout x3] | [gate_error, gate_length]``
    (the last two groups optional), plus one in/out node per wire with feature ``[0, 0]``.  Edges follow
    qubit wires only, bucketed by endpoint types, each with ``edge_attr = [t1, t2, readout_error]`` of the wire.
    """
    circ = Circuit.from_any(circuit)
    type_slot = {g: i for i, g in enumerate(list(properties["gates_set"]) + ADDITIONAL_GATE_TYPES)}
    n_types = len(type_slot)
    nq, nc = circ.num_qubits, circ.num_clbits

    # ---- node features ---------------------------------------------------------------------------
    op_features: List[List[float]] = []
    for op in circ.ops:
        if op.name != "barrier" and len(op.qubits) > 3:
            raise Exception("Non barrier gate that has more than 3 qubits."
                            "Those tyoe of gates are not supported yet.")
        regidx = [circ.qubit_reg_index[q] for q in op.qubits]

        t1, t2, ro = [0.0] * 3, [0.0] * 3, [0.0] * 3
        if op.name != "barrier":  # barriers may span any number of wires: zero qubit features
            for slot, qi in enumerate(regidx):
                qp = _qubit_props(properties, qi)
                t1[slot], t2[slot], ro[slot] = qp.get("t1", 0.0), qp.get("t2", 0.0), qp.get("readout_error", 0.0)

        onehot = [0.0] * n_types
        onehot[type_slot[op.name]] = 1.0  # KeyError for a gate outside the backend basis, as in the reference

        pvec = [0.0, 0.0, 0.0]
        for k, p in enumerate(op.params):
            pvec[k] = float(p)  # IndexError beyond 3 parameters, as in the reference

        feat = pvec + onehot
        if use_qubit_features:
            feat += t1 + t2 + ro
        if use_gate_features:
            gp = properties["gate_props"].get("_".join([op.name] + [str(i) for i in regidx]), {})
            feat += [gp.get("gate_error", 0.0), gp.get("gate_length", 0.0)]
        op_features.append(feat)

    # ---- replay the DAG appends -------------------------------------------------------------------
    # node ids: wire w (qubits first, then clbits) -> in-node 2w, out-node 2w+1; op k -> 2*(nq+nc)+k
    n_wires = nq + nc
    op_base = 2 * n_wires
    out_adj: List[List[Tuple[int, int]]] = [[] for _ in range(op_base + len(circ.ops))]  # (dst, wire)
    last = [2 * w for w in range(n_wires)]
    for w in range(n_wires):
        out_adj[2 * w].append((2 * w + 1, w))
    for k, op in enumerate(circ.ops):
        node = op_base + k
        for w in list(op.qubits) + [nq + c for c in op.clbits]:
            pred, sink = last[w], 2 * w + 1
            out_adj[pred].remove((sink, w))
            out_adj[pred].append((node, w))
            out_adj[node].append((sink, w))
            last[w] = node

    def describe(node_id: int) -> Tuple[str, int]:
        if node_id >= op_base:
            return "DAGOpNode", node_id - op_base
        return ("DAGOutNode", node_id // 2) if node_id & 1 else ("DAGInNode", node_id // 2)

    edges: Dict[str, Dict[str, List]] = {}
    for src_id in range(len(out_adj)):
        s_type, s_idx = describe(src_id)
        for dst_id, w in reversed(out_adj[src_id]):  # most recently inserted edge first
            if w >= nq:
                continue  # classical wires carry no edge
            d_type, d_idx = describe(dst_id)
            qp = _qubit_props(properties, circ.qubit_reg_index[w])
            bucket = edges.setdefault(f"{s_type}_wire_{d_type}", {"src": [], "dst": [], "edge_attr": []})
            bucket["src"].append(s_idx)
            bucket["dst"].append(d_idx)
            bucket["edge_attr"].append([qp["t1"], qp["t2"], qp["readout_error"]])

    return {
        "nodes": {
            "DAGOpNode": op_features,
            "DAGInNode": [[0, 0] for _ in range(n_wires)],
            "DAGOutNode": [[0, 0] for _ in range(n_wires)],
        },
        "edges": {
            key: {"edge_index": [b["src"], b["dst"]], "edge_attr": b["edge_attr"]} for key, b in edges.items()
        },
    }


_PAULI_SLOT = {"I": 0, "Z": 1, "Y": 2, "X": 3}


def encode_pauli_sum_op(op: Any) -> List[List[float]]:
    """One row per term: ``[Re(coeff)] | 4-way one-hot per Pauli character`` with I,Z,Y,X -> slots 0..3,
    characters in label order (reference: utils.py:447-474)."""
    rows = []
    for label, coeff in pauli_terms(op):
        row: List[float] = [complex(coeff).real]
        for ch in label:
            cell = [0, 0, 0, 0]
            if ch in _PAULI_SLOT:
                cell[_PAULI_SLOT[ch]] = 1
            row += cell
        rows.append(row)
    return rows


def circuit_to_pyg_data(circuit: Any, gate_set: Optional[List[str]] = None):
    """Legacy homogeneous encoder (reference: utils.py:52-123): feature =
    one-hot(gate_set + barrier/measure/delay) | qubit incidence | 3 params; op->op edges only.

    Unlike the reference this does not append to the caller's (or the module's) gate list, so the feature
    width stays 26+3+n+3 on every call (SURVEY.md appendix A.1 documents the reference's growth bug).
    """
    import numpy as np
    import torch

    from .graph import Data

    circ = Circuit.from_any(circuit)
    vocab = list(gate_set or available_gate_names) + ["barrier", "measure", "delay"]
    feats = []
    for op in circ.ops:
        enc = [0.0] * len(vocab)
        enc[vocab.index(op.name)] = 1.0
        touched = [0.0] * circ.num_qubits
        for q in op.qubits:
            touched[circ.qubit_reg_index[q]] = 1.0
        pv = [0.0, 0.0, 0.0]
        for i, p in enumerate(op.params):
            pv[i] = p
        feats.append(enc + touched + pv)

    # the legacy encoder keeps every op->op DAG edge, classical wires included
    pairs = _op_to_op_edges_all_wires(circ)
    return Data(
        x=torch.tensor(feats, dtype=torch.float),
        edge_index=torch.tensor(np.transpose(pairs).reshape(2, -1), dtype=torch.long),
        edge_attr=torch.zeros((1, len(pairs)), dtype=torch.float),
        circuit_depth=torch.tensor([[circ.depth()]], dtype=torch.long),
    )


def _op_to_op_edges_all_wires(circ: Circuit) -> List[List[int]]:
    n_wires = circ.num_qubits + circ.num_clbits
    last: List[Optional[int]] = [None] * n_wires
    out_adj: List[List[int]] = [[] for _ in circ.ops]
    for k, op in enumerate(circ.ops):
        for w in list(op.qubits) + [circ.num_qubits + c for c in op.clbits]:
            if last[w] is not None:
                out_adj[last[w]].append(k)
            last[w] = k
    return [[s, d] for s in range(len(circ.ops)) for d in reversed(out_adj[s])]
