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


def random_circuit(nq: int, depth: int, seed: int, two_q: str = "cx", measure: bool = True) -> Circuit:
    """Layers of random disjoint one- and two-qubit gates, like the reference's use of
    ``qiskit.circuit.random.random_circuit`` (blackwater/data/generators/exp_val.py:116-120), already in the backend
    basis {rz, sx, x, two_q}; two-qubit gates act on neighbours of a line so that calibration entries exist."""
    rng = np.random.default_rng(seed)
    ops: List[CircuitOp] = []
    for _ in range(depth):
        q = 0
        while q < nq:
            if q + 1 < nq and rng.random() < 0.35:
                a, b = (q, q + 1) if rng.random() < 0.5 else (q + 1, q)
                ops.append(CircuitOp(two_q, (a, b)))
                q += 2
                continue
            kind = rng.integers(0, 4)
            if kind == 0:
                ops.append(CircuitOp("rz", (q,), (), (float(rng.uniform(-math.pi, math.pi)),)))
            elif kind == 1:
                ops.append(CircuitOp("sx", (q,)))
            elif kind == 2:
                ops.append(CircuitOp("x", (q,)))
            q += 1  # kind 3: idle wire in this layer
    if measure:
        ops.append(CircuitOp("barrier", tuple(range(nq))))
        ops.extend(CircuitOp("measure", (q,), (q,)) for q in range(nq))
    return Circuit(nq, nq if measure else 0, ops)


def pauli_twirl(circ: Circuit, seed: int, two_q=("cx", "ecr")) -> Circuit:
    """Random single-qubit Paulis (x, or rz(pi) for Z, or both for Y) before and after every two-qubit gate -- the
    structural effect of Pauli twirling on the circuit graph (cfg5 of BASELINE.json)."""
    rng = np.random.default_rng(seed)

    def pauli(q):
        k = rng.integers(0, 4)
        out = []
        if k in (1, 2):
            out.append(CircuitOp("x", (q,)))
        if k in (2, 3):
            out.append(CircuitOp("rz", (q,), (), (math.pi,)))
        return out

    ops: List[CircuitOp] = []
    for op in circ.ops:
        if op.name in two_q:
            for q in op.qubits:
                ops += pauli(q)
            ops.append(op)
            for q in op.qubits:
                ops += pauli(q)
        else:
            ops.append(op)
    return Circuit(circ.num_qubits, circ.num_clbits, ops)


def encode_corpus(circuits, nq: int, two_q: str = "cx", seed: int = 7, exp_value_size: int = 1,
                  add_self_loops: bool = True) -> Dict[str, list]:
    """Arena-ready arrays (same keys as :func:`tfim_corpus`) for arbitrary circuits, encoded by the native encoder."""
    from .circuit import circuit_to_qasm
    from .native_encoder import NativeEncoder

    props = get_backend_properties_v1(synthetic_backend(nq, two_q))
    enc = NativeEncoder(props)
    rng = np.random.default_rng(seed)
    xs, eis, ys, noisy, depth, obs = [], [], [], [], [], []
    for circ in circuits:
        x, ei, _, d = enc.encode(circuit_to_qasm(circ))
        if add_self_loops:
            loops = np.arange(x.shape[0], dtype=np.int64)
            ei = np.concatenate([ei, np.stack([loops, loops])], axis=1)
        ideal = rng.uniform(-1, 1, size=exp_value_size)
        n2q = sum(1 for op in circ.ops if op.name == two_q)
        xs.append(x.astype(np.float32))
        eis.append(ei)
        ys.append(ideal)
        noisy.append(ideal * math.exp(-5e-3 * n2q) + rng.normal(0, 0.01, size=exp_value_size))
        depth.append([float(d)])
        o = np.zeros((1, 4 * nq + 1), dtype=np.float32)
        o[0, 0], o[0, 1::4] = 1.0, 1.0
        qz = int(rng.integers(0, nq))
        o[0, 1 + 4 * qz], o[0, 2 + 4 * qz] = 0.0, 1.0
        obs.append(o)
    return {"x": xs, "edge_index": eis, "y": np.asarray(ys, np.float32), "noisy": np.asarray(noisy, np.float32),
            "depth": np.asarray(depth, np.float32), "observable": np.asarray(obs, np.float32)}


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


class TfimCorpus:
    """The (steps x J) grid of encoded TFIM-Trotter graphs, WITHOUT holding one feature matrix per circuit.

    The graph of a TFIM circuit depends on J only through the rz angle of the bond rotations (feature column 0), so each
    step count is encoded once with the real encoder ("template") and a circuit is (template, J).  Graph ids run
    step-major: ``g = steps_index * n_J + j``; J ~ U(0, 0.66 pi) from ``np.random.RandomState(seed)`` as in the reference's
    ``get_Js`` (h24 notebook cell [7]).  ``host_graphs(ids)`` materialises plain numpy graphs (what the CPU oracle and
    the host-side loaders consume); ``arena(device, ids)`` replicates the templates ON THE DEVICE straight into a
    :class:`GraphArena` -- an 8 200-circuit / 90 M-node corpus costs ten template uploads instead of 9 GB of host
    arrays."""

    def __init__(self, nq: int, steps_list, n_J: int, seed: int = 42, two_q: str = "ecr", exp_value_size: int = 1,
                 add_self_loops: bool = True):
        props = get_backend_properties_v1(synthetic_backend(nq, two_q))
        self.nq, self.n_J, self.steps_list = nq, int(n_J), list(steps_list)
        rs = np.random.RandomState(seed)
        self.Js = rs.uniform(0, 0.66 * math.pi, size=n_J)
        label_rng = np.random.default_rng(seed + 1)
        self.templates = []
        ys, noisy, depth, obs_z = [], [], [], []
        for steps in self.steps_list:
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
            self.templates.append({"x": x0, "edge_index": ei, "bond_nodes": bond_nodes, "depth": float(d)})
            for _ in range(self.n_J):   # same draw order as ever: ideal, noise, observable qubit -- per circuit
                ideal = label_rng.uniform(-1, 1, size=exp_value_size)
                ys.append(ideal)
                noisy.append(ideal * math.exp(-2e-4 * n2q) + label_rng.normal(0, 0.01, size=exp_value_size))
                depth.append([float(d)])
                obs_z.append(int(label_rng.integers(0, nq)))
        self.y, self.noisy = np.asarray(ys, np.float32), np.asarray(noisy, np.float32)
        self.depth, self.obs_z = np.asarray(depth, np.float32), np.asarray(obs_z, np.int64)
        self.node_counts = np.repeat([t["x"].shape[0] for t in self.templates], self.n_J).astype(np.int64)

    def __len__(self):
        return len(self.templates) * self.n_J

    def observables(self, ids) -> np.ndarray:
        """[len(ids), 1, 4 nq + 1]: coefficient 1, identity everywhere except one Z."""
        ids = np.asarray(ids, dtype=np.int64)
        o = np.zeros((len(ids), 1, 4 * self.nq + 1), dtype=np.float32)
        o[:, 0, 0] = 1.0
        o[:, 0, 1::4] = 1.0
        rows = np.arange(len(ids))
        o[rows, 0, 1 + 4 * self.obs_z[ids]] = 0.0
        o[rows, 0, 2 + 4 * self.obs_z[ids]] = 1.0
        return o

    def graph_x(self, g: int) -> np.ndarray:
        t = self.templates[g // self.n_J]
        x = t["x"].copy()
        x[t["bond_nodes"], 0] = np.float32(-2 * self.Js[g % self.n_J] * 0.5)
        return x

    def host_graphs(self, ids=None) -> Dict[str, list]:
        ids = np.arange(len(self)) if ids is None else np.asarray(ids, dtype=np.int64)
        return {"x": [self.graph_x(int(g)) for g in ids],
                "edge_index": [self.templates[int(g) // self.n_J]["edge_index"] for g in ids],
                "y": self.y[ids], "noisy": self.noisy[ids], "depth": self.depth[ids], "observable": self.observables(ids)}

    def arena(self, device, ids=None, filler_nodes: int = 0):
        """The graphs ``ids`` (ascending global ids; default all) as a device-resident arena, built by replicating the
        templates on the device.  Under data parallelism every rank passes its own shard of ids."""
        import torch

        from .arena import GraphArena

        ids = np.arange(len(self)) if ids is None else np.asarray(ids, dtype=np.int64)
        if len(ids) > 1 and (np.diff(ids) <= 0).any():
            raise ValueError("TfimCorpus.arena: ids must be strictly ascending")
        dev = torch.device(device)
        f = self.templates[0]["x"].shape[1]
        f4 = (f + 3) // 4 * 4
        tmpl_of = ids // self.n_J
        n_total = int(self.node_counts[ids].sum())
        x = torch.zeros((max(n_total, 1), f4), dtype=torch.float32, device=dev)
        ei_parts, base = [], 0
        for t_idx, t in enumerate(self.templates):
            mine = ids[tmpl_of == t_idx]
            c = len(mine)
            if c == 0:
                continue
            n_t = t["x"].shape[0]
            x0 = torch.from_numpy(t["x"]).to(dev)
            block = x[base:base + c * n_t].view(c, n_t, f4)
            block[:, :, :f] = x0.unsqueeze(0)
            bond = torch.from_numpy(t["bond_nodes"]).to(dev)
            jv = torch.from_numpy((-2 * self.Js[mine % self.n_J] * 0.5).astype(np.float32)).to(dev)
            block[:, bond, 0] = jv.unsqueeze(1)
            ei_t = torch.from_numpy(t["edge_index"]).to(dev)
            offs = base + torch.arange(c, device=dev, dtype=torch.int64) * n_t
            ei_parts.append((ei_t.unsqueeze(1) + offs.view(1, c, 1)).reshape(2, -1))
            base += c * n_t
        ei = torch.cat(ei_parts, dim=1) if ei_parts else torch.zeros((2, 0), dtype=torch.int64, device=dev)
        return GraphArena.from_device(x[:n_total, :f], self.node_counts[ids], ei, self.y[ids], self.noisy[ids],
                                      self.depth[ids], self.observables(ids), filler_nodes=filler_nodes)


def tfim_corpus(nq: int, steps_list, n_J: int, seed: int = 42, two_q: str = "ecr", exp_value_size: int = 1,
                add_self_loops: bool = True) -> Dict[str, list]:
    """Encoded graphs for every (steps, J) pair as plain host arrays: ``steps_list`` x ``n_J`` circuits
    (see :class:`TfimCorpus`)."""
    return TfimCorpus(nq, steps_list, n_J, seed, two_q, exp_value_size, add_self_loops).host_graphs()
