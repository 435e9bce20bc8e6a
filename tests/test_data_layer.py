"""Host-side data layer against the reference's own known answers."""
import json
import os

import numpy as np
import pytest
import torch

from blackwater.data.backends import PauliObservable
from blackwater.data.circuit import Circuit, CircuitOp
from blackwater.data.generators.exp_val import ExpValueEntry
from blackwater.data.graph import AddSelfLoops, Batch, Data, DataLoader
from blackwater.data.loaders.exp_val import CircuitGraphExpValMitigationDataset
from blackwater.data.utils import circuit_to_graph_data_json, circuit_to_pyg_data, encode_pauli_sum_op


def test_legacy_encoder_known_answer():
    # reference tests/data/test_utils.py:13-25: h(0); cx(0,1); measure_all() -> x (5,34), edge_index (2,5)
    c = Circuit(2, 2, [CircuitOp("h", (0,)), CircuitOp("cx", (0, 1)), CircuitOp("barrier", (0, 1)),
                       CircuitOp("measure", (0,), (0,)), CircuitOp("measure", (1,), (1,))])
    d = circuit_to_pyg_data(c)
    assert tuple(d.x.shape) == (5, 34) and tuple(d.edge_index.shape) == (2, 5)
    assert circuit_to_pyg_data(c).x.shape[1] == 34  # stays 34 on a second call (the reference grows by 3)
    assert d.edge_index.t().tolist() == [[0, 1], [1, 2], [1, 2], [2, 4], [2, 3]]
    assert d.circuit_depth.item() == 3


def test_encode_pauli_sum_op():
    rows = encode_pauli_sum_op(PauliObservable([("IZYX", 0.5), ("XXII", -1 + 0j)]))
    assert rows[0] == [0.5, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    assert rows[1][:9] == [-1.0, 0, 0, 0, 1, 0, 0, 0, 1]
    assert encode_pauli_sum_op("ZZ") == [[1.0, 0, 1, 0, 0, 0, 1, 0, 0]]


def test_entry_to_pyg_shapes(golden_dir):
    e = json.load(open(os.path.join(golden_dir, "encoder_goldens.json")))[20]
    entry = ExpValueEntry.from_json({k: v for k, v in e.items() if k not in ("circuit", "source_file")})
    d = entry.to_pyg_data()
    n, m = len(e["circuit_graph"]["nodes"]["DAGOpNode"]), len(e["circuit_graph"]["edges"]["DAGOpNode_wire_DAGOpNode"]["edge_attr"])
    assert tuple(d.x.shape) == (n, 22) and d.x.dtype == torch.float32
    assert tuple(d.edge_index.shape) == (2, m) and d.edge_index.dtype == torch.int64
    assert tuple(d.edge_attr.shape) == (m, 3)
    assert tuple(d.y.shape) == (1, 1, 4) and tuple(d.noisy_0.shape) == (1, 1, 4)
    assert tuple(d.circuit_depth.shape) == (1, 1) and tuple(d.observable.shape) == (1, 0)
    assert d.batch is None
    assert entry.to_dict()["circuit_depth"] == e["circuit_depth"]


def test_dataset_roundtrip_and_silent_drop(tmp_path, golden_dir):
    entries = json.load(open(os.path.join(golden_dir, "encoder_goldens.json")))[:6]
    for e in entries:
        e.pop("source_file")
    broken = json.loads(json.dumps(entries[0]))
    del broken["circuit_graph"]["edges"]["DAGOpNode_wire_DAGOpNode"]  # graph without op->op wires -> KeyError -> dropped
    path = tmp_path / "set.json"
    path.write_text(json.dumps(entries + [broken]))
    ds = CircuitGraphExpValMitigationDataset(str(path))
    assert ds.len() == 6 and len(ds) == 6
    g = ds.get(0)
    n = g.x.shape[0]
    raw = len(entries[0]["circuit_graph"]["edges"]["DAGOpNode_wire_DAGOpNode"]["edge_attr"])
    assert g.edge_index.shape[1] == raw + n  # default transform = AddSelfLoops
    assert g.edge_attr.shape[0] == raw       # ... which leaves edge_attr alone (01_ngem.ipynb:186)
    assert torch.equal(g.edge_index[:, -n:], torch.arange(n).repeat(2, 1))
    assert CircuitGraphExpValMitigationDataset([str(path)], num_samples=2).len() == 2
    assert CircuitGraphExpValMitigationDataset(str(path), transforms=[lambda d: d]).get(0).edge_index.shape[1] == raw


def test_batch_collate_rules():
    a = Data(x=torch.ones(3, 2), edge_index=torch.tensor([[0, 1], [1, 2]]), y=torch.zeros(1, 1, 4),
             noisy_0=torch.ones(1, 1, 4), circuit_depth=torch.tensor([[5.0]]))
    b = Data(x=torch.zeros(2, 2), edge_index=torch.tensor([[0], [1]]), y=torch.ones(1, 1, 4),
             noisy_0=torch.zeros(1, 1, 4), circuit_depth=torch.tensor([[7.0]]))
    batch = Batch.from_data_list([a, b])
    assert batch.edge_index.tolist() == [[0, 1, 3], [1, 2, 4]]
    assert batch.batch.tolist() == [0, 0, 0, 1, 1] and batch.ptr.tolist() == [0, 3, 5]
    assert tuple(batch.y.shape) == (2, 1, 4) and tuple(batch.circuit_depth.shape) == (2, 1)
    assert batch.num_graphs == 2 and "DataBatch(" in repr(batch)
    loader = DataLoader([a, b, a], batch_size=2, shuffle=False)
    sizes = [bt.num_graphs for bt in loader]
    assert sizes == [2, 1]


def test_feature_encoders_match_oracle_and_shapes(g1, lima_props):
    from blackwater.library.learning.features import encode_data, encode_data_v2_ecr
    from oracle.features import encode_data_rows

    circs = [Circuit.from_qasm_str(t) for t in g1["qasm"][:40]]
    X, y = encode_data(circs, lima_props, g1["ideal"][:40].tolist(), g1["noisy"][:40].tolist(), 4)
    ops = [[(o.name, len(o.qubits), o.params[0] if o.params else None) for o in c.ops] for c in circs]
    assert torch.equal(X, encode_data_rows(ops, lima_props, g1["noisy"][:40].tolist(), 4))
    assert tuple(y.shape) == (40, 4)
    X1, _ = encode_data([circs[0]], lima_props, [[0.0]], [[0.25]], 1, meas_bases=encode_pauli_sum_op("IIIIZ"))
    assert X1.shape[1] == 8 + 6 + 40 + 1 + 21  # the TorchLearningModelProcessor call shape (learning/estimator.py:174-181)
    X2, _ = encode_data_v2_ecr(circs[:3], [[0.0] * 4] * 3, g1["noisy"][:3].tolist(), 4, two_q_gate="cx")
    assert X2.shape == (3, 5 + 160 + 4)  # 169-d demo2 feature set
    assert X2[0, 0].item() == pytest.approx(0.01 * circs[0].count_ops().get("cx", 0))


def test_stratified_batches_quotas_and_epochs():
    """train.StratifiedBatches (host logic): quotas are the classes' shares of the batch, remainders go to the sizes nearest
    the mean, every batch has the same node total, and a class is walked through completely before any of it repeats."""
    import numpy as np

    from blackwater.train import StratifiedBatches

    nodes = np.repeat(np.arange(1, 11) * 2000 + 89, 820)
    edges = nodes * 5 // 4
    sb = StratifiedBatches(nodes, edges, 1024, seed=1)
    assert sb.quota.sum() == 1024 and sorted(set(sb.quota.tolist())) == [102, 103]
    assert sb.quota[3:7].tolist() == [103, 103, 103, 103]            # the four spare places: the sizes nearest the mean
    assert sb.nodes_per_batch == int(1024 * nodes.mean())
    seen = []
    for _ in range(8):                                               # 8 x 102 = 816 <= 820: no class wraps yet
        ids = sb.draw()
        assert len(ids) == 1024 and nodes[ids].sum() == sb.nodes_per_batch
        seen.append(ids)
    first_class = np.concatenate([ids[nodes[ids] == nodes[0]] for ids in seen])
    assert len(set(first_class.tolist())) == len(first_class)        # no repeats inside an epoch of the class
    with np.testing.assert_raises(ValueError):
        StratifiedBatches(nodes[:50], edges[:50], 1024)


def test_deferred_structure_builds_its_connectivity_on_first_use():
    """native.structure.GraphStructure.deferred: graph boundaries are there at once, the CSR arrays appear when a layer first
    reads one of them (ASAPooling's second coarsening is never read by the reference's models: gnn.py:112-114)."""
    import torch

    from blackwater.native.structure import GraphStructure

    calls = []

    def build():
        calls.append(1)
        z = torch.zeros(4, dtype=torch.int32)
        return (z, z[:0], z, z[:0], z[:3], 0, z[:0])

    s = GraphStructure.deferred(3, torch.tensor([0, 3], dtype=torch.int32), 1, build, graph_sizes=[3])
    assert s.num_nodes == 3 and s.num_graphs == 1 and s.graph_sizes == [3] and not s.connectivity_built and not calls
    assert s.in_ptr.shape[0] == 4 and calls == [1] and s.connectivity_built
    assert s.edge_count() == 0 and s.out_eid.numel() == 0 and calls == [1]      # built once
    try:
        s.no_such_attribute
    except AttributeError:
        pass
    else:
        raise AssertionError("unknown attributes must raise")
