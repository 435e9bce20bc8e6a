"""Device-side expansion of the op stream (csrc/encode_expand.hip behind NativeEncoder.encode_batch_expand)
against the host fill (mlqem_qasm_batch_fill behind NativeEncoder.encode_batch), which tests/test_native_encoder.py and
tests/test_encoder_goldens.py pin to the reference's own input/output pairs: ``x``, ``edge_index`` (the reference's edge ORDER),
``batch``, node counts and depths must be equal BIT FOR BIT -- the encoder is integer / copy work, there is no tolerance."""
import json
import os

import numpy as np
import pytest
import torch

from blackwater.data.circuit import circuit_to_qasm
from blackwater.data.native_encoder import NativeEncoder
from blackwater.data.synthetic import synthetic_backend, tfim_circuit
from blackwater.data.utils import get_backend_properties_v1
from helpers import G1_GATES_ORDER, g1_graph, infer_gates_order

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _same(enc, texts, **kw):
    x, ei, batch, counts, depths = enc.encode_batch(texts, **kw)
    xd, eid, bd, counts_d, depths_d = enc.encode_batch_expand(texts, DEV, **kw)
    torch.cuda.synchronize()
    assert xd.dtype == torch.float32 and eid.dtype == torch.int64 and bd.dtype == torch.int64
    assert tuple(xd.shape) == tuple(x.shape) and tuple(eid.shape) == tuple(ei.shape)
    assert torch.equal(xd.cpu(), x), "x differs"
    assert torch.equal(eid.cpu(), ei), "edge_index differs"
    assert torch.equal(bd.cpu(), batch), "batch differs"
    assert np.array_equal(counts_d, counts) and list(depths_d) == list(depths)
    return x, ei


def test_the_references_circuits_bit_for_bit(g1, golden_dir, lima_props):
    """Every circuit text the repository holds from the reference: the 300 G1 circuits (gnn1's validation set), the 600 ising
    train / validation circuits and the JSON goldens (QASM with its encoded graph) -- as one run() each, all three feature options."""
    goldens = [e["circuit"] for e in json.load(open(os.path.join(golden_dir, "encoder_goldens.json")))]
    ising = json.load(open(os.path.join(golden_dir, "ising_trainval_circuits.json")))
    for props in (lima_props, dict(lima_props, gates_set=G1_GATES_ORDER)):
        enc = NativeEncoder(props)
        for texts in (list(g1["qasm"]), ising, goldens):
            for gate_f, qubit_f in ((True, True), (False, False), (True, False)):
                _same(enc, texts, use_gate_features=gate_f, use_qubit_features=qubit_f)


def test_device_encoder_against_the_references_stored_graphs(g1, golden_dir, lima_props):
    """The pin without a hop through the host fill: what the device expansion writes for the reference's own circuits
    equals the graphs the REFERENCE stored next to them -- the 30 JSON input/output pairs of encoder_goldens.json (the op -> op wire
    edges in the reference's order, exactly; the feature rows exactly but for the three angle columns, which the stored QASM prints
    rounded: 1e-6, as float32) and the 300 G1 graphs (x, edge_index as stored, exactly)."""
    from blackwater.data.circuit import Circuit

    for e in json.load(open(os.path.join(golden_dir, "encoder_goldens.json"))):
        want = e["circuit_graph"]
        rows = np.array(want["nodes"]["DAGOpNode"])
        props = dict(lima_props, gates_set=infer_gates_order(Circuit.from_qasm_str(e["circuit"]), want["nodes"]["DAGOpNode"], lima_props["gates_set"]))
        x, ei, _, counts, depths = NativeEncoder(props).encode_batch_expand([e["circuit"]], DEV)
        torch.cuda.synchronize()
        x = x.cpu().numpy()
        assert x.shape == rows.shape and counts[0] == rows.shape[0] and depths[0] == e["circuit_depth"]
        assert np.array_equal(x[:, 3:], rows[:, 3:].astype(np.float32))
        assert np.abs(x[:, :3] - rows[:, :3]).max() < 1e-6
        assert np.array_equal(ei.cpu().numpy(), np.array(want["edges"]["DAGOpNode_wire_DAGOpNode"]["edge_index"], dtype=np.int64).reshape(2, -1))
    enc = NativeEncoder(dict(lima_props, gates_set=G1_GATES_ORDER))
    x, ei, batch, counts, _ = enc.encode_batch_expand(list(g1["qasm"]), DEV)
    torch.cuda.synchronize()
    x, ei = x.cpu().numpy(), ei.cpu().numpy()
    node0 = np.concatenate([[0], np.cumsum(counts)])
    e0 = 0
    for i in range(len(g1["qasm"])):
        xs, eis, _ = g1_graph(g1, i)
        assert np.array_equal(x[node0[i]:node0[i + 1]], np.asarray(xs, dtype=np.float32))
        assert np.array_equal(ei[:, e0:e0 + eis.shape[1]] - node0[i], eis)
        e0 += eis.shape[1]
    assert e0 == ei.shape[1]


@pytest.mark.parametrize("nq,two_q,count", [(100, "ecr", 48), (20, "cx", 100), (4, "cx", 700)])
def test_synthetic_corpora_bit_for_bit(nq, two_q, count):
    """TFIM-Trotter circuits of the bench's shapes (100 qubits: barriers over 100 wires, 2-20 k ops per circuit): a barrier's
    out-edges -- one per wire, ranked by its wave -- land in the reference's latest-first order."""
    props = get_backend_properties_v1(synthetic_backend(nq, two_q))
    rng = np.random.RandomState(nq)
    steps = list(range(1, 11)) if nq == 100 else list(range(15))
    texts = [circuit_to_qasm(tfim_circuit(nq, steps[k % len(steps)], float(rng.uniform(0, 2.0)), two_q=two_q)) for k in range(count)]
    x, ei = _same(NativeEncoder(props), texts)
    assert x.shape[0] > count and ei.shape[1] > x.shape[0] // 2


def test_patches_registers_and_odd_circuits(lima_props):
    """What the 16-byte record has no room for, and the shapes a collated batch must survive: gates with two and three parameters
    (patches), a calibrated three-qubit gate (patch), several registers, whole-register broadcasts, barriers of one and two qubits,
    a circuit without ops, a circuit of one op, repeated texts, bytes instead of str."""
    props = dict(lima_props)
    props["gates_set"] = list(lima_props["gates_set"]) + ["u2", "u3", "ccx"]
    props["gate_props"] = dict(lima_props["gate_props"])
    props["gate_props"]["ccx_0_1_2"] = {"index": 99, "gate_error": 0.0123, "gate_length": 7.5e-7}
    props["gate_props"]["u3_2"] = {"index": 98, "gate_error": 0.004, "gate_length": 3.5e-8}
    head = 'OPENQASM 2.0;\ninclude "qelib1.inc";\n'
    texts = [
        head + "qreg q[5];\ncreg c[5];\nu3(0.1,-0.2,pi/3) q[2];\nu2(1.5,2.5) q[0];\nccx q[0],q[1],q[2];\nccx q[2],q[1],q[0];\ncx q[0],q[1];\n"
               "barrier q[0];\nbarrier q[1],q[2];\nbarrier q;\nmeasure q -> c;\n",
        head + "qreg a[2];\nqreg b[3];\ncreg m[2];\nx a;\ncx a[0],b[2];\nrz(0.25) b;\nbarrier a,b;\nsx b[1];\nmeasure a -> m;\n",
        head + "qreg q[3];\n",
        head + "qreg q[1];\nx q[0];\n",
        head + "qreg q[4];\ncx q[0],q[1];\ncx q[0],q[1];\ncx q[1],q[0];\nid q[3];\nreset q[2];\nrz(-7.25) q[3];\n",
    ]
    texts = texts + [texts[0], texts[4].encode()]
    enc = NativeEncoder(props)
    for gate_f, qubit_f in ((True, True), (False, True), (False, False)):
        x, ei = _same(enc, texts, use_gate_features=gate_f, use_qubit_features=qubit_f)
    assert x[0, 0].item() == np.float32(0.1) and x[0, 1].item() == np.float32(-0.2) and x[0, 2].item() == np.float32(np.pi / 3)
    _same(enc, [])
    _same(enc, [texts[2]])
    with pytest.raises(Exception, match="circuit 1: "):
        enc.encode_batch_expand([texts[0], head + "qreg q[2];\nrz(1 q[0];\n"], DEV)
    _same(enc, texts[:2])                                # usable after a rejected run


def test_batched_decorator_on_the_device_expansion_equals_the_host_fill(g1, lima_backend):
    """ngem(..., batched=True) expands the batch on the device: the values of a run() are those of the same model call on the batch
    the host encoder fills (same arrays, same model call)."""
    from blackwater.data.backends import PauliObservable
    from blackwater.data.native_encoder import NativeEncoder
    from blackwater.data.utils import encode_pauli_sum_op, get_backend_properties_v1
    from blackwater.library.ngem.estimator import ngem
    from blackwater.nn import ExpValCircuitGraphModelA

    from test_estimators import FakeEstimator

    torch.manual_seed(0)
    model = ExpValCircuitGraphModelA(5, 22, 10).to(DEV).eval()
    circuits = [g1["qasm"][i] for i in range(0, 300, 7)]
    obs = PauliObservable("ZIIII")
    job = ngem(FakeEstimator, model, lima_backend, batched=True)().run(circuits, [obs] * len(circuits))
    base = job._base_job.result().values
    got = job.result().values
    x, edge_index, batch, _, _ = NativeEncoder(get_backend_properties_v1(lima_backend)).encode_batch(circuits)
    noisy = torch.tensor([[float(v)] for v in base], dtype=torch.float)
    observable = torch.tensor([encode_pauli_sum_op(obs)] * len(circuits), dtype=torch.float)
    with torch.no_grad():
        want = model(*[a.to(DEV) for a in (noisy, observable, torch.zeros(len(circuits), 1), x, edge_index, batch)])
    assert np.array_equal(np.asarray(got), want.reshape(len(circuits), -1)[:, 0].cpu().numpy().astype(np.asarray(got).dtype))
