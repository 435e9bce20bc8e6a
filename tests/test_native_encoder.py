"""The C++ encoder (CPU-only entry point of the C ABI) against the Python encoder and the reference's goldens."""
import json
import os

import numpy as np
import pytest

from blackwater.data.circuit import Circuit, circuit_to_qasm
from blackwater.data.native_encoder import NativeEncoder
from blackwater.data.synthetic import synthetic_backend, tfim_circuit
from blackwater.data.utils import circuit_to_graph_data_json, get_backend_properties_v1
from helpers import G1_GATES_ORDER, g1_graph


def test_g1_circuits_bit_exact(g1, lima_props):
    props = dict(lima_props, gates_set=G1_GATES_ORDER)
    enc = NativeEncoder(props)
    for i, text in enumerate(g1["qasm"]):
        x, ei, ea, depth = enc.encode(text)
        want_x, want_ei, want_ea = g1_graph(g1, i)
        assert np.array_equal(x, want_x) and np.array_equal(ei, want_ei) and np.array_equal(ea, want_ea)
        assert depth == g1["depth"][i]


def test_matches_python_encoder_on_json_goldens_and_options(golden_dir, lima_props):
    entries = json.load(open(os.path.join(golden_dir, "encoder_goldens.json")))
    enc = NativeEncoder(lima_props)
    for e in entries[::3]:
        for gate_f, qubit_f in ((True, True), (False, False), (True, False)):
            want = circuit_to_graph_data_json(e["circuit"], lima_props, use_gate_features=gate_f, use_qubit_features=qubit_f)
            x, ei, ea, depth = enc.encode(e["circuit"], use_gate_features=gate_f, use_qubit_features=qubit_f)
            assert np.array_equal(x, np.array(want["nodes"]["DAGOpNode"], dtype=np.float64))
            wires = want["edges"]["DAGOpNode_wire_DAGOpNode"]
            assert np.array_equal(ei, np.array(wires["edge_index"])) and np.array_equal(ea, np.array(wires["edge_attr"]))
            assert depth == e["circuit_depth"]


def test_100_qubit_circuit_and_errors():
    props = get_backend_properties_v1(synthetic_backend(100))
    circ = tfim_circuit(100, 3, J=0.4)
    text = circuit_to_qasm(circ)
    x, ei, ea, depth = NativeEncoder(props).encode(text)
    want = circuit_to_graph_data_json(Circuit.from_qasm_str(text), props, use_gate_features=True, use_qubit_features=True)
    assert x.shape == (len(circ.ops), 22)
    assert np.array_equal(x, np.array(want["nodes"]["DAGOpNode"]))
    assert np.array_equal(ei, np.array(want["edges"]["DAGOpNode_wire_DAGOpNode"]["edge_index"]))
    assert depth == circ.depth()
    enc = NativeEncoder(props)
    with pytest.raises(KeyError):
        enc.encode('OPENQASM 2.0;\nqreg q[2];\nh q[0];\n')
    with pytest.raises(Exception, match="more than 3 qubits"):
        enc.encode('OPENQASM 2.0;\nqreg q[5];\necr q[0],q[1],q[2],q[3];\n')
    x, ei, _, _ = enc.encode('OPENQASM 2.0;\ninclude "qelib1.inc";\nqreg q[3];\ncreg c[3];\nrz(-3*pi/4) q; // all\n'
                             'barrier q;\nmeasure q -> c;\n')
    assert x.shape[0] == 7 and x[0, 0] == -3 * np.pi / 4 and ei.shape[1] == 6
