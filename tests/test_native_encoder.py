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


def test_circuit_features_match_python_feature_rows(g1, golden_dir, lima_props):
    """mlqem_circuit_features_qasm: gate counts and the rotation-angle histogram of the MLP
    feature rows from the C++ op scan equal the Python path, bit for bit, on the reference's circuits."""
    import torch

    from blackwater.data.native_encoder import circuit_features
    from blackwater.library.learning.features import count_gates_by_rotation_angle, encode_data, encode_data_v2_ecr

    qasms = list(g1["qasm"][:40]) + [e["circuit"] for e in json.load(open(os.path.join(golden_dir, "encoder_goldens.json")))]
    noisy = [[0.1 * (k % 7)] for k in range(len(qasms))]
    ideal = [[0.0]] * len(qasms)
    a, _ = encode_data(qasms, lima_props, ideal, noisy, num_qubits=1)
    b, _ = encode_data(qasms, lima_props, ideal, noisy, num_qubits=1, native=True)
    assert torch.equal(a, b) and a[:, 8:54].abs().sum() > 0
    a, _ = encode_data_v2_ecr(qasms, ideal, noisy, obs_size=1, two_q_gate="cx")
    b, _ = encode_data_v2_ecr(qasms, ideal, noisy, obs_size=1, two_q_gate="cx", native=True)
    assert torch.equal(a, b) and a.shape[1] == 5 + 160 + 1
    # bin edges: left-closed bins, closed last bin, outside values dropped, two-qubit rotations ignored
    text = ('OPENQASM 2.0;\ninclude "qelib1.inc";\nqreg q[2];\nrz(0) q[0];\nrx(-2*pi) q[0];\nry(2*pi) q[1];\n'
            'rz(7) q[0];\nrzz(0.3) q[0],q[1];\nrz(0.5) q;\nbarrier q;\n')
    edges = np.array([-2 * np.pi, -1.0, 0.0, 0.5, 2 * np.pi])
    counts, hist = circuit_features(text, ["rz", "barrier", "h", "rz"], edges)
    assert counts.tolist() == [4, 1, 0, 4] and hist.tolist() == [1, 0, 1, 3]
    bins = count_gates_by_rotation_angle(text, 0.1 * np.pi)
    _, hist = circuit_features(text, [], np.arange(-2 * np.pi, 2 * np.pi + 0.1 * np.pi, 0.1 * np.pi))
    assert hist.tolist() == bins
    with pytest.raises(TypeError):
        encode_data([Circuit.from_qasm_str(text)], lima_props, [[0.0]], [[0.1]], num_qubits=1, native=True)


def test_batch_encoder_equals_the_collate_of_per_circuit_encodings(g1, lima_props):
    """mlqem_qasm_batch_parse / _fill (every circuit of one estimator run() at once): the
    collated batch equals ``Batch.from_data_list`` of the per-circuit encodings -- the reference's path
    (blackwater/library/ngem/estimator.py:61-66, then PyG's collate) -- bit for bit, whatever the number of worker threads;
    includes a circuit that broadcasts over whole registers, comments and a gate definition."""
    import torch

    from blackwater.data.graph import Batch, Data

    props = dict(lima_props, gates_set=G1_GATES_ORDER)
    enc = NativeEncoder(props)
    extra = ('OPENQASM 2.0;\ninclude "qelib1.inc";\ngate foo a { x a; }\nqreg q[3];\ncreg c[3];\nrz(-3*pi/4) q; // all three\n'
             'barrier q;\nx q[1];\nmeasure q -> c;\n')
    texts = list(g1["qasm"][:40]) + [extra] + list(g1["qasm"][40:60])
    entries = []
    for t in texts:
        x, ei, _, depth = enc.encode(t)
        entries.append(Data(x=torch.from_numpy(x).float(), edge_index=torch.from_numpy(ei), y=torch.zeros(1, 1)))
    want = Batch.from_data_list(entries)
    for threads in (1, 3, 0):
        x, ei, batch, counts, depths = enc.encode_batch(texts, threads=threads)
        assert x.dtype == torch.float32 and torch.equal(x, want.x)
        assert ei.dtype == torch.int64 and torch.equal(ei, want.edge_index)
        assert torch.equal(batch, want.batch)
        assert counts.tolist() == [e.x.shape[0] for e in entries]
        assert depths[:40] == [int(d) for d in g1["depth"][:40]]
    # the empty run()
    x, ei, batch, counts, depths = enc.encode_batch([])
    assert x.shape[0] == 0 and ei.shape == (2, 0) and batch.numel() == 0 and len(counts) == 0 and depths == []


def test_batch_encoder_names_the_first_bad_circuit(lima_props):
    enc = NativeEncoder(lima_props)
    good = 'OPENQASM 2.0;\nqreg q[2];\nx q[0];\n'
    with pytest.raises(Exception, match="circuit 2: "):
        enc.encode_batch([good, good, 'OPENQASM 2.0;\nqreg q[2];\nrz(1 q[0];\n', good, 'OPENQASM 2.0;\nqreg q[2];\nrz((1) q[0];\n'], threads=1)
    with pytest.raises(KeyError):           # a gate outside gates_set stays a KeyError, as in encode()
        enc.encode_batch([good, 'OPENQASM 2.0;\nqreg q[2];\nh q[0];\n'])
    x, _, _, _, _ = enc.encode_batch([good])        # the encoder is usable after a rejected batch
    assert x.shape[0] == 1


def test_batch_feature_scan_equals_the_per_circuit_scan(g1, lima_props):
    """mlqem_circuit_features_qasm_batch (the op scan of every circuit of a run() on host threads) against the per-circuit entry
    point, and ``encode_data(native=True)`` rows against the Python path on the reference's circuits."""
    import torch

    from blackwater.data.native_encoder import circuit_features, circuit_features_batch
    from blackwater.library.learning.features import encode_data

    texts = list(g1["qasm"][:50])
    gates = sorted(lima_props["gates_set"])
    edges = np.arange(-2 * np.pi, 2 * np.pi + 0.1 * np.pi, 0.1 * np.pi)
    for threads in (1, 4, 0):
        counts, hist = circuit_features_batch(texts, gates, edges, threads=threads)
        for i, t in enumerate(texts):
            c, h = circuit_features(t, gates, edges)
            assert np.array_equal(counts[i], c) and np.array_equal(hist[i], h)
    assert circuit_features_batch([], gates, edges)[0].shape == (0, len(gates))
    with pytest.raises(Exception, match="circuit 1: "):
        circuit_features_batch([texts[0], "OPENQASM 2.0;\nqreg q[2];\nrz(1 q[0];\n"], gates, edges)
    noisy = [[0.1 * k] for k in range(len(texts))]
    bases = [[1.0, 0, 1, 0, 0] for _ in texts]
    a, _ = encode_data(texts, lima_props, [[0.0]] * len(texts), noisy, 1, meas_bases=bases, native=True)
    b, _ = encode_data(texts, lima_props, [[0.0]] * len(texts), noisy, 1, meas_bases=bases, native=False)
    assert torch.equal(a, b)


def _literal_corpus():
    """Decimal literals that exercise the scanner's own number path: shortest round-trip reprs, the same values cut to 1-19
    digits, exponent forms, more than 19 digits (-> strtod), and the decimal expansions of points HALFWAY between two adjacent
    doubles cut to 17-19 digits (where rounding an extended-precision result a second time would go wrong)."""
    from decimal import Decimal, getcontext

    getcontext().prec = 60
    rng = np.random.RandomState(7)
    lits = ["0", "0.0", "1", "3.", ".5", "0.001", "000012.5000", "1e3", "1E-3", "2.5e+2", "6.283185307179586", "1.5707963267948966",
            "7.853981633974483", "123456789012345678", "1234567890123456789", "12345678901234567890123", "0.30000000000000004",
            "9007199254740993", "9007199254740992.5", "4.35", "0.1e-10", "1e22", "1e23", "8.5e-27", "1.7976931348623157e308", "5e-324"]
    vals = np.concatenate([rng.uniform(-10, 10, 3000), rng.lognormal(0, 6, 3000), rng.uniform(0, 2 * np.pi, 3000)])
    for v in np.abs(vals):
        r = repr(float(v))
        lits.append(r)
        d = Decimal(r)
        for k in (3, 9, 15, 16, 17, 18, 19):
            lits.append(format(d, f".{k}f") if d < 1000 else format(d, f".{k}e"))
        lo = float(v)
        hi = np.nextafter(lo, np.inf)
        mid = (Decimal(lo) + Decimal(float(hi))) / 2                  # exact: both are dyadic rationals
        for k in (17, 18, 19, 20):
            s = format(mid, "f")
            digits = 0
            cut = []
            for ch in s:                                              # k significant digits of the midpoint's expansion
                cut.append(ch)
                if ch.isdigit() and (digits or ch != "0"):
                    digits += 1
                if digits == k:
                    break
            lits.append("".join(cut))
    return [l for l in lits if "e" not in l.lower() or "." in l or l[0].isdigit()]


def test_scanner_literals_are_correctly_rounded(lima_props):
    """Every parameter the fast statement path parses itself (csrc/encode_qasm.cpp: fast_literal) equals Python's float() of the
    same text bit for bit -- i.e. strtod's correctly rounded value, which the general path uses -- also with a sign."""
    props = dict(lima_props)
    lits = _literal_corpus()
    assert len(lits) > 60000
    text = 'OPENQASM 2.0;\ninclude "qelib1.inc";\nqreg q[5];\n' + "".join(
        f"rz({'-' if k % 3 == 0 else ''}{l}) q[{k % 5}];\n" for k, l in enumerate(lits))
    x, _, _, _ = NativeEncoder(props).encode(text, edge_attr=False)
    want = np.array([(-1.0 if k % 3 == 0 else 1.0) * float(l) for k, l in enumerate(lits)])
    got = x[:, 0]
    bad = np.nonzero(got.view(np.int64) != want.view(np.int64))[0]
    assert bad.size == 0, [(lits[i], got[i], want[i]) for i in bad[:5]]


def test_fast_statement_path_equals_the_general_path(g1, lima_props, tmp_path):
    """MLQEM_QASM_FAST=0 (read once per process: a child interpreter) sends every statement through the general path, 1 through
    round 4's fast_gate first, 2 (the default) through the word scanner in front of that; rows, edges and depths are identical on
    transpiled circuits, hand-written statement shapes and refused inputs alike."""
    import subprocess
    import sys

    texts = [circuit_to_qasm(tfim_circuit(6, s, 0.3 * s + 0.1, two_q="cx")) for s in range(4)]
    texts.append('OPENQASM 2.0;\ninclude "qelib1.inc";\nqreg q[5];\ncreg c[5];\nrz( -0.5 ) q[ 1 ] ;\nsx q[1];cx q[0] , q[1];\nrz(pi/2) q[2];\n'
                 'x q;\nbarrier q[0],q[1];\nrz(+1.25) q[3];\nrz(1.5.3e) q[4];\n')
    texts.append('OPENQASM 2.0;\nqreg q[5];\nsx q[7];\n')
    texts.append('OPENQASM 2.0;\nqreg q[5];\nsx r[0];\n')
    texts.append('OPENQASM 2.0;\nqreg q[5];\nrz(0.25) q[1]junk;\nsx q[0];\n')
    texts.append('OPENQASM 2.0;\nqreg q[5];\nmeasurez q[0];\n')
    # the word scanner (MLQEM_QASM_FAST=2, the default): parameters of 1 .. 25 bytes seen once and again (its table is keyed by
    # their bytes), spaces and signs inside the parentheses, two registers taking turns, a last statement without ';', names of
    # 8 and 9 bytes, keyword prefixes, unbalanced parentheses, a text shorter than its look-ahead
    texts.append('OPENQASM 2.0;\nqreg q[3];\nqreg r[2];\nrz(1) q[0];rz(1.5) r[1];rz(1.5) q[1];rz(0.12345678) q[0];rz(0.123456789) q[0];'
                 'rz(0.1234567890123456) q[2];rz(0.12345678901234567) q[2];rz(0.12345678901234567) q[1];rz(1.234567890123456789e-3) q[0];'
                 'rz(1.2345678901234567890123) q[1];rz(1.2345678901234567890123) q[1];rz(1.23456789012345678901234) q[1];rz( 0.5 ) q[0];'
                 'rz(-0.5) q[0];rz(- 0.5) q[0];rz(0.5 ) q[0];rz(0.5) q[0];cx q[0],r[1];cx r[0],q[2];cx q[10],q[0];sx q[2]')
    texts.append('OPENQASM 2.0;\nqreg q[5];\nsx q[0];\nabcdefgh q[0];\n')
    texts.append('OPENQASM 2.0;\nqreg q[5];\nsx q[0];\nabcdefghi q[0];\n')
    texts.append('OPENQASM 2.0;\nqreg q[5];\nresetx q[0];\nsx q[1];\n')
    texts.append('OPENQASM 2.0;\nqreg q[5];\nrz(0.5 q[0];sx q[1];\nsx q[2];\n')
    texts.append('OPENQASM 2.0;\nqreg q[5];\nrz(0.5)) q[0];sx q[1];\n')
    texts.append('OPENQASM 2.0;\nqreg q[5];\nrz(0.5)(0.25) q[0];sx q[1];\n')
    texts.append('OPENQASM 2.0;qreg q[1];x q[0];')
    texts.append('OPENQASM 2.0;\nqreg verylongregister[4];\nsx verylongregister[3];\nx verylongregister[0];\nrz(0.5) verylongregister[1];\n')
    (tmp_path / "texts.json").write_text(json.dumps(texts))
    (tmp_path / "props.json").write_text(json.dumps(lima_props))
    script = (
        "import json, sys, numpy as np\n"
        f"sys.path[:0] = {[p for p in sys.path if p]!r}\n"
        "from blackwater.data.native_encoder import NativeEncoder\n"
        f"texts = json.load(open({str(tmp_path / 'texts.json')!r}))\n"
        f"enc = NativeEncoder(json.load(open({str(tmp_path / 'props.json')!r})))\n"
        "out = {}\n"
        "for k, t in enumerate(texts):\n"
        "    try:\n"
        "        x, ei, ea, d = enc.encode(t)\n"
        "        out[f'x{k}'], out[f'e{k}'], out[f'a{k}'], out[f'd{k}'] = x, ei, ea, np.array(d)\n"
        "    except Exception as err:\n"
        "        out[f'err{k}'] = np.array(type(err).__name__ + ': ' + str(err))\n"
        "np.savez(sys.argv[1], **out)\n")
    res = {}
    for mode in ("2", "1", "0"):
        path = tmp_path / f"out{mode}.npz"
        env = dict(os.environ, MLQEM_QASM_FAST=mode)
        subprocess.run([sys.executable, "-c", script, str(path)], check=True, env=env, timeout=300)
        res[mode] = dict(np.load(path))
    assert sorted(res["1"]) == sorted(res["0"]) == sorted(res["2"])
    assert any(k.startswith("err") for k in res["1"]) and any(k.startswith("x") for k in res["1"])
    for key in res["1"]:
        for mode in ("1", "2"):
            a, b = res[mode][key], res["0"][key]
            assert a.shape == b.shape and (a == b).all(), (mode, key)


def test_encoding_after_fork_does_not_wait_for_the_parents_workers(g1, lima_props):
    """the process-wide worker pool must not be inherited across fork() -- the child has none of its threads.  The parent
    encodes with helper threads, forks, and the child encodes the same batch again (a DataLoader / multiprocessing worker started by
    fork does this); the child must finish (an alarm ends a hung one) and produce the parent's arrays."""
    import signal

    import torch

    enc = NativeEncoder(dict(lima_props, gates_set=G1_GATES_ORDER))
    texts = list(g1["qasm"][:48])
    want = enc.encode_batch(texts, threads=4)
    pid = os.fork()
    if pid == 0:                                  # the child: exit codes only, no pytest machinery
        code = 3
        try:
            signal.alarm(20)
            got = enc.encode_batch(texts, threads=4)
            code = 0 if all(torch.equal(a, b) for a, b in zip(want[:3], got[:3])) else 2
        finally:
            os._exit(code)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, f"child status {status} (ended by the alarm = it hung)"
    again = enc.encode_batch(texts, threads=4)   # the parent's pool is untouched
    assert all(torch.equal(a, b) for a, b in zip(want[:3], again[:3]))
