"""Regenerates the golden fixtures under tests/golden/ from the read-only reference checkout.

Runs ONLY in the build container (needs /root/reference); the GPU box uses the committed outputs.
Nothing here imports reference code: the reference's *data* artefacts (datasets, checkpoints, one
notebook cell output) are read and re-packed into compact fixtures.

Outputs (all data, no source):
  fake_lima_backend_props.json   BackendProperties.to_dict() of FakeLima as printed in
                                 docs/demos/fake_backend_info.ipynb cell [1] (dates dropped)
  g1_dataset.npz                 300 graphs + labels of
                                 docs/tutorials/data/ising_init_from_qasm_no_readout/val_extra/step_0.pk
  g1_circuits.json               the same 300 circuits as OpenQASM-2 text (rebuilt from the pickled op lists)
  encoder_goldens.json           dataset-format entries (QASM + reference-encoded circuit_graph) sampled from
                                 docs/tutorials/data/mbd_datasets2/theta_0.05pi/{train,val}/*.json
  ising_trainval.npz             the graphs + labels of docs/tutorials/data/ising_init_from_qasm_no_readout/
                                 {train/step_0 (300), val/step_0, val/step_1, val/step_2 (100 each)}.pk, in that order
                                 (``split`` = 0 for the train file, 1..3 for the val files) -- the training-to-accuracy set
  ising_trainval_circuits.json   the same 600 circuits as OpenQASM-2 text (input of the MLP feature encoder)
  ref_loss_curves.json           the loss curves the reference recorded next to its checkpoints
                                 (docs/tutorials/model/ising_init_from_qasm_no_readout/{gnn1,mlp1_smaller_2}.pk)
  ckpt/*.pth                     reference state-dicts (one per architecture used by the parity tests)
  ckpt_manifest.json             key -> shape for all 63 reference checkpoints (strict-load test)
  g4_digest.json                 tally + digests of ALL 1 100 QASM -> graph pairs of
                                 docs/tutorials/data/mbd_datasets2/theta_0.05pi/ run through the Python and the C++ encoder
"""
import datetime
import glob
import json
import os
import pickle
import shutil
import sys

import numpy as np
import torch

REF = "/root/reference"
TUT = os.path.join(REF, "docs/tutorials")
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))


# ---------------------------------------------------------------------------------------------
# stub unpickler: the .pk datasets embed qiskit objects; qiskit is not installed, so every class
# rooted at qiskit/symengine/rustworkx is replaced by a state-capturing dummy.
class _Stub:
    def __init__(self, *a, **k):
        self._args, self._kw = a, k

    def __setstate__(self, state):
        self._state = state

    def __call__(self, *a, **k):
        return _Stub()


def _stub_fn(*a, **k):
    return _Stub(*a, **k)


class _StubUnpickler(pickle.Unpickler):
    ROOTS = {"qiskit", "qiskit_aer", "symengine", "rustworkx", "qiskit_ibm_runtime", "qiskit_ibm_provider"}

    def find_class(self, module, name):
        if module.split(".")[0] in self.ROOTS:
            return _stub_fn if name[0].islower() else type(name, (_Stub,), {})
        return super().find_class(module, name)


def load_pk(path):
    with open(path, "rb") as fh:
        return _StubUnpickler(fh).load()


# ---------------------------------------------------------------------------------------------
def make_backend_props():
    nb = json.load(open(os.path.join(REF, "docs/demos/fake_backend_info.ipynb")))
    text = "".join(nb["cells"][1]["outputs"][0]["data"]["text/plain"])
    raw = eval(text, {"datetime": datetime, "tzoffset": lambda *a: None})  # a printed python literal

    def strip(o):
        if isinstance(o, dict):
            return {k: strip(v) for k, v in o.items() if not isinstance(v, datetime.datetime)}
        if isinstance(o, list):
            return [strip(v) for v in o]
        return o

    raw = strip(raw)
    with open(os.path.join(OUT, "fake_lima_backend_props.json"), "w") as fh:
        json.dump(raw, fh, indent=0)
    return raw


def _instr_fields(ins):
    s = ins._state[1]
    op = s["operation"]._state
    qs = [q._state[1]["_index"] for q in s["qubits"]]
    cs = [c._state[1]["_index"] for c in s["clbits"]]
    return op["_name"], qs, cs, op["_params"]


def circuit_to_qasm(circ, angle_from_graph=None):
    st = circ._state
    nq, nc = len(st["_qubits"]), len(st["_clbits"])
    creg = st["cregs"][0]._state[0] if st["cregs"] else "c"
    lines = ["OPENQASM 2.0;", 'include "qelib1.inc";', f"qreg q[{nq}];"]
    if nc:
        lines.append(f"creg {creg}[{nc}];")
    for k, ins in enumerate(st["_data"]):
        name, qs, cs, params = _instr_fields(ins)
        if name == "measure":
            lines.append(f"measure q[{qs[0]}] -> {creg}[{cs[0]}];")
            continue
        qtxt = ",".join(f"q[{i}]" for i in qs)
        if params:
            vals = []
            for j, p in enumerate(params):
                if isinstance(p, (int, float)):
                    vals.append(repr(float(p)))
                else:  # bound ParameterExpression stub -> the encoded graph holds its float value
                    vals.append(repr(float(angle_from_graph[k][j])))
            lines.append(f"{name}({','.join(vals)}) {qtxt};")
        else:
            lines.append(f"{name} {qtxt};")
    return "\n".join(lines) + "\n"


def make_g1():
    path = os.path.join(TUT, "data/ising_init_from_qasm_no_readout/val_extra/step_0.pk")
    entries = load_pk(path)
    xs, eis, eas, nptr, eptr = [], [], [], [0], [0]
    noisy, ideal, depth, qasm = [], [], [], []
    for e in entries:
        g = e["circuit_graph"]
        x = np.asarray(g["nodes"]["DAGOpNode"], dtype=np.float64)
        ed = g["edges"]["DAGOpNode_wire_DAGOpNode"]
        ei = np.asarray(ed["edge_index"], dtype=np.int32)
        xs.append(x)
        eis.append(ei)
        eas.append(np.asarray(ed["edge_attr"], dtype=np.float64))
        nptr.append(nptr[-1] + x.shape[0])
        eptr.append(eptr[-1] + ei.shape[1])
        assert len(e["noisy_exp_values"]) == 1 and e["observable"] == []
        noisy.append(e["noisy_exp_values"][0])
        ideal.append(e["ideal_exp_value"])
        depth.append(e["circuit_depth"])
        qasm.append(circuit_to_qasm(e["circuit"], angle_from_graph=x[:, :3]))
        assert len(e["circuit"]._state["_data"]) == x.shape[0]
    np.savez_compressed(
        os.path.join(OUT, "g1_dataset.npz"),
        x=np.concatenate(xs), edge_index=np.concatenate(eis, axis=1), edge_attr=np.concatenate(eas),
        node_ptr=np.asarray(nptr, np.int64), edge_ptr=np.asarray(eptr, np.int64),
        noisy=np.asarray(noisy, np.float64), ideal=np.asarray(ideal, np.float64),
        depth=np.asarray(depth, np.int64),
    )
    with open(os.path.join(OUT, "g1_circuits.json"), "w") as fh:
        json.dump(qasm, fh)
    print("g1:", len(entries), "graphs", nptr[-1], "nodes", eptr[-1], "edges")


def make_trainval():
    base = os.path.join(TUT, "data/ising_init_from_qasm_no_readout")
    xs, eis, nptr, eptr, noisy, ideal, depth, split, qasm = [], [], [0], [0], [], [], [], [], []
    for k, rel in enumerate(["train/step_0.pk", "val/step_0.pk", "val/step_1.pk", "val/step_2.pk"]):
        for e in load_pk(os.path.join(base, rel)):
            g = e["circuit_graph"]
            x = np.asarray(g["nodes"]["DAGOpNode"], dtype=np.float64)
            ei = np.asarray(g["edges"]["DAGOpNode_wire_DAGOpNode"]["edge_index"], dtype=np.int32)
            xs.append(x)
            eis.append(ei)
            nptr.append(nptr[-1] + x.shape[0])
            eptr.append(eptr[-1] + ei.shape[1])
            assert len(e["noisy_exp_values"]) == 1 and e["observable"] == []
            noisy.append(e["noisy_exp_values"][0])
            ideal.append(e["ideal_exp_value"])
            depth.append(e["circuit_depth"])
            split.append(k)
            qasm.append(circuit_to_qasm(e["circuit"], angle_from_graph=x[:, :3]))
    np.savez_compressed(os.path.join(OUT, "ising_trainval.npz"), x=np.concatenate(xs).astype(np.float32),
                        edge_index=np.concatenate(eis, axis=1), node_ptr=np.asarray(nptr, np.int64),
                        edge_ptr=np.asarray(eptr, np.int64), noisy=np.asarray(noisy, np.float64),
                        ideal=np.asarray(ideal, np.float64), depth=np.asarray(depth, np.int64),
                        split=np.asarray(split, np.int8))
    with open(os.path.join(OUT, "ising_trainval_circuits.json"), "w") as fh:
        json.dump(qasm, fh)
    curves = {}
    for name in ("gnn1", "mlp1_smaller_2"):
        d = load_pk(os.path.join(TUT, "model/ising_init_from_qasm_no_readout", name + ".pk"))
        curves[name] = {k: [float(v) for v in d[k]] for k in ("train_losses", "val_losses")}
    with open(os.path.join(OUT, "ref_loss_curves.json"), "w") as fh:
        json.dump(curves, fh)
    print("trainval:", len(split), "graphs", nptr[-1], "nodes")


def make_encoder_goldens():
    base = os.path.join(TUT, "data/mbd_datasets2/theta_0.05pi")
    picks = [("train/step_0.json", 6), ("val/step_0.json", 6), ("val/step_1.json", 10), ("val/step_2.json", 8)]
    out = []
    for rel, n in picks:
        data = json.load(open(os.path.join(base, rel)))
        step = max(1, len(data) // n)
        for e in data[::step][:n]:
            e = dict(e)
            e["source_file"] = "docs/tutorials/data/mbd_datasets2/theta_0.05pi/" + rel
            out.append(e)
    with open(os.path.join(OUT, "encoder_goldens.json"), "w") as fh:
        json.dump(out, fh)
    print("encoder goldens:", len(out))


G4_FILES = ["train/step_0.json", "val/step_0.json", "val/step_1.json", "val/step_2.json"]


def g4_pair_check(entry, lima_props, native_by_order):
    """One QASM -> graph pair of golden G4 through BOTH encoders of the build.  Returns (python_ok, native_ok, digest, pair name): the Python
    encoder against the stored graph (edges, edge_attr and every non-angle column exactly; angles within 1e-9: the stored QASM prints
    angles near k pi / n symbolically), the C++ encoder against the Python one bit for bit, and the sha256 of the native arrays."""
    import hashlib

    import numpy as np

    sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd"), os.path.join(ROOT, "tests")]
    from blackwater.data.circuit import Circuit
    from blackwater.data.native_encoder import NativeEncoder
    from blackwater.data.utils import circuit_to_graph_data_json
    from helpers import infer_gates_order

    circ = Circuit.from_qasm_str(entry["circuit"])
    want = entry["circuit_graph"]
    props = dict(lima_props)
    props["gates_set"] = infer_gates_order(circ, want["nodes"]["DAGOpNode"], lima_props["gates_set"])
    got = circuit_to_graph_data_json(circ, props, use_gate_features=True, use_qubit_features=True)
    a, b = np.array(got["nodes"]["DAGOpNode"]), np.array(want["nodes"]["DAGOpNode"])
    py_ok = (got["edges"] == want["edges"] and got["nodes"]["DAGInNode"] == want["nodes"]["DAGInNode"]
             and got["nodes"]["DAGOutNode"] == want["nodes"]["DAGOutNode"] and a.shape == b.shape
             and bool(np.array_equal(a[:, 3:], b[:, 3:])) and float(np.abs(a[:, :3] - b[:, :3]).max()) < 1e-9
             and circ.depth() == entry["circuit_depth"])
    key = tuple(props["gates_set"])
    enc = native_by_order.get(key)
    if enc is None:
        enc = native_by_order[key] = NativeEncoder(props)
    x, ei, ea, depth = enc.encode(entry["circuit"])
    wires = got["edges"]["DAGOpNode_wire_DAGOpNode"]
    nat_ok = (bool(np.array_equal(x, a)) and bool(np.array_equal(ei, np.array(wires["edge_index"])))
              and bool(np.array_equal(ea, np.array(wires["edge_attr"]))) and depth == entry["circuit_depth"])
    h = hashlib.sha256()
    h.update(entry["circuit"].encode())
    h.update(",".join(props["gates_set"]).encode())
    for arr, dt in ((x, np.float64), (ei, np.int64), (ea, np.float64)):
        h.update(np.ascontiguousarray(arr, dtype=dt).tobytes())
    # the pair's name: the text AND the one-hot column order of its stored graph (hash-random per generation run in the reference:
    # the same circuit text occurs with different orders)
    name = hashlib.sha256((entry["circuit"] + "|" + ",".join(props["gates_set"])).encode()).hexdigest()
    return py_ok, nat_ok, h.hexdigest(), name


def compute_g4_digest():
    """ALL 1 100 QASM -> graph pairs the reference holds through the Python and the C++ encoder, here in the
    build container; what is committed is the tally and one digest per file (and per pair for the 30 committed pairs), which
    tests/test_encoder_goldens.py compares against -- the pairs themselves (13 MB of JSON) stay in the reference."""
    import hashlib

    base = os.path.join(TUT, "data/mbd_datasets2/theta_0.05pi")
    sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
    from blackwater.data.backends import StaticBackend
    from blackwater.data.utils import get_backend_properties_v1

    lima = get_backend_properties_v1(StaticBackend.from_json(os.path.join(OUT, "fake_lima_backend_props.json")))
    with open(os.path.join(OUT, "encoder_goldens.json")) as fh:
        committed = {e["circuit"] for e in json.load(fh)}
    native, files, committed_digests = {}, {}, {}
    tot = py = nat = 0
    for rel in G4_FILES:
        data = json.load(open(os.path.join(base, rel)))
        fh_ = hashlib.sha256()
        n_py = n_nat = 0
        for e in data:
            p_ok, n_ok, dig, name = g4_pair_check(e, lima, native)
            n_py += p_ok
            n_nat += n_ok
            fh_.update(bytes.fromhex(dig))
            if e["circuit"] in committed:
                committed_digests[name] = dig
        files["docs/tutorials/data/mbd_datasets2/theta_0.05pi/" + rel] = {"pairs": len(data), "python_encoder_matches": n_py,
                                                                          "native_encoder_matches": n_nat, "sha256_of_pair_digests": fh_.hexdigest()}
        tot, py, nat = tot + len(data), py + n_py, nat + n_nat
    out = {"what": "golden G4: every QASM -> circuit_graph pair the reference stores, through the build's Python encoder (vs the stored "
                   "graph) and C++ encoder (vs the Python one), run by tests/golden/make_fixtures.py make_g4_digest() in the build container",
           "pairs": tot, "python_encoder_matches": py, "native_encoder_matches": nat, "files": files,
           "committed_pair_digests": committed_digests}
    return out


def make_g4_digest():
    out = compute_g4_digest()
    with open(os.path.join(OUT, "g4_digest.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("G4:", out["pairs"], "pairs; python", out["python_encoder_matches"], "native", out["native_encoder_matches"])


def make_ckpts():
    keep = {
        "model/ising_init_from_qasm_no_readout/gnn1.pth": "gnn1.pth",
        "model/ising_init_from_qasm_no_readout/mlp1_smaller_2.pth": "mlp1_smaller_2.pth",
        "model/ising/gnn3_ising.pth": "gnn3_ising.pth",
        "model/finetuning/iskandar.pth": "iskandar.pth",
        "model/finetuning/train_fakelima.pth": "train_fakelima.pth",
        "model/haoran_mbd2/mlp2_mbd.pth": "mlp2_mbd.pth",
        "model/ising/mlp3_ising.pth": "mlp3_ising.pth",
    }
    for src, dst in keep.items():
        shutil.copyfile(os.path.join(TUT, src), os.path.join(OUT, "ckpt", dst))
    manifest = {}
    for p in sorted(glob.glob(os.path.join(TUT, "model/**/*.pth"), recursive=True)):
        sd = torch.load(p, map_location="cpu", weights_only=True)
        manifest[os.path.relpath(p, TUT)] = {k: list(v.shape) for k, v in sd.items()}
    with open(os.path.join(OUT, "ckpt_manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=0)
    print("checkpoints in manifest:", len(manifest))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs the reference checkout at /root/reference")
    make_backend_props()
    make_g1()
    make_trainval()
    make_encoder_goldens()
    make_g4_digest()
    make_ckpts()
