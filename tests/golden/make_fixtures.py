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
    make_ckpts()
