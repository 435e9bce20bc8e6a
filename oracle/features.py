"""Oracle restatement of the MLP feature encoders (docs/tutorials/mlp.py:124-252); test infra only.

Takes plain op lists ``[(name, n_qubits, first_param_or_None), ...]`` so it shares no code with the product's
circuit IR.  Arithmetic follows the reference line by line: float64 means x100 cast to f32 (:210-218,:232),
integer counts promoted to f32 then x0.01 (:236-238,:242).
"""
import numpy as np
import torch


def _collect(d, parent, key1, key2, out):
    for k, v in d.items():
        if isinstance(v, dict):
            _collect(v, k, key1, key2, out)
        elif parent and key1 in str(parent) and k == key2:
            out.append(v)
    return out


def encode_data_rows(op_lists, properties, noisy, num_vals, bases=None):
    gates = sorted(properties["gates_set"])
    sel = [("cx", "gate_error"), ("id", "gate_error"), ("sx", "gate_error"), ("x", "gate_error"),
           ("rz", "gate_error"), ("", "readout_error"), ("", "t1"), ("", "t2")]
    vec = torch.tensor([np.mean(_collect(properties, None, a, b, []) or 0.) for a, b in sel]) * 100
    bin_size = 0.1 * np.pi
    nb = int(np.ceil(4 * np.pi / bin_size))
    nbase = len(bases[0]) if bases else 0
    X = torch.zeros(len(op_lists), 8 + len(gates) + nb + num_vals + nbase)
    X[:, :8] = vec[None, :]
    for i, ops in enumerate(op_lists):
        cnt = {}
        for name, _, _ in ops:
            cnt[name] = cnt.get(name, 0) + 1
        X[i, 8:8 + len(gates)] = torch.tensor([cnt.get(g, 0) for g in gates]) * 0.01
        angles = [p for name, nq, p in ops if name in ("rx", "ry", "rz") and nq == 1]
        hist, _ = np.histogram(angles, bins=np.arange(-2 * np.pi, 2 * np.pi + bin_size, bin_size))
        X[i, 8 + len(gates):8 + len(gates) + nb] = torch.tensor(list(hist)) * 0.01
        X[i, 8 + len(gates) + nb:8 + len(gates) + nb + num_vals] = torch.tensor(noisy[i])
        if bases:
            X[i, 8 + len(gates) + nb + num_vals:] = torch.tensor(bases[i])
    return X
