"""Pins the CPU oracle to the reference's printed numbers (G1, G2, G3, G6)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import g1_graph, mean_l2
from oracle.features import encode_data_rows
from oracle.models import MLP1, MLP2, MLP3, family_b_from_state_dict


def _family_b_outputs(model, g1, dtype):
    outs = []
    with torch.no_grad():
        for i in range(len(g1["depth"])):
            x, ei, _ = g1_graph(g1, i)
            x = torch.tensor(x, dtype=torch.float32).to(dtype)  # the reference builds f32 tensors from the lists
            noisy = torch.tensor(g1["noisy"][i], dtype=torch.float32).to(dtype).view(1, 1, -1)
            depth = torch.tensor([[float(g1["depth"][i])]], dtype=dtype)
            # inference sites pass the raw graph: no self-loops, batch=None (ngem/estimator.py:68-82)
            out = model(noisy, None, depth, x, torch.tensor(ei, dtype=torch.long), None)
            outs.append(out.numpy().ravel())
    return np.stack(outs)


def test_g1_family_b_golden(golden_dir, g1):
    sd = torch.load(os.path.join(golden_dir, "ckpt", "gnn1.pth"), weights_only=True)
    assert sum(v.numel() for v in sd.values()) == 13645  # docs/tutorials/h01_mbd.ipynb:297
    model = family_b_from_state_dict(sd).double().eval()
    l2 = mean_l2(g1["ideal"], _family_b_outputs(model, g1, torch.float64))
    assert round(l2, 6) == 0.117838  # docs/tutorials/h17_compare_over_steps.ipynb:513, L2_gnn step 0
    model32 = family_b_from_state_dict(sd).eval()
    out32 = _family_b_outputs(model32, g1, torch.float32)
    assert np.abs(out32 - _family_b_outputs(model, g1, torch.float64)).max() < 1e-5


def test_g3_noisy_golden(g1):
    assert round(mean_l2(g1["ideal"], g1["noisy"]), 6) == 0.027510  # same cell, L2_noisy


def test_g2_mlp_golden(golden_dir, g1, lima_props):
    from blackwater.data.circuit import Circuit

    ops = [[(o.name, len(o.qubits), o.params[0] if o.params else None) for o in Circuit.from_qasm_str(t).ops]
           for t in g1["qasm"]]
    X = encode_data_rows(ops, lima_props, g1["noisy"].tolist(), 4)
    assert X.shape == (300, 58)
    want_vec = [1.113405, 0.035497, 0.035497, 0.514567, 0.0, 3.64, 0.0061991, 0.0068301]
    assert np.allclose(X[0, :8].numpy(), want_vec, rtol=2e-5, atol=1e-7)
    sd = torch.load(os.path.join(golden_dir, "ckpt", "mlp1_smaller_2.pth"), weights_only=True)
    assert sum(v.numel() for v in sd.values()) == 4036  # docs/tutorials/h10_mlp.ipynb:331
    model = MLP1(58, 64, 4)
    model.load_state_dict(sd, strict=True)
    out = model.eval()(X).detach().numpy()
    assert round(mean_l2(g1["ideal"], out), 6) == 0.032910  # h17 cell [14], L2_mlp step 0


def test_g6_every_reference_checkpoint_loads_strictly(golden_dir):
    manifest = json.load(open(os.path.join(golden_dir, "ckpt_manifest.json")))
    assert len(manifest) == 63
    n_gnn = n_mlp = 0
    for name, shapes in manifest.items():
        sd = {k: torch.zeros(s) for k, s in shapes.items()}
        if "transformer1.lin_key.weight" in sd:
            family_b_from_state_dict(sd)
            n_gnn += 1
        else:
            i, h = shapes["fc1.weight"][1], shapes["fc1.weight"][0]
            if "fc4.weight" in shapes:
                m = MLP3(i, h, shapes["fc4.weight"][0])
            elif "fc3.weight" in shapes:
                m = MLP2(i, h, shapes["fc3.weight"][0])
            else:
                m = MLP1(i, h, shapes["fc2.weight"][0])
            m.load_state_dict(sd, strict=True)
            n_mlp += 1
    assert (n_gnn, n_mlp) == (42, 21)
