"""Shared helpers for the tests (not collected)."""
import numpy as np

# one-hot column order of the G1 dataset (inferred from op names vs node features; 'id'/'reset' never occur
# in those circuits so their two slots are interchangeable)
G1_GATES_ORDER = ["id", "reset", "sx", "x", "cx", "rz"]


def infer_gates_order(circ, x_rows, default_sorted):
    """Column order of the gate one-hot in a reference-encoded graph (hash-random in the reference)."""
    slot = {}
    for op, row in zip(circ.ops, x_rows):
        slot[op.name] = list(row[3:3 + len(default_sorted) + 2]).index(1.0)
    names = [None] * len(default_sorted)
    for name, s in slot.items():
        if name not in ("barrier", "measure"):
            names[s] = name
    rest = [g for g in default_sorted if g not in names]
    return [n if n is not None else rest.pop(0) for n in names]


def g1_graph(z, i):
    s, e = z["node_ptr"][i], z["node_ptr"][i + 1]
    es, ee = z["edge_ptr"][i], z["edge_ptr"][i + 1]
    return z["x"][s:e], z["edge_index"][:, es:ee], z["edge_attr"][es:ee]


def mean_l2(ideal, pred):
    return float(np.mean(np.linalg.norm(np.asarray(ideal) - np.asarray(pred), axis=1)))


def g1_batch(z, indices, self_loops=True, obs_terms=1, n_qubits=5, first_only=True):
    """Collates golden graphs the way the reference's training path does (AddSelfLoops + DataLoader collate,
    blackwater/data/loaders/exp_val.py:33, docs/tutorials/__ml_models.py:105) into plain CPU tensors.
    Family A consumes a scalar noisy value and an observable, so ``first_only`` keeps qubit 0's values and a
    deterministic synthetic observable is attached."""
    import torch

    xs, eis, bs, off = [], [], [], 0
    for b, i in enumerate(indices):
        x, ei, _ = g1_graph(z, i)
        n = x.shape[0]
        ei = torch.tensor(ei, dtype=torch.long)
        if self_loops:
            ei = torch.cat([ei, torch.arange(n).unsqueeze(0).repeat(2, 1)], dim=1)
        xs.append(torch.tensor(x, dtype=torch.float32))
        eis.append(ei + off)
        bs.append(torch.full((n,), b, dtype=torch.long))
        off += n
    idx = list(indices)
    noisy = torch.tensor(z["noisy"][idx], dtype=torch.float32)
    ideal = torch.tensor(z["ideal"][idx], dtype=torch.float32)
    gen = torch.Generator().manual_seed(1234)
    obs = torch.zeros(len(idx), obs_terms, 4 * n_qubits + 1)
    obs[:, :, 0] = torch.rand(len(idx), obs_terms, generator=gen) * 2 - 1
    pick = torch.randint(0, 4, (len(idx), obs_terms, n_qubits), generator=gen)
    obs[:, :, 1:] = torch.nn.functional.one_hot(pick, 4).reshape(len(idx), obs_terms, -1).float()
    return {
        "x": torch.cat(xs), "edge_index": torch.cat(eis, dim=1), "batch": torch.cat(bs),
        "noisy": noisy[:, :1] if first_only else noisy.unsqueeze(1),
        "y": ideal[:, :1] if first_only else ideal,
        "depth": torch.tensor(z["depth"][idx], dtype=torch.float32).unsqueeze(1),
        "observable": obs,
    }
