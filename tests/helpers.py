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
