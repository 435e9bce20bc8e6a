"""Family-A graph convolutions on the CSR aggregation kernel.

These stand in for ``torch_geometric.nn.{GCNConv, ChebConv, SAGEConv}`` as the reference uses them in
docs/tutorials/01_ngem.ipynb cell [9] (and 05/06 notebooks), with PyG's parameter names so state-dicts are
interchangeable: ``lin.weight``/``bias`` (GCN), ``lins.k.weight``/``bias`` (Cheb), ``lin_l.*``/``lin_r.weight`` (SAGE).
Each ``forward`` takes the batch's :class:`GraphStructure` instead of a raw ``edge_index``.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from ..native import functional as F
from ..native.structure import GraphStructure


def _glorot(out_f, in_f):
    w = torch.empty(out_f, in_f)
    nn.init.xavier_uniform_(w)
    return nn.Parameter(w)


def _kaiming_linear(out_f, in_f, bias=True):
    """torch.nn.Linear's default initialisation (what PyG's Linear falls back to)."""
    w = torch.empty(out_f, in_f)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    b = None
    if bias:
        bound = 1 / math.sqrt(in_f) if in_f > 0 else 0
        b = nn.Parameter(torch.empty(out_f).uniform_(-bound, bound))
    return nn.Parameter(w), b


class _WeightOnly(nn.Module):
    def __init__(self, weight, bias=None):
        super().__init__()
        self.weight = weight
        if bias is not None:
            self.bias = bias
        else:
            self.register_parameter("bias", None)


class GCNConv(nn.Module):
    """out = D^-1/2 (A+I) D^-1/2 (x W^T) + b, optional fused ReLU + dropout epilogue."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.lin = _WeightOnly(_glorot(out_channels, in_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, struct: GraphStructure, relu=False, drop_p=0.0, seed=0, **handover):
        """``handover``: ``defer_mask`` / ``x_gate_scale`` (native/functional.py, "Mask hand-over")."""
        return F.gcn_layer(x, self.lin.weight, self.bias, struct, relu=relu, drop_p=drop_p, seed=seed, **handover)


class SAGEConv(nn.Module):
    """out = act(lin_l(mean_{j->i} x_j) + lin_r(x_i)), optional fused ReLU + dropout epilogue."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        w, b = _kaiming_linear(out_channels, in_channels, bias=True)
        self.lin_l = _WeightOnly(w, b)
        w, _ = _kaiming_linear(out_channels, in_channels, bias=False)
        self.lin_r = _WeightOnly(w)

    def forward(self, x, struct: GraphStructure, relu=False, drop_p=0.0, seed=0, **handover):
        return F.sage_layer(x, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight, struct, relu=relu, drop_p=drop_p,
                            seed=seed, **handover)


class ChebConv(nn.Module):
    """out = act(sum_k lins[k](T_k) + b) with T_0 = x, T_1 = L^x, T_k = 2 L^ T_{k-1} - T_{k-2}, L^ = -D^-1/2 A D^-1/2."""

    def __init__(self, in_channels: int, out_channels: int, K: int):
        super().__init__()
        self.lins = nn.ModuleList([_WeightOnly(_glorot(out_channels, in_channels)) for _ in range(K)])
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, struct: GraphStructure, relu=False, drop_p=0.0, seed=0, **handover):
        return F.cheb_layer(x, [lin.weight for lin in self.lins], self.bias, struct, relu=relu, drop_p=drop_p, seed=seed,
                            **handover)
