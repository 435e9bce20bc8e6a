"""Family B: TransformerConv x2 + ASAPooling x2 + global mean pool + head, the architecture of every GNN
checkpoint the reference ships (docs/tutorials/gnn.py:70-276; census in SURVEY.md section 2.3).

State-dict keys equal the reference's (``transformer1.lin_key.weight`` ... ``pooling1.gnn_score.lin2.weight`` ...
``body_seq.0.weight``).  Forward and backward run on the native kernels (attention with dropout on the weights in train mode; ASAPooling's
top-k and coarsened connectivity are structural and carry no gradient, as in PyG).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from ..native import functional as F
from ..native import ops
from ..native.structure import GraphStructure
from .conv import _kaiming_linear, _WeightOnly
from .mlp import MLP2, MLP3
from .models import _Seq, as_structure, dropout_key


def _lin(in_f, out_f, bias=True):
    w, b = _kaiming_linear(out_f, in_f, bias=bias)
    return _WeightOnly(w, b)


class TransformerConv(nn.Module):
    """heads=H, concat=True, beta=False, edge_dim=None, root_weight=True, dropout on the attention weights in train
    mode (reference construction: gnn.py:80-91)."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, dropout: float = 0.0):
        super().__init__()
        self.heads, self.out_channels, self.dropout = heads, out_channels, dropout
        hc = heads * out_channels
        self.lin_key, self.lin_query = _lin(in_channels, hc), _lin(in_channels, hc)
        self.lin_value, self.lin_skip = _lin(in_channels, hc), _lin(in_channels, hc)

    def forward(self, x, struct: GraphStructure):
        # one projection for query | key | value | skip: x is read once; the four parameters are fused (and padded per head) by one
        # launch inside the autograd node
        w = [self.lin_query.weight, self.lin_key.weight, self.lin_value.weight, self.lin_skip.weight]
        b = [self.lin_query.bias, self.lin_key.bias, self.lin_value.bias, self.lin_skip.bias]
        self._calls = getattr(self, "_calls", 0) + 1
        drop = self.dropout if self.training else 0.0
        # static_dropout_key: a device-resident step counter varies the masks instead (train.BucketedTrainer, hipGraph replay)
        call = 0 if getattr(self, "static_dropout_key", False) else self._calls
        return F.transformer_conv(x, w, b, struct, self.heads, self.out_channels, drop_p=drop,
                                  seed=dropout_key(call, salt=self.heads * 1000003 + self.out_channels))


class _LEConv(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.lin1, self.lin2, self.lin3 = _lin(in_channels, 1), _lin(in_channels, 1, bias=False), _lin(in_channels, 1)


class ASAPooling(nn.Module):
    """ASAPooling(in_channels, ratio) with GNN=None, dropout=0, negative_slope=0.2, add_self_loops=False
    (reference construction: gnn.py:85,92).  Returns (x, structure of the pooled graph, perm)."""

    def __init__(self, in_channels: int, ratio: float = 0.5, negative_slope: float = 0.2):
        super().__init__()
        self.in_channels, self.ratio, self.negative_slope = in_channels, ratio, negative_slope
        self.lin = _lin(in_channels, in_channels)
        self.att = _lin(2 * in_channels, 1)
        self.gnn_score = _LEConv(in_channels)

    def forward(self, x, struct: GraphStructure):
        return F.asap_pool(x, self, struct)


class _FamilyB(nn.Module):
    heads = (3, 2)
    # ASAPooling sizes its outputs from the per-graph node counts (k_g = ceil(n_g / 2), the capacity of the coarsened edge
    # list): a captured step (train.BucketedTrainer) is only valid for batches with the SAME sequence of graph sizes
    needs_size_pattern = True
    accepts_device_batches = True

    def _build(self, num_node_features, hidden_channels):
        h1, h2 = self.heads
        self.transformer1 = TransformerConv(num_node_features, hidden_channels, heads=h1, dropout=0.1)
        self.pooling1 = ASAPooling(hidden_channels * h1, 0.5)
        self.transformer2 = TransformerConv(hidden_channels * h1, hidden_channels, heads=h2, dropout=0.1)
        self.pooling2 = ASAPooling(hidden_channels * h2, 0.5)
        return hidden_channels * h2

    def forward(self, exp_value, observable, circuit_depth, nodes, edge_index, batch):
        # a device batch hands over rows-of-the-arena (ops.RowsOf): the first projection and its weight gradient read them through
        # the row map, no per-batch copy of the features is made
        b = exp_value.shape[0]
        s = as_structure(edge_index, nodes.shape[0], batch, b)
        real = getattr(s, "num_real", None)
        self.transformer1.static_dropout_key = self.transformer2.static_dropout_key = getattr(self, "static_dropout_key", False)
        self.body_seq.static_dropout_key = getattr(self, "static_dropout_key", False)      # an MLP2 / MLP3 head draws masks too
        g = self.transformer1(nodes, s)
        g, s, _ = self.pooling1(g, s)
        g = self.transformer2(g, s)
        g, s, _ = self.pooling2(g, s)
        g = F.segment_mean(g, s)
        merged = torch.cat((g, torch.squeeze(exp_value, 1), circuit_depth), dim=1)
        if real is not None and real < b and isinstance(self.body_seq, (MLP2, MLP3)):
            # a bucket-padded batch ends in edgeless filler graphs: a head with BatchNorm must not see their rows (its batch
            # statistics are over the circuits: docs/tutorials/gnn.py:150-170); the other heads are row-wise and the trainer cuts the
            # filler rows off before the loss
            merged = merged[:real]
        return self.body_seq(merged)


class ExpValCircuitGraphModel(_FamilyB):
    """gnn.py:70-122: heads 3/2, head = Linear -> Dropout -> Linear."""

    def __init__(self, num_node_features: int, hidden_channels: int, exp_value_size: int = 4, dropout: float = 0.2):
        super().__init__()
        pooled = self._build(num_node_features, hidden_channels)
        self.body_seq = _Seq([pooled + 1 + exp_value_size, hidden_channels, exp_value_size], dropout=dropout)


class ExpValCircuitGraphModel_2(_FamilyB):
    """gnn.py:126-173 (heads 3/2, MLP2 head of width hidden_channels; shapes from cliffords_and_mbd3.pth)."""

    def __init__(self, num_node_features: int, hidden_channels: int, exp_value_size: int = 4, dropout: float = 0.3):
        super().__init__()
        pooled = self._build(num_node_features, hidden_channels)
        self.body_seq = MLP2(pooled + 1 + exp_value_size, hidden_channels, exp_value_size, dropout)


class ExpValCircuitGraphModel_3(_FamilyB):
    """gnn.py:178-224: heads 5/3, MLP3 head of width 5 * hidden_channels."""

    heads = (5, 3)

    def __init__(self, num_node_features: int, hidden_channels: int, exp_value_size: int = 4, dropout: float = 0.3):
        super().__init__()
        pooled = self._build(num_node_features, hidden_channels)
        self.body_seq = MLP3(pooled + 1 + exp_value_size, hidden_channels * 5, exp_value_size, dropout)


class ExpValCircuitGraphModel_4(_FamilyB):
    """gnn.py:229-276: heads 5/3, MLP3 head of width hidden_channels."""

    heads = (5, 3)

    def __init__(self, num_node_features: int, hidden_channels: int, exp_value_size: int = 4, dropout: float = 0.3):
        super().__init__()
        pooled = self._build(num_node_features, hidden_channels)
        self.body_seq = MLP3(pooled + 1 + exp_value_size, hidden_channels, exp_value_size, dropout)


def family_b_from_state_dict(sd) -> _FamilyB:
    """Instantiates the variant whose shapes match a reference checkpoint and loads it with ``strict=True``."""
    f = sd["transformer1.lin_key.weight"].shape[1]
    hc1, hc2 = sd["transformer1.lin_key.weight"].shape[0], sd["transformer2.lin_key.weight"].shape[0]
    if "body_seq.0.weight" in sd:
        hidden, out = sd["body_seq.0.weight"].shape[0], sd["body_seq.2.weight"].shape[0]
        model = ExpValCircuitGraphModel(f, hidden, out)
    else:
        head_hidden = sd["body_seq.fc1.weight"].shape[0]
        if "body_seq.fc4.weight" in sd:
            out, hidden = sd["body_seq.fc4.weight"].shape[0], hc1 // 5
            model = (ExpValCircuitGraphModel_3 if head_hidden == hidden * 5 else ExpValCircuitGraphModel_4)(f, hidden, out)
        else:
            out, hidden = sd["body_seq.fc3.weight"].shape[0], hc1 // 3
            model = ExpValCircuitGraphModel_2(f, hidden, out)
    model.load_state_dict(sd, strict=True)
    return model
