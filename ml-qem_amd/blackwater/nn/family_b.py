"""Family B: TransformerConv x2 + ASAPooling x2 + global mean pool + head, the architecture of every GNN
checkpoint the reference ships (docs/tutorials/gnn.py:70-276; census in SURVEY.md section 2.3).

State-dict keys equal the reference's (``transformer1.lin_key.weight`` ... ``pooling1.gnn_score.lin2.weight`` ...
``body_seq.0.weight``).  Forward runs on the native kernels; the backward kernels of the attention / pooling ops are
not written yet, so these modules are inference-only for now and raise if a gradient is requested through them.
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
from .models import _Seq, as_structure


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
        # one projection for query | key | value | skip: x is read once
        w = torch.cat([self.lin_query.weight, self.lin_key.weight, self.lin_value.weight, self.lin_skip.weight], 0)
        b = torch.cat([self.lin_query.bias, self.lin_key.bias, self.lin_value.bias, self.lin_skip.bias], 0)
        qkvs = F.linear(x, w, b)
        if self.training and self.dropout > 0:
            raise NotImplementedError("TransformerConv: train-mode attention dropout needs the backward kernels (next)")
        fn = lambda t: ops.transformer_attention(t, struct.in_ptr, struct.in_src, struct.loops, self.heads,
                                                 self.out_channels)
        return F.forward_only(fn, "transformer_attention", qkvs)


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
        d, n = self.in_channels, struct.num_nodes
        s = struct

        def pool(x):
            xq = ops.linear(ops.csr_segment_max(x, s.in_ptr, s.in_src, ell=s.in_ell), self.lin.weight, self.lin.bias)
            att_w = self.att.weight
            a_dst = ops.linear(xq, att_w[:, :d].contiguous(), self.att.bias)[:, 0].contiguous()
            c_src = ops.linear(x, att_w[:, d:].contiguous())[:, 0].contiguous()
            x_new = ops.csr_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, self.negative_slope)
            g = self.gnn_score
            w3 = torch.cat([g.lin1.weight, g.lin2.weight, g.lin3.weight], 0)
            b3 = torch.cat([g.lin1.bias, torch.zeros_like(g.lin1.bias), g.lin3.bias], 0)
            fitness = ops.leconv_fitness(ops.linear(x_new, w3, b3).contiguous(), s.in_ptr, s.in_src)
            # k_g = ceil(ratio * n_g) evaluated in float32 like PyG's topk
            keep = [int(math.ceil(float(torch.tensor(self.ratio * float(m), dtype=torch.float32)))) for m in s.graph_sizes]
            new_ptr_host = [0]
            for k in keep:
                new_ptr_host.append(new_ptr_host[-1] + k)
            new_ptr = torch.tensor(new_ptr_host, dtype=torch.int32, device=x.device)
            perm = ops.segment_topk(fitness, s.graph_ptr, new_ptr, n, s.num_graphs, new_ptr_host[-1])
            x_out = ops.gather_scale_rows(x_new, perm, fitness)
            ei = ops.asap_coarsen(s.in_ptr, s.in_src, s.out_ptr, s.out_dst, perm, n)
            in_ptr, in_src, out_ptr, out_dst, loops = ops.csr_build(ei, new_ptr_host[-1])
            pooled = GraphStructure(new_ptr_host[-1], in_ptr, in_src, out_ptr, out_dst, loops, new_ptr, s.num_graphs,
                                    num_edges=int(ei.shape[1]), graph_sizes=keep)
            return x_out, pooled, perm

        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("ASAPooling: backward kernels not implemented yet (forward/inference only); "
                                      "call under torch.no_grad()")
        return pool(ops.rowmajor(x))


class _FamilyB(nn.Module):
    heads = (3, 2)

    def _build(self, num_node_features, hidden_channels):
        h1, h2 = self.heads
        self.transformer1 = TransformerConv(num_node_features, hidden_channels, heads=h1, dropout=0.1)
        self.pooling1 = ASAPooling(hidden_channels * h1, 0.5)
        self.transformer2 = TransformerConv(hidden_channels * h1, hidden_channels, heads=h2, dropout=0.1)
        self.pooling2 = ASAPooling(hidden_channels * h2, 0.5)
        return hidden_channels * h2

    def forward(self, exp_value, observable, circuit_depth, nodes, edge_index, batch):
        b = exp_value.shape[0]
        s = as_structure(edge_index, nodes.shape[0], batch, b)
        g = self.transformer1(nodes, s)
        g, s, _ = self.pooling1(g, s)
        g = self.transformer2(g, s)
        g, s, _ = self.pooling2(g, s)
        g = F.segment_mean(g, s)
        merged = torch.cat((g, torch.squeeze(exp_value, 1), circuit_depth), dim=1)
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
