"""The reference's model families as modules over the native kernels.

All models keep the reference's six-positional-argument protocol
``model(exp_value, observable, circuit_depth, nodes, edge_index, batch)`` (pinned by
tests/library/ngem/test_estimator.py:24 in the reference).  ``edge_index`` may be the reference's [2,E] int64
tensor (the structure is then built on the device on the fly) or a prebuilt :class:`GraphStructure`
(what the device-resident dataset hands out, so nothing is rebuilt per step).
"""
from __future__ import annotations

from typing import Optional, Union

import torch
from torch import nn

from ..native import functional as F
from ..native import ops
from ..native.structure import GraphStructure
from .conv import ChebConv, GCNConv, SAGEConv, _kaiming_linear, _WeightOnly


def as_structure(edge_index: Union[torch.Tensor, GraphStructure], num_nodes: int, batch: Optional[torch.Tensor],
                 num_graphs: int) -> GraphStructure:
    if isinstance(edge_index, GraphStructure):
        return edge_index
    return GraphStructure.from_edge_index(edge_index, num_nodes, batch=batch, num_graphs=num_graphs)


def dropout_key(call: int, salt: int = 0) -> int:
    """Key of the counter-based dropout masks of one forward call: a function of torch's seed (so runs are reproducible
    under ``torch.manual_seed`` and differ between seeds, like the reference's ``nn.Dropout``), of the data-parallel
    rank (replicas draw different masks) and of the module's call counter."""
    rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
    z = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (rank + 1) + 0xBF58476D1CE4E5B9 * (call + 1) + salt) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return (z ^ (z >> 31)) & 0x7FFFFFFFFFFFFFFF


class _Seq(nn.Module):
    """Two Linear layers at indices 0 and 2 (index 1 is the reference's Dropout), PyG/torch key names."""

    def __init__(self, sizes, dropout=None, second_index=2):
        super().__init__()
        w, b = _kaiming_linear(sizes[1], sizes[0])
        self.add_module("0", _WeightOnly(w, b))
        w, b = _kaiming_linear(sizes[2], sizes[1])
        self.add_module(str(second_index), _WeightOnly(w, b))
        self._second, self.p = str(second_index), dropout

    def forward(self, x):
        first, second = getattr(self, "0"), getattr(self, self._second)
        if F.seq2_fused_ok(x, first.weight, second.weight):
            # one launch per direction (csrc/seq2.hip); the dropout mask comes from the counter-keyed hash of the native layers:
            # the module's call count, or the device-resident step counter under hipGraph replay (static_dropout_key)
            self._calls = getattr(self, "_calls", 0) + 1
            p = self.p if (self.p and self.training) else 0.0
            call = 0 if getattr(self, "static_dropout_key", False) else self._calls
            return F.seq2(x, first.weight, first.bias, second.weight, second.bias, drop_p=p,
                          seed=dropout_key(call, salt=first.weight.shape[1] * 1009 + first.weight.shape[0]))
        h = F.linear(x, first.weight, first.bias)
        if self.p and self.training:
            h = nn.functional.dropout(h, self.p, True)
        return F.linear(h, second.weight, second.bias)


class ExpValCircuitGraphModelA(nn.Module):
    """Family A: GCNx3 || Chebx2 || SAGEx2 -> mean pools; observable MLP; 6-wide body
    (reference: docs/tutorials/01_ngem.ipynb cell [9], constructed with n_qubits=5, 22 features, hidden 10)."""

    # ``nodes`` may be rows of a device-resident arena (ops.RowsOf) and ``edge_index`` a GraphStructure: the estimators' serial path
    # then replays one captured forward per size bucket (train.BucketedPredictor) instead of enqueueing every launch per circuit
    accepts_device_batches = True

    def __init__(self, n_qubits: int, num_node_features: int, hidden_channels: int):
        super().__init__()
        hc = hidden_channels
        self.conv1, self.conv2, self.conv3 = GCNConv(num_node_features, hc), GCNConv(hc, hc), GCNConv(hc, 1)
        self.cheb_conv1, self.cheb_conv2 = ChebConv(num_node_features, hc, K=3), ChebConv(hc, 1, K=2)
        self.sage_conv1, self.sage_conv2 = SAGEConv(num_node_features, hc), SAGEConv(hc, 1)
        self.obs_seq = _Seq([n_qubits * 4 + 1, hc, 1], dropout=0.2)
        self.body_seq = _Seq([6, hc, 1], second_index=1)
        self._step = 0

    def forward(self, exp_value, observable, circuit_depth, nodes, edge_index, batch):
        return self.forward_layers(exp_value, observable, circuit_depth, nodes, edge_index, batch)

    def _graph_params(self):
        c1, c2, s1, s2 = self.cheb_conv1, self.cheb_conv2, self.sage_conv1, self.sage_conv2
        return (self.conv1.lin.weight, self.conv1.bias, self.conv2.lin.weight, self.conv2.bias, self.conv3.lin.weight,
                self.conv3.bias, c1.lins[0].weight, c1.lins[1].weight, c1.lins[2].weight, c1.bias, c2.lins[0].weight,
                c2.lins[1].weight, c2.bias, s1.lin_l.weight, s1.lin_l.bias, s1.lin_r.weight, s2.lin_l.weight,
                s2.lin_l.bias, s2.lin_r.weight)

    def _single_node_ok(self, nodes):
        """The one-node form covers the reference's configuration (K = 3 / 2, node features that need no gradient)."""
        return (getattr(self, "single_node", True) and not nodes.requires_grad and len(self.cheb_conv1.lins) == 3
                and len(self.cheb_conv2.lins) == 2)

    def forward_layers(self, exp_value, observable, circuit_depth, nodes, edge_index, batch):
        if isinstance(nodes, ops.RowsOf) and not self._single_node_ok(nodes):
            nodes = nodes.materialize()      # the per-layer autograd nodes save plain tensors
        b = exp_value.shape[0]
        s = as_structure(edge_index, nodes.shape[0], batch, b)
        train = self.training
        self._step += 1
        # static_dropout_key: a device-resident step counter varies the masks instead (train.BucketedTrainer, hipGraph replay)
        seed = dropout_key(0 if getattr(self, "static_dropout_key", False) else self._step)
        # Each hidden activation has exactly one consumer -- the next layer of its branch -- so the ReLU/dropout mask of
        # its backward is applied by that consumer's data-gradient GEMM (native/functional.py, "Mask hand-over").
        p1, p2 = (0.1, 0.2) if train else (0.0, 0.0)
        k1, k2 = 1.0 / (1.0 - p1), 1.0 / (1.0 - p2)
        self.obs_seq.static_dropout_key = self.body_seq.static_dropout_key = getattr(self, "static_dropout_key", False)
        if self._single_node_ok(nodes):
            # the same seven layers and three pools as below, as ONE autograd node (saves ~0.7 ms of host time per step)
            pooled = F.family_a_graph(nodes, s, p1, p2, seed, self._graph_params())
            obs = self.obs_seq(observable)
            obs = obs.squeeze(1) if obs.shape[1] == 1 else torch.mean(obs, dim=1)      # one Pauli term: the mean is the term (a view)
            return self.body_seq(torch.cat((pooled, obs, circuit_depth, exp_value), dim=1))
        g = self.conv1(nodes, s, relu=True, drop_p=p1, seed=seed + 1, defer_mask=True)
        g = self.conv2(g, s, relu=True, drop_p=p1, seed=seed + 2, defer_mask=True, x_gate_scale=k1)
        g = F.segment_mean(self.conv3(g, s, x_gate_scale=k1), s)
        c = self.cheb_conv1(nodes, s, relu=True, drop_p=p2, seed=seed + 3, defer_mask=True)
        c = F.segment_mean(self.cheb_conv2(c, s, x_gate_scale=k2), s)
        sg = self.sage_conv1(nodes, s, relu=True, drop_p=p2, seed=seed + 4, defer_mask=True)
        sg = F.segment_mean(self.sage_conv2(sg, s, x_gate_scale=k2), s)
        obs = torch.mean(self.obs_seq(observable), dim=1)
        merged = torch.cat((g, c, sg, obs, circuit_depth, exp_value), dim=1)
        return self.body_seq(merged)
