"""Third-party (PyG) message-passing semantics restated with plain torch ops (oracle; test infra only).

Call sites in the reference: docs/tutorials/gnn.py:51-65,80-92,104-114 (TransformerConv, ASAPooling,
global_mean_pool), docs/tutorials/01_ngem.ipynb cell [9] (GCNConv, ChebConv, SAGEConv),
blackwater/data/loaders/exp_val.py:33 (AddSelfLoops).  Formulas: SURVEY.md appendix B.  Conventions:
``src = edge_index[0]``, ``dst = edge_index[1]``, messages flow src -> dst, every ``Linear`` is
``y = x @ W.T + b`` with ``W: [out, in]`` exactly as the reference checkpoints store it.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import Tensor, nn


# ------------------------------------------------------------------------------------- scatter helpers
def scatter_sum(src: Tensor, index: Tensor, n: int) -> Tensor:
    out = src.new_zeros((n,) + tuple(src.shape[1:]))
    return out.index_add_(0, index, src)


def scatter_max(src: Tensor, index: Tensor, n: int) -> Tensor:
    """Segment max; rows that receive nothing stay 0 (torch_scatter's fill for empty segments)."""
    out = src.new_zeros((n,) + tuple(src.shape[1:]))
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    return out.scatter_reduce(0, idx, src, reduce="amax", include_self=False)


def segment_softmax(score: Tensor, index: Tensor, n: int) -> Tensor:
    """exp(s - max_seg) / (sum_seg + 1e-16), per destination segment."""
    smax = scatter_max(score.detach(), index, n)
    ex = (score - smax[index]).exp()
    denom = scatter_sum(ex, index, n) + 1e-16
    return ex / denom[index]


def add_self_loops(edge_index: Tensor, n: int) -> Tensor:
    loops = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, loops.unsqueeze(0).repeat(2, 1)], dim=1)


def add_remaining_self_loops(edge_index: Tensor, n: int) -> Tensor:
    """Drops existing self-loops, then appends (i, i) for every node."""
    keep = edge_index[0] != edge_index[1]
    return add_self_loops(edge_index[:, keep], n)


def global_mean_pool(x: Tensor, batch: Optional[Tensor], num_graphs: Optional[int] = None) -> Tensor:
    if batch is None:
        return x.mean(dim=0, keepdim=True)
    b = int(batch.max()) + 1 if num_graphs is None else num_graphs
    total = scatter_sum(x, batch, b)
    count = scatter_sum(torch.ones_like(batch, dtype=x.dtype), batch, b).clamp(min=1)
    return total / count.unsqueeze(-1)


def _linear(in_f: int, out_f: int, bias: bool = True) -> nn.Linear:
    return nn.Linear(in_f, out_f, bias=bias)


# --------------------------------------------------------------------------------------- TransformerConv
class TransformerConv(nn.Module):
    """heads=H, concat=True, beta=False, edge_dim=None, root_weight=True."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, dropout: float = 0.0):
        super().__init__()
        self.heads, self.out_channels, self.dropout = heads, out_channels, dropout
        self.lin_key = _linear(in_channels, heads * out_channels)
        self.lin_query = _linear(in_channels, heads * out_channels)
        self.lin_value = _linear(in_channels, heads * out_channels)
        self.lin_skip = _linear(in_channels, heads * out_channels)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        n, h, c = x.shape[0], self.heads, self.out_channels
        src, dst = edge_index[0], edge_index[1]
        q = self.lin_query(x).view(n, h, c)
        k = self.lin_key(x).view(n, h, c)
        v = self.lin_value(x).view(n, h, c)
        alpha = (q[dst] * k[src]).sum(-1) / math.sqrt(c)          # [E, H]
        alpha = segment_softmax(alpha, dst, n)
        alpha = nn.functional.dropout(alpha, p=self.dropout, training=self.training)
        out = scatter_sum(v[src] * alpha.unsqueeze(-1), dst, n).reshape(n, h * c)
        return out + self.lin_skip(x)


# --------------------------------------------------------------------------------------------- ASAPooling
class LEConv(nn.Module):
    """out_i = sum_{j->i} (lin1(x_j) - lin2(x_i)) + lin3(x_i); lin2 has no bias."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.lin1 = _linear(in_channels, out_channels)
        self.lin2 = _linear(in_channels, out_channels, bias=False)
        self.lin3 = _linear(in_channels, out_channels)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        src, dst = edge_index[0], edge_index[1]
        a, b = self.lin1(x), self.lin2(x)
        return scatter_sum(a[src] - b[dst], dst, x.shape[0]) + self.lin3(x)


def topk_per_graph(fitness: Tensor, ratio: float, batch: Tensor) -> Tensor:
    """Indices of the ceil(ratio*n_g) largest entries per graph, graphs in order, descending inside a graph
    (ties: lower index first)."""
    out = []
    num_graphs = int(batch.max()) + 1 if batch.numel() else 0
    for g in range(num_graphs):
        idx = (batch == g).nonzero().view(-1)
        k = int(math.ceil(ratio * idx.numel()))
        order = torch.sort(fitness[idx], descending=True, stable=True).indices[:k]
        out.append(idx[order])
    return torch.cat(out) if out else batch.new_zeros(0)


class ASAPooling(nn.Module):
    """ratio-pooling with learned soft cluster assignment (GNN=None, dropout=0,
    negative_slope=0.2, add_self_loops=False)."""

    def __init__(self, in_channels: int, ratio: float = 0.5, negative_slope: float = 0.2):
        super().__init__()
        self.ratio, self.negative_slope = ratio, negative_slope
        self.lin = _linear(in_channels, in_channels)
        self.att = _linear(2 * in_channels, 1)
        self.gnn_score = LEConv(in_channels, 1)

    def forward(self, x: Tensor, edge_index: Tensor, batch: Optional[Tensor] = None):
        n = x.shape[0]
        edge_index = add_remaining_self_loops(edge_index, n)
        if batch is None:
            batch = edge_index.new_zeros(n)
        src, dst = edge_index[0], edge_index[1]

        x_q = self.lin(scatter_max(x[src], dst, n))[dst]
        score = self.att(torch.cat([x_q, x[src]], dim=-1)).view(-1)
        score = nn.functional.leaky_relu(score, self.negative_slope)
        score = segment_softmax(score, dst, n)
        x_new = scatter_sum(x[src] * score.view(-1, 1), dst, n)

        fitness = self.gnn_score(x_new, edge_index).sigmoid().view(-1)
        perm = topk_per_graph(fitness, self.ratio, batch)
        x_out = x_new[perm] * fitness[perm].view(-1, 1)

        # coarsened connectivity: pattern of S^T A S restricted to the kept clusters, diagonal removed.
        # All soft assignments are > 0, so only the structure matters (the models discard the weights).
        k = perm.numel()
        slot = torch.full((n,), -1, dtype=torch.long)
        slot[perm] = torch.arange(k)
        member_cluster = slot[dst]                      # edge (u -> c): u belongs to cluster slot[c]
        valid = member_cluster >= 0
        mem_u, mem_c = src[valid], member_cluster[valid]
        # (p, v) : some member u of cluster p has A[u, v] != 0
        by_u = [[] for _ in range(n)]
        for u, v in zip(src.tolist(), dst.tolist()):
            by_u[u].append(v)
        clusters_of = [[] for _ in range(n)]
        for u, c in zip(mem_u.tolist(), mem_c.tolist()):
            clusters_of[u].append(c)
        pairs = set()
        for u in range(n):
            if not clusters_of[u]:
                continue
            for v in by_u[u]:
                for q in clusters_of[v]:
                    for p in clusters_of[u]:
                        if p != q:
                            pairs.add((p, q))
        pairs = sorted(pairs)
        new_ei = torch.tensor(pairs, dtype=torch.long).t().reshape(2, -1) if pairs else torch.zeros((2, 0), dtype=torch.long)
        return x_out, new_ei, None, batch[perm], perm


# ---------------------------------------------------------------------------------------------- Family A
class GCNConv(nn.Module):
    """X' = D^-1/2 (A + I) D^-1/2 X W^T + b, degree by destination."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.lin = _linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.xavier_uniform_(self.lin.weight)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        n = x.shape[0]
        ei = add_remaining_self_loops(edge_index, n)
        src, dst = ei[0], ei[1]
        deg = scatter_sum(torch.ones(ei.shape[1], dtype=x.dtype), dst, n)
        dinv = deg.pow(-0.5)
        dinv = torch.where(torch.isinf(dinv), torch.zeros_like(dinv), dinv)
        norm = dinv[src] * dinv[dst]
        h = self.lin(x)
        return scatter_sum(h[src] * norm.unsqueeze(-1), dst, n) + self.bias


class SAGEConv(nn.Module):
    """lin_l(mean_{j->i} x_j) + lin_r(x_i); lin_r has no bias."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.lin_l = _linear(in_channels, out_channels)
        self.lin_r = _linear(in_channels, out_channels, bias=False)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        n = x.shape[0]
        src, dst = edge_index[0], edge_index[1]
        total = scatter_sum(x[src], dst, n)
        count = scatter_sum(torch.ones(src.numel(), dtype=x.dtype), dst, n).clamp(min=1)
        return self.lin_l(total / count.unsqueeze(-1)) + self.lin_r(x)


class ChebConv(nn.Module):
    """Chebyshev filter, sym normalisation, lambda_max = 2: self-loops removed, degree by
    SOURCE, L^ = -D^-1/2 A D^-1/2 (net-zero diagonal), T_0 = x, T_1 = L^ x, T_k = 2 L^ T_{k-1} - T_{k-2}."""

    def __init__(self, in_channels: int, out_channels: int, K: int):
        super().__init__()
        self.lins = nn.ModuleList([_linear(in_channels, out_channels, bias=False) for _ in range(K)])
        self.bias = nn.Parameter(torch.zeros(out_channels))
        for lin in self.lins:
            nn.init.xavier_uniform_(lin.weight)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        n = x.shape[0]
        keep = edge_index[0] != edge_index[1]
        src, dst = edge_index[0][keep], edge_index[1][keep]
        deg = scatter_sum(torch.ones(src.numel(), dtype=x.dtype), src, n)
        dinv = deg.pow(-0.5)
        dinv = torch.where(torch.isinf(dinv), torch.zeros_like(dinv), dinv)
        w = -dinv[src] * dinv[dst]

        def lap(t):
            return scatter_sum(t[src] * w.unsqueeze(-1), dst, n)

        tx0 = x
        out = self.lins[0](tx0)
        if len(self.lins) > 1:
            tx1 = lap(x)
            out = out + self.lins[1](tx1)
            for lin in self.lins[2:]:
                tx2 = 2.0 * lap(tx1) - tx0
                out = out + lin(tx2)
                tx0, tx1 = tx1, tx2
        return out + self.bias
