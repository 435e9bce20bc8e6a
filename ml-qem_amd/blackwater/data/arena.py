"""Device-resident circuit-graph dataset ("arena") and on-device batch assembly.

The reference collates a batch on the host every step: PyG's ``DataLoader`` concatenates ~9 attributes of 32
``Data`` objects and offsets ``edge_index`` (docs/tutorials/__ml_models.py:105-119,148).  Here the whole encoded
dataset is uploaded once: node features, both CSR structures, self-loop counts and the per-node normalisation
scalars live in HBM (a 1M-circuit corpus is a few tens of GB of the 288 GB), and a batch is gathered by one
native call from a list of graph ids -- no host-side tensor work beyond two prefix sums over B integers.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from ..native import _lib, ops
from ..native.structure import GraphStructure


class DeviceBatch:
    """One assembled batch: node features, structure and the per-graph tensors of the model protocol.

    The node features are NOT copied when the batch is assembled: ``nodes`` is ``ops.RowsOf(arena.x, src_node)`` -- the
    batch's rows of the arena -- which the first layers of the models read through the row map.  ``x`` materialises (and
    caches) the gathered [N, F] matrix for anything that wants a plain tensor."""

    def __init__(self, nodes, structure, y, noisy, depth, observable, graph_ids, num_real=None):
        self.nodes, self.structure = nodes, structure
        self.y, self.noisy_0, self.circuit_depth, self.observable = y, noisy, depth, observable
        self.graph_ids = graph_ids
        self.num_graphs = structure.num_graphs
        # a batch padded to a size bucket ends in a slice of the arena's edgeless filler graph: its output row is not a circuit
        self.num_real = self.num_graphs if num_real is None else int(num_real)
        self._x = nodes if isinstance(nodes, torch.Tensor) else None

    @property
    def x(self) -> torch.Tensor:
        if self._x is None:
            self._x = self.nodes.materialize()
        return self._x

    def model_args(self):
        """The six positional arguments of the reference's model protocol (``nodes`` may be a ``RowsOf``)."""
        nodes = self._x if self._x is not None else self.nodes
        return self.noisy_0, self.observable, self.circuit_depth, nodes, self.structure, None


def _ranges(starts, lengths):
    """Concatenation of arange(starts[i], starts[i] + lengths[i]) without a Python loop."""
    lengths = np.asarray(lengths, dtype=np.int64)
    total = int(lengths.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    first = np.cumsum(lengths) - lengths
    return np.repeat(np.asarray(starts, dtype=np.int64) - first, lengths) + np.arange(total, dtype=np.int64)


def _coarse_capacities(out_ptr, out_dst, gptr, n_total, in_ptr=None, in_src=None) -> np.ndarray:
    """Per graph, an upper bound on the edges of the graph ASAPooling coarsens it to, from the structure alone (computed once
    per arena, on the device).  Cluster p is its centre and the centre's in-neighbours; (p, q) is a coarsened edge iff some
    member u of p has u -> v or u = v for a member v of q.  A node w belongs to at most
    1 + outdeg(w) clusters (itself, if kept, and its kept out-neighbours), so whatever top-k keeps,

        #edges  <=  sum_u (1 + outdeg u) * sum_{v in N+[u] or v = u} (1 + outdeg v).

    This is also the sum over ALL nodes c of the per-row bound the list form of the coarsening places its out-rows by
    (csrc/asap.hip: cap_o(c) = sum over u in N-[c] of h_out(u)); with ``in_ptr`` / ``in_src`` the same is computed for its
    in-rows (h_in(u) = sum over v in N-[u] of (1 + outdeg v)) and the larger of the two is returned, so that one number per
    graph bounds both scratch lists and the edge arrays.

    On 100-qubit TFIM circuits this is ~100 per node against ~27 real coarsened edges per node (barrier nodes have 100
    out-edges): memory, not time -- the kernels walk the real rows.  What it buys: the coarsened edge arrays are sized on the
    host, so the one device->host read of the wave-per-cluster coarsening (the edge total) disappears and a Family B step on
    the headline graphs can be captured in a hipGraph."""
    if n_total == 0:
        return np.zeros(max(int(gptr.numel()) - 1, 0), dtype=np.int64)
    optr = out_ptr[: n_total + 1].to(torch.int64)
    outdeg = optr[1:] - optr[:-1]
    od1 = outdeg + 1
    e = int(optr[-1].item())
    g = gptr.long()

    def per_graph(ptr, idx):
        deg = ptr[1:] - ptr[:-1]
        s_u = od1.clone()
        if e > 0:
            own = torch.repeat_interleave(torch.arange(n_total, device=optr.device), deg)
            s_u.index_add_(0, own, od1[idx[:e].long()])
        c = torch.zeros(n_total + 1, dtype=torch.int64, device=optr.device)
        torch.cumsum(od1 * s_u, 0, out=c[1:])
        return c[g[1:]] - c[g[:-1]]

    caps = per_graph(optr, out_dst)
    if in_ptr is not None and in_src is not None:
        caps = torch.maximum(caps, per_graph(in_ptr[: n_total + 1].to(torch.int64), in_src))
    return caps.cpu().numpy()


class GraphArena:
    def __init__(self, x, node_counts, structure_arrays, nscal, y, noisy, depth, observable, edge_counts, ell=None,
                 filler_nodes=0, coarse_caps=None):
        self.x = x
        # per graph: a structural upper bound on the number of edges ASAPooling's coarsened graph can have (see
        # _coarse_capacities): lets a batch size its pooled edge arrays without reading the count back from the device
        self.coarse_caps = None if coarse_caps is None else np.asarray(coarse_caps, dtype=np.int64)
        # the LAST graph is an edgeless, all-zero filler of `filler_nodes` nodes when filler_nodes > 0: batches padded to a
        # size bucket (batch(..., bucket=)) end in a slice of it; it is not part of len(arena)
        self.filler_nodes = int(filler_nodes)
        self.node_counts = np.asarray(node_counts, dtype=np.int64)
        self.edge_counts = np.asarray(edge_counts, dtype=np.int64)
        self.gptr, self.in_ptr, self.in_src, self.out_ptr, self.out_dst, self.loops, self.out_eid = structure_arrays
        n_total = int(self.node_counts.sum())
        # ELL side tables of the whole arena: a batch's tables are these rows, rebased (no per-batch CSR walk)
        if ell is None:
            ell = (ops.ell_from_csr(self.in_ptr, self.in_src, n_total), ops.ell_from_csr(self.out_ptr, self.out_dst, n_total))
        self.in_ell, self.out_ell = ell
        self.nscal = nscal  # [N,6]: gcn_dinv, sage_rinv, cheb_dinv, then the column sums P^T 1 of the three convs (structure.colsum)
        self.y, self.noisy, self.depth, self.observable = y, noisy, depth, observable
        self.device = x.device

    def __len__(self):
        return len(self.node_counts) - (1 if self.filler_nodes else 0)

    @property
    def num_nodes(self):
        return int(self.node_counts.sum()) - self.filler_nodes

    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def from_arrays(xs: Sequence[np.ndarray], edge_indices: Sequence[np.ndarray], y, noisy, depth, observable,
                    device="cuda", filler_nodes: int = 0) -> "GraphArena":
        """xs[g]: [n_g,F] float; edge_indices[g]: [2,e_g] graph-local int (self-loops allowed, e.g. after
        ``AddSelfLoops``); y/noisy/depth/observable: per-graph arrays with leading dimension G."""
        node_counts = np.array([a.shape[0] for a in xs], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(node_counts)])
        x_host = np.ascontiguousarray(np.concatenate(xs, axis=0), dtype=np.float32)
        ei = np.concatenate([np.asarray(e, dtype=np.int64) + o for e, o in zip(edge_indices, offs[:-1])], axis=1)
        return GraphArena._from_flat(x_host, node_counts, ei, y, noisy, depth, observable, device, filler_nodes)

    @staticmethod
    def from_shards(shards, device="cuda", rank: int = 0, world: int = 1) -> "GraphArena":
        """From binary shards (``data/shards.py``: paths or ``GraphShard`` objects).  With ``world > 1`` rank ``r``
        keeps the graphs ``r, r + world, ...`` of the concatenated corpus (the data-parallel split by circuit)."""
        from .shards import GraphShard, read_shard

        parts = [s if isinstance(s, GraphShard) else read_shard(s) for s in shards]
        if not parts:
            raise ValueError("from_shards: no shards given")
        xs, eis, counts, labels = [], [], [], {k: [] for k in ("y", "noisy", "depth", "observable")}
        first = 0
        for sh in parts:
            keep = np.arange(len(sh))
            keep = keep[(first + keep) % world == rank] if world > 1 else keep
            first += len(sh)
            n_of = np.diff(sh.node_ptr)[keep]
            e_of = np.diff(sh.edge_ptr)[keep]
            if world > 1:   # gather the kept graphs' rows / edges
                rows = _ranges(np.asarray(sh.node_ptr)[keep], n_of)
                cols = _ranges(np.asarray(sh.edge_ptr)[keep], e_of)
                xs.append(np.asarray(sh.x)[rows])
                eis.append(np.asarray(sh.edge_index)[:, cols].astype(np.int64))
            else:
                xs.append(np.asarray(sh.x))
                eis.append(np.asarray(sh.edge_index).astype(np.int64))
            counts.append((n_of, e_of))
            for k in labels:
                labels[k].append(np.asarray(sh.arrays[k])[keep])
        node_counts = np.concatenate([c[0] for c in counts]).astype(np.int64)
        edge_counts = np.concatenate([c[1] for c in counts]).astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(node_counts)])[:-1]
        ei = np.concatenate(eis, axis=1) + np.repeat(offs, edge_counts)[None, :]   # graph-local -> arena-global ids
        lab = {k: np.concatenate(v, axis=0) for k, v in labels.items()}
        return GraphArena._from_flat(np.ascontiguousarray(np.concatenate(xs, axis=0), dtype=np.float32), node_counts,
                                     ei, lab["y"], lab["noisy"], lab["depth"], lab["observable"], device)

    @staticmethod
    def _from_flat(x_host, node_counts, ei, y, noisy, depth, observable, device, filler_nodes=0) -> "GraphArena":
        """x_host [N,F] f32 (graphs back to back), ei [2,E] int64 with arena-global node ids."""
        f = x_host.shape[1]
        f4 = (f + 3) // 4 * 4                                    # rows padded to a multiple of 4 floats, pads zero
        x = torch.zeros((x_host.shape[0], f4), dtype=torch.float32, device=torch.device(device))[:, :f]
        x.copy_(torch.from_numpy(x_host))
        ei_dev = torch.from_numpy(np.ascontiguousarray(ei)).to(device)
        return GraphArena.from_device(x, node_counts, ei_dev, y, noisy, depth, observable, filler_nodes)

    @staticmethod
    def from_device(x: torch.Tensor, node_counts, ei_dev: torch.Tensor, y, noisy, depth, observable,
                    filler_nodes: int = 0) -> "GraphArena":
        """From tensors already on the device: ``x`` [N,F] fp32 in the padded row layout (row stride a multiple of 4
        floats, pad columns zero), graphs back to back; ``ei_dev`` [2,E] int64 with arena-global node ids; labels as
        host arrays with leading dimension G.  ``filler_nodes`` > 0 appends the edgeless all-zero filler graph that
        bucket-padded batches draw from."""
        device = x.device
        node_counts = np.asarray(node_counts, dtype=np.int64)
        if filler_nodes > 0:
            f = x.shape[1]
            f4 = (f + 3) // 4 * 4
            grown = torch.zeros((x.shape[0] + filler_nodes, f4), dtype=torch.float32, device=device)[:, :f]
            grown[:x.shape[0]].copy_(x)
            x = grown
            node_counts = np.concatenate([node_counts, [filler_nodes]])
            zrow = lambda a: np.concatenate([np.asarray(a), np.zeros((1,) + np.asarray(a).shape[1:], dtype=np.asarray(a).dtype)])
            y, noisy, depth, observable = zrow(y), zrow(noisy), zrow(depth), zrow(observable)
        offs = np.concatenate([[0], np.cumsum(node_counts)])
        n_total = int(offs[-1])
        if x.shape[0] != n_total or x.dtype != torch.float32 or (n_total > 1 and (x.stride(0) % 4 or x.stride(1) != 1)):
            raise ValueError("GraphArena.from_device: x must be [sum(node_counts), F] fp32 with rows padded to 4 floats")
        csr = ops.csr_build(ei_dev, n_total)
        in_ptr, in_src, out_ptr, out_dst, loops = csr
        gcn, sage, cheb = ops.graph_norms(in_ptr, out_ptr, loops, n_total)
        gptr = torch.from_numpy(offs.astype(np.int32)).to(device)
        # column sums of the three propagation matrices: structural, so computed once for the whole arena
        ell = (ops.ell_from_csr(in_ptr, in_src, n_total), ops.ell_from_csr(out_ptr, out_dst, n_total))
        whole = GraphStructure(n_total, in_ptr, in_src, out_ptr, out_dst, loops, gptr, len(node_counts), norms=(gcn, sage, cheb),
                               ell=ell)
        nscal = torch.stack([gcn, sage, cheb, whole.colsum("gcn"), whole.colsum("sage"), whole.colsum("cheb")], dim=1).contiguous()
        del whole
        # edges per graph (self-loops excluded): one read-back at build time
        edge_counts = np.diff(in_ptr[gptr.long()].cpu().numpy()).astype(np.int64)
        t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a), dtype=dt).to(device)
        return GraphArena(x, node_counts, (gptr, in_ptr, in_src, out_ptr, out_dst, loops, csr.out_eid), nscal, t(y), t(noisy),
                          t(depth), t(observable), edge_counts, ell=ell, filler_nodes=filler_nodes,
                          coarse_caps=_coarse_capacities(out_ptr, out_dst, gptr, n_total, in_ptr, in_src))

    @staticmethod
    def from_data_list(graphs, device="cuda") -> "GraphArena":
        """From the host dataset's ``Data`` entries (CircuitGraphExpValMitigationDataset)."""
        xs = [g.x.numpy() for g in graphs]
        eis = [g.edge_index.numpy() for g in graphs]
        y = np.stack([g.y.numpy().reshape(-1) for g in graphs])
        noisy = np.stack([g.noisy_0.numpy().reshape(-1) for g in graphs])
        depth = np.stack([g.circuit_depth.numpy().reshape(-1) for g in graphs])
        obs = np.stack([g.observable.numpy()[0] for g in graphs])
        return GraphArena.from_arrays(xs, eis, y, noisy, depth, obs, device=device)

    # ------------------------------------------------------------------------------------------------
    _DEVICE_ARRAYS = ("x", "nscal", "gptr", "in_ptr", "in_src", "out_ptr", "out_dst", "out_eid", "loops", "in_ell", "out_ell",
                      "y", "noisy", "depth", "observable")

    def with_capacity(self, factor: float = 2.0) -> "GraphArena":
        """A copy of this arena whose device arrays are the leading parts of allocations ``factor`` times their size: ``refill_from``
        then puts ANOTHER arena's content at the same addresses, which is what a hipGraph captured over ``assemble`` needs to be
        replayed for the circuits of a later run() (train.BucketedPredictor)."""
        import copy

        grown = copy.copy(self)
        grown._base = {}
        for name in self._DEVICE_ARRAYS:
            t = getattr(self, name)
            rows = max(int(t.shape[0] * factor), t.shape[0] + 1)
            if name == "x":
                f4 = t.stride(0) if t.shape[0] > 1 else (t.shape[1] + 3) // 4 * 4
                base = torch.zeros((rows, f4), dtype=t.dtype, device=t.device)
                base[:t.shape[0], :t.shape[1]].copy_(t)
            else:
                base = torch.zeros((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                base[:t.shape[0]].copy_(t)
            grown._base[name] = base
            setattr(grown, name, base[:t.shape[0], :t.shape[1]] if name == "x" else base[:t.shape[0]])
        return grown

    def refill_from(self, other: "GraphArena") -> bool:
        """Takes over ``other``'s graphs in place (an arena made by ``with_capacity``): False -- and nothing changed -- when an array
        of ``other`` is longer than its allocation here or of another row shape / type."""
        base = getattr(self, "_base", None)
        if base is None or other.filler_nodes != self.filler_nodes:
            return False
        for name in self._DEVICE_ARRAYS:
            t, b = getattr(other, name), base[name]
            trailing = (t.shape[1],) if name == "x" else tuple(t.shape[1:])
            mine = (getattr(self, name).shape[1],) if name == "x" else tuple(b.shape[1:])
            if t.shape[0] > b.shape[0] or trailing != mine or t.dtype != b.dtype:
                return False
        for name in self._DEVICE_ARRAYS:
            t, b = getattr(other, name), base[name]
            view = b[:t.shape[0], :t.shape[1]] if name == "x" else b[:t.shape[0]]
            view.copy_(t)
            setattr(self, name, view)
        self.node_counts, self.edge_counts, self.coarse_caps = other.node_counts, other.edge_counts, other.coarse_caps
        return True

    def selection(self, graph_ids, bucket=None, filler_sizes=None):
        """Host side of a batch: (sel, nptr, eptr, Nb, Eb, number of real graphs).  With ``bucket = (n_pad, e_pad)`` the
        batch is padded to exactly n_pad nodes by a slice of the filler graph (appended as one more, edgeless, graph) and
        Eb = e_pad is a capacity: every kernel of a step then launches with the same shapes for all selections that fit the
        bucket -- what a captured hipGraph needs.  ``filler_sizes``: pad with SEVERAL edgeless graphs of these sizes (slices of
        the same filler; they must add up to n_pad minus the selection's nodes) instead of one: train.stable_padding sizes them
        so that the node totals after each ASAPooling are functions of the bucket too."""
        sel = np.asarray(graph_ids, dtype=np.int64)
        b = int(sel.shape[0])
        if b == 0:
            raise ValueError("empty batch")
        if sel.min() < 0 or sel.max() >= len(self):
            raise IndexError("graph id out of range")
        n_of, e_of = self.node_counts[sel], self.edge_counts[sel]
        if bucket is not None:
            n_pad, e_pad = int(bucket[0]), int(bucket[1])
            nb, eb = int(n_of.sum()), int(e_of.sum())
            if not self.filler_nodes:
                raise ValueError("bucketed batches need an arena built with filler_nodes > 0")
            if nb > n_pad or eb > e_pad or (filler_sizes is None and n_pad - nb > self.filler_nodes):
                raise ValueError(f"selection ({nb} nodes, {eb} edges) does not fit bucket {bucket} (filler {self.filler_nodes})")
            fill = np.asarray([n_pad - nb] if filler_sizes is None else filler_sizes, dtype=np.int64)
            if int(fill.sum()) != n_pad - nb or (filler_sizes is not None and (fill.min() < 1 or fill.max() > self.filler_nodes)):
                raise ValueError(f"filler sizes {fill.tolist()} do not pad {nb} nodes to {n_pad}")
            sel = np.concatenate([sel, np.full(len(fill), len(self), dtype=np.int64)])
            n_of, e_of = np.concatenate([n_of, fill]), np.concatenate([e_of, np.zeros(len(fill), dtype=np.int64)])
        nptr = np.zeros(len(sel) + 1, dtype=np.int64)
        eptr = np.zeros(len(sel) + 1, dtype=np.int64)
        np.cumsum(n_of, out=nptr[1:])
        np.cumsum(e_of, out=eptr[1:])
        nb, eb = int(nptr[-1]), int(eptr[-1])
        if bucket is not None:
            eb = int(bucket[1])
        return sel, nptr, eptr, nb, eb, b

    def coarse_capacity(self, sel) -> Optional[int]:
        """Upper bound on the coarsened edges of a batch of the graphs ``sel`` (None when unknown or beyond int32 indexing)."""
        if self.coarse_caps is None:
            return None
        cap = int(self.coarse_caps[np.asarray(sel, dtype=np.int64)].sum())
        return cap if cap < (1 << 30) else None

    def batch(self, graph_ids, bucket=None) -> DeviceBatch:
        """Gathers the graphs ``graph_ids`` (host ints, any order, repeats allowed) into one batch on the device."""
        sel, nptr, eptr, nb, eb, b_real = self.selection(graph_ids, bucket)
        packed = torch.from_numpy(np.concatenate([sel, nptr, eptr]).astype(np.int32)).to(self.device, non_blocking=True)
        return self.assemble(packed, len(sel), nb, eb, self.node_counts[sel] if bucket is None else nptr[1:] - nptr[:-1],
                             sel, b_real, coarse_capacity=self.coarse_capacity(sel))

    def assemble(self, packed: torch.Tensor, b: int, nb: int, eb: int, graph_sizes, sel_host=None, num_real=None,
                 coarse_capacity=None, pool_plan=None) -> DeviceBatch:
        """Device side of a batch: ``packed`` = [sel (b) | nptr (b + 1) | eptr (b + 1)] int32 on the device.  Nothing
        here reads a host value other than the shapes, so with a persistent ``packed`` buffer the whole call can sit
        inside a captured hipGraph and be replayed for another selection of the same bucket."""
        sel_d, nptr_d, eptr_d = packed[:b], packed[b:2 * b + 1], packed[2 * b + 1:3 * b + 2]
        dev, f = self.device, self.x.shape[1]
        f4 = (f + 3) // 4 * 4
        k = int(self.nscal.shape[1])
        nscal_b = torch.empty((k, max(nb, 1)), dtype=torch.float32, device=dev)   # planar: one contiguous vector per scalar
        derived_b = torch.empty((3, max(nb, 1)), dtype=torch.float32, device=dev)
        mk = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        in_ptr, out_ptr, loops, src_node = mk(nb + 1), mk(nb + 1), mk(nb), mk(nb)
        in_src, out_dst, out_eid = mk(eb), mk(eb), mk(eb)
        in_ell = torch.empty((max(nb, 1), 2), dtype=torch.int32, device=dev)
        out_ell = torch.empty((max(nb, 1), 2), dtype=torch.int32, device=dev)
        p = ops._p
        code = _lib.load().mlqem_batch_assemble(
            p(self.x), self.x.stride(0), f4, p(self.nscal), k, p(self.gptr), p(self.in_ptr), p(self.in_src),
            p(self.out_ptr), p(self.out_dst), p(self.out_eid), p(self.loops), p(self.in_ell), p(self.out_ell), p(sel_d),
            p(nptr_d), p(eptr_d), b, nb, eb, None, 0, p(nscal_b), p(derived_b), p(src_node), p(in_ptr), p(in_src),
            p(out_ptr), p(out_dst), p(out_eid), p(loops), p(in_ell), p(out_ell), ops._stream())
        _lib.check(code, "mlqem_batch_assemble")
        norms = (nscal_b[0, :nb], nscal_b[1, :nb], nscal_b[2, :nb])
        s = GraphStructure(nb, in_ptr, in_src, out_ptr, out_dst, loops, nptr_d, b, num_edges=eb, norms=norms,
                           graph_sizes=graph_sizes, out_eid=out_eid, ell=(in_ell, out_ell),
                           colsums=(nscal_b[3, :nb], nscal_b[4, :nb], nscal_b[5, :nb]),
                           derived={"gcn_dself": derived_b[0, :nb], "sage_dself": derived_b[1, :nb], "cheb_neg": derived_b[2, :nb]})
        s.coarse_capacity = coarse_capacity
        s.pool_plan = pool_plan           # size-stable batch: graph_sizes is None and the poolings go by this plan
        s.num_real = num_real
        nodes = ops.RowsOf(self.x, src_node[:nb])     # the feature rows stay in the arena
        # the per-graph inputs in ONE launch (five torch gathers before: a captured step of 32 small circuits is ~80 launches of ~5 us)
        labels = (self.y, self.noisy, self.depth, self.observable)
        if all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] > 0 and t[0].numel() > 0 for t in labels):
            y, noisy, depth, observable = ops.gather_rows(labels, sel_d)
        else:
            idx = sel_d.to(torch.int64)
            y, noisy, depth, observable = self.y[idx], self.noisy[idx], self.depth[idx], self.observable[idx]
        return DeviceBatch(nodes, s, y, noisy, depth, observable, sel_host, num_real)
