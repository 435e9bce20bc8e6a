"""Device-resident connectivity of one batch of graphs, in the layout the kernels consume.

``GraphStructure`` is what a conv needs instead of PyG's raw ``edge_index``: CSR by destination (forward),
CSR by source (backward), self-loop counts, graph boundaries, and the per-node normalisation scalars.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


_CSR_FIELDS = ("in_ptr", "in_src", "out_ptr", "out_dst", "loops", "num_edges", "out_eid")


class GraphStructure:
    def __init__(self, num_nodes, in_ptr, in_src, out_ptr, out_dst, loops, graph_ptr, num_graphs, num_edges=None,
                 norms=None, graph_sizes=None, out_eid=None, ell=None, colsums=None, derived=None):
        self.num_nodes = int(num_nodes)
        self.in_ptr, self.in_src, self.out_ptr, self.out_dst, self.loops = in_ptr, in_src, out_ptr, out_dst, loops
        self.graph_ptr, self.num_graphs = graph_ptr, int(num_graphs)
        self.num_edges = num_edges  # non-self-loop edges, host int when known without a sync
        self.out_eid = out_eid      # in-CSR position of every out-CSR entry (edge-softmax backward)
        self._init_rest(norms, graph_sizes, ell, colsums, derived)

    @classmethod
    def deferred(cls, num_nodes, graph_ptr, num_graphs, build, graph_sizes=None) -> "GraphStructure":
        """A structure whose CONNECTIVITY is built on first use: ``build()`` returns (in_ptr, in_src, out_ptr, out_dst, loops,
        num_edges, out_eid) and runs when one of those attributes is first read.  Graph boundaries and sizes are there at
        once.  ASAPooling hands its coarsened graph back this way: every model of the reference follows its second pooling
        with ``global_mean_pool`` (docs/tutorials/gnn.py:112-114), which reads the boundaries only, so the second coarsening
        -- the most expensive kernel of a step on 100-qubit circuits -- is never computed unless someone looks at it."""
        self = cls.__new__(cls)
        self.num_nodes = int(num_nodes)
        self.graph_ptr, self.num_graphs = graph_ptr, int(num_graphs)
        self._build = build
        self._init_rest(None, graph_sizes, None, None, None)
        return self

    def __getattr__(self, name):
        # only reached for attributes that are not set: the connectivity of a deferred structure
        if name in _CSR_FIELDS and "_build" in self.__dict__:
            build = self.__dict__.pop("_build")
            for key, val in zip(_CSR_FIELDS, build()):
                self.__dict__[key] = val
            return self.__dict__[name]
        raise AttributeError(name)

    # ------------------------------------------------------------------------------------------------
    def set_block_order(self, make):
        """Marks the structure as one whose rows share their sources when taken in a certain order (ASAPooling's coarsened graphs
        of large circuits, rows by the program position of their centres): the edge walks over it then take its long rows as
        dense blocks (csrc/dense_block.hpp).  ``make()`` -> (order, max_span) runs when the first plan is built (a structure no
        layer reads never pays for it); ``max_span``: a bound on the id range of one block's entries."""
        self._block_order = make
        self._dense_plans = {}

    @property
    def blocked(self) -> bool:
        return self._block_order is not None

    def dense_plan(self, direction: str):
        """The dense blocks of the in- ("in") or out-structure ("out") (csrc/dense_block.hpp), built on first use; None for a
        structure without a block order (its rows are short, or nothing is known about their order)."""
        if self._block_order is None:
            return None
        if direction not in self._dense_plans:
            if callable(self._block_order):
                self._block_order = self._block_order()
            order, max_span = self._block_order
            ptr, idx = (self.in_ptr, self.in_src) if direction == "in" else (self.out_ptr, self.out_dst)
            self._dense_plans[direction] = ops.dense_plan_build(ptr, idx, self.loops, self.num_nodes, self.graph_ptr, self.num_graphs,
                                                                order, max_span)
        return self._dense_plans[direction]

    @property
    def connectivity_built(self) -> bool:
        return "_build" not in self.__dict__

    def _init_rest(self, norms, graph_sizes, ell, colsums, derived):
        self.coarse_capacity = None   # host-side upper bound on the edges of this structure's ASAPooling coarsening (data/arena.py)
        # Size-stable batches (train.BucketedTrainer): what the poolings that follow may know of the batch WITHOUT its per-graph sizes
        # -- one (k_total, nmax, kmax) per pooling level: the exact number of kept nodes (the batch's filler graphs are sized to
        # make it a function of the bucket) and bounds on the largest graph before / after.  None: sizes come from graph_sizes.
        self.pool_plan = None
        self.num_real = None          # graphs of the batch that are circuits (the rest: edgeless fillers at the end)
        self._block_order = None      # (order, max_span): what the dense-block plans are built from (set_block_order)
        self._dense_plans = {}
        self._norms = norms
        self._derived = {} if derived is None else dict(derived)
        self._colsum = {} if colsums is None else dict(zip(("gcn", "sage", "cheb"), colsums))
        self._ell = {} if ell is None else {"in": ell[0], "out": ell[1]}   # side tables, built on demand otherwise
        self._graph_sizes = None if graph_sizes is None else [int(v) for v in graph_sizes]

    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def from_edge_index(edge_index: torch.Tensor, num_nodes: int, batch: Optional[torch.Tensor] = None,
                        num_graphs: Optional[int] = None, graph_ptr: Optional[torch.Tensor] = None) -> "GraphStructure":
        """Builds the structure on the device from a [2,E] int64 edge list (the reference's model protocol)."""
        csr = ops.csr_build(edge_index, num_nodes)
        in_ptr, in_src, out_ptr, out_dst, loops = csr
        out_eid = csr.out_eid
        dev = edge_index.device
        if graph_ptr is None:
            if batch is None:
                graph_ptr = torch.tensor([0, num_nodes], dtype=torch.int32, device=dev)
                num_graphs = 1
                return GraphStructure(num_nodes, in_ptr, in_src, out_ptr, out_dst, loops, graph_ptr, 1,
                                      graph_sizes=[num_nodes], out_eid=out_eid)
            else:
                if num_graphs is None:
                    raise ValueError("num_graphs is required with `batch` (avoids a device sync)")
                counts = torch.bincount(batch, minlength=num_graphs)
                graph_ptr = torch.zeros(num_graphs + 1, dtype=torch.int32, device=dev)
                graph_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        else:
            graph_ptr = graph_ptr.to(device=dev, dtype=torch.int32)
            num_graphs = graph_ptr.numel() - 1
        return GraphStructure(num_nodes, in_ptr, in_src, out_ptr, out_dst, loops, graph_ptr, num_graphs, out_eid=out_eid)

    @property
    def graph_sizes(self):
        """Host-side node count of every graph (pooling needs it to size its output); read back once if the
        structure was built from a device-side ``batch`` vector."""
        if self._graph_sizes is None:
            ptr = self.graph_ptr.cpu().tolist()
            self._graph_sizes = [b - a for a, b in zip(ptr[:-1], ptr[1:])]
        return self._graph_sizes

    # ------------------------------------------------------------------------------------------------
    def _base_norms(self):
        if self._norms is None:
            self._norms = ops.graph_norms(self.in_ptr, self.out_ptr, self.loops, self.num_nodes)
        return self._norms

    @property
    def gcn_dinv(self):
        return self._base_norms()[0]

    @property
    def sage_rinv(self):
        return self._base_norms()[1]

    @property
    def cheb_dinv(self):
        return self._base_norms()[2]

    def edge_count(self) -> int:
        """Number of stored (non-self-loop) edges; one device read if the structure was built without it."""
        if self.num_edges is None:
            self.num_edges = int(self.in_ptr[self.num_nodes].item())
        return self.num_edges

    @property
    def in_ell(self):
        """First two sources of every destination row (fast path of the forward aggregation)."""
        if "in" not in self._ell:
            self._ell["in"] = ops.ell_from_csr(self.in_ptr, self.in_src, self.num_nodes)
        return self._ell["in"]

    @property
    def out_ell(self):
        """First two destinations of every source row (fast path of the backward aggregation)."""
        if "out" not in self._ell:
            self._ell["out"] = ops.ell_from_csr(self.out_ptr, self.out_dst, self.num_nodes)
        return self._ell["out"]

    def colsum(self, kind):
        """t = P^T 1, the column sums of a conv's propagation matrix P (``gcn``: D^-1/2 (A+I) D^-1/2, ``sage``: the in-edge
        mean incl. listed self-loops, ``cheb``: L^ = -D^-1/2 A D^-1/2) -- one structural scalar per node.  A conv that
        feeds a mean pool directly collapses onto it: mean_pool(P (h W^T)) = wmean_t(h) W^T (csrc/pool.hip).  Batches of
        a :class:`GraphArena` carry these from the arena (graphs are disjoint, so a node's column sum does not depend
        on the batch it is in); otherwise they are one transposed aggregation of the ones vector, cached."""
        if kind not in self._colsum:
            ones = torch.ones((max(self.num_nodes, 1), 1), dtype=torch.float32, device=self.in_ptr.device)[:self.num_nodes]
            if kind == "gcn":
                kw = dict(cscale=self.gcn_dinv, rscale=self.gcn_dinv, dself=self.derived("gcn_dself"))
            elif kind == "sage":
                kw = dict(cscale=self.sage_rinv, dself=self.derived("sage_dself"))
            elif kind == "cheb":
                kw = dict(cscale=self.derived("cheb_neg"), rscale=self.cheb_dinv)
            else:
                raise KeyError(kind)
            out = torch.empty_like(ones)
            ops.csr_aggregate(ones, self.out_ptr, self.out_dst, ell=self.out_ell, out=out, **kw)
            self._colsum[kind] = out[:, 0]
        return self._colsum[kind]

    def derived(self, key):
        """Per-node scalars derived from the base norms, cached per structure (tiny elementwise torch ops)."""
        if key not in self._derived:
            if key == "gcn_dself":      # weight of the (i,i) term of D^-1/2 (A+I) D^-1/2
                v = self.gcn_dinv * self.gcn_dinv
            elif key == "sage_dself":   # self-loops of the raw edge list take part in the mean
                v = self.loops.to(torch.float32) * self.sage_rinv
            elif key == "cheb_neg":     # row factor of L^ = -D^-1/2 A D^-1/2
                v = -self.cheb_dinv
            else:
                raise KeyError(key)
            self._derived[key] = v
        return self._derived[key]
