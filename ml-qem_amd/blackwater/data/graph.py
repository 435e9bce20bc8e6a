"""Graph containers of the path: ``Data`` / ``Batch`` (host tensors), ``AddSelfLoops`` and ``DataLoader``.

These replace the torch_geometric objects the reference passes around
(blackwater/data/generators/exp_val.py:80-89 builds a ``Data``; blackwater/data/loaders/exp_val.py:33 applies
``AddSelfLoops``; docs/tutorials/__ml_models.py:105-119 batches with PyG's ``DataLoader``), keeping the
attribute names and collate rules the models rely on: every tensor attribute is concatenated on dim 0,
``edge_index`` on dim 1 with a cumulative node offset, and ``batch`` / ``ptr`` are added.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional, Sequence

import torch


class Data:
    """Attribute bag for one graph. ``batch`` is ``None`` for a single graph (what the estimators pass on)."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, **kwargs):
        self._store: Dict[str, Any] = {}
        for k, v in dict(x=x, edge_index=edge_index, edge_attr=edge_attr, y=y, **kwargs).items():
            if v is not None:
                self._store[k] = v

    def __getattr__(self, key):
        store = self.__dict__.get("_store", {})
        if key in store:
            return store[key]
        if key == "batch" or key == "ptr":
            return None
        raise AttributeError(f"{type(self).__name__} has no attribute {key!r}")

    def __setattr__(self, key, value):
        if key == "_store":
            object.__setattr__(self, key, value)
        else:
            self._store[key] = value

    def __contains__(self, key):
        return key in self._store

    def keys(self) -> List[str]:
        return list(self._store.keys())

    @property
    def num_nodes(self) -> int:
        if "x" in self._store:
            return int(self._store["x"].shape[0])
        ei = self._store.get("edge_index")
        return int(ei.max()) + 1 if ei is not None and ei.numel() else 0

    @property
    def num_edges(self) -> int:
        ei = self._store.get("edge_index")
        return int(ei.shape[1]) if ei is not None else 0

    def to(self, device, non_blocking: bool = False):
        out = type(self).__new__(type(self))
        object.__setattr__(out, "_store", {k: (v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v)
                                           for k, v in self._store.items()})
        return out

    def __repr__(self):
        parts = [f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v!r}" for k, v in self._store.items()]
        return f"{type(self).__name__}({', '.join(parts)})"


class Batch(Data):
    """Disjoint union of graphs with ``batch`` (graph id per node) and ``ptr`` (node offsets)."""

    @staticmethod
    def from_data_list(graphs: Sequence[Data]) -> "Batch":
        if len(graphs) == 0:
            raise ValueError("cannot batch an empty list of graphs")
        keys = graphs[0].keys()
        sizes = [g.num_nodes for g in graphs]
        ptr = torch.zeros(len(graphs) + 1, dtype=torch.long)
        ptr[1:] = torch.cumsum(torch.tensor(sizes, dtype=torch.long), 0)
        merged: Dict[str, Any] = {}
        for k in keys:
            vals = [g._store[k] for g in graphs]
            if not torch.is_tensor(vals[0]):
                merged[k] = vals
            elif k == "edge_index":
                merged[k] = torch.cat([v + off for v, off in zip(vals, ptr[:-1].tolist())], dim=1)
            else:
                merged[k] = torch.cat(vals, dim=0)
        merged["batch"] = torch.repeat_interleave(torch.arange(len(graphs)), torch.tensor(sizes, dtype=torch.long))
        merged["ptr"] = ptr
        out = Batch()
        object.__setattr__(out, "_store", merged)
        return out

    @property
    def num_graphs(self) -> int:
        return int(self._store["ptr"].numel()) - 1

    def __repr__(self):
        return "Data" + super().__repr__()


class AddSelfLoops:
    """Appends one (i, i) edge per node; ``edge_attr`` is left as is (the dataset default transform,
    reference: blackwater/data/loaders/exp_val.py:33; evidence for the untouched ``edge_attr``:
    docs/tutorials/01_ngem.ipynb:186 prints edge_index=[2, 866] next to edge_attr=[477, 3])."""

    def __call__(self, data: Data) -> Data:
        n = data.num_nodes
        loops = torch.arange(n, dtype=data.edge_index.dtype).unsqueeze(0).repeat(2, 1)
        data.edge_index = torch.cat([data.edge_index, loops], dim=1)
        return data

    def __repr__(self):
        return "AddSelfLoops()"


class DataLoader(torch.utils.data.DataLoader):
    """torch ``DataLoader`` whose collate builds a :class:`Batch` (same sampler/RNG behaviour as PyG's)."""

    def __init__(self, dataset, batch_size: int = 1, shuffle: bool = False, **kwargs):
        kwargs.pop("collate_fn", None)
        super().__init__(dataset, batch_size=batch_size, shuffle=shuffle,
                         collate_fn=lambda items: Batch.from_data_list(items), **kwargs)
