"""Binary dataset shards: the encoded-graph corpus as flat, memory-mappable arrays.

The reference stores datasets as ``.json`` / ``.pk`` lists of ``ExpValueEntry`` dicts -- nested Python lists per node
and per edge (``blackwater/data/generators/exp_val.py:31-89``, read back by ``loaders/exp_val.py:45-66``).  That is
fine for 10^3 four-qubit circuits and untenable for 10^6 circuits or 20 000-node graphs.
A shard holds the same content -- node features, op->op edges in the encoder's order, the per-circuit labels -- as
seven flat arrays behind a small JSON header, each array 64-byte aligned so it maps straight into ``numpy.memmap``
and from there into one host-to-device copy (``GraphArena.from_shards``).

File layout (little endian)::

    0   8 bytes   magic  b"MLQEMSH1"
    8   u64       header length H
    16  H bytes   UTF-8 JSON: {"version", "graphs", "nodes", "edges", "features", "meta",
                               "arrays": {name: {"dtype", "shape", "offset"}}}
    ... arrays, each starting on a multiple of 64 bytes:
        x          f32 [nodes, features]     node features, graphs back to back
        node_ptr   i64 [graphs + 1]          first row of every graph
        edge_ptr   i64 [graphs + 1]          first edge of every graph
        edge_index i32 [2, edges]            graph-LOCAL node ids, (source row, destination row), encoder order
        y, noisy   f32 [graphs, Y]           ideal / noisy expectation values
        depth      f32 [graphs, 1]
        observable f32 [graphs, T, P]        encoded Pauli terms (``encode_pauli_sum_op``)
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, Iterable, List, Optional, Sequence

import numpy as np

MAGIC = b"MLQEMSH1"
VERSION = 1
_ALIGN = 64
_ORDER = ("x", "node_ptr", "edge_ptr", "edge_index", "y", "noisy", "depth", "observable")
_DTYPES = {"x": "<f4", "node_ptr": "<i8", "edge_ptr": "<i8", "edge_index": "<i4", "y": "<f4", "noisy": "<f4",
           "depth": "<f4", "observable": "<f4"}


class ShardFormatError(ValueError):
    """The file is not a shard, is truncated, or its header contradicts its size."""


def _round_up(v: int, a: int = _ALIGN) -> int:
    return (v + a - 1) // a * a


class GraphShard:
    """The arrays of one shard (``numpy.memmap`` views when read from disk)."""

    def __init__(self, arrays: Dict[str, np.ndarray], meta: Optional[Dict[str, Any]] = None):
        self.arrays, self.meta = arrays, dict(meta or {})
        for name in _ORDER:
            setattr(self, name, arrays[name])
        self._validate()

    def __len__(self) -> int:
        return int(self.node_ptr.shape[0]) - 1

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    def _validate(self) -> None:
        g = len(self)
        if g < 0 or self.edge_ptr.shape[0] != g + 1:
            raise ShardFormatError("node_ptr / edge_ptr lengths disagree")
        if g >= 0 and (int(self.node_ptr[0]) != 0 or int(self.edge_ptr[0]) != 0):
            raise ShardFormatError("node_ptr / edge_ptr must start at 0")
        if int(self.node_ptr[-1]) != self.x.shape[0] or int(self.edge_ptr[-1]) != self.edge_index.shape[1]:
            raise ShardFormatError("node_ptr / edge_ptr do not cover x / edge_index")
        if np.any(np.diff(self.node_ptr) < 0) or np.any(np.diff(self.edge_ptr) < 0):
            raise ShardFormatError("node_ptr / edge_ptr must be non-decreasing")
        for name in ("y", "noisy", "depth", "observable"):
            if self.arrays[name].shape[0] != g:
                raise ShardFormatError(f"{name} has {self.arrays[name].shape[0]} rows for {g} graphs")
        if self.num_edges:
            # every edge must stay inside its graph: a bad id would become an out-of-bounds gather on the device
            sizes = np.repeat(np.diff(self.node_ptr), np.diff(self.edge_ptr))
            ei = np.asarray(self.edge_index)
            if ei.min() < 0 or np.any(ei[0] >= sizes) or np.any(ei[1] >= sizes):
                raise ShardFormatError("edge_index refers to a node outside its graph")

    def graph(self, i: int):
        """(x, edge_index, y, noisy, depth, observable) of graph ``i`` -- views, nothing is copied."""
        s, e = int(self.node_ptr[i]), int(self.node_ptr[i + 1])
        es, ee = int(self.edge_ptr[i]), int(self.edge_ptr[i + 1])
        return self.x[s:e], self.edge_index[:, es:ee], self.y[i], self.noisy[i], self.depth[i], self.observable[i]


def pack_graphs(xs: Sequence[np.ndarray], edge_indices: Sequence[np.ndarray], y, noisy, depth, observable,
                meta: Optional[Dict[str, Any]] = None) -> GraphShard:
    """Flat arrays from per-graph lists (the arguments of ``GraphArena.from_arrays``)."""
    g = len(xs)
    if len(edge_indices) != g:
        raise ValueError("xs and edge_indices differ in length")
    node_ptr = np.zeros(g + 1, dtype=np.int64)
    edge_ptr = np.zeros(g + 1, dtype=np.int64)
    np.cumsum([a.shape[0] for a in xs], out=node_ptr[1:])
    np.cumsum([np.asarray(e).shape[1] for e in edge_indices], out=edge_ptr[1:])
    feats = {a.shape[1] for a in xs}
    if len(feats) > 1:
        raise ValueError(f"graphs disagree on the feature width: {sorted(feats)}")
    f = feats.pop() if feats else 0
    x = np.concatenate(xs, axis=0).astype(np.float32) if g else np.zeros((0, f), np.float32)
    ei = (np.concatenate([np.asarray(e) for e in edge_indices], axis=1) if g else np.zeros((2, 0))).astype(np.int32)

    def lab(a, min_ndim):
        a = np.asarray(a, dtype=np.float32)
        if a.ndim < min_ndim:
            a = a.reshape((g, -1)) if min_ndim == 2 else a.reshape((g, 1, -1))
        return np.ascontiguousarray(a)

    arrays = {"x": np.ascontiguousarray(x), "node_ptr": node_ptr, "edge_ptr": edge_ptr,
              "edge_index": np.ascontiguousarray(ei), "y": lab(y, 2), "noisy": lab(noisy, 2), "depth": lab(depth, 2),
              "observable": lab(observable, 3)}
    return GraphShard(arrays, meta)


def write_shard(path: str, shard: GraphShard) -> None:
    """Writes ``shard`` to ``path`` (atomically: a temporary file is renamed into place)."""
    arrays = {k: np.ascontiguousarray(shard.arrays[k], dtype=np.dtype(_DTYPES[k])) for k in _ORDER}

    def header_for(base: int):
        entries, off = {}, base
        for k in _ORDER:
            off = _round_up(off)
            entries[k] = {"dtype": _DTYPES[k], "shape": list(arrays[k].shape), "offset": off}
            off += arrays[k].nbytes
        head = {"version": VERSION, "graphs": len(shard), "nodes": shard.num_nodes, "edges": shard.num_edges,
                "features": int(arrays["x"].shape[1]), "meta": shard.meta, "arrays": entries}
        return json.dumps(head, sort_keys=True).encode(), off

    # the offsets depend on the header's length and vice versa: reserve room, then pad the JSON with spaces
    probe, _ = header_for(0)
    room = _round_up(16 + len(probe) + 64)
    blob, total = header_for(room)
    if 16 + len(blob) > room:
        raise AssertionError("shard header outgrew its reservation")
    blob = blob + b" " * (room - 16 - len(blob))
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(MAGIC)
        fh.write(np.uint64(len(blob)).tobytes())
        fh.write(blob)
        pos = room
        head = json.loads(blob)
        for k in _ORDER:
            off = head["arrays"][k]["offset"]
            fh.write(b"\0" * (off - pos))
            fh.write(arrays[k].tobytes())
            pos = off + arrays[k].nbytes
        if pos != total:
            raise AssertionError("shard size bookkeeping is off")
    os.replace(tmp, path)


def read_shard(path: str, mmap: bool = True) -> GraphShard:
    """Maps (or, with ``mmap=False``, reads) a shard.  Raises ``ShardFormatError`` on anything inconsistent."""
    size = os.path.getsize(path)
    with open(path, "rb") as fh:
        if fh.read(8) != MAGIC:
            raise ShardFormatError(f"{path}: not an ml-qem shard (bad magic)")
        raw = fh.read(8)
        if len(raw) != 8:
            raise ShardFormatError(f"{path}: truncated header")
        hlen = int(np.frombuffer(raw, dtype="<u8")[0])
        if 16 + hlen > size:
            raise ShardFormatError(f"{path}: truncated header")
        try:
            head = json.loads(fh.read(hlen))
        except ValueError as err:
            raise ShardFormatError(f"{path}: unreadable header ({err})") from None
    if head.get("version") != VERSION:
        raise ShardFormatError(f"{path}: shard version {head.get('version')} (this reader: {VERSION})")
    arrays = {}
    for k in _ORDER:
        try:
            ent = head["arrays"][k]
            dt, shape, off = np.dtype(ent["dtype"]), tuple(int(v) for v in ent["shape"]), int(ent["offset"])
        except (KeyError, TypeError, ValueError):
            raise ShardFormatError(f"{path}: header lacks array '{k}'") from None
        if dt != np.dtype(_DTYPES[k]) or off % _ALIGN or min(shape, default=0) < 0:
            raise ShardFormatError(f"{path}: array '{k}' has dtype {dt} / offset {off}")
        nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        if off + nbytes > size:
            raise ShardFormatError(f"{path}: truncated ('{k}' needs bytes up to {off + nbytes}, file has {size})")
        if nbytes == 0:
            arrays[k] = np.zeros(shape, dtype=dt)
        elif mmap:
            arrays[k] = np.memmap(path, dtype=dt, mode="r", offset=off, shape=shape)
        else:
            with open(path, "rb") as fh:
                fh.seek(off)
                arrays[k] = np.frombuffer(fh.read(nbytes), dtype=dt).reshape(shape)
    return GraphShard(arrays, head.get("meta"))


def shard_from_dataset(entries: Iterable[Any], meta: Optional[Dict[str, Any]] = None) -> GraphShard:
    """From the host dataset's ``Data`` entries (``CircuitGraphExpValMitigationDataset``: the reference's ``.json`` /
    ``.pk`` files after ``to_pyg_data`` and the transforms)."""
    xs, eis, y, noisy, depth, obs = [], [], [], [], [], []
    for g in entries:
        xs.append(g.x.numpy())
        eis.append(g.edge_index.numpy())
        y.append(g.y.numpy().reshape(-1))
        noisy.append(g.noisy_0.numpy().reshape(-1))
        depth.append(g.circuit_depth.numpy().reshape(-1))
        o = g.observable.numpy()
        obs.append(o.reshape(o.shape[-2:]) if o.ndim >= 2 else o.reshape(1, -1))   # [T, P]; an absent observable is [1, 0]
    return pack_graphs(xs, eis, np.stack(y), np.stack(noisy), np.stack(depth), np.stack(obs), meta)


def shard_from_qasm(qasms: Sequence[str], properties: Dict[str, Any], y, noisy, observable,
                    add_self_loops: bool = True, meta: Optional[Dict[str, Any]] = None) -> GraphShard:
    """Encodes OpenQASM-2 circuits with the native encoder (``mlqem_encode_qasm``) straight into shard arrays -- the
    fast replacement of ``circuit_to_graph_data_json`` + ``json.dump`` (``generators/exp_val.py``).  ``add_self_loops``
    applies the training path's dataset transform (``loaders/exp_val.py:33``) at write time."""
    from .native_encoder import NativeEncoder

    enc = NativeEncoder(properties)
    xs: List[np.ndarray] = []
    eis: List[np.ndarray] = []
    depth = []
    for text in qasms:
        x, ei, _, d = enc.encode(text)
        if add_self_loops:
            loops = np.arange(x.shape[0], dtype=np.int64)
            ei = np.concatenate([ei, np.stack([loops, loops])], axis=1)
        xs.append(x.astype(np.float32))
        eis.append(ei)
        depth.append([float(d)])
    return pack_graphs(xs, eis, y, noisy, np.asarray(depth, np.float32), observable, meta)
