"""ctypes front-end of the native (C++) circuit encoder ``mlqem_encode_qasm`` -- the fast path behind
``circuit_to_graph_data_json`` for the arrays the models actually consume (x, op->op edges, edge_attr, depth)."""
from __future__ import annotations

import ctypes
import os
from typing import Any, Dict, Tuple

import numpy as np

from ..native import _lib


class _Props(ctypes.Structure):
    _fields_ = [("num_qubits", ctypes.c_int), ("t1", ctypes.POINTER(ctypes.c_double)),
                ("t2", ctypes.POINTER(ctypes.c_double)), ("readout", ctypes.POINTER(ctypes.c_double)),
                ("num_gate_types", ctypes.c_int), ("gate_names", ctypes.POINTER(ctypes.c_char_p)),
                ("num_gate_props", ctypes.c_int), ("gate_keys", ctypes.POINTER(ctypes.c_char_p)),
                ("gate_error", ctypes.POINTER(ctypes.c_double)), ("gate_length", ctypes.POINTER(ctypes.c_double))]


class NativeEncoder:
    """Holds the C view of one ``properties`` dict (as returned by ``get_backend_properties_v1``)."""

    def __init__(self, properties: Dict[str, Any]):
        qp = properties["qubits_props"]
        n = len(qp)
        get = lambda i: qp[i] if i in qp else qp[str(i)]
        dbl = lambda vals: np.ascontiguousarray(vals, dtype=np.float64)
        self._t1, self._t2 = dbl([get(i)["t1"] for i in range(n)]), dbl([get(i)["t2"] for i in range(n)])
        self._ro = dbl([get(i)["readout_error"] for i in range(n)])
        names = [g.encode() for g in properties["gates_set"]]
        keys = list(properties["gate_props"].keys())
        self._ge = dbl([properties["gate_props"][k].get("gate_error", 0.0) for k in keys])
        self._gl = dbl([properties["gate_props"][k].get("gate_length", 0.0) for k in keys])
        self._names = (ctypes.c_char_p * len(names))(*names)
        self._keys = (ctypes.c_char_p * max(len(keys), 1))(*[k.encode() for k in keys])
        dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        self._props = _Props(n, dp(self._t1), dp(self._t2), dp(self._ro), len(names), self._names, len(keys),
                             self._keys, dp(self._ge), dp(self._gl))
        self._lib = _lib.load()

    def encode(self, qasm: str, use_gate_features: bool = True, use_qubit_features: bool = True, edge_attr: bool = True):
        """-> (x [N,F] f64, edge_index [2,E] int64, edge_attr [E,3] f64 (None with ``edge_attr=False``), depth)."""
        lib, text = self._lib, qasm.encode()
        f, d = ctypes.c_int(0), ctypes.c_int(0)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        n_feat = 3 + self._props.num_gate_types + 2 + (9 if use_qubit_features else 0) + (2 if use_gate_features else 0)
        # ONE parse: buffers sized from the text (a statement per ';', a qubit argument per '['); only a circuit that
        # broadcasts over whole registers can need more, and then the call reports the sizes it wants
        cap_n, cap_e = text.count(b";"), text.count(b"[")
        for _ in range(2):
            n, e = ctypes.c_int64(cap_n), ctypes.c_int64(cap_e)
            x = np.empty((cap_n, n_feat), dtype=np.float64)
            src, dst = np.empty(cap_e, dtype=np.int32), np.empty(cap_e, dtype=np.int32)
            attr = np.empty((cap_e, 3), dtype=np.float64) if edge_attr else None
            code = lib.mlqem_encode_qasm(text, ctypes.byref(self._props), int(use_qubit_features), int(use_gate_features),
                                         ctypes.byref(n), ctypes.byref(e), ctypes.byref(f), ctypes.byref(d), vp(x), vp(src), vp(dst),
                                         vp(attr) if edge_attr else None)
            if code != _lib.ERR_WORKSPACE:
                break
            cap_n, cap_e = max(n.value, cap_n), max(e.value, cap_e)
        if code != 0:
            self._raise(code)
        assert f.value == n_feat
        x, src, dst = x[:n.value], src[:e.value], dst[:e.value]
        return x, np.stack([src, dst]).astype(np.int64), (attr[:e.value] if edge_attr else None), d.value

    def encode_many(self, texts, threads: int = 0, **kwargs):
        """``encode`` of every text, on a pool of host threads (the C call releases the GIL; results in input order)."""
        import os
        from concurrent.futures import ThreadPoolExecutor

        threads = threads or min(16, os.cpu_count() or 1, max(len(texts), 1))
        if threads <= 1 or len(texts) <= 1:
            return [self.encode(t, **kwargs) for t in texts]
        with ThreadPoolExecutor(max_workers=threads) as pool:
            return list(pool.map(lambda t: self.encode(t, **kwargs), texts))

    def encode_batch(self, texts, threads: int = 0, pin: bool = False, use_gate_features: bool = True,
                     use_qubit_features: bool = True):
        """The collated batch the models consume, straight from the texts of one ``run()`` (``mlqem_qasm_batch_parse`` /
        ``_fill``: every circuit scanned on a pool of host threads, rows written once, as float32, at their place in the
        batch): ``x`` [sum N, F] float32, ``edge_index`` [2, sum E] int64 with node offsets applied, ``batch`` [sum N] int64,
        per-circuit node counts and depths.  Equal to ``Batch.from_data_list`` of the per-circuit encodings (no self-loops:
        blackwater/library/ngem/estimator.py:61-66 encodes at inference without the training transform).  ``pin``: page-locked
        outputs (one DMA to the GPU)."""
        import torch

        lib, count = self._lib, len(texts)
        raw = [t if isinstance(t, bytes) else t.encode() for t in texts]
        arr = (ctypes.c_char_p * max(count, 1))(*raw)
        node_ptr, edge_ptr = np.zeros(count + 1, dtype=np.int64), np.zeros(count + 1, dtype=np.int64)
        depths = np.zeros(max(count, 1), dtype=np.int32)
        handle, f, failed = ctypes.c_void_p(None), ctypes.c_int(0), ctypes.c_int64(-1)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        code = lib.mlqem_qasm_batch_parse(arr, count, ctypes.byref(self._props), int(use_qubit_features), int(use_gate_features),
                                          int(threads), ctypes.byref(handle), vp(node_ptr), vp(edge_ptr), vp(depths), ctypes.byref(f),
                                          ctypes.byref(failed))
        if code != 0:
            self._raise(code)
        try:
            n, e = int(node_ptr[-1]), int(edge_ptr[-1])
            x = torch.empty((n, f.value), dtype=torch.float32, pin_memory=pin)
            ei = torch.empty((2, e), dtype=torch.int64, pin_memory=pin)
            batch = torch.empty(n, dtype=torch.int64, pin_memory=pin)
            code = lib.mlqem_qasm_batch_fill(handle, int(threads), x.data_ptr(), ei[0].data_ptr(), ei[1].data_ptr(), batch.data_ptr())
            if code != 0:
                self._raise(code)
        finally:
            lib.mlqem_qasm_batch_free(handle)
        return x, ei, batch, np.diff(node_ptr), depths[:count].tolist()

    def encode_batch_to_device(self, texts, device, threads: int = 0, chunks: int = 4, use_gate_features: bool = True,
                               use_qubit_features: bool = True, group_bytes: int = 32 << 20):
        """``encode_batch`` with the collated batch left ON THE DEVICE, the upload overlapped with the encoding: the texts are parsed
        in ``chunks`` contiguous groups (sizes first), then every group is filled into its own pinned buffers and copied to its slice
        of the device tensors without blocking, so group k travels over PCIe while group k + 1 is being written (a 1024-circuit run()
        of 100-qubit circuits is 1.3 GB of rows and indices: the copy is a third of its host time).  Node offsets of a group's
        ``edge_index`` and graph numbers of its ``batch`` vector are shifted on the device.  Same arrays as ``encode_batch``."""
        import torch

        lib, count = self._lib, len(texts)
        # groups only pay when a group's copy is long against the fixed cost of a group (a parse call, three pinned buffers): about
        # 32 MB of text (~0.15 GB of rows) per group; a run() of small circuits is one group
        k = max(1, min(int(chunks), count, sum(len(t) for t in texts) // max(int(group_bytes), 1))) if count else 1
        bounds = [count * i // k for i in range(k + 1)]
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        groups, f = [], ctypes.c_int(0)
        try:
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                raw = [t.encode() for t in texts[lo:hi]]
                n_txt = hi - lo
                arr = (ctypes.c_char_p * max(n_txt, 1))(*raw)
                node_ptr, edge_ptr = np.zeros(n_txt + 1, dtype=np.int64), np.zeros(n_txt + 1, dtype=np.int64)
                depths = np.zeros(max(n_txt, 1), dtype=np.int32)
                handle, failed = ctypes.c_void_p(None), ctypes.c_int64(-1)
                code = lib.mlqem_qasm_batch_parse(arr, n_txt, ctypes.byref(self._props), int(use_qubit_features), int(use_gate_features),
                                                  int(threads), ctypes.byref(handle), vp(node_ptr), vp(edge_ptr), vp(depths), ctypes.byref(f),
                                                  ctypes.byref(failed))
                groups.append([handle, node_ptr, edge_ptr, depths[:n_txt], raw])
                if code != 0:
                    self._raise(code, first=lo)
            n = sum(int(g[1][-1]) for g in groups)
            e = sum(int(g[2][-1]) for g in groups)
            x = torch.empty((n, f.value), dtype=torch.float32, device=device)
            ei = torch.empty((2, e), dtype=torch.int64, device=device)
            batch = torch.empty(n, dtype=torch.int64, device=device)
            n0 = e0 = g0 = 0
            keep = []
            for g in groups:
                handle, node_ptr, edge_ptr = g[0], g[1], g[2]
                gn, ge = int(node_ptr[-1]), int(edge_ptr[-1])
                xs = torch.empty((gn, f.value), dtype=torch.float32, pin_memory=True)
                es = torch.empty((2, ge), dtype=torch.int64, pin_memory=True)
                bs = torch.empty(gn, dtype=torch.int64, pin_memory=True)
                code = lib.mlqem_qasm_batch_fill(handle, int(threads), xs.data_ptr(), es[0].data_ptr(), es[1].data_ptr(), bs.data_ptr())
                if code != 0:
                    self._raise(code)
                x[n0:n0 + gn].copy_(xs, non_blocking=True)
                ei[:, e0:e0 + ge].copy_(es, non_blocking=True)
                batch[n0:n0 + gn].copy_(bs, non_blocking=True)
                if n0:
                    ei[:, e0:e0 + ge] += n0
                if g0:
                    batch[n0:n0 + gn] += g0
                keep.append((xs, es, bs))
                n0, e0, g0 = n0 + gn, e0 + ge, g0 + len(node_ptr) - 1
        finally:
            for g in groups:
                if g[0]:
                    lib.mlqem_qasm_batch_free(g[0])
        counts = np.concatenate([np.diff(g[1]) for g in groups]) if groups else np.zeros(0, dtype=np.int64)
        depths = [int(d) for g in groups for d in g[3]]
        return x, ei, batch, counts, depths

    # ------------------------------------------------------------------------------------------------ device-side expansion
    def _device_tables(self, device):
        """The calibration tables the expansion kernel indexes, on ``device`` (built once per encoder and device): t1 / t2 / readout /
        gate_error / gate_length as float32 (the rounding the rows get), g1 / g2 = mlqem_props_gate_tables."""
        import torch

        key = str(device)
        tabs = getattr(self, "_tables", None)
        if tabs is None:
            tabs = self._tables = {}
        if key not in tabs:
            nq, slots = int(self._props.num_qubits), int(self._props.num_gate_types) + 2
            g1 = np.empty(max(slots * nq, 1), dtype=np.int32)
            g2 = np.empty(max(slots * nq * nq, 1), dtype=np.int32)
            code = self._lib.mlqem_props_gate_tables(ctypes.byref(self._props), g1.ctypes.data_as(ctypes.c_void_p), g2.ctypes.data_as(ctypes.c_void_p))
            if code != 0:
                self._raise(code)
            f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32) if len(a) else np.zeros(1, dtype=np.float32)).to(device)
            tabs[key] = dict(t1=f32(self._t1), t2=f32(self._t2), ro=f32(self._ro), ge=f32(self._ge), gl=f32(self._gl),
                             g1=torch.from_numpy(g1).to(device), g2=torch.from_numpy(g2).to(device), nq=nq, slots=slots)
        return tabs[key]

    @staticmethod
    def _text_pointers(texts):
        """char* of every text WITHOUT copying it: a ``str`` hands out its cached UTF-8 buffer (for ASCII text -- OpenQASM -- that is
        the string's own storage), ``bytes`` its buffer.  The caller keeps ``texts`` alive while the pointers are in use."""
        api = ctypes.pythonapi.PyUnicode_AsUTF8AndSize
        api.restype, api.argtypes = ctypes.c_void_p, [ctypes.py_object, ctypes.c_void_p]
        arr = (ctypes.c_void_p * max(len(texts), 1))()
        keep = []
        for i, t in enumerate(texts):
            if isinstance(t, str):
                ptr = api(t, None)
                if not ptr:
                    raise ValueError(f"circuit {i}: text cannot be read as UTF-8")
                arr[i] = ptr
            else:
                b = bytes(t)
                keep.append(b)
                arr[i] = ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p).value
        return arr, keep

    def encode_batch_expand(self, texts, device, threads: int = 0, use_gate_features: bool = True, use_qubit_features: bool = True):
        """``encode_batch_to_device`` with the rows, edges and offsets made ON THE DEVICE (round 4): the host scans the texts
        (``mlqem_qasm_batch_parse``) and writes a compact op stream -- 16 bytes per op, 2 per qubit argument
        (``mlqem_qasm_batch_stream_fill``) -- into pinned memory; ONE upload (0.2 GB for 1024 100-qubit circuits instead of 1.3 GB
        of float32 rows and int64 indices); ``mlqem_encode_expand`` builds ``x`` [sum N, F] float32, ``edge_index`` [2, sum E] int64
        (the reference's edge order) and ``batch`` [sum N] int64.  Same arrays as ``encode_batch``, bit for bit.  The two halves are
        callable on their own -- ``scan_to_stream`` (host only) and ``expand_stream`` (device only) -- so that a caller can scan the
        next slice of a run() on another thread while this one is expanded and evaluated (library/ngem/estimator.py)."""
        return self.expand_stream(self.scan_to_stream(texts, threads, use_gate_features, use_qubit_features), device)

    def scan_to_stream(self, texts, threads: int = 0, use_gate_features: bool = True, use_qubit_features: bool = True):
        """The host half of ``encode_batch_expand``: the op stream of ``texts`` in one pinned staging buffer (no device call)."""
        import time

        import torch

        trace = [] if os.environ.get("MLQEM_ENCODE_TRACE") else None       # phase times of this call, printed by expand_stream
        mark = (lambda name: trace.append((name, time.perf_counter()))) if trace is not None else (lambda name: None)
        mark("start")
        lib, count = self._lib, len(texts)
        arr, keep = self._text_pointers(texts)
        mark("pointers")
        node_ptr, edge_ptr = np.zeros(count + 1, dtype=np.int64), np.zeros(count + 1, dtype=np.int64)
        wire_ptr, patch_ptr = np.zeros(count + 1, dtype=np.int64), np.zeros(count + 1, dtype=np.int64)
        depths = np.zeros(max(count, 1), dtype=np.int32)
        handle, f, failed, widest = ctypes.c_void_p(None), ctypes.c_int(0), ctypes.c_int64(-1), ctypes.c_int(0)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        code = lib.mlqem_qasm_batch_parse(arr, count, ctypes.byref(self._props), int(use_qubit_features), int(use_gate_features),
                                          int(threads), ctypes.byref(handle), vp(node_ptr), vp(edge_ptr), vp(depths), ctypes.byref(f),
                                          ctypes.byref(failed))
        if code != 0:
            self._raise(code)
        mark("parse")
        try:
            code = lib.mlqem_qasm_batch_stream_sizes(handle, vp(wire_ptr), vp(patch_ptr), ctypes.byref(widest))
            if code != 0:
                self._raise(code)
            n, e, w, p = int(node_ptr[-1]), int(edge_ptr[-1]), int(wire_ptr[-1]), int(patch_ptr[-1])
            # ONE pinned staging buffer: [ops 16 n | patches 12 p | wires 2 w | node_ptr 8 (count + 1)], every part 16-byte aligned
            up16 = lambda v: (v + 15) // 16 * 16
            o_ops, o_pat = 0, up16(16 * n)
            o_wir = o_pat + up16(12 * p)
            o_ptr = o_wir + up16(2 * w)
            total = o_ptr + 8 * (count + 1)
            mark("sizes")
            stage = torch.empty(max(total, 16), dtype=torch.uint8, pin_memory=True)
            mark("pinned buffer")
            base = stage.data_ptr()
            code = lib.mlqem_qasm_batch_stream_fill(handle, int(threads), vp(wire_ptr), vp(patch_ptr), base + o_ops, base + o_wir, base + o_pat)
            if code != 0:
                self._raise(code)
        finally:
            lib.mlqem_qasm_batch_free(handle)
        mark("stream fill")
        del keep
        stage[o_ptr:o_ptr + 8 * (count + 1)].view(torch.int64).copy_(torch.from_numpy(node_ptr))
        return dict(stage=stage, offsets=(o_ops, o_pat, o_wir, o_ptr), sizes=(n, e, w, p), count=count, f=f.value, widest=int(widest.value),
                    node_ptr=node_ptr, depths=depths, use=(int(use_qubit_features), int(use_gate_features)), trace=trace)

    def expand_stream(self, scan, device):
        """The device half: one upload of the staging buffer, ``mlqem_encode_expand`` -> (x, edge_index, batch, node counts, depths)."""
        import time

        import torch

        lib = self._lib
        stage, (o_ops, o_pat, o_wir, o_ptr), (n, e, w, p), count = scan["stage"], scan["offsets"], scan["sizes"], scan["count"]
        trace = scan["trace"]
        mark = (lambda name: trace.append((name, time.perf_counter()))) if trace is not None else (lambda name: None)
        dev_stage = stage.to(device, non_blocking=True)
        mark("upload enqueued")
        tabs = self._device_tables(device)
        x = torch.empty((n, scan["f"]), dtype=torch.float32, device=device)
        ei = torch.empty((2, e), dtype=torch.int64, device=device)
        batch = torch.empty(n, dtype=torch.int64, device=device)
        need = lib.mlqem_encode_expand_workspace_bytes(n, w)
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=device)
        mark("device buffers")
        d = dev_stage.data_ptr()
        stream = torch.cuda.current_stream(device).cuda_stream
        code = lib.mlqem_encode_expand(d + o_ops, d + o_wir, d + o_pat, p, d + o_ptr, n, w, e, count, scan["widest"],
                                       tabs["t1"].data_ptr(), tabs["t2"].data_ptr(), tabs["ro"].data_ptr(), tabs["nq"],
                                       tabs["g1"].data_ptr(), tabs["g2"].data_ptr(), tabs["ge"].data_ptr(), tabs["gl"].data_ptr(),
                                       tabs["slots"], scan["use"][0], scan["use"][1], x.data_ptr(), scan["f"],
                                       ei[0].data_ptr() if e else None, ei[1].data_ptr() if e else None, batch.data_ptr(), ws.data_ptr(), need,
                                       stream)
        _lib.check(code, "mlqem_encode_expand")
        # the staging buffers must outlive the asynchronous copy and kernels: record them on the stream
        dev_stage.record_stream(torch.cuda.current_stream(device))
        if trace is not None:
            mark("expand enqueued")
            torch.cuda.synchronize(device)
            mark("device done")
            print("encode_batch_expand: " + ", ".join(f"{b[0]} {1e3 * (b[1] - a[1]):.1f} ms" for a, b in zip(trace[:-1], trace[1:])), flush=True)
        return x, ei, batch, np.diff(scan["node_ptr"]), scan["depths"][:count].tolist()

    def _raise(self, code, first: int = 0):
        msg = self._lib.mlqem_encode_last_error().decode()
        if first:       # a group of a larger run(): the batch entry points number circuits from the group's first one
            import re
            msg = re.sub(r"^circuit (\d+):", lambda m: f"circuit {int(m.group(1)) + first}:", msg)
        if "not in the backend's gates_set" in msg:
            raise KeyError(msg)
        raise Exception(msg or f"mlqem_encode_qasm failed with code {code}")


def circuit_features(qasm: str, gate_names, bin_edges) -> Tuple[np.ndarray, np.ndarray]:
    """(gate counts [len(gate_names)], rotation-angle histogram [len(bin_edges) - 1]) of one OpenQASM-2 circuit through
    ``mlqem_circuit_features_qasm`` -- the per-circuit part of ``encode_data`` / ``encode_data_v2_ecr``."""
    lib = _lib.load()
    names = [g.encode() for g in gate_names]
    c_names = (ctypes.c_char_p * max(len(names), 1))(*names)
    edges = np.ascontiguousarray(bin_edges, dtype=np.float64)
    counts = np.zeros(len(names), dtype=np.int64)
    hist = np.zeros(max(len(edges) - 1, 0), dtype=np.int64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    code = lib.mlqem_circuit_features_qasm(qasm.encode(), c_names, len(names), vp(edges), len(edges), vp(counts), vp(hist))
    if code != 0:
        raise Exception(lib.mlqem_encode_last_error().decode() or f"mlqem_circuit_features_qasm failed with code {code}")
    return counts, hist


def circuit_features_batch(texts, gate_names, bin_edges, threads: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """``circuit_features`` of every text of one ``run()`` on a pool of host threads (``mlqem_circuit_features_qasm_batch``):
    (gate counts [n, len(gate_names)], angle histograms [n, len(bin_edges) - 1])."""
    lib = _lib.load()
    n = len(texts)
    names = [g.encode() for g in gate_names]
    c_names = (ctypes.c_char_p * max(len(names), 1))(*names)
    # the texts' own buffers (no 0.24 GB of ``str.encode`` copies per 1024 100-qubit circuits: that was 60 % of a run()), and every
    # DISTINCT buffer scanned once (a run() names one bound circuit once per Pauli term: learning/estimator.py process_batch)
    all_ptrs, keep = NativeEncoder._text_pointers(texts)
    first, inverse, order = {}, np.empty(n, dtype=np.int64), []
    for i in range(n):
        ptr = all_ptrs[i]
        j = first.get(ptr)
        if j is None:
            j = first[ptr] = len(order)
            order.append(ptr)
        inverse[i] = j
    m = len(order)
    arr = (ctypes.c_void_p * max(m, 1))(*order)
    edges = np.ascontiguousarray(bin_edges, dtype=np.float64)
    counts = np.zeros((m, len(names)), dtype=np.int64)
    hist = np.zeros((m, max(len(edges) - 1, 0)), dtype=np.int64)
    failed = ctypes.c_int64(-1)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    code = lib.mlqem_circuit_features_qasm_batch(arr, m, c_names, len(names), vp(edges), len(edges), int(threads), vp(counts), vp(hist),
                                                 ctypes.byref(failed))
    del keep
    if code != 0:
        msg = lib.mlqem_encode_last_error().decode() or f"mlqem_circuit_features_qasm_batch failed with code {code}"
        if failed.value >= 0:          # name the circuit by its position in the run(), not among the distinct buffers
            import re
            pos = int(np.nonzero(inverse == failed.value)[0][0])
            msg = re.sub(r"^circuit \d+:", f"circuit {pos}:", msg)
        raise Exception(msg)
    return (counts, hist) if m == n else (counts[inverse], hist[inverse])
