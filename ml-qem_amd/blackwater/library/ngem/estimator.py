"""NGEM estimator: wraps any qiskit-style Estimator class so that ``job.result()`` returns expectation values
mitigated by a GNN (reference: blackwater/library/ngem/estimator.py:23-158).

Per (value, circuit, observable, parameters): transpile + bind, encode the circuit as a graph with qubit and gate
features, build the ``ExpValueEntry`` tensors (no self-loops, ``batch=None`` -- the reference's inference convention)
and call ``model(noisy_0, observable, circuit_depth, x, edge_index, batch)``; tensors are moved to the model's device.
"""
from __future__ import annotations

import weakref
from functools import wraps
from typing import Any, Callable, Optional, Type

import os

import numpy as np
import torch

from ...data.backends import is_pauli_observable
from ...data.generators.exp_val import ExpValueEntry
from ...data.graph import Batch
from ...data.utils import circuit_to_graph_data_json, encode_pauli_sum_op, get_backend_properties_v1
from ...exception import BlackwaterException
from ..primitives import job_base, make_estimator_result, model_device, transpile_and_bind




# False: the serial path encodes with the Python walk (circuit_to_graph_data_json) whatever the circuit's type -- the form the native
# encoder is compared with (tests/test_estimators.py sets it)
_NATIVE_SERIAL = True
# circuits per slice of a large batched run() (NgemJob._batched_in_slices); MLQEM_NGEM_SLICE overrides (A/B; a huge value = one batch)
_SLICE = int(os.environ.get("MLQEM_NGEM_SLICE", "512"))      # measured on 1024 100-qubit circuits: one batch 32.9 ms, 512: 29.5, 256: 33.6, 128: 37.4


# model -> train.BucketedPredictor: captured forwards outlive a result() call, not the model -- the predictor holds its model through a
# weak reference (train.BucketedPredictor.model), so the entry, its hipGraphs, their memory pool and its arena go with the model
_predictors = weakref.WeakKeyDictionary()
_encoders = {}          # content of a backend-properties dict -> NativeEncoder (at most 8, oldest dropped first)


def _encoder_for(properties):
    """The native encoder of these backend properties, kept between result() calls: the reference re-reads the backend's calibration
    on every result() (:46) and so does this path, but an encoder -- its calibration tables on the host and, for the device
    expansion, on the GPU -- is rebuilt only when the CONTENT of the properties changed (a VQE loop calls result() thousands of
    times against one calibration)."""
    import hashlib
    import json

    from ...data.native_encoder import NativeEncoder

    key = hashlib.sha256(json.dumps(properties, sort_keys=True, default=str).encode()).hexdigest()      # the content, not hash() of it
    enc = _encoders.get(key)
    if enc is None:
        while len(_encoders) >= 8:
            _encoders.pop(next(iter(_encoders)))
        enc = _encoders[key] = NativeEncoder(properties)
    return enc


def _qasm_text(bound):
    """OpenQASM-2 text of a bound circuit the native encoder can scan (text as it came, or this package's ``Circuit``); None for
    anything else (a qiskit circuit goes through the Python encoder, which walks its DAG)."""
    from ...data.circuit import Circuit, circuit_to_qasm

    if isinstance(bound, str):
        return bound
    if isinstance(bound, Circuit):
        return circuit_to_qasm(bound)
    return None


def _options_dict(options) -> dict:
    if options is None:
        return {}
    return dict(options.__dict__) if hasattr(options, "__dict__") and not isinstance(options, dict) else dict(options)


class NgemJob(job_base()):  # type: ignore[misc]
    """Delegating job whose ``result()`` post-processes the base job's values with the model."""

    def __init__(self, base_job, model, backend, circuits, observables, parameter_values, options=None,
                 batched: bool = False) -> None:  # pylint: disable=super-init-not-called
        self._batched = batched
        self._base_job = base_job
        self._model = model
        self._backend = backend
        self._circuits = circuits
        self._observables = observables
        self._parameter_values = parameter_values
        self._options = options

    def result(self):
        result = self._base_job.result()
        properties = get_backend_properties_v1(self._backend)  # recomputed per call, like the reference (:46)
        device = model_device(self._model)
        mitigated = []
        if self._batched:
            return make_estimator_result(np.array(self._result_batched_native(result, properties, device)), result.metadata)
        # (a model left in train() mode is called in train() mode by the reference's loop -- dropout on -- whatever the number of
        # circuits; the replayed path evaluates in eval mode, so it serves eval-mode models only and the plain loop takes the rest)
        if (_NATIVE_SERIAL and device is not None and torch.device(device).type == "cuda" and len(self._circuits) > 1
                and not getattr(self._model, "training", False)
                and getattr(self._model, "accepts_device_batches", False) and not getattr(self._model, "needs_size_pattern", False)):
            replayed = self._result_serial_replayed(result, properties, device)
            if replayed is not None:
                return make_estimator_result(np.array(replayed), result.metadata)
        encoder = None      # the C++ encoder, built on first need: same arrays as the Python walk, bit for bit
        encoded = {}        # an observable OBJECT that appears many times in a run() is encoded once
        for value, circuit, obs, params in zip(result.values, self._circuits, self._observables,
                                               self._parameter_values):
            if not is_pauli_observable(obs):
                raise BlackwaterException("Only `PauliSumOp` observables are supported by NGEM.")
            bound = transpile_and_bind(circuit, self._backend, params, _options_dict(self._options), keep_text=_NATIVE_SERIAL)
            text = _qasm_text(bound) if _NATIVE_SERIAL else None
            if text is not None:
                # the reference's loop, one model call per circuit (:49-84), with the encoding done natively: OpenQASM text (or
                # this package's Circuit) -> x, op->op edges in one C call instead of the dict-of-lists walk of
                # circuit_to_graph_data_json (88 ms per 100-qubit circuit in Python, ~2 ms here)
                if encoder is None:
                    from ...data.native_encoder import NativeEncoder

                    encoder = _encoder_for(properties)
                # float32 rows and int64 indices straight from the C call (the batch entry points with one circuit: no float64
                # staging array, no casts), in pinned memory when they are about to be uploaded
                on_gpu = device is not None and torch.device(device).type == "cuda"
                x, edge_index, _, _, _ = encoder.encode_batch([text], threads=1, pin=on_gpu)
                enc = encoded.get(id(obs))
                if enc is None:
                    enc = encoded[id(obs)] = torch.tensor([encode_pauli_sum_op(obs)], dtype=torch.float)
                args = [torch.tensor([[value]], dtype=torch.float), enc, torch.zeros(1, 1), x, edge_index, None]
                if device is not None:
                    args = [a if a is None else a.to(device, non_blocking=on_gpu) for a in args]
            else:
                graph = circuit_to_graph_data_json(circuit=bound, properties=properties, use_qubit_features=True,
                                                   use_gate_features=True)
                data = ExpValueEntry(circuit_graph=graph, observable=encode_pauli_sum_op(obs), ideal_exp_value=0.0,
                                     noisy_exp_values=[value]).to_pyg_data()
                if device is not None:
                    data = data.to(device)
                args = [data.noisy_0, data.observable, data.circuit_depth, data.x, data.edge_index, data.batch]
            with torch.no_grad():
                out = self._model(*args)
            mitigated.append(out.item())
        return make_estimator_result(np.array(mitigated), result.metadata)

    def _result_serial_replayed(self, result, properties, device):
        """The serial loop of the reference -- one model call per circuit, values read back one by one -- for this package's own
        models on the GPU: the run()'s circuits are scanned natively into a device-resident arena once, and every per-circuit model
        call is the replay of a forward captured per size bucket (train.BucketedPredictor) instead of ~45 launches enqueued from
        Python.  None when a circuit is not OpenQASM text / a ``Circuit`` (the caller then walks the loop as before)."""
        from ...data.arena import GraphArena
        from ...train import BucketedPredictor

        texts, values, observables, encoded = [], [], [], {}
        for value, circuit, obs, params in zip(result.values, self._circuits, self._observables, self._parameter_values):
            if not is_pauli_observable(obs):
                raise BlackwaterException("Only `PauliSumOp` observables are supported by NGEM.")
            bound = transpile_and_bind(circuit, self._backend, params, _options_dict(self._options), keep_text=True)
            text = _qasm_text(bound)
            if text is None:
                return None
            texts.append(text)
            values.append([float(value)])
            enc = encoded.get(id(obs))
            if enc is None:
                enc = encoded[id(obs)] = np.asarray(encode_pauli_sum_op(obs), dtype=np.float32)
            observables.append(enc)
        if len({o.shape for o in observables}) != 1:
            return None                       # observables of several shapes cannot share one arena: the plain loop takes them
        x, edge_index, _, counts, _ = _encoder_for(properties).encode_batch_expand(texts, device)
        n = len(texts)
        arena = GraphArena.from_device(x, counts, edge_index, np.zeros((n, 1), np.float32), np.asarray(values, np.float32),
                                       np.zeros((n, 1), np.float32), np.stack(observables), filler_nodes=256)
        predictor = _predictors.get(self._model)
        if predictor is None:
            predictor = _predictors[self._model] = BucketedPredictor(self._model, arena)
        else:
            predictor.load(arena)           # the captures of earlier run()s, over this run()'s circuits
        mitigated = []
        for i in range(n):              # one model call per circuit; a bucket's output is overwritten by its next replay, so each is kept
            mitigated.append(predictor.predict_ids([i]).reshape(-1)[:1].clone())
        return torch.cat(mitigated).tolist()        # ... and the values are read back together

    def _result_batched_native(self, result, properties, device):
        """``batched=True``: every circuit of this run() encoded by the C++ encoder on a pool of host threads straight into ONE
        collated batch (pinned when the model is on the GPU), ONE upload, ONE model call.  The reference's loop
        (:49-84) encodes in Python and calls the model per circuit; the arrays are bit-identical (tests/test_gpu_family_b.py)."""
        from ...data.circuit import Circuit, circuit_to_qasm
        from ...data.native_encoder import NativeEncoder

        texts, values, observables, encoded = [], [], [], {}
        for value, circuit, obs, params in zip(result.values, self._circuits, self._observables, self._parameter_values):
            if not is_pauli_observable(obs):
                raise BlackwaterException("Only `PauliSumOp` observables are supported by NGEM.")
            bound = transpile_and_bind(circuit, self._backend, params, _options_dict(self._options), keep_text=_NATIVE_SERIAL)
            texts.append(bound if isinstance(bound, str) else circuit_to_qasm(Circuit.from_any(bound)))
            values.append([float(value)])
            # needs observables of one shape, which a run() over one operator family has; an operator OBJECT that appears many times
            # (the usual run(): one Hamiltonian term for all circuits) is encoded once -- its 1 + 4 n floats per term as Python lists
            # for every circuit were a third of this method's host time on 100-qubit circuits
            enc = encoded.get(id(obs))
            if enc is None:
                enc = encoded[id(obs)] = np.asarray(encode_pauli_sum_op(obs), dtype=np.float32)
            observables.append(enc)
        on_gpu = device is not None and torch.device(device).type == "cuda"
        encoder = _encoder_for(properties)
        noisy = torch.tensor(values, dtype=torch.float)
        observable = torch.from_numpy(np.stack(observables)) if observables else torch.zeros((0, 0, 0))
        depth = torch.zeros(len(texts), 1)
        if on_gpu and len(texts) >= 2 * _SLICE and getattr(self._model, "accepts_device_batches", False):
            # the per-graph inputs go up once, BEFORE the loop (a copy from pageable memory queues behind whatever the stream still
            # has to do, and the host waits for it); the values are read back once, after it
            up = [t.to(device) for t in (noisy, observable, depth)]
            return self._batched_in_slices(encoder, texts, *up, device).tolist()
        if on_gpu:
            # the host scans the texts into a compact op stream (16 bytes per op); rows, edges and offsets are made on the device
            x, edge_index, batch, counts, _ = encoder.encode_batch_expand(texts, device)
        else:
            x, edge_index, batch, counts, _ = encoder.encode_batch(texts)
        args = [noisy, observable, depth, x, edge_index, batch]
        if device is not None:
            args = [a.to(device, non_blocking=on_gpu) for a in args]
        with torch.no_grad():
            out = self._model(*args)
        return out.reshape(len(texts), -1)[:, 0].tolist()

    def _batched_in_slices(self, encoder, texts, noisy, observable, depth, device):
        """A large run() as slices of ``_SLICE`` circuits: while this thread has the device expand slice k into rows and edges, builds
        its structure and enqueues the model's launches for it (3-4 ms of host time a slice), a second thread has the C++ pool scan
        slice k + 1 (the scan releases the interpreter lock: 17 ms of host time per 1024 100-qubit circuits, the largest part of a
        run()).  Nothing in here waits for the device -- the per-graph inputs are on it already (the caller's one upload), a slice's
        graph boundaries come from the scan (no count on the device), and the values are read back once, by the caller.  Per-circuit
        values do not depend on the batch a circuit is in (mean pools per graph)."""
        from concurrent.futures import ThreadPoolExecutor

        from ...native.structure import GraphStructure

        bounds = [(lo, min(lo + _SLICE, len(texts))) for lo in range(0, len(texts), _SLICE)]
        outs = []
        with ThreadPoolExecutor(max_workers=1) as scanner, torch.no_grad():
            ahead = scanner.submit(encoder.scan_to_stream, texts[bounds[0][0]:bounds[0][1]])
            for k, (lo, hi) in enumerate(bounds):
                scan = ahead.result()
                if k + 1 < len(bounds):
                    ahead = scanner.submit(encoder.scan_to_stream, texts[bounds[k + 1][0]:bounds[k + 1][1]])
                x, edge_index, batch, counts, _ = encoder.expand_stream(scan, device)
                ptr = np.zeros(hi - lo + 1, dtype=np.int32)
                np.cumsum(counts, out=ptr[1:])
                ptr_d = torch.from_numpy(ptr).pin_memory().to(device, non_blocking=True)
                struct = GraphStructure.from_edge_index(edge_index, int(x.shape[0]), graph_ptr=ptr_d)
                struct._graph_sizes = [int(v) for v in counts]
                outs.append(self._model(noisy[lo:hi], observable[lo:hi], depth[lo:hi], x, struct, batch).reshape(hi - lo, -1)[:, 0])
        return torch.cat(outs)

    def submit(self):
        return self._base_job.submit()

    def status(self):
        return self._base_job.status()

    def cancel(self):
        return self._base_job.cancel()

    def __repr__(self):
        return f"<NgemJob: {self._base_job.job_id()}>"


def patch_run(run: Callable, model, backend, options=None, batched: bool = False) -> Callable:
    """Wraps an Estimator's ``_run`` so that it returns an :class:`NgemJob`."""

    @wraps(run)
    def ngem_run(self, circuits, observables, parameter_values, **run_options):
        job = run(self, circuits=circuits, observables=observables, parameter_values=parameter_values, **run_options)
        return NgemJob(job, model=model, backend=backend, circuits=circuits, observables=observables,
                       parameter_values=parameter_values, options=options, batched=batched)

    return ngem_run


def ngem(cls: Type, model, backend, options=None, *, batched: bool = False):
    """Decorator turning an Estimator class into an NGEM estimator class ``NGEM<cls.__name__>``.

    ``batched=True`` (an addition; the default reproduces the reference exactly) post-processes all circuits of one
    ``run()`` with a single model call on the collated batch instead of one call per circuit."""
    new_class: type = type(f"NGEM{cls.__name__}", (cls,), {})
    new_class._run = patch_run(new_class._run, model, backend, options, batched)  # pylint: disable=protected-access
    return new_class
