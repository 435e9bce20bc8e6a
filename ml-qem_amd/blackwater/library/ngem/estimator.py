"""NGEM estimator: wraps any qiskit-style Estimator class so that ``job.result()`` returns expectation values
mitigated by a GNN (reference: blackwater/library/ngem/estimator.py:23-158).

Per (value, circuit, observable, parameters): transpile + bind, encode the circuit as a graph with qubit and gate
features, build the ``ExpValueEntry`` tensors (no self-loops, ``batch=None`` -- the reference's inference convention)
and call ``model(noisy_0, observable, circuit_depth, x, edge_index, batch)``; tensors are moved to the model's device.
"""
from __future__ import annotations

from functools import wraps
from typing import Any, Callable, Optional, Type

import numpy as np
import torch

from ...data.backends import is_pauli_observable
from ...data.generators.exp_val import ExpValueEntry
from ...data.graph import Batch
from ...data.utils import circuit_to_graph_data_json, encode_pauli_sum_op, get_backend_properties_v1
from ...exception import BlackwaterException
from ..primitives import job_base, make_estimator_result, model_device, transpile_and_bind


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
        entries = []
        native = None
        if self._batched:  # the fast mode also encodes with the C++ encoder (bit-identical arrays, ~10x faster)
            from ...data.circuit import Circuit, circuit_to_qasm
            from ...data.graph import Data
            from ...data.native_encoder import NativeEncoder

            native = NativeEncoder(properties)
        for value, circuit, obs, params in zip(result.values, self._circuits, self._observables,
                                               self._parameter_values):
            if not is_pauli_observable(obs):
                raise BlackwaterException("Only `PauliSumOp` observables are supported by NGEM.")
            bound = transpile_and_bind(circuit, self._backend, params, _options_dict(self._options))
            if native is not None:
                text = bound if isinstance(bound, str) else circuit_to_qasm(Circuit.from_any(bound))
                x, ei, ea, _ = native.encode(text)
                entries.append(Data(x=torch.from_numpy(x).float(), edge_index=torch.from_numpy(ei),
                                    edge_attr=torch.from_numpy(ea).float(), y=torch.zeros(1, 1),
                                    observable=torch.tensor([encode_pauli_sum_op(obs)], dtype=torch.float),
                                    circuit_depth=torch.zeros(1, 1), noisy_0=torch.tensor([[value]], dtype=torch.float)))
                continue
            graph = circuit_to_graph_data_json(circuit=bound, properties=properties, use_qubit_features=True,
                                               use_gate_features=True)
            data = ExpValueEntry(circuit_graph=graph, observable=encode_pauli_sum_op(obs), ideal_exp_value=0.0,
                                 noisy_exp_values=[value]).to_pyg_data()
            if self._batched:
                entries.append(data)
                continue
            if device is not None:
                data = data.to(device)
            with torch.no_grad():
                out = self._model(data.noisy_0, data.observable, data.circuit_depth, data.x, data.edge_index,
                                  data.batch)
            mitigated.append(out.item())
        if self._batched and entries:
            # one collate + ONE model call for every circuit of this run() (the reference loops circuit by circuit,
            # :49-84); needs observables of one shape, which a run() over one operator family has
            batch = Batch.from_data_list(entries)
            if device is not None:
                batch = batch.to(device)
            with torch.no_grad():
                out = self._model(batch.noisy_0, batch.observable, batch.circuit_depth, batch.x, batch.edge_index,
                                  batch.batch)
            mitigated = out.reshape(len(entries), -1)[:, 0].tolist()
        return make_estimator_result(np.array(mitigated), result.metadata)

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
