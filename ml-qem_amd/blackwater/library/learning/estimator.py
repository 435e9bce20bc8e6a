"""Learning-based estimator: wraps any qiskit-style Estimator class so that ``job.result()`` returns values
post-processed by a pluggable processor (reference: blackwater/library/learning/estimator.py:22-30,151-328).

In scope: the decorator, the job wrapper, the processor protocol, ``TorchLearningModelProcessor``,
``EmptyProcessor`` and ``ScikitLearningModelProcessor`` (the random-forest / OLS baselines: same feature rows, the
model stays on the host CPU by design).  ``ZNEProcessor`` (runs extra noisy circuits
through a ZNE estimator; no model) is out of scope.
"""
from __future__ import annotations

from functools import wraps
from typing import Any, Callable, Optional, Type

import numpy as np
import torch

from ...data.backends import is_pauli_observable
from ...data.utils import encode_pauli_sum_op, get_backend_properties_v1
from ...exception import BlackwaterException
from ..primitives import job_base, make_estimator_result, model_device, transpile_and_bind
from .features import encode_data


class LearningMethodEstimatorProcessor:
    """Protocol: ``process(expectation_value, circuits, observables, parameter_values) -> float | ndarray``, called
    once per (value, bound circuit, observable, parameters) tuple."""

    def process(self, expectation_value, circuits, observables, parameter_values):
        raise NotImplementedError


class TorchLearningModelProcessor(LearningMethodEstimatorProcessor):
    """Feeds ``encode_data`` rows to a torch model, one Pauli term at a time, and sums ``output * coeff``.

    As in the reference (:170-187) the WHOLE-observable expectation value is the noisy input of every term, so this is
    meant for single-Pauli observables (``separate_observables=True`` in the VQE drivers)."""

    accepts_qasm_text = True     # process_batch scans OpenQASM text natively: PostProcessedJob hands text over unparsed

    def __init__(self, model: torch.nn.Module, backend):
        self._model = model
        self._backend = backend
        self._properties = get_backend_properties_v1(backend)

    def process(self, expectation_value, circuits, observables, parameter_values):
        device = model_device(self._model)
        results = []
        for term in observables:
            coeff, label = term.coeffs, str(term.paulis[0])
            # OpenQASM text takes the C++ op scan (mlqem_circuit_features_qasm): same row, no Python circuit objects
            model_input, _ = encode_data(circuits=[circuits], properties=self._properties, ideal_exp_vals=[[0.0]],
                                         noisy_exp_vals=[[expectation_value]], num_qubits=1,
                                         meas_bases=encode_pauli_sum_op([(label, 1.0)]), native=isinstance(circuits, str))
            if device is not None:
                model_input = model_input.to(device)
            with torch.no_grad():
                output = self._model(model_input).item()
            results.append(output * coeff[0])
        return np.sum(results)

    def process_batch(self, expectation_values, circuits, observables, parameter_values):
        """All (circuit, Pauli term) rows of one ``run()`` through ONE model call (an addition: ``PostProcessedJob``
        uses it when the processor has it; results equal ``process`` applied circuit by circuit)."""
        rows, owners, coeffs, values, bases = [], [], [], [], []
        for k, (value, circuit, obs) in enumerate(zip(expectation_values, circuits, observables)):
            for term in obs:
                rows.append(circuit)
                owners.append(k)
                coeffs.append(term.coeffs[0])
                values.append([float(value)])
                bases.append(encode_pauli_sum_op([(str(term.paulis[0]), 1.0)])[0])
        if not rows:
            return [0.0] * len(circuits)
        # one feature matrix for the whole run(); OpenQASM text goes through the C++ op scan (mlqem_circuit_features_qasm)
        native = all(isinstance(c, str) for c in rows)
        model_input, _ = encode_data(circuits=rows, properties=self._properties, ideal_exp_vals=[[0.0]] * len(rows),
                                     noisy_exp_vals=values, num_qubits=1, meas_bases=bases, native=native)
        device = model_device(self._model)
        if device is not None:
            model_input = model_input.to(device)
        with torch.no_grad():
            out = self._model(model_input).reshape(len(rows), -1)[:, 0].cpu().tolist()
        totals = [0.0] * len(circuits)
        for k, o, c in zip(owners, out, coeffs):
            totals[k] = totals[k] + o * c
        return totals


class ScikitLearningModelProcessor(LearningMethodEstimatorProcessor):
    """The same per-term ``encode_data`` rows into anything with scikit-learn's ``predict`` (reference :90-148; the
    demos fit ``RandomForestRegressor`` / OLS on these rows).  Host CPU only: there is no tensor work to move."""

    def __init__(self, model, backend):
        if not hasattr(model, "predict"):
            raise BlackwaterException("ScikitLearningModelProcessor needs a fitted estimator with .predict(X)")
        self._model = model
        self._backend = backend
        self._properties = get_backend_properties_v1(backend)

    def process(self, expectation_value, circuits, observables, parameter_values):
        total = 0.0
        for term in observables:
            row, _ = encode_data(circuits=[circuits], properties=self._properties, ideal_exp_vals=[[0.0]],
                                 noisy_exp_vals=[[expectation_value]], num_qubits=1,
                                 meas_bases=encode_pauli_sum_op([(str(term.paulis[0]), 1.0)]))
            output = float(np.ravel(self._model.predict(row.numpy()))[0])
            total = total + output * float(np.real(term.coeffs[0]))
        return total


class EmptyProcessor(LearningMethodEstimatorProcessor):
    def process(self, expectation_value, circuits, observables, parameter_values):
        return expectation_value


def _options_dict(options) -> dict:
    if options is None:
        return {}
    return dict(options.__dict__) if hasattr(options, "__dict__") and not isinstance(options, dict) else dict(options)


class PostProcessedJob(job_base()):  # type: ignore[misc]
    def __init__(self, base_job, processor: LearningMethodEstimatorProcessor, circuits, observables, parameter_values,
                 skip_transpile: bool, backend, job_id: str, options=None, **kwargs) -> None:
        try:
            super().__init__(backend, job_id, **kwargs)
        except TypeError:  # plain ``object`` base when qiskit is absent
            self._backend, self._job_id = backend, job_id
        self._base_job = base_job
        self._processor = processor
        self._circuits = circuits
        self._observables = observables
        self._parameter_values = parameter_values
        self._options = options
        self._skip_transpile = skip_transpile
        self._wrapped_backend = backend

    def result(self):
        result = self._base_job.result()
        mitigated, metadata = [], []
        batch_fn = getattr(self._processor, "process_batch", None)
        bound_all = []
        for value, circuit, obs, params, meta in zip(result.values, self._circuits, self._observables,
                                                     self._parameter_values, result.metadata):
            if not is_pauli_observable(obs):
                raise BlackwaterException("Only `PauliSumOp` observables are supported by learning primitive.")
            opts = dict(optimization_level=3, **_options_dict(self._options))
            bound = transpile_and_bind(circuit, self._wrapped_backend, params, opts, do_transpile=not self._skip_transpile,
                                       keep_text=batch_fn is not None and getattr(self._processor, "accepts_qasm_text", False))
            metadata.append({**meta, "original_value": value})
            if batch_fn is not None:
                bound_all.append(bound)
                continue
            mitigated.append(self._processor.process(expectation_value=value, circuits=bound, observables=obs,
                                                     parameter_values=params))
        if batch_fn is not None and bound_all:
            mitigated = batch_fn(list(result.values), bound_all, list(self._observables), list(self._parameter_values))
        return make_estimator_result(np.array(mitigated), metadata)

    def submit(self):
        return self._base_job.submit()

    def status(self):
        return self._base_job.status()

    def cancel(self):
        return self._base_job.cancel()

    def __repr__(self):
        return f"<NgemJob: {self._base_job.job_id()}>"  # sic: the reference prints this name here too (:258-259)


def patch_run(run: Callable, processor: LearningMethodEstimatorProcessor, skip_transpile: bool, backend=None,
              options=None) -> Callable:
    @wraps(run)
    def patched_run(self, circuits, observables, parameter_values, **run_options):
        job = run(self, circuits=circuits, observables=observables, parameter_values=parameter_values, **run_options)
        return PostProcessedJob(job, job_id=job.job_id(), backend=backend, processor=processor, circuits=circuits,
                                observables=observables, parameter_values=parameter_values,
                                skip_transpile=skip_transpile, options=options)

    return patched_run


def learning(cls: Type, processor: LearningMethodEstimatorProcessor, skip_transpile: bool = False, backend=None,
             options=None):
    """Decorator turning an Estimator class into ``Learning<cls.__name__>``."""
    new_class: type = type(f"Learning{cls.__name__}", (cls,), {})
    new_class._run = patch_run(new_class._run, processor, skip_transpile, backend, options)  # type: ignore[attr-defined]
    return new_class
