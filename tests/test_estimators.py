"""Boundary tests of the two Estimator decorators with a fake Estimator (no qiskit): same structure as the
reference's tests/library/ngem/test_estimator.py (DummyModel pins the six-argument model protocol)."""
import numpy as np
import pytest
import torch

from blackwater.data.backends import PauliObservable
from blackwater.exception import BlackwaterException
from blackwater.library.learning.estimator import (EmptyProcessor, LearningMethodEstimatorProcessor, PostProcessedJob,
                                                   ScikitLearningModelProcessor, TorchLearningModelProcessor, learning)
from blackwater.library.ngem.estimator import NgemJob, ngem

QASM = ('OPENQASM 2.0;\ninclude "qelib1.inc";\nqreg q[5];\ncreg meas[2];\nrz(0.3) q[0];\nsx q[0];\ncx q[0],q[1];\n'
        'barrier q[0],q[1];\nmeasure q[0] -> meas[0];\nmeasure q[1] -> meas[1];\n')


class _Result:
    def __init__(self, values):
        self.values, self.metadata = np.asarray(values, dtype=float), [{"shots": 7} for _ in values]


class _Job:
    def __init__(self, values):
        self._values = values

    def result(self):
        return _Result(self._values)

    def job_id(self):
        return "job-42"

    def status(self):
        return "DONE"


class FakeEstimator:
    """Stand-in for a qiskit BaseEstimator: ``run`` forwards to ``_run`` with keyword arguments."""

    def run(self, circuits, observables, parameter_values=None, **opts):
        parameter_values = parameter_values or [()] * len(circuits)
        return self._run(circuits, observables, parameter_values, **opts)

    def _run(self, circuits, observables, parameter_values, **opts):
        return _Job([0.5 + 0.1 * k for k in range(len(circuits))])


class DummyModel(torch.nn.Module):
    def forward(self, exp_value, observable, circuit_depth, nodes, edge_index, batch):
        assert tuple(exp_value.shape) == (1, 1) and tuple(circuit_depth.shape) == (1, 1)
        assert observable.dim() == 3 and nodes.shape[1] == 22 and edge_index.shape[0] == 2 and batch is None
        self.seen = (nodes.shape[0], edge_index.shape[1])
        return exp_value * 2


def test_ngem_wraps_class_and_postprocesses(lima_backend):
    model = DummyModel()
    cls = ngem(FakeEstimator, model, lima_backend)
    assert cls.__name__ == "NGEMFakeEstimator" and issubclass(cls, FakeEstimator)
    job = cls().run([QASM, QASM], [PauliObservable("ZIIII"), PauliObservable([("IZIII", 0.5)])])
    assert isinstance(job, NgemJob) and repr(job) == "<NgemJob: job-42>" and job.status() == "DONE"
    res = job.result()
    assert np.allclose(res.values, [1.0, 1.2]) and res.metadata[0] == {"shots": 7}
    assert model.seen == (6, 6)  # 6 op nodes, 6 op->op qubit wires (cx->barrier twice), no self-loops at inference
    assert np.allclose(job.result().values, [1.0, 1.2])  # result() is recomputed on every call
    assert FakeEstimator()._run([QASM], ["x"], [()]).result().values[0] == 0.5  # base class untouched


def test_serial_ngem_path_hands_the_model_the_same_tensors_natively_and_from_python(lima_backend, monkeypatch):
    """the default (serial, reference-shaped) loop encodes OpenQASM text and ``Circuit`` objects with the C++
    encoder; the six model arguments are equal, bit for bit, to the ones the Python walk (circuit_to_graph_data_json ->
    ExpValueEntry.to_pyg_data) builds.  Circuits: the boundary text above and three of the reference's G1 circuits."""
    import json
    import os

    import blackwater.library.ngem.estimator as mod
    from blackwater.data.circuit import Circuit

    class Recorder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.calls = []

        def forward(self, *args):
            self.calls.append([None if a is None else a.clone() for a in args])
            return args[0] * 2

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g1_circuits.json")))
    texts = [QASM, gold[0], gold[7], gold[299]]
    circuits = texts + [Circuit.from_any(QASM)]
    obs = [PauliObservable("ZIIII")] * len(circuits)
    native, python = Recorder(), Recorder()
    monkeypatch.setattr(mod, "_NATIVE_SERIAL", True)
    a = ngem(FakeEstimator, native, lima_backend)().run(circuits, obs).result()
    monkeypatch.setattr(mod, "_NATIVE_SERIAL", False)
    b = ngem(FakeEstimator, python, lima_backend)().run(circuits, obs).result()
    assert np.array_equal(a.values, b.values) and len(native.calls) == len(python.calls) == len(circuits)
    for ca, cb in zip(native.calls, python.calls):
        for ta, tb in zip(ca, cb):
            assert (ta is None and tb is None) or (ta.dtype == tb.dtype and ta.shape == tb.shape and torch.equal(ta, tb))


def test_ngem_rejects_non_pauli_observable(lima_backend):
    job = ngem(FakeEstimator, DummyModel(), lima_backend)().run([QASM], ["not an operator"])
    with pytest.raises(BlackwaterException, match="Only `PauliSumOp` observables are supported by NGEM"):
        job.result()


def test_learning_with_empty_and_custom_processor(lima_backend):
    cls = learning(FakeEstimator, EmptyProcessor(), backend=lima_backend, skip_transpile=True)
    assert cls.__name__ == "LearningFakeEstimator"
    job = cls().run([QASM], [PauliObservable("ZIIII")])
    assert isinstance(job, PostProcessedJob)
    res = job.result()
    assert res.values.tolist() == [0.5] and res.metadata == [{"shots": 7, "original_value": 0.5}]

    class Doubler(LearningMethodEstimatorProcessor):
        def process(self, expectation_value, circuits, observables, parameter_values):
            assert len(circuits.ops) == 6 and parameter_values == ()
            return 2 * expectation_value

    assert learning(FakeEstimator, Doubler())().run([QASM], [PauliObservable("ZIIII")]).result().values[0] == 1.0
    with pytest.raises(BlackwaterException, match="learning primitive"):
        learning(FakeEstimator, EmptyProcessor())().run([QASM], [3.14]).result()
    with pytest.raises(NotImplementedError):
        LearningMethodEstimatorProcessor().process(0.0, None, None, None)


def test_torch_processor_feature_row_and_coefficients(lima_backend):
    class Probe(torch.nn.Module):
        def forward(self, x):
            assert tuple(x.shape) == (1, 8 + 6 + 40 + 1 + 21) and x.dtype == torch.float32
            self.row = x.clone()
            return x[:, 54:55] + 1.0  # noisy value sits right after backend | gate counts | angle bins

    probe = Probe()
    proc = TorchLearningModelProcessor(probe, lima_backend)
    obs = PauliObservable([("ZIIII", 2.0), ("IXIII", -1.0)])
    out = proc.process(0.25, QASM, obs, ())
    assert out == pytest.approx((0.25 + 1.0) * 2.0 + (0.25 + 1.0) * -1.0)
    assert probe.row[0, 55:].tolist() == [1.0] + [1, 0, 0, 0] + [0, 0, 0, 1] + [1, 0, 0, 0] * 3  # IXIII, coeff 1
    assert probe.row[0, 8 + 0].item() == pytest.approx(0.01)  # one cx


def test_batched_postprocessing_equals_serial(lima_backend):
    """``ngem(..., batched=True)`` and ``TorchLearningModelProcessor.process_batch`` give the per-circuit results."""

    class MeanModel(torch.nn.Module):  # uses every argument incl. ``batch``
        def forward(self, exp_value, observable, circuit_depth, nodes, edge_index, batch):
            b = exp_value.shape[0]
            idx = torch.zeros(nodes.shape[0], dtype=torch.long) if batch is None else batch
            pooled = torch.zeros(b, nodes.shape[1]).index_add_(0, idx, nodes)
            counts = torch.zeros(b).index_add_(0, idx, torch.ones(nodes.shape[0]))
            return exp_value + (pooled / counts[:, None]).sum(1, keepdim=True) + observable.sum((1, 2)).unsqueeze(1)

    q2 = QASM.replace("rz(0.3) q[0];", "rz(0.3) q[0];\nx q[1];\nsx q[1];")
    circuits, obs = [QASM, q2, QASM], [PauliObservable("ZIIII"), PauliObservable("IZIII"), PauliObservable("IIZII")]
    serial = ngem(FakeEstimator, MeanModel(), lima_backend)().run(circuits, obs).result().values
    batched = ngem(FakeEstimator, MeanModel(), lima_backend, batched=True)().run(circuits, obs).result().values
    assert np.allclose(serial, batched, rtol=1e-6)

    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(76, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))
    proc = TorchLearningModelProcessor(mlp, lima_backend)
    two_terms = PauliObservable([("ZIIII", 0.5), ("IXIII", -2.0)])
    want = [proc.process(0.5, QASM, two_terms, ()), proc.process(0.6, q2, PauliObservable("IIIIZ"), ())]
    got = learning(FakeEstimator, proc, skip_transpile=True)().run([QASM, q2], [two_terms, PauliObservable("IIIIZ")]).result()
    assert np.allclose(got.values, want, rtol=1e-6) and got.metadata[1]["original_value"] == pytest.approx(0.6)


def test_scikit_processor_matches_torch_processor_on_a_linear_model(lima_backend):
    """The RF / OLS baseline path (reference learning/estimator.py:90-148): the same 76-wide rows go to ``predict``; an
    OLS fit and a torch Linear carrying its coefficients must agree through the decorator."""
    from sklearn.linear_model import LinearRegression

    rng = np.random.default_rng(0)
    ols = LinearRegression().fit(rng.normal(size=(200, 76)), rng.normal(size=200))
    lin = torch.nn.Linear(76, 1)
    with torch.no_grad():
        lin.weight.copy_(torch.tensor(ols.coef_, dtype=torch.float32)[None])
        lin.bias.fill_(float(ols.intercept_))
    obs = [PauliObservable([("ZIIII", 0.5), ("IXIII", -2.0)]), PauliObservable("IIIIZ")]
    sk = learning(FakeEstimator, ScikitLearningModelProcessor(ols, lima_backend), skip_transpile=True)
    th = learning(FakeEstimator, TorchLearningModelProcessor(lin, lima_backend), skip_transpile=True)
    a = sk().run([QASM, QASM], obs).result().values
    b = th().run([QASM, QASM], obs).result().values
    assert np.allclose(a, b, rtol=1e-4, atol=1e-4)
    with pytest.raises(BlackwaterException):
        ScikitLearningModelProcessor(object(), lima_backend)
