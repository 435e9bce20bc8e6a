"""SURVEY.md section 8 row f1 on the device: every circuit of one ``run()`` post-processed by ONE model call.

* ``ngem(..., batched=True)`` (native C++ encoder -> ``Batch.from_data_list`` -> one call) with Family A, 12 QASM
  circuits, against the per-circuit fp64 oracle (the reference's serial loop, blackwater/library/ngem/estimator.py:49-84);
* the same collate with Family B (gnn1.pth), inference convention (raw graphs, no self-loops), against the per-circuit
  oracle.  Family B does not go through the ``ngem`` wrapper itself in the reference either: the wrapper hands the model
  ``noisy_0`` of shape [1, 1] (estimator.py:68-73) and gnn.py:117 squeezes dim 1 of a [B, 1, k] tensor;
* ``TorchLearningModelProcessor.process_batch`` through ``learning`` with MLP1, 10 circuits x multi-term observables,
  against the per-circuit oracle MLP.
"""
import os

import numpy as np
import pytest
import torch

from helpers import G1_GATES_ORDER, g1_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
IDX = [0, 3, 17, 42, 77, 101, 150, 200, 222, 250, 280, 299]


def test_ngem_batched_family_a_equals_per_circuit_oracle(g1, lima_backend):
    from blackwater.data.backends import PauliObservable
    from blackwater.data.utils import encode_pauli_sum_op
    from blackwater.library.ngem.estimator import ngem
    from blackwater.nn import ExpValCircuitGraphModelA
    from oracle.models import FamilyA
    from test_estimators import FakeEstimator, _Job
    import blackwater.library.ngem.estimator as mod

    torch.manual_seed(4)
    model = ExpValCircuitGraphModelA(5, 22, 10)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bias"):
                p.uniform_(-0.5, 0.5)
    ref = FamilyA(5, 22, 10).double().eval()
    ref.load_state_dict(model.state_dict())
    model = model.to(DEV).eval()
    obs = PauliObservable([("IIZIZ", 0.75)])

    class Est(FakeEstimator):
        def _run(self, circuits, observables, parameter_values, **opts):
            return _Job([g1["noisy"][i][0] for i in IDX])

    orig = mod.get_backend_properties_v1  # the fixture graphs use the reference's (hash-random) gate column order
    mod.get_backend_properties_v1 = lambda b: orig(b, gates_order=G1_GATES_ORDER)
    try:
        circuits = [g1["qasm"][i] for i in IDX]
        batched = ngem(Est, model, lima_backend, batched=True)().run(circuits, [obs] * len(IDX)).result()
        serial = ngem(Est, model, lima_backend)().run(circuits, [obs] * len(IDX)).result()
    finally:
        mod.get_backend_properties_v1 = orig
    want = []
    for i in IDX:
        x, ei, _ = g1_graph(g1, i)
        out = ref(torch.tensor([[g1["noisy"][i][0]]], dtype=torch.float64),
                  torch.tensor([encode_pauli_sum_op(obs)], dtype=torch.float64), torch.zeros(1, 1, dtype=torch.float64),
                  torch.tensor(x, dtype=torch.float32).double(), torch.tensor(ei, dtype=torch.long), None)
        want.append(out.item())
    assert batched.values.shape == (len(IDX),)
    assert np.abs(batched.values - np.array(want)).max() < 1e-5      # north_star tolerance, per circuit
    assert np.abs(batched.values - serial.values).max() < 1e-5


def test_batched_collate_family_b_equals_per_circuit_oracle(golden_dir, g1):
    from blackwater.data.graph import Batch, Data
    from blackwater.nn import family_b_from_state_dict
    from oracle.models import family_b_from_state_dict as oracle_from_sd

    sd = torch.load(os.path.join(golden_dir, "ckpt", "gnn1.pth"), weights_only=True)
    model = family_b_from_state_dict(sd).to(DEV).eval()
    ref = oracle_from_sd(sd).double().eval()
    entries, want = [], []
    with torch.no_grad():
        for i in IDX:
            x, ei, ea = g1_graph(g1, i)
            noisy = torch.tensor(g1["noisy"][i], dtype=torch.float32).view(1, 1, -1)
            depth = torch.tensor([[float(g1["depth"][i])]])
            entries.append(Data(x=torch.tensor(x, dtype=torch.float32), edge_index=torch.tensor(ei, dtype=torch.long),
                                edge_attr=torch.tensor(ea, dtype=torch.float32), y=torch.zeros(1, 1),
                                observable=torch.zeros(1, 0), circuit_depth=depth, noisy_0=noisy))
            want.append(ref(noisy.double(), None, depth.double(), torch.tensor(x, dtype=torch.float32).double(),
                            torch.tensor(ei, dtype=torch.long), None).numpy().ravel())
        batch = Batch.from_data_list(entries).to(DEV)
        got = model(batch.noisy_0, batch.observable, batch.circuit_depth, batch.x, batch.edge_index, batch.batch)
    assert got.shape == (len(IDX), 4)
    assert np.abs(got.cpu().numpy() - np.stack(want)).max() < 1e-5


def test_learning_process_batch_equals_per_circuit_oracle(g1, lima_backend):
    from blackwater.data.backends import PauliObservable
    from blackwater.data.utils import encode_pauli_sum_op, get_backend_properties_v1
    from blackwater.library.learning.estimator import TorchLearningModelProcessor, learning
    from blackwater.library.learning.features import encode_data
    from blackwater.library.learning.mlp import MLP1
    from oracle.models import MLP1 as OracleMLP1
    from test_estimators import FakeEstimator

    torch.manual_seed(0)
    model = MLP1(8 + 6 + 40 + 1 + 21, 64, 1).to(DEV).eval()
    proc = TorchLearningModelProcessor(model, lima_backend)
    assert hasattr(proc, "process_batch")
    est = learning(FakeEstimator, proc, skip_transpile=True)()
    idx = IDX[:10]
    terms = [[("IIIIZ", 1.0)], [("IIIZI", 0.5), ("ZIIII", -0.25)], [("IZIZI", 2.0)]]
    observables = [PauliObservable(terms[k % 3]) for k in range(len(idx))]
    res = est.run([g1["qasm"][i] for i in idx], observables).result()
    ref = OracleMLP1(76, 64, 1).double()
    ref.load_state_dict({k: v.cpu().double() for k, v in model.state_dict().items()})
    props = get_backend_properties_v1(lima_backend)
    for k, i in enumerate(idx):
        value = 0.5 + 0.1 * k                                   # FakeEstimator's noisy values
        total = 0.0
        for label, coeff in terms[k % 3]:
            X, _ = encode_data([g1["qasm"][i]], props, [[0.0]], [[value]], 1, meas_bases=encode_pauli_sum_op(label))
            total += coeff * ref(X.double()).item()
        assert res.values[k] == pytest.approx(total, abs=1e-5)
        assert res.metadata[k]["original_value"] == pytest.approx(value)


@pytest.mark.parametrize("count,chunks", [(300, 4), (130, 2), (7, 4), (64, 1)])
def test_encode_to_device_in_groups_equals_the_one_shot_batch(lima_props, count, chunks):
    """NativeEncoder.encode_batch_to_device (groups of circuits filled into pinned buffers and copied to their slices of the device
    tensors while the next group is written; node offsets and graph numbers shifted on the device) returns the arrays of
    encode_batch bit for bit, for every grouping."""
    from blackwater.data.circuit import circuit_to_qasm
    from blackwater.data.native_encoder import NativeEncoder
    from blackwater.data.synthetic import tfim_circuit

    texts = [circuit_to_qasm(tfim_circuit(5, k % 7, 0.1 + 0.01 * k, two_q="cx")) for k in range(count)]
    enc = NativeEncoder(lima_props)
    x, ei, batch, counts, depths = enc.encode_batch(texts)
    xd, eid, bd, counts_d, depths_d = enc.encode_batch_to_device(texts, DEV, chunks=chunks, group_bytes=4096)
    torch.cuda.synchronize()
    assert torch.equal(xd.cpu(), x) and torch.equal(eid.cpu(), ei) and torch.equal(bd.cpu(), batch)
    assert np.array_equal(counts_d, counts) and list(depths_d) == list(depths)


def test_encode_to_device_names_a_bad_circuit_by_its_position_in_the_run(lima_props):
    from blackwater.data.circuit import circuit_to_qasm
    from blackwater.data.native_encoder import NativeEncoder
    from blackwater.data.synthetic import tfim_circuit

    texts = [circuit_to_qasm(tfim_circuit(5, k % 5, 0.2, two_q="cx")) for k in range(200)]
    texts[150] = 'OPENQASM 2.0;\nqreg q[2];\nrz(1 q[0];\n'
    enc = NativeEncoder(lima_props)
    with pytest.raises(Exception, match="circuit 150: "):
        enc.encode_batch_to_device(texts, DEV, chunks=4, group_bytes=4096)
    x, *_ = enc.encode_batch_to_device(texts[:100], DEV, chunks=4, group_bytes=4096)      # usable after a rejected run
    assert x.shape[0] > 0


def test_serial_decorator_replays_one_captured_forward_per_size_bucket(g1, lima_backend):
    """the serial ``ngem`` loop with this package's Family A on the GPU makes every per-circuit model call by
    replaying a forward captured per size bucket (train.BucketedPredictor).  The values equal the eager per-circuit loop's; the
    captures are kept with the model and reused by later run()s over OTHER circuits (their arrays are copied to the addresses the
    captures read), including a run() that outgrows the allocation, and by one after the model's parameters moved (captures dropped)."""
    import blackwater.library.ngem.estimator as mod
    from blackwater.data.backends import PauliObservable
    from blackwater.library.ngem.estimator import ngem
    from blackwater.nn import ExpValCircuitGraphModelA
    from test_estimators import FakeEstimator

    torch.manual_seed(11)
    model = ExpValCircuitGraphModelA(5, 22, 10).to(DEV).eval()
    eager = ExpValCircuitGraphModelA(5, 22, 10).to(DEV).eval()
    eager.load_state_dict(model.state_dict())
    eager.accepts_device_batches = False          # the plain loop: one encode, one upload, ~45 launches per circuit
    obs = PauliObservable([("ZIIIZ", 1.0), ("IXIII", -0.5)])

    def both(ids):
        circuits = [g1["qasm"][i] for i in ids]
        a = ngem(FakeEstimator, model, lima_backend)().run(circuits, [obs] * len(ids)).result().values
        b = ngem(FakeEstimator, eager, lima_backend)().run(circuits, [obs] * len(ids)).result().values
        assert np.abs(a - b).max() < 1e-6, (ids, np.abs(a - b).max())
        return a

    both(IDX)
    predictor = mod._predictors[model]
    first = predictor.captures
    assert 1 <= first <= len(IDX) and eager not in mod._predictors
    both(IDX[::-1])                                # the same circuits in another order: no new capture
    assert predictor.captures == first
    both([1, 5, 9, 33, 64, 128, 256, 290])         # other circuits, same allocation
    assert mod._predictors[model] is predictor
    seen = predictor.captures
    both(list(range(0, 300, 2)))                   # 150 circuits: outgrows the first allocation (2 x 12 circuits)
    assert mod._predictors[model] is predictor and predictor.captures > seen
    keep = [p.data for p in model.parameters()]    # (held, so that the allocator cannot hand the same addresses back)
    model.to(torch.float64).to(torch.float32)      # parameters at new addresses: the captures of the old ones are dropped
    again = predictor.captures
    both(IDX)
    assert predictor.captures > again and len(keep) > 0
    # a single circuit keeps the plain call (nothing to amortise a capture over)
    one = ngem(FakeEstimator, model, lima_backend)().run([g1["qasm"][3]], [obs]).result().values
    assert np.abs(one - both([3, 3])[:1]).max() < 1e-6


def test_serial_decorator_keeps_a_train_mode_model_in_train_mode_and_lets_go_of_dead_models(g1, lima_backend):
    """(1) a model left in train() mode (dropout on) is called in train() mode by the reference's loop
    (blackwater/library/ngem/estimator.py:75-82) whatever the number of circuits -- the replayed path, which evaluates in eval
    mode, must leave such a model to the plain loop: two runs over the same circuits then differ (fresh masks) for one circuit AND
    for several; (2) the predictor kept per model holds it weakly: when the model goes, so do its captures and arena."""
    import gc
    import weakref

    import blackwater.library.ngem.estimator as mod
    from blackwater.data.backends import PauliObservable
    from blackwater.library.ngem.estimator import ngem
    from blackwater.nn import ExpValCircuitGraphModelA
    from test_estimators import FakeEstimator

    torch.manual_seed(3)
    model = ExpValCircuitGraphModelA(5, 22, 10).to(DEV).train()
    obs = PauliObservable([("ZIIIZ", 1.0)])
    for ids in ([4], [4, 7, 19]):
        circuits = [g1["qasm"][i] for i in ids]
        a = ngem(FakeEstimator, model, lima_backend)().run(circuits, [obs] * len(ids)).result().values
        b = ngem(FakeEstimator, model, lima_backend)().run(circuits, [obs] * len(ids)).result().values
        assert model.training and np.abs(a - b).max() > 0, ids          # dropout drew twice: train-mode semantics on both paths
    assert model not in mod._predictors                                 # the replayed path never took it
    model.eval()
    ngem(FakeEstimator, model, lima_backend)().run([g1["qasm"][i] for i in (4, 7, 19)], [obs] * 3).result()
    assert model in mod._predictors
    alive = weakref.ref(mod._predictors[model])
    del model
    gc.collect()
    assert alive() is None                                              # predictor, captures, pool and arena went with the model


def test_a_large_batched_run_in_slices_equals_the_one_batch_form(g1, lima_backend, monkeypatch):
    """library/ngem/estimator.py ``_batched_in_slices``: a run() of at least two slices is scanned, expanded and evaluated slice by
    slice (host and device overlap) -- the values are those of the one-batch form (a circuit's value does not depend on the batch it
    is in; fp32 summation order of the pooled means: 1e-6), for a slice size that does not divide the run, and nothing in the slice
    loop waits for the device (torch's sync-debug mode raises on a blocking call)."""
    from blackwater.data.backends import PauliObservable
    from blackwater.library.ngem.estimator import ngem
    from blackwater.nn import ExpValCircuitGraphModelA
    from test_estimators import FakeEstimator, _Job
    import blackwater.library.ngem.estimator as mod

    torch.manual_seed(9)
    model = ExpValCircuitGraphModelA(5, 22, 10).to(DEV).eval()
    idx = list(range(0, 150, 3))                     # 50 circuits
    obs = PauliObservable([("IIZIZ", 0.75)])

    class Est(FakeEstimator):
        def _run(self, circuits, observables, parameter_values, **opts):
            return _Job([g1["noisy"][i][0] for i in idx])

    orig = mod.get_backend_properties_v1
    monkeypatch.setattr(mod, "get_backend_properties_v1", lambda b: orig(b, gates_order=G1_GATES_ORDER))
    circuits = [g1["qasm"][i] for i in idx]
    monkeypatch.setattr(mod, "_SLICE", 10 ** 6)
    whole = ngem(Est, model, lima_backend, batched=True)().run(circuits, [obs] * len(idx)).result().values
    calls = []
    real = mod.NgemJob._batched_in_slices

    def watched(self, *a, **k):
        calls.append(1)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")      # a blocking call inside the slice loop raises
        try:
            return real(self, *a, **k)
        finally:
            torch.cuda.set_sync_debug_mode("default")

    monkeypatch.setattr(mod.NgemJob, "_batched_in_slices", watched)
    monkeypatch.setattr(mod, "_SLICE", 16)           # 50 circuits: slices of 16, 16, 16, 2
    sliced = ngem(Est, model, lima_backend, batched=True)().run(circuits, [obs] * len(idx)).result().values
    assert calls == [1]
    assert whole.shape == sliced.shape == (len(idx),)
    assert np.abs(whole - sliced).max() <= 1e-6 * max(1.0, np.abs(whole).max())
