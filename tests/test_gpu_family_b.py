"""Family B and the MLP path on the GPU against the reference's goldens and the CPU oracle:
G1 through the product model AND through the ``ngem`` decorator from QASM text, G2 through the product MLP1 and
through ``TorchLearningModelProcessor``; kernel-level checks of the attention / pooling pieces."""
import os

import numpy as np
import pytest
import torch

from helpers import G1_GATES_ORDER, g1_batch, g1_graph, mean_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ckpt(golden_dir, name):
    return torch.load(os.path.join(golden_dir, "ckpt", name), weights_only=True)


def _oracle_outputs(sd, g1, idx, dtype=torch.float64):
    from oracle.models import family_b_from_state_dict

    model = family_b_from_state_dict(sd).to(dtype).eval()
    outs = []
    with torch.no_grad():
        for i in idx:
            x, ei, _ = g1_graph(g1, i)
            out = model(torch.tensor(g1["noisy"][i], dtype=torch.float32).to(dtype).view(1, 1, -1), None,
                        torch.tensor([[float(g1["depth"][i])]], dtype=dtype),
                        torch.tensor(x, dtype=torch.float32).to(dtype), torch.tensor(ei, dtype=torch.long), None)
            outs.append(out.numpy().ravel())
    return np.stack(outs)


def _gpu_outputs(model, g1, idx):
    outs = []
    with torch.no_grad():
        for i in idx:
            x, ei, _ = g1_graph(g1, i)
            out = model(torch.tensor(g1["noisy"][i], dtype=torch.float32, device=DEV).view(1, 1, -1), None,
                        torch.tensor([[float(g1["depth"][i])]], device=DEV),
                        torch.tensor(x, dtype=torch.float32, device=DEV),
                        torch.tensor(ei, dtype=torch.long, device=DEV), None)
            outs.append(out.cpu().numpy().ravel())
    return np.stack(outs)


def test_g1_golden_on_gpu(golden_dir, g1):
    from blackwater.nn import family_b_from_state_dict

    sd = _ckpt(golden_dir, "gnn1.pth")
    model = family_b_from_state_dict(sd).to(DEV).eval()
    idx = range(300)
    got = _gpu_outputs(model, g1, idx)
    assert round(mean_l2(g1["ideal"], got), 6) == 0.117838  # docs/tutorials/h17_compare_over_steps.ipynb:513
    want = _oracle_outputs(sd, g1, idx)
    assert np.abs(got - want).max() < 1e-5  # north_star tolerance, per circuit, vs the fp64 oracle


@pytest.mark.parametrize("name", ["gnn3_ising.pth", "train_fakelima.pth"])
def test_other_reference_architectures(golden_dir, g1, name):
    """Heads 5/3 with the MLP3 head (hidden 25, 103 465 parameters) and the single-output variant."""
    from blackwater.nn import family_b_from_state_dict

    sd = _ckpt(golden_dir, name)
    out_size = sd["body_seq.fc4.weight"].shape[0] if "body_seq.fc4.weight" in sd else sd["body_seq.2.weight"].shape[0]
    model = family_b_from_state_dict(sd).to(DEV).eval()
    sub = dict(g1)
    sub["noisy"] = g1["noisy"][:, :out_size]
    idx = range(0, 300, 7)
    got, want = _gpu_outputs(model, sub, idx), _oracle_outputs(sd, sub, idx)
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())


def _widen_features(x, width):
    """The G1 graphs (F = 22: FakeLima's six gates) as graphs of a backend with more gates: zero one-hot columns inserted after the
    existing gate columns, the way `circuit_to_graph_data_json` lays a wider gates_set out (blackwater/data/utils.py:214-222)."""
    out = np.zeros((x.shape[0], width), dtype=x.dtype)
    out[:, :11] = x[:, :11]
    out[:, width - 11:] = x[:, 11:]
    return out


def test_iskandar_checkpoint_f28_hidden20_on_gpu(golden_dir, g1):
    """The reference's lima-backend checkpoint (docs/tutorials/model/finetuning/iskandar.pth: F = 28, hidden 20, heads 3/2 ->
    60 / 40 channels, one output; 03_experiments_on_lima_backend.ipynb): a 28 -> 60 -> 40 model on the device against the fp64
    oracle, per circuit (inference convention) and batched with self-loops (training convention), outputs and every gradient."""
    from blackwater.nn import family_b_from_state_dict
    from oracle.models import family_b_from_state_dict as oracle_from_sd

    sd = _ckpt(golden_dir, "iskandar.pth")
    assert sd["transformer1.lin_key.weight"].shape == (60, 28) and sd["body_seq.0.weight"].shape == (20, 42)
    wide = dict(g1)
    wide["x"] = _widen_features(g1["x"], 28)
    wide["noisy"] = g1["noisy"][:, :1]
    wide["ideal"] = g1["ideal"][:, :1]
    model = family_b_from_state_dict(sd).to(DEV).eval()
    idx = range(0, 300, 5)
    got, want = _gpu_outputs(model, wide, idx), _oracle_outputs(sd, wide, idx)
    assert got.shape == (60, 1) and np.abs(got - want).max() < 1e-5
    ref = oracle_from_sd(sd).double().eval()
    batch = g1_batch(wide, range(40, 72), self_loops=True, first_only=False)
    out = model(batch["noisy"].to(DEV), None, batch["depth"].to(DEV), batch["x"].to(DEV), batch["edge_index"].to(DEV), batch["batch"].to(DEV))
    torch.nn.functional.mse_loss(out, batch["y"].to(DEV)).backward()
    ref_out = ref(batch["noisy"].double(), None, batch["depth"].double(), batch["x"].double(), batch["edge_index"], batch["batch"])
    assert (out.detach().cpu().double() - ref_out.detach()).abs().max().item() < 1e-5
    torch.nn.functional.mse_loss(ref_out, batch["y"].double()).backward()
    ref_grads = {k: p.grad for k, p in ref.named_parameters()}
    overall = max(g.abs().max().item() for g in ref_grads.values())
    for name, p in model.named_parameters():
        scale = max(ref_grads[name].abs().max().item(), 1e-3 * overall)
        err = (p.grad.cpu().double() - ref_grads[name]).abs().max().item() / scale
        assert err < 2e-4, f"{name}: relative gradient error {err}"


def test_batched_with_self_loops_matches_oracle(golden_dir, g1):
    """The training-time convention: AddSelfLoops + collate (the attention then includes the self-loop)."""
    from blackwater.nn import family_b_from_state_dict
    from oracle.models import family_b_from_state_dict as oracle_from_sd

    sd = _ckpt(golden_dir, "gnn1.pth")
    batch = g1_batch(g1, range(100, 148), self_loops=True, first_only=False)
    model = family_b_from_state_dict(sd).to(DEV).eval()
    with torch.no_grad():
        got = model(batch["noisy"].to(DEV), None, batch["depth"].to(DEV), batch["x"].to(DEV),
                    batch["edge_index"].to(DEV), batch["batch"].to(DEV))
        want = oracle_from_sd(sd).double().eval()(batch["noisy"].double(), None, batch["depth"].double(),
                                                  batch["x"].double(), batch["edge_index"], batch["batch"])
    assert got.shape == (48, 4)
    assert (got.cpu().double() - want).abs().max().item() < 1e-5


def test_pooling_pieces_match_oracle(golden_dir, g1):
    """perm (top-k order) and the coarsened edge list are integer outputs: exact match with the oracle."""
    from blackwater.native.structure import GraphStructure
    from blackwater.nn import family_b_from_state_dict
    from oracle.models import family_b_from_state_dict as oracle_from_sd

    sd = _ckpt(golden_dir, "gnn1.pth")
    model = family_b_from_state_dict(sd).to(DEV).eval()
    ref = oracle_from_sd(sd).eval()
    batch = g1_batch(g1, [3, 250, 77], self_loops=False, first_only=False)
    x, ei, bvec = batch["x"], batch["edge_index"], batch["batch"]
    with torch.no_grad():
        s = GraphStructure.from_edge_index(ei.to(DEV), x.shape[0], batch=bvec.to(DEV), num_graphs=3)
        g = model.transformer1(x.to(DEV), s)
        g_ref = ref.transformer1(x, ei)
        assert (g.cpu() - g_ref).abs().max().item() < 1e-4
        xo, s2, perm = model.pooling1(g, s)
        xo_ref, ei_ref, _, b_ref, perm_ref = ref.pooling1(g_ref, ei, batch=bvec)
    assert perm.cpu().tolist() == perm_ref.tolist()
    assert (xo.cpu() - xo_ref).abs().max().item() < 1e-4
    # pooled structure: CSR by source of the GPU result lists exactly the oracle's sorted (src, dst) pairs
    k = perm.numel()
    out_ptr, out_dst = s2.out_ptr.cpu().numpy(), s2.out_dst.cpu().numpy()
    pairs = [(p, int(q)) for p in range(k) for q in out_dst[out_ptr[p]:out_ptr[p + 1]]]
    assert pairs == [tuple(e) for e in ei_ref.t().tolist()]
    assert s2.graph_ptr.cpu().tolist() == np.concatenate([[0], np.cumsum(np.bincount(b_ref.numpy(), minlength=3))]).tolist()


def test_g2_golden_mlp_on_gpu(golden_dir, g1, lima_props):
    from blackwater.data.circuit import Circuit
    from blackwater.library.learning.mlp import MLP1, encode_data

    sd = _ckpt(golden_dir, "mlp1_smaller_2.pth")
    model = MLP1(58, 64, 4)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    X, _ = encode_data([Circuit.from_qasm_str(t) for t in g1["qasm"]], lima_props, g1["ideal"].tolist(),
                       g1["noisy"].tolist(), 4)
    with torch.no_grad():
        out = model(X.to(DEV)).cpu().numpy()
    assert round(mean_l2(g1["ideal"], out), 6) == 0.032910  # h17 cell [14], L2_mlp step 0


@pytest.mark.parametrize("name,cls", [("mlp2_mbd.pth", "MLP2"), ("mlp3_ising.pth", "MLP3")])
def test_mlp2_mlp3_match_oracle(golden_dir, name, cls):
    import blackwater.nn as bnn
    import oracle.models as om

    sd = _ckpt(golden_dir, name)
    i, h = sd["fc1.weight"].shape[1], sd["fc1.weight"].shape[0]
    o = sd["fc4.weight"].shape[0] if "fc4.weight" in sd else sd["fc3.weight"].shape[0]
    gpu, ref = getattr(bnn, cls)(i, h, o), getattr(om, cls)(i, h, o).double()
    gpu.load_state_dict(sd, strict=True)
    ref.load_state_dict(sd, strict=True)
    x = torch.randn(257, i, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        got = gpu.to(DEV).eval()(x.to(DEV)).cpu().double()
        want = ref.eval()(x.double())
    assert (got - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())


def test_ngem_decorator_end_to_end_from_qasm(g1, lima_backend):
    """QASM text -> encoder -> graph -> GPU model through the estimator wrapper, with the model family the reference
    uses the decorator with (docs/tutorials/01_ngem.ipynb / 04_ngem_vqe.ipynb: scalar exp_value, an observable)."""
    from blackwater.data.backends import PauliObservable
    from blackwater.data.utils import encode_pauli_sum_op
    from blackwater.library.ngem.estimator import ngem
    from blackwater.nn import ExpValCircuitGraphModelA
    from oracle.models import FamilyA
    from test_estimators import FakeEstimator, _Job
    import blackwater.library.ngem.estimator as mod

    torch.manual_seed(4)
    model = ExpValCircuitGraphModelA(5, 22, 10)
    ref = FamilyA(5, 22, 10).double().eval()
    ref.load_state_dict(model.state_dict())
    model = model.to(DEV).eval()
    idx = [0, 17, 150, 299]
    obs = PauliObservable([("IIZIZ", 0.75)])

    class Est(FakeEstimator):
        def _run(self, circuits, observables, parameter_values, **opts):
            return _Job([g1["noisy"][i][0] for i in idx])

    orig = mod.get_backend_properties_v1  # the fixture graphs use the reference's (hash-random) gate column order
    mod.get_backend_properties_v1 = lambda b: orig(b, gates_order=G1_GATES_ORDER)
    try:
        res = ngem(Est, model, lima_backend)().run([g1["qasm"][i] for i in idx], [obs] * len(idx)).result()
    finally:
        mod.get_backend_properties_v1 = orig
    want = []
    for i in idx:
        x, ei, _ = g1_graph(g1, i)  # == what the encoder yields for this QASM (tests/test_encoder_goldens.py)
        out = ref(torch.tensor([[g1["noisy"][i][0]]], dtype=torch.float64),
                  torch.tensor([encode_pauli_sum_op(obs)], dtype=torch.float64),
                  torch.zeros(1, 1, dtype=torch.float64),  # NgemJob builds the entry without a depth (reference :68-73)
                  torch.tensor(x, dtype=torch.float32).double(), torch.tensor(ei, dtype=torch.long), None)
        want.append(out.item())
    assert np.abs(res.values - np.array(want)).max() < 1e-5


def test_learning_decorator_end_to_end(golden_dir, g1, lima_backend):
    from blackwater.data.backends import PauliObservable
    from blackwater.library.learning.estimator import TorchLearningModelProcessor, learning
    from blackwater.library.learning.mlp import MLP1
    from test_estimators import FakeEstimator

    torch.manual_seed(0)
    model = MLP1(8 + 6 + 40 + 1 + 21, 64, 1).to(DEV).eval()
    est = learning(FakeEstimator, TorchLearningModelProcessor(model, lima_backend), skip_transpile=True)()
    res = est.run([g1["qasm"][0], g1["qasm"][1]], [PauliObservable("IIIIZ"), PauliObservable([("IIIZI", 0.5)])]).result()
    assert res.values.shape == (2,) and np.isfinite(res.values).all()
    assert res.metadata[1]["original_value"] == pytest.approx(0.6)
    # same rows through the CPU oracle MLP
    from oracle.models import MLP1 as OracleMLP1
    from blackwater.library.learning.features import encode_data
    from blackwater.data.utils import encode_pauli_sum_op, get_backend_properties_v1

    ref = OracleMLP1(76, 64, 1)
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    props = get_backend_properties_v1(lima_backend)
    X, _ = encode_data([g1["qasm"][1]], props, [[0.0]], [[0.6]], 1, meas_bases=encode_pauli_sum_op("IIIZI"))
    assert res.values[1] == pytest.approx(0.5 * ref(X).item(), abs=1e-6)


def _family_b_pair(golden_dir, name="gnn1.pth", perturb=0.0):
    from blackwater.nn import family_b_from_state_dict
    from oracle.models import family_b_from_state_dict as oracle_from_sd

    sd = _ckpt(golden_dir, name)
    model = family_b_from_state_dict(sd).to(DEV)
    ref = oracle_from_sd(sd).double()
    return model, ref


@pytest.mark.parametrize("self_loops", [True, False])
def test_family_b_gradients_match_oracle(golden_dir, g1, self_loops):
    """Every parameter gradient of the reference's own architecture (gnn1.pth weights), MSE loss on a 24-graph batch,
    dropout off: GPU backward kernels vs autograd through the fp64 oracle."""
    model, ref = _family_b_pair(golden_dir)
    model.eval(), ref.eval()
    batch = g1_batch(g1, range(200, 224), self_loops=self_loops, first_only=False)
    out = model(batch["noisy"].to(DEV), None, batch["depth"].to(DEV), batch["x"].to(DEV),
                batch["edge_index"].to(DEV), batch["batch"].to(DEV))
    torch.nn.functional.mse_loss(out, batch["y"].to(DEV)).backward()
    want = ref(batch["noisy"].double(), None, batch["depth"].double(), batch["x"].double(), batch["edge_index"],
               batch["batch"])
    assert (out.detach().cpu().double() - want.detach()).abs().max().item() < 1e-5
    torch.nn.functional.mse_loss(want, batch["y"].double()).backward()
    ref_grads = {k: p.grad for k, p in ref.named_parameters()}
    overall = max(g.abs().max().item() for g in ref_grads.values())
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        g_ref = ref_grads[name]
        # lin_key.bias has an analytically ZERO gradient (a constant added to every key cancels in the softmax), so
        # errors are measured against the larger of the parameter's own scale and 1e-3 of the overall gradient scale
        scale = max(g_ref.abs().max().item(), 1e-3 * overall)
        err = (p.grad.cpu().double() - g_ref).abs().max().item() / scale
        assert err < 2e-4, f"{name}: relative gradient error {err}"


def test_family_b_input_gradient_and_train_mode(golden_dir, g1):
    model, ref = _family_b_pair(golden_dir)
    batch = g1_batch(g1, range(5, 13), self_loops=True, first_only=False)
    x_gpu = batch["x"].to(DEV).requires_grad_(True)
    x_ref = batch["x"].double().requires_grad_(True)
    model.eval(), ref.eval()
    model(batch["noisy"].to(DEV), None, batch["depth"].to(DEV), x_gpu, batch["edge_index"].to(DEV),
          batch["batch"].to(DEV)).square().sum().backward()
    ref(batch["noisy"].double(), None, batch["depth"].double(), x_ref, batch["edge_index"],
        batch["batch"]).square().sum().backward()
    scale = x_ref.grad.abs().max().item()
    assert (x_gpu.grad.cpu().double() - x_ref.grad).abs().max().item() / scale < 2e-4
    # train mode: attention dropout + head dropout run, gradients stay finite, two steps differ (fresh masks)
    model.train()
    outs = []
    for _ in range(2):
        model.zero_grad()
        o = model(batch["noisy"].to(DEV), None, batch["depth"].to(DEV), batch["x"].to(DEV),
                  batch["edge_index"].to(DEV), batch["batch"].to(DEV))
        o.sum().backward()
        outs.append(o.detach().clone())
        assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    assert not torch.equal(outs[0], outs[1])


def test_family_b_trains_through_the_reference_loop(golden_dir, g1):
    """The reference's own train loop shape (gnn.py:318-378): loss.backward(); optimizer.step() on model.parameters()."""
    from blackwater.nn import ExpValCircuitGraphModel

    torch.manual_seed(0)
    model = ExpValCircuitGraphModel(num_node_features=22, hidden_channels=15).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    batch = g1_batch(g1, range(0, 64), self_loops=True, first_only=False)
    args = (batch["noisy"].to(DEV), None, batch["depth"].to(DEV), batch["x"].to(DEV), batch["edge_index"].to(DEV),
            batch["batch"].to(DEV))
    y = batch["y"].to(DEV)
    model.eval()  # deterministic loss curve for the assertion below
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(model(*args), y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0]


def test_dense_sync_free_coarsening_equals_the_two_hop_path(golden_dir, g1):
    """ASAPooling's coarsened connectivity built as per-graph bit matrices in LDS (no device->host copy; the default for
    graphs that pool to <= 512 clusters) against the general two-hop path (count -> read back -> fill -> sort-unique):
    identical CSR arrays at both pooling levels, hence bit-identical model outputs."""
    from blackwater.native import functional as F
    from blackwater.native.structure import GraphStructure
    from blackwater.nn import family_b_from_state_dict

    sd = _ckpt(golden_dir, "gnn1.pth")
    model = family_b_from_state_dict(sd).to(DEV).eval()
    for self_loops in (True, False):
        batch = g1_batch(g1, list(range(60, 100)) + [299, 0, 150], self_loops=self_loops, first_only=False)
        n, b = batch["x"].shape[0], batch["noisy"].shape[0]
        results = {}
        for dense in (True, False):
            F._ASAP_DENSE, keep_link = dense, F._ASAP_LINK
            F._ASAP_LINK = True                    # out_eid is compared below (the list form skips the link pass by default)
            try:
                with torch.no_grad():
                    s = GraphStructure.from_edge_index(batch["edge_index"].to(DEV), n, batch=batch["batch"].to(DEV), num_graphs=b)
                    h = model.transformer1(batch["x"].to(DEV), s)
                    h1, s1, perm1 = model.pooling1(h, s)
                    h2, s2, perm2 = model.pooling2(model.transformer2(h1, s1), s1)
                    out = model(batch["noisy"].to(DEV), None, batch["depth"].to(DEV), batch["x"].to(DEV),
                                batch["edge_index"].to(DEV), batch["batch"].to(DEV))
            finally:
                F._ASAP_DENSE, F._ASAP_LINK = True, keep_link
            results[dense] = (s1, s2, perm1, perm2, out)
        for lvl in (0, 1):
            a, c = results[True][lvl], results[False][lvl]
            e = int(c.in_ptr[c.num_nodes].item())
            assert e == c.num_edges and a.num_edges >= e            # the dense path only knows a capacity on the host
            assert torch.equal(a.in_ptr[:a.num_nodes + 1], c.in_ptr[:c.num_nodes + 1])
            assert torch.equal(a.out_ptr[:a.num_nodes + 1], c.out_ptr[:c.num_nodes + 1])
            assert torch.equal(a.in_src[:e], c.in_src[:e]) and torch.equal(a.out_dst[:e], c.out_dst[:e])
            assert torch.equal(a.out_eid[:e], c.out_eid[:e])
            assert not a.loops[:a.num_nodes].any()
        assert torch.equal(results[True][2], results[False][2]) and torch.equal(results[True][3], results[False][3])
        assert torch.equal(results[True][4], results[False][4])


def _pooled_levels(model, x, ei, batch_vec, b):
    from blackwater.native.structure import GraphStructure

    with torch.no_grad():
        s = GraphStructure.from_edge_index(ei, x.shape[0], batch=batch_vec, num_graphs=b)
        h = model.transformer1(x, s)
        h1, s1, perm1 = model.pooling1(h, s)
        h2, s2, perm2 = model.pooling2(model.transformer2(h1, s1), s1)
    return s1, s2, perm1, perm2, h2


def test_wave_per_cluster_coarsening_equals_the_two_hop_path(golden_dir, g1):
    """The large-graph form of ASAPooling's coarsening (one wave per cluster, LDS bitsets, one host read) against the
    two-hop path (four reads, two 64-bit sorts): identical CSR arrays and out_eid at both pooling levels -- on the
    reference's small graphs (dense form switched off) and on 100-qubit circuits whose second pooling has hub clusters
    with hundreds of neighbours."""
    from blackwater.data.synthetic import TfimCorpus
    from blackwater.native import functional as F
    from blackwater.nn import ExpValCircuitGraphModel, family_b_from_state_dict

    cases = []
    small = g1_batch(g1, list(range(60, 90)) + [299, 0, 150], self_loops=True, first_only=False)
    cases.append((family_b_from_state_dict(_ckpt(golden_dir, "gnn1.pth")).to(DEV).eval(), small["x"].to(DEV),
                  small["edge_index"].to(DEV), small["batch"].to(DEV), small["noisy"].shape[0]))
    corpus = TfimCorpus(100, [1, 4, 7], 1, seed=3, exp_value_size=4)
    hg = corpus.host_graphs()
    counts = np.asarray([g.shape[0] for g in hg["x"]])
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    xs = np.concatenate(hg["x"], 0).astype(np.float32)
    eis = np.concatenate([np.asarray(e, dtype=np.int64) + o for e, o in zip(hg["edge_index"], offs)], 1)
    torch.manual_seed(1)
    cases.append((ExpValCircuitGraphModel(22, 15).to(DEV).eval(), torch.from_numpy(xs).to(DEV), torch.from_numpy(eis).to(DEV),
                  torch.from_numpy(np.repeat(np.arange(len(counts)), counts)).to(DEV), len(counts)))
    for model, x, ei, bv, b in cases:
        res = {}
        # "lists": sorted lists from persistent waves (round 4, the default); "rows": dense bit matrices (round 3); "hop": two-hop
        for mode, (rows, lists) in {"lists": (True, True), "rows": (True, False), "hop": (False, False)}.items():
            F._ASAP_DENSE, F._ASAP_ROWS, F._ASAP_LISTS, keep_link = False, rows, lists, F._ASAP_LINK
            F._ASAP_LINK = True                    # out_eid is compared below (the default skips the link pass)
            try:
                res[mode] = _pooled_levels(model, x, ei, bv, b)
                for lvl in (0, 1):
                    res[mode][lvl].in_ptr          # the second level is deferred: build it under THIS mode's switches
            finally:
                F._ASAP_DENSE, F._ASAP_ROWS, F._ASAP_LISTS, F._ASAP_LINK = True, True, True, keep_link
        for mode in ("lists", "rows"):
            for lvl in (0, 1):
                a, c = res[mode][lvl], res["hop"][lvl]
                e = int(c.in_ptr[c.num_nodes].item())
                assert e == c.num_edges == a.num_edges and e > 0
                assert torch.equal(a.in_ptr[:a.num_nodes + 1], c.in_ptr[:c.num_nodes + 1])
                assert torch.equal(a.out_ptr[:a.num_nodes + 1], c.out_ptr[:c.num_nodes + 1])
                assert torch.equal(a.in_src[:e], c.in_src[:e]) and torch.equal(a.out_dst[:e], c.out_dst[:e])
                assert torch.equal(a.out_eid[:e], c.out_eid[:e])
                assert not a.loops[:a.num_nodes].any()
            assert torch.equal(res[mode][2], res["hop"][2]) and torch.equal(res[mode][3], res["hop"][3])
            assert torch.equal(res[mode][4], res["hop"][4])


def test_list_coarsening_with_a_structural_capacity_equals_the_exact_one(golden_dir):
    """An arena batch sizes the list form's scratch and edge arrays by GraphArena.coarse_capacity (no device->host copy); the arrays'
    first in_ptr[K] entries equal the ones built with read-back sizes, the capacity bounds both row-bound totals, and the
    structure of a batch is the same whichever capacity it was built with -- 100-qubit circuits, both pooling levels."""
    from blackwater.data.arena import GraphArena
    from blackwater.data.synthetic import TfimCorpus
    from blackwater.native import _lib
    from blackwater.nn import ExpValCircuitGraphModel

    corpus = TfimCorpus(100, [1, 3, 6, 10], 2, seed=5, exp_value_size=4)
    h = corpus.host_graphs()
    arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"], device=DEV)
    batch = arena.batch(np.arange(len(arena)))
    s = batch.structure
    assert s.coarse_capacity is not None and s.coarse_capacity > 0
    torch.manual_seed(2)
    model = ExpValCircuitGraphModel(22, 15).to(DEV).eval()
    from blackwater.native import functional as F
    keep_link, F._ASAP_LINK = F._ASAP_LINK, True
    with torch.no_grad():
        g = model.transformer1(batch.nodes.materialize() if hasattr(batch.nodes, "materialize") else batch.nodes, s)
        cap = s.coarse_capacity
        _, s1_cap, perm_cap = model.pooling1(g, s)
        s1_cap.in_ptr
        s.coarse_capacity = None
        _, s1_exact, perm_exact = model.pooling1(g, s)
        s1_exact.in_ptr
        s.coarse_capacity = cap
    F._ASAP_LINK = keep_link
    assert torch.equal(perm_cap, perm_exact)
    k = s1_exact.num_nodes
    e = int(s1_exact.in_ptr[k].item())
    assert e == s1_exact.num_edges and e <= cap and s1_cap.num_edges == cap
    for name in ("in_ptr", "out_ptr"):
        assert torch.equal(getattr(s1_cap, name)[:k + 1], getattr(s1_exact, name)[:k + 1])
    for name in ("in_src", "out_dst", "out_eid"):
        assert torch.equal(getattr(s1_cap, name)[:e], getattr(s1_exact, name)[:e])
    # the capacity really bounds the scratch the rows are placed in
    lib = _lib.load()
    from blackwater.native import ops
    need = lib.mlqem_asap_coarsen_lists_workspace_bytes(s.num_nodes, k, 0, 0)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    totals = torch.zeros(4, dtype=torch.int64, device=DEV)
    code = lib.mlqem_asap_coarsen_lists_caps(s.in_ptr.data_ptr(), s.in_src.data_ptr(), s.out_ptr.data_ptr(), s.out_dst.data_ptr(),
                                             s1_exact.graph_ptr.data_ptr(), perm_exact.data_ptr(), s.num_nodes, k, s.num_graphs,
                                             totals.data_ptr(), ws.data_ptr(), need, torch.cuda.current_stream().cuda_stream)
    assert code == 0
    t = totals.tolist()           # row bounds (out, in), per-node list sizes (out, in)
    assert e <= min(t[:2]) and max(t) <= cap


def test_list_coarsening_raises_its_overflow_flag_when_the_capacity_is_no_bound():
    """with a capacity the list form reads nothing back and DROPS a list that would leave its buffer; the flag the kernels
    raise must reach the host (ops.check_overflow_flags: epoch ends, MLQEM_SYNC_OPS) instead of a silently truncated graph."""
    from blackwater.data.arena import GraphArena
    from blackwater.data.synthetic import TfimCorpus
    from blackwater.native import _lib, ops
    from blackwater.nn import ExpValCircuitGraphModel

    h = TfimCorpus(100, [2, 5], 1, seed=5, exp_value_size=4).host_graphs()
    arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"], device=DEV)
    batch = arena.batch(np.arange(len(arena)))
    s = batch.structure
    torch.manual_seed(2)
    model = ExpValCircuitGraphModel(22, 15).to(DEV).eval()
    ops.check_overflow_flags()                         # whatever earlier tests left
    with torch.no_grad():
        g = model.transformer1(batch.nodes, s)
        _, s1, _ = model.pooling1(g, s)
        s1.in_ptr                                      # builds the coarsened graph within the structural capacity
        ops.check_overflow_flags()                     # ... which is a bound: nothing raised
        s.coarse_capacity = 4096                       # far below the real edge total
        _, s1_small, _ = model.pooling1(g, s)
        s1_small.in_ptr
    with pytest.raises(_lib.NativeLibraryError, match="capacity overflow"):
        ops.check_overflow_flags()
    ops.check_overflow_flags()                         # read once, cleared


def test_overflow_flag_is_sticky_across_replays_of_a_captured_launch():
    """a captured step registers nothing on the host when it is replayed, so the flag its kernels raise must be ONE
    persistent per-device word that launches OR into and only the host's check resets: every replay that overflows is seen, not
    only the launch that was enqueued (or captured) by the Python wrapper."""
    from blackwater.data.arena import GraphArena
    from blackwater.data.synthetic import TfimCorpus
    from blackwater.native import _lib, ops
    from blackwater.nn import ExpValCircuitGraphModel

    h = TfimCorpus(100, [2, 5], 1, seed=5, exp_value_size=4).host_graphs()
    arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"], device=DEV)
    batch = arena.batch(np.arange(len(arena)))
    s = batch.structure
    torch.manual_seed(2)
    model = ExpValCircuitGraphModel(22, 15).to(DEV).eval()
    ops.check_overflow_flags()
    with torch.no_grad():
        g = model.transformer1(batch.nodes, s)
        s.coarse_capacity = 4096                       # far below the real edge total
        _, s1, _ = model.pooling1(g, s)
        s1.in_ptr                                      # the eager pass (it also leaves the pooled graph boundaries on the device)
    with pytest.raises(_lib.NativeLibraryError, match="capacity overflow"):
        ops.check_overflow_flags()
    torch.cuda.synchronize()
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(graph, stream=side):
        with torch.no_grad():
            _, s1, _ = model.pooling1(g, s)
            s1.in_ptr
    ops.check_overflow_flags()                         # a capture runs nothing: the flag is still clear
    for _ in range(3):                                 # every replay raises it again, with no host-side registration in between
        graph.replay()
        graph.replay()
        with pytest.raises(_lib.NativeLibraryError, match="capacity overflow"):
            ops.check_overflow_flags()
        ops.check_overflow_flags()                     # read once, reset


def test_family_b_train_step_makes_no_device_to_host_copy(golden_dir, g1):
    """A Family B train step on an arena batch of small graphs enqueues without a single device->host read: checked by
    making every synchronising call an error (torch.cuda.set_sync_debug_mode) around the step."""
    from blackwater.data.arena import GraphArena
    from blackwater.nn import ExpValCircuitGraphModel
    from blackwater.train import Trainer

    xs, eis = [], []
    for i in range(64):
        x, ei, _ = g1_graph(g1, i)
        loops = np.arange(x.shape[0])
        xs.append(x.astype(np.float32))
        eis.append(np.concatenate([ei, np.stack([loops, loops])], axis=1))
    arena = GraphArena.from_arrays(xs, eis, g1["ideal"][:64, None, :].astype(np.float32),
                                   g1["noisy"][:64, None, :].astype(np.float32),
                                   g1["depth"][:64, None].astype(np.float32), np.zeros((64, 1, 1), np.float32), device=DEV)
    torch.manual_seed(0)
    model = ExpValCircuitGraphModel(22, 15, 4).to(DEV)
    trainer = Trainer(model, lr=1e-3)
    trainer.step(arena.batch(np.arange(32)))          # warm-up: allocator growth, lazily built tables
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        loss = trainer.step(arena.batch(np.arange(32, 64)))
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert torch.isfinite(loss).item()


def _big_arena(steps, n_j, filler=0, seed=3):
    from blackwater.data.arena import GraphArena
    from blackwater.data.synthetic import TfimCorpus

    hb = TfimCorpus(100, steps, n_j, seed=seed, exp_value_size=4).host_graphs()
    return GraphArena.from_arrays(hb["x"], hb["edge_index"], hb["y"][:, None, :], hb["noisy"][:, None, :], hb["depth"],
                                  hb["observable"], device=DEV, filler_nodes=filler)


def test_coarsened_edge_capacity_bounds_the_real_count_on_100_qubit_graphs():
    """GraphArena.coarse_caps: the structural bound sum_u (1 + outdeg u) sum_{v in N+[u]} (1 + outdeg v) really is an upper
    bound on ASAPooling's coarsened edge count (whatever the top-k keeps), per graph, and the capacity-sized sync-free form of
    the wave-per-cluster coarsening gives the same arrays as the form that reads the count back."""
    from blackwater.native import functional as F
    from blackwater.nn import ExpValCircuitGraphModel

    arena = _big_arena([1, 3, 6, 10], 2)
    assert arena.coarse_caps is not None and len(arena.coarse_caps) == len(arena.node_counts)
    torch.manual_seed(1)
    model = ExpValCircuitGraphModel(22, 15, 4).to(DEV).eval()
    sel = np.arange(len(arena))
    res = {}
    for use_cap in (True, False):
        b = arena.batch(sel)
        s = b.structure
        assert s.coarse_capacity == int(arena.coarse_caps[sel].sum())
        if not use_cap:
            s.coarse_capacity = None
        with torch.no_grad():
            g = model.transformer1(b.x, s)
            if use_cap:
                torch.cuda.synchronize()
                torch.cuda.set_sync_debug_mode("error")      # no device->host read in the pooling + coarsening
            try:
                g, s1, perm = model.pooling1(g, s)
                _ = s1.in_ptr                                  # builds the (deferred) coarsened connectivity
            finally:
                torch.cuda.set_sync_debug_mode("default")
        res[use_cap] = (s1, perm)
    a, c = res[True][0], res[False][0]
    e = int(c.in_ptr[c.num_nodes].item())
    assert c.num_edges == e and a.num_edges == int(arena.coarse_caps[sel].sum()) >= e > 0
    assert torch.equal(a.in_ptr[:a.num_nodes + 1], c.in_ptr[:c.num_nodes + 1])
    assert torch.equal(a.out_ptr[:a.num_nodes + 1], c.out_ptr[:c.num_nodes + 1])
    assert torch.equal(a.in_src[:e], c.in_src[:e]) and torch.equal(a.out_dst[:e], c.out_dst[:e])
    assert a.out_eid is None and c.out_eid is None     # the list form links nothing by default (recomputed backward forms)
    assert torch.equal(res[True][1], res[False][1])
    # per graph: real coarsened edges <= the graph's own bound
    gp = a.graph_ptr.cpu().numpy()
    per_graph = np.diff(c.out_ptr.cpu().numpy()[gp])
    assert (per_graph <= arena.coarse_caps[sel]).all() and per_graph.sum() == e


def test_family_b_step_on_100_qubit_graphs_is_captured_and_replays_bit_for_bit():
    """The reference's GNN on the headline graphs as a captured step: with the coarsened edge arrays sized by the structural
    bound nothing in the step reads the device, so train.BucketedTrainer captures it (size-stratified batches repeat their
    size pattern) -- replayed losses equal the eagerly enqueued bucketed step's, bit for bit, and an eager step passes under
    torch's sync-debug mode."""
    from blackwater.native import ops
    from blackwater.nn import ExpValCircuitGraphModel
    from blackwater.train import BucketedTrainer, StratifiedBatches

    arena = _big_arena([1, 2, 5], 6, filler=1024)
    n = len(arena)
    runs = {}
    for graphs in (False, True):
        torch.manual_seed(0)
        sampler = StratifiedBatches(arena.node_counts[:n], arena.edge_counts[:n], 6, seed=5)
        bt = BucketedTrainer(ExpValCircuitGraphModel(22, 15, 4).to(DEV), arena, lr=1e-3, graphs=graphs, node_quantum=1024,
                             edge_quantum=2048)
        losses = []
        for k in range(6):
            ids = sampler.draw()
            if not graphs and k == 3:
                torch.cuda.synchronize()
                torch.cuda.set_sync_debug_mode("error")
                try:
                    loss = bt.step_ids(ids)
                finally:
                    torch.cuda.set_sync_debug_mode("default")
            else:
                loss = bt.step_ids(ids)
            losses.append(float(loss))
        runs[graphs] = (losses, bt.flat_param.detach().clone(), len(bt._entries))
        ops.set_seed_counter(None)
        del bt
    assert runs[True][2] == 1                      # one pattern, seen eagerly once, then captured
    assert runs[False][0] == runs[True][0]
    assert torch.equal(runs[False][1], runs[True][1])
    assert all(np.isfinite(runs[True][0]))


def test_family_b_reads_the_arena_rows_through_the_row_map(g1):
    """The first TransformerConv takes a device batch's ``RowsOf`` nodes as they are (mlqem_linear_f32 / mlqem_linear_wgrad_f32 with
    x_rows, 180 output columns: the 16-byte GEMM kernel): outputs and every gradient bit-identical to the materialised copy."""
    from blackwater.data.arena import GraphArena
    from blackwater.native import ops
    from blackwater.nn import ExpValCircuitGraphModel

    xs, eis = [], []
    for i in range(48):
        x, ei, _ = g1_graph(g1, i)
        loops = np.arange(x.shape[0])
        xs.append(x.astype(np.float32))
        eis.append(np.concatenate([ei, np.stack([loops, loops])], axis=1))
    arena = GraphArena.from_arrays(xs, eis, g1["ideal"][:48, None, :].astype(np.float32), g1["noisy"][:48, None, :].astype(np.float32),
                                   g1["depth"][:48, None].astype(np.float32), np.zeros((48, 1, 1), np.float32), device=DEV)
    batch = arena.batch(np.arange(47, -1, -1))            # reversed: the row map is not the identity
    args = list(batch.model_args())
    assert isinstance(args[3], ops.RowsOf)
    torch.manual_seed(0)
    model = ExpValCircuitGraphModel(22, 15, 4).to(DEV).eval()      # eval: no dropout draws, gradients still flow
    res = []
    for nodes in (args[3], args[3].materialize()):
        model.zero_grad(set_to_none=True)
        out = model(args[0], args[1], args[2], nodes, args[4], args[5])
        out.square().sum().backward()
        res.append((out.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]))
    assert torch.equal(res[0][0], res[1][0])
    for ga, gb in zip(res[0][1], res[1][1]):
        assert torch.equal(ga, gb)


@pytest.mark.parametrize("c", [30, 45, 7])
def test_segment_max_backward_takes_its_tie_counts_from_the_softmax_walk(g1, c):
    """mlqem_csr_softmax_aggregate_bwd_f32 with xmax / tie_count and mlqem_csr_segment_max_bwd_f32 with tie_count: the same gx,
    bit for bit, as the segment max's own destination-side walk -- on features quantised so that maxima are attained several
    times (circuit graphs do that: identical gates on one qubit have identical rows)."""
    from blackwater.native import ops
    from blackwater.native.structure import GraphStructure

    b = g1_batch(g1, list(range(40)), self_loops=False, first_only=False)
    n = b["x"].shape[0]
    s = GraphStructure.from_edge_index(b["edge_index"].to(DEV), n, batch=b["batch"].to(DEV), num_graphs=40)
    g = torch.Generator().manual_seed(c)
    x = ops.padded_copy((torch.randint(0, 3, (n, c), generator=g).float() * 0.5).to(DEV))
    xmax = ops.csr_segment_max(x, s.in_ptr, s.in_src, ell=s.in_ell)
    a_dst, c_src = torch.randn(n, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    x_new = ops.csr_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, 0.2)
    gnew = ops.padded_copy(torch.randn(n, c, generator=g).to(DEV))
    gmax = ops.padded_copy(torch.randn(n, c, generator=g).to(DEV))
    e = s.edge_count()
    gx0, ga0, gc0 = ops.csr_softmax_aggregate_bwd(x, x_new, gnew, s, e, a_dst, c_src, 0.2)
    gx1, ga1, gc1, ties = ops.csr_softmax_aggregate_bwd(x, x_new, gnew, s, e, a_dst, c_src, 0.2, xmax=xmax)
    assert torch.equal(gx0, gx1) and torch.equal(ga0, ga1) and torch.equal(gc0, gc1)
    # the counts against a direct enumeration
    src, dst = b["edge_index"][0].to(DEV), b["edge_index"][1].to(DEV)
    want = (x == xmax).float()
    want.index_add_(0, dst, (x[src] == xmax[dst]).float())
    assert torch.equal(ties, want) and ties.max().item() >= 2
    ops.csr_segment_max_bwd_(gx0, x, xmax, gmax, s)
    ops.csr_segment_max_bwd_(gx1, x, xmax, gmax, s, ties=ties)
    assert torch.equal(gx0, gx1)


@pytest.mark.parametrize("heads,ch", [(3, 15), (2, 15), (5, 25), (3, 25), (1, 16), (2, 32), (4, 3), (1, 1), (2, 17)])
def test_attention_kernels_against_a_dense_reference(heads, ch):
    """mlqem_transformer_attention_train_f32 / _bwd_f32 / the inference entry point (four channels per lane: 4 or 8 lanes per
    (row, head), the last lane of a head partly filled for C = 15 / 25 / 17 / 3 / 1) against TransformerConv's formulas in
    fp64 through torch autograd: rows of 0-40 in-edges (one to eleven chunks), repeated edges, rows with and without a
    self-loop entry, NaN-poisoned row padding."""
    from blackwater.native import ops
    from blackwater.native.structure import GraphStructure

    rng = np.random.RandomState(heads * 100 + ch)
    n = 300
    deg = rng.choice([0, 1, 2, 3, 4, 5, 8, 9, 17, 40], size=n)
    dst = np.repeat(np.arange(n), deg)
    src = rng.randint(0, n, size=dst.shape[0])
    src = np.where(src == dst, (src + 1) % n, src)                      # self-loops are the structure's own entry
    loops = rng.randint(0, 3, size=n)                                   # multiplicity 0, 1, 2 of the self entry
    ei = np.concatenate([np.stack([src, dst]), np.repeat(np.stack([np.arange(n)] * 2), loops, axis=1)], axis=1)
    s = GraphStructure.from_edge_index(torch.from_numpy(ei).to(DEV), n)
    assert int(s.loops.sum().item()) == int(loops.sum()) and s.edge_count() == dst.shape[0]
    hc = heads * ch
    g = torch.Generator().manual_seed(ch)
    qkvs_h = torch.randn(n, 4 * hc, generator=g)
    gout_h = torch.randn(n, hc, generator=g)

    def poisoned(t):
        buf = torch.full((t.shape[0], (t.shape[1] + 3) // 4 * 4 + 4), float("nan"), device=DEV)     # pads AND slack after the row
        buf[:, :t.shape[1]] = t.to(DEV)
        return buf[:, :t.shape[1]]

    qkvs = ops.padded_copy(qkvs_h.to(DEV))            # [N, 4 H C]: no padding inside (4 H C is a multiple of 4)
    out, attn, m, den = ops.transformer_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, s.edge_count(), heads, ch)
    gq = ops.transformer_attention_bwd(qkvs, poisoned(gout_h), attn, m, den, s, s.edge_count(), heads, ch)
    inf = ops.transformer_attention(qkvs, s.in_ptr, s.in_src, s.loops, heads, ch)
    # reference: softmax over the in-edges and the self entry (multiplicity = repeated entry), fp64
    x = qkvs_h.double().requires_grad_(True)
    q, k, v, skip = (x[:, i * hc:(i + 1) * hc].reshape(n, heads, ch) for i in range(4))
    e_src = torch.from_numpy(ei[0]); e_dst = torch.from_numpy(ei[1])
    score = (q[e_dst] * k[e_src]).sum(-1) / ch ** 0.5                                   # [E', H]
    mx = torch.full((n, heads), -float("inf"), dtype=torch.float64).scatter_reduce(0, e_dst[:, None].expand(-1, heads), score.detach(), "amax")
    p = torch.exp(score - mx[e_dst])
    denom = torch.zeros(n, heads, dtype=torch.float64).index_add(0, e_dst, p) + 1e-16
    alpha = p / denom[e_dst]
    agg = torch.zeros(n, heads, ch, dtype=torch.float64).index_add(0, e_dst, alpha[:, :, None] * v[e_src])
    want = (agg + skip).reshape(n, hc)
    want.backward(gout_h.double())
    scale = max(1.0, want.detach().abs().max().item())
    assert (out.cpu().double() - want.detach()).abs().max().item() < 2e-5 * scale
    assert (inf.cpu().double() - want.detach()).abs().max().item() < 2e-5 * scale
    assert torch.equal(out, inf)
    gscale = max(1.0, x.grad.abs().max().item())
    assert (gq.cpu().double() - x.grad).abs().max().item() < 5e-5 * gscale


@pytest.mark.parametrize("heads,ch,linked", [(3, 15, True), (2, 15, False), (2, 13, True)])
def test_attention_with_a_head_pitch_of_16_equals_the_compact_layout(heads, ch, linked):
    """``head_pitch = 16``: q / k / v / skip as [N, 4 H 16] with zero pads (what a projection with padded weight rows writes) -- the same
    output bit for bit, the same gradient in the real channels, zeros in the pads; stored and recomputed backward forms, with dropout."""
    from blackwater.native import ops
    from blackwater.native.structure import GraphStructure

    rng = np.random.RandomState(heads * 10 + ch)
    n, hc, cp = 700, heads * ch, 16
    pairs = set()
    for i in range(n):
        for j in rng.choice(n, size=rng.randint(0, 40 if i % 7 == 0 else 3), replace=False):
            if i != j:
                pairs.add((int(j), i))
    ei = np.array(sorted(pairs)).T
    s = GraphStructure.from_edge_index(torch.from_numpy(ei).to(DEV), n)
    if not linked:
        s.out_eid = None
    pk = not linked
    g = torch.Generator().manual_seed(ch)
    qkvs_h, gout_h = torch.randn(n, 4 * hc, generator=g), torch.randn(n, hc, generator=g)
    qkvs = ops.padded_copy(qkvs_h.to(DEV))
    qp_h = torch.zeros(n, 4 * heads, cp)
    qp_h[:, :, :ch] = qkvs_h.view(n, 4 * heads, ch)
    qkvs_p = ops.padded_copy(qp_h.view(n, 4 * heads * cp).to(DEV))
    e = s.edge_count()
    for drop_p in (0.0, 0.2):
        a = ops.transformer_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, drop_p, 7, pair_key=pk, ell=s.in_ell)
        b = ops.transformer_attention_train(qkvs_p, s.in_ptr, s.in_src, s.loops, e, heads, ch, drop_p, 7, pair_key=pk, ell=s.in_ell, head_pitch=cp)
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        ga = ops.transformer_attention_bwd(qkvs, gout_h.to(DEV), a[1], a[2], a[3], s, e, heads, ch, drop_p, 7, pair_key=pk)
        gb = ops.transformer_attention_bwd(qkvs_p, gout_h.to(DEV), b[1], b[2], b[3], s, e, heads, ch, drop_p, 7, pair_key=pk, head_pitch=cp)
        gb3 = gb.view(n, 4 * heads, cp)
        assert torch.equal(gb3[:, :, :ch].reshape(n, 4 * hc), ga)
        assert (gb3[:, :, ch:] == 0).all()


@pytest.mark.parametrize("heads,ch", [(2, 15), (3, 15), (3, 25), (1, 16), (2, 17), (4, 3)])
def test_recomputed_attention_backward_equals_the_stored_form(heads, ch):
    """A structure without out_eid (ASAPooling's coarsened graphs since round 4) takes the RECOMPUTED source side of
    mlqem_transformer_attention_bwd_f32: no per-edge buffers, weights recomputed from m / den / g . attn_out per (row, head).  Same
    gradients as the stored form (through out_eid) within 2e-5 of the gradient scale -- without dropout, and with dropout keyed by
    (destination, head, source) in both forms (same masks from either end of an edge); the forward under that key drops the
    stated fraction of the weights.  Graph: no parallel edges, rows of 0-60 in-edges, self entries of multiplicity 0 / 1."""
    import copy

    from blackwater.native import ops
    from blackwater.native.structure import GraphStructure

    rng = np.random.RandomState(heads * 100 + ch)
    n = 400
    deg = rng.choice([0, 1, 2, 3, 4, 5, 8, 9, 17, 40, 60], size=n)
    src = np.concatenate([rng.choice(n - 1, size=d, replace=False) for d in deg]) if deg.sum() else np.zeros(0, dtype=np.int64)
    dst = np.repeat(np.arange(n), deg)
    src = np.where(src >= dst, src + 1, src)                            # distinct sources per row, never the row itself
    loops = rng.randint(0, 2, size=n)
    ei = np.concatenate([np.stack([src, dst]), np.repeat(np.stack([np.arange(n)] * 2), loops, axis=1)], axis=1)
    s = GraphStructure.from_edge_index(torch.from_numpy(ei).to(DEV), n)
    s_rc = copy.copy(s)
    s_rc.out_eid = None
    hc = heads * ch
    g = torch.Generator().manual_seed(ch)
    qkvs = ops.padded_copy(torch.randn(n, 4 * hc, generator=g).to(DEV))
    gout = ops.padded_copy(torch.randn(n, hc, generator=g).to(DEV))
    e = s.edge_count()
    for drop_p in (0.0, 0.3):
        out, attn, m, den = ops.transformer_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, drop_p, 99, pair_key=True)
        stored = ops.transformer_attention_bwd(qkvs, gout, attn, m, den, s, e, heads, ch, drop_p, 99, pair_key=True)
        again = ops.transformer_attention_bwd(qkvs, gout, attn, m, den, s_rc, e, heads, ch, drop_p, 99, pair_key=True)
        scale = max(1.0, stored[:, :4 * hc].abs().max().item())
        assert (stored[:, :4 * hc] - again[:, :4 * hc]).abs().max().item() < 2e-5 * scale, drop_p
        assert torch.isfinite(again[:, :4 * hc]).all()
        if drop_p == 0.0:
            by_pos = ops.transformer_attention_bwd(qkvs, gout, attn, m, den, s, e, heads, ch)      # position key: no draws, same numbers
            assert torch.equal(by_pos[:, :4 * hc], stored[:, :4 * hc])
            plain = out
        else:
            assert not torch.equal(out, plain)                         # weights were dropped
    with pytest.raises(Exception):                                     # a position-keyed draw cannot be found from the source side
        ops.transformer_attention_bwd(qkvs, gout, attn, m, den, s_rc, e, heads, ch, 0.3, 99, pair_key=False)


@pytest.mark.parametrize("c", [30, 45, 7, 64, 100])
def test_recomputed_softmax_aggregate_backward_equals_the_stored_form(c):
    """mlqem_csr_softmax_aggregate_bwd_f32 without out_eid: the source side recomputes al / gp from one record {a_i, m_i, 1 / den_i,
    delta_i} per destination row.  Same gx, g_a, g_c as the stored form within 2e-5 of their scales; tie counts unchanged."""
    import copy

    from blackwater.native import ops
    from blackwater.native.structure import GraphStructure

    rng = np.random.RandomState(c)
    n = 500
    deg = rng.choice([0, 1, 2, 3, 5, 8, 16, 17, 33, 70], size=n)
    src = np.concatenate([rng.choice(n - 1, size=d, replace=False) for d in deg])
    dst = np.repeat(np.arange(n), deg)
    src = np.where(src >= dst, src + 1, src)
    s = GraphStructure.from_edge_index(torch.from_numpy(np.stack([src, dst])).to(DEV), n)
    s_rc = copy.copy(s)
    s_rc.out_eid = None
    g = torch.Generator().manual_seed(c)
    x = ops.padded_copy(torch.randn(n, c, generator=g).to(DEV))
    a_dst, c_src = torch.randn(n, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    x_new = ops.csr_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, 0.2)
    xmax = ops.csr_segment_max(x, s.in_ptr, s.in_src, ell=s.in_ell)
    gnew = ops.padded_copy(torch.randn(n, c, generator=g).to(DEV))
    e = s.edge_count()
    gx0, ga0, gc0, t0 = ops.csr_softmax_aggregate_bwd(x, x_new, gnew, s, e, a_dst, c_src, 0.2, xmax=xmax)
    gx1, ga1, gc1, t1 = ops.csr_softmax_aggregate_bwd(x, x_new, gnew, s_rc, e, a_dst, c_src, 0.2, xmax=xmax)
    for a, b in ((gx0[:, :c], gx1[:, :c]), (ga0, ga1), (gc0, gc1)):
        assert (a - b).abs().max().item() < 2e-5 * max(1.0, a.abs().max().item())
    assert torch.equal(t0[:, :c], t1[:, :c])


@pytest.mark.parametrize("d", [45, 30, 7, 64])
def test_fused_pooling_forward_equals_the_chain_of_kernels_it_stands_for(d):
    """Round 5: on graphs of short rows ASAPooling's forward up to the fitness projections is one pass (mlqem_asap_scores_fused_f32) --
    against the chain it replaces (segment max, the composed score projection, c = x att_x, score softmax + cluster sum, LEConv's three
    projections) on a graph of rows with 0-2 entries and a few rows of up to 150: the maxima exactly, everything else to 1e-5 of
    its scale; and the model-level results agree with the switch off (functional._ASAP_FUSED)."""
    import numpy as np

    from blackwater.native import ops
    from blackwater.native.structure import GraphStructure

    rng = np.random.RandomState(d)
    n = 4000
    deg = rng.choice([0, 1, 2, 2, 2, 1], size=n)
    deg[rng.choice(n, 12, replace=False)] = rng.randint(3, 150, size=12)
    dst = np.repeat(np.arange(n), deg)
    src = np.concatenate([rng.choice(n, k, replace=False) for k in deg]) if deg.sum() else np.zeros(0, np.int64)
    keep = src != dst
    ei = torch.from_numpy(np.stack([src[keep], dst[keep]]).astype(np.int64)).to(DEV)
    s = GraphStructure.from_edge_index(ei, n)
    x = ops.padded_copy(torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(DEV))
    w_comp = torch.from_numpy(rng.standard_normal((1, d)).astype(np.float32)).to(DEV)
    b_comp = torch.from_numpy(rng.standard_normal(1).astype(np.float32)).to(DEV)
    att_x = torch.from_numpy(rng.standard_normal((1, d)).astype(np.float32)).to(DEV)
    w3 = torch.from_numpy(rng.standard_normal((3, d)).astype(np.float32)).to(DEV)
    b3 = torch.from_numpy(rng.standard_normal(3).astype(np.float32)).to(DEV)
    xmax_r = ops.csr_segment_max(x, s.in_ptr, s.in_src)
    a_r = ops.linear(xmax_r, w_comp, b_comp)[:, 0].contiguous()
    c_r = ops.linear(x, att_x)[:, 0].contiguous()
    xnew_r = ops.csr_softmax_aggregate(x, s.in_ptr, s.in_src, a_r, c_r, 0.2)
    pqr_r = ops.linear(xnew_r, w3, b3)[:, :3]
    xmax, a_dst, c_src, xnew, pqr = ops.asap_scores_fused(x, s.in_ptr, s.in_src, w_comp, b_comp, att_x, w3, b3, 0.2)
    assert torch.equal(xmax[:, :d], xmax_r[:, :d])
    tol = lambda t: 1e-5 * max(1.0, t.abs().max().item())
    assert (a_dst - a_r).abs().max().item() < tol(a_r) and (c_src - c_r).abs().max().item() < tol(c_r)
    assert (xnew[:, :d] - xnew_r[:, :d]).abs().max().item() < tol(xnew_r) and (pqr - pqr_r).abs().max().item() < tol(pqr_r)
    # ... and the backward: the segment max's gradient carried by the source-side walk of the cluster sum's backward (fuse_max_col)
    # against the two walks it stands for; x quantised so that maxima tie (the gradient is split evenly)
    xq = ops.padded_copy(torch.from_numpy(rng.randint(0, 4, size=(n, d)).astype(np.float32)).to(DEV))
    xmax_q = ops.csr_segment_max(xq, s.in_ptr, s.in_src)
    xnew_q = ops.csr_softmax_aggregate(xq, s.in_ptr, s.in_src, a_r, c_r, 0.2)
    gnew = ops.padded_copy(torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(DEV))
    e = s.edge_count()
    gx_r, ga_r, gc_r, ties_r = ops.csr_softmax_aggregate_bwd(xq, xnew_q, gnew, s, e, a_r, c_r, 0.2, xmax=xmax_q, gx_rank1=att_x[0])
    ops.csr_segment_max_bwd_(gx_r, xq, xmax_q, None, s, ties=ties_r, gmax_rank1=(ga_r, w_comp[0].contiguous()))
    gx, ga, gc, ties = ops.csr_softmax_aggregate_bwd(xq, xnew_q, gnew, s, e, a_r, c_r, 0.2, xmax=xmax_q, gx_rank1=att_x[0],
                                                     fuse_max_col=w_comp[0].contiguous())
    assert torch.equal(ties[:, :d], ties_r[:, :d]) and torch.equal(ga, ga_r) and torch.equal(gc, gc_r)
    assert (gx[:, :d] - gx_r[:, :d]).abs().max().item() < tol(gx_r[:, :d])


def test_family_b_with_and_without_the_fused_pooling_forward(g1):
    import blackwater.native.functional as F
    from blackwater.data.arena import GraphArena
    from blackwater.nn import ExpValCircuitGraphModel

    xs, eis = zip(*[g1_graph(g1, i)[:2] for i in range(40)])
    noisy = np.asarray([g1["noisy"][i] for i in range(40)], dtype=np.float32)[:, None, :]
    arena = GraphArena.from_arrays([np.asarray(a, np.float32) for a in xs], [np.asarray(e) for e in eis], noisy, noisy, np.zeros((40, 1), np.float32),
                                   np.zeros((40, 1, 1), np.float32), device=DEV)
    res = {}
    for fused in (True, False):
        torch.manual_seed(3)
        model = ExpValCircuitGraphModel(22, 15, noisy.shape[2]).to(DEV).train()
        was = F._ASAP_FUSED
        F._ASAP_FUSED = fused
        try:
            out = model(*arena.batch(np.arange(40)).model_args())
            out.square().sum().backward()
        finally:
            F._ASAP_FUSED = was
        res[fused] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
    (oa, ga), (ob, gb) = res[True], res[False]
    assert (oa - ob).abs().max().item() < 2e-5 * max(1.0, ob.abs().max().item())
    gmax = max(v.abs().max().item() for v in gb.values())
    for k in gb:
        assert (ga[k] - gb[k]).abs().max().item() < 2e-4 * gmax, k


def test_list_coarsening_equals_the_bit_matrix_form_on_a_64_circuit_batch():
    """Scale matters to the persistent kernels of the list coarsening (batches of 64 clusters per wave, partial last batches, several
    batches per wave): the first pooling of 64 100-qubit circuits (363 k clusters, ~20 M coarsened edges) must give the arrays of the
    round-3 bit-matrix form exactly.  (A variant of coarsen_unique_kernel passed every smaller comparison and failed this one.)"""
    from blackwater.data.arena import GraphArena
    from blackwater.data.synthetic import TfimCorpus
    from blackwater.native import functional as F
    from blackwater.nn import ExpValCircuitGraphModel

    h = TfimCorpus(100, list(range(1, 11)), 7, seed=42, exp_value_size=4).host_graphs()
    arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"], device=DEV)
    b = arena.batch(np.random.RandomState(0).randint(0, len(arena), size=64))
    torch.manual_seed(0)
    model = ExpValCircuitGraphModel(22, 15, 4).to(DEV).eval()
    res = {}
    keep = F._ASAP_LISTS
    try:
        with torch.no_grad():
            g = model.transformer1(b.nodes, b.structure)
            for lists in (True, False):
                F._ASAP_LISTS = lists
                _, s, perm = model.pooling1(g, b.structure)
                k = s.num_nodes
                e = int(s.in_ptr[k].item())
                res[lists] = (s.in_ptr[:k + 1].clone(), s.in_src[:e].clone(), s.out_ptr[:k + 1].clone(), s.out_dst[:e].clone(), perm.clone())
                del s
    finally:
        F._ASAP_LISTS = keep
    for a, c in zip(res[True], res[False]):
        assert a.shape == c.shape and torch.equal(a, c)
