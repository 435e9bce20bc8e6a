"""Family A (GCN || Cheb || SAGE) on the GPU vs the CPU oracle: forward within 1e-5, gradients of every
parameter, on batches collated the way the reference's training path collates them."""
import os

import numpy as np
import pytest
import torch

from helpers import g1_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ARGS = ("noisy", "observable", "depth", "x", "edge_index", "batch")


def _models(seed=0, hidden=10):
    from blackwater.nn import ExpValCircuitGraphModelA
    from oracle.models import FamilyA

    torch.manual_seed(seed)
    model = ExpValCircuitGraphModelA(5, 22, hidden)
    # GCN/Cheb biases start at zero; give them values so their gradients/paths are exercised
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bias"):
                p.uniform_(-0.5, 0.5)
    ref = FamilyA(5, 22, hidden).double()
    ref.load_state_dict(model.state_dict(), strict=True)
    return model.to(DEV), ref


@pytest.mark.parametrize("self_loops", [True, False])
@pytest.mark.parametrize("count", [1, 32, 300])
def test_forward_matches_oracle(g1, self_loops, count):
    model, ref = _models()
    batch = g1_batch(g1, range(count), self_loops=self_loops)
    out = model.eval()(*[batch[k].to(DEV) for k in ARGS])
    want = ref.eval()(*[batch[k].double() if batch[k].is_floating_point() else batch[k] for k in ARGS])
    assert out.shape == (count, 1)
    assert (out.detach().cpu().double() - want.detach()).abs().max().item() < 1e-5  # north_star tolerance


def test_gradients_match_oracle(g1):
    model, ref = _models(seed=1)
    batch = g1_batch(g1, range(40, 104))
    model.eval(), ref.eval()  # dropout off; gradients still flow
    out = model(*[batch[k].to(DEV) for k in ARGS])
    torch.nn.functional.mse_loss(out, batch["y"].to(DEV)).backward()
    want = ref(*[batch[k].double() if batch[k].is_floating_point() else batch[k] for k in ARGS])
    torch.nn.functional.mse_loss(want, batch["y"].double()).backward()
    ref_grads = dict(ref.named_parameters())
    for name, p in model.named_parameters():
        g_ref = ref_grads[name].grad
        scale = g_ref.abs().max().item() + 1e-9
        err = (p.grad.cpu().double() - g_ref).abs().max().item() / scale
        assert err < 1e-4, f"{name}: relative grad error {err}"


@pytest.mark.parametrize("hidden", [3, 40, 80])
def test_other_hidden_widths(g1, hidden):
    """Hidden widths around the limits of the column-block GEMM (64 concatenated input columns; 16-column output tiles):
    3 (one partial tile), 40 (three blocks = 120 output columns), 80 (per-block fallback launches)."""
    model, ref = _models(seed=2, hidden=hidden)
    batch = g1_batch(g1, range(100, 148))
    model.eval(), ref.eval()
    out = model(*[batch[k].to(DEV) for k in ARGS])
    want = ref(*[batch[k].double() if batch[k].is_floating_point() else batch[k] for k in ARGS])
    assert (out.detach().cpu().double() - want.detach()).abs().max().item() < 1e-5
    torch.nn.functional.mse_loss(out, batch["y"].to(DEV)).backward()
    torch.nn.functional.mse_loss(want, batch["y"].double()).backward()
    ref_grads = dict(ref.named_parameters())
    for name, p in model.named_parameters():
        g_ref = ref_grads[name].grad
        scale = g_ref.abs().max().item() + 1e-9
        assert (p.grad.cpu().double() - g_ref).abs().max().item() / scale < 1e-4, name


def test_single_node_and_per_layer_paths_agree(g1):
    """The graph part as ONE autograd node (default; the last conv of every branch folded into its pool as a weighted
    mean, native/functional.py _FamilyAGraph) and as one node per layer (conv3 / cheb_conv2 / sage_conv2 run as layers,
    then a plain mean pool) compute the same function with the same dropout masks: outputs and every gradient agree to
    fp32 rounding, in train mode (dropout on) too."""
    model, _ = _models(seed=4)
    batch = g1_batch(g1, range(20, 84))
    args = [batch[k].to(DEV) for k in ARGS]
    results = []
    for single in (True, False):
        model.single_node = single
        model.train()
        model._step = 0                      # same dropout seeds in both runs: the conv layers' counter-based masks, ...
        model.obs_seq._calls = model.body_seq._calls = 0      # ... the call counters the fused heads key their masks with ...
        torch.manual_seed(123)               # ... and torch's generator for the unfused nn.Dropout path
        model.zero_grad()
        out = model(*args)
        out.square().mean().backward()
        results.append((out.detach().clone(), [p.grad.clone() for p in model.parameters()]))
    assert (results[0][0] - results[1][0]).abs().max().item() < 2e-6 * max(1.0, results[1][0].abs().max().item())
    for (name, _), a, b in zip(model.named_parameters(), results[0][1], results[1][1]):
        assert (a - b).abs().max().item() <= 2e-5 * (b.abs().max().item() + 1e-9), name


def test_train_mode_dropout_runs_and_is_seeded(g1):
    """Dropout masks are keyed by torch's seed (nn/models.py dropout_key): the same seed reproduces a run bit for bit,
    another seed draws other masks -- the behaviour of the reference's nn.Dropout under torch.manual_seed."""
    batch = g1_batch(g1, range(32))
    outs = []
    for seed in (11, 11, 12):
        model, _ = _models()
        args = [batch[k].to(DEV) for k in ARGS]
        torch.manual_seed(seed)
        model.train()
        out = model(*args)
        torch.nn.functional.mse_loss(out, batch["y"].to(DEV)).backward()
        assert all(torch.isfinite(p.grad).all() for p in model.parameters())
        outs.append(out.detach().clone())
    assert torch.equal(outs[0], outs[1])
    assert not torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("hidden", [10, 16])
def test_pooled_gradient_computed_in_its_aggregation_equals_the_written_one(g1, hidden, monkeypatch):
    """native/functional.py _POOLED_GRAD: the gradient of every branch's last hidden activation computed inside its first transposed
    aggregation (ops.PooledGrad, csrc/pooled_grad.hip) instead of written by ops.segment_pool_bwd and gathered: same arithmetic in
    the same order, so every parameter gradient is bit-equal -- but the bias of the GCN branch's second conv, whose gradient is a
    column sum taken from the gate bits in another order (the gradient matrix is not written at all there).  Train mode (dropout
    gates in the bits), batch of 64 graphs."""
    from blackwater.native import functional as F, ops

    model, _ = _models(seed=5, hidden=hidden)
    batch = g1_batch(g1, range(10, 74))
    args = [batch[k].to(DEV) for k in ARGS]
    calls = []
    real = ops.PooledGrad.aggregate
    monkeypatch.setattr(ops.PooledGrad, "aggregate", lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1])
    results = []
    monkeypatch.setattr(F, "_POOLED_GRAD_MIN_NODES", 0)      # small batches keep the written form by default (launch counts)
    for on in (True, False):
        monkeypatch.setattr(F, "_POOLED_GRAD", on)
        model.train()
        model._step = 0
        model.obs_seq._calls = model.body_seq._calls = 0
        torch.manual_seed(7)
        model.zero_grad()
        out = model(*args)
        out.square().mean().backward()
        results.append([p.grad.clone() for p in model.parameters()])
    assert len(calls) == 3                      # the three branches, in the first run only
    inexact = 0
    for (name, _), a, b in zip(model.named_parameters(), results[0], results[1]):
        if not torch.equal(a, b):
            inexact += 1
            assert name.endswith("bias") and (a - b).abs().max().item() <= 1e-6 * (b.abs().max().item() + 1e-12), name
    assert inexact <= 1
