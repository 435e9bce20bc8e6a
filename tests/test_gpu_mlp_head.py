"""The one-launch MLP head (csrc/mlp_head.hip: mlqem_mlp1_forward / mlqem_mlp1_backward) against torch fp64 algebra.

Reference: docs/tutorials/mlp.py:18-30 (MLP1 = fc2(relu(fc1 x))).  Stated tolerances:
* fp32 mode (exact fp32 on v_mfma_f32_16x16x4_f32): outputs within 1e-5 of the output scale of the fp64 value, parameter
  gradients within 1e-5 of their scale (Frobenius) -- the same bar as every other fp32 kernel of the path;
* bf16 mode (operands rounded to bf16, fp32 accumulation, bf16 stash): every product equals "round both operands to bf16,
  multiply exactly, accumulate" to 1e-5 of its scale GIVEN the stash the device wrote; the stash itself equals
  bf16(relu(bf16(x) bf16(W1)^T + b1)) up to one bf16 ulp on values that sit on a rounding boundary (fp32 vs fp64 sums).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(170, 128, 1), (58, 64, 4), (169, 64, 1), (7, 5, 2), (175, 128, 3), (64, 128, 1), (33, 17, 4)]
ROWS = [1, 33, 1000, 4099]


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _make(i, h, o2, n, seed=0, nan_pads=True):
    g = torch.Generator().manual_seed(seed + 1000 * i + h + n)
    x = torch.randn(n, i, generator=g)
    w1 = torch.randn(h, i, generator=g) / i ** 0.5
    b1 = torch.randn(h, generator=g) * 0.1
    w2 = torch.randn(o2, h, generator=g) / h ** 0.5
    b2 = torch.randn(o2, generator=g) * 0.1
    gout = torch.randn(n, o2, generator=g)
    from blackwater.native import ops

    xd = ops.padded_empty(n, i, DEV)
    if nan_pads and n > 1 and i % 4:       # pad columns may hold anything: poison them
        torch.as_strided(xd, (n, (i + 3) // 4 * 4), (xd.stride(0), 1)).fill_(float("nan"))
    xd.copy_(x)
    return x, w1, b1, w2, b2, gout, xd


@pytest.mark.parametrize("i,h,o2", SHAPES)
def test_fp32_head_forward_and_backward_equal_fp64_algebra(i, h, o2):
    from blackwater.native import ops

    for n in ROWS:
        x, w1, b1, w2, b2, gout, xd = _make(i, h, o2, n)
        out, hs, xp = ops.mlp1_forward(xd, w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV))
        X, W1, B1, W2, B2, G = (t.double() for t in (x, w1, b1, w2, b2, gout))
        H = torch.relu(X @ W1.T + B1)
        want = H @ W2.T + B2
        scale = want.abs().max().item() + 1e-30
        assert (out.cpu().double() - want).abs().max().item() <= 1e-5 * scale
        assert tuple(hs.shape) == (n, 128) and hs.dtype == torch.float32
        assert (hs[:, :h].cpu().double() - H).abs().max().item() <= 1e-5 * (H.abs().max().item() + 1e-30)
        assert (hs[:, h:] == 0).all()       # the padded hidden units are exact zeros
        gw1, gb1, gw2, gb2 = ops.mlp1_backward(gout.to(DEV), xp, hs, w2.to(DEV), i, h)
        GH = (G @ W2) * (H > 0)
        for got, ref, name in ((gw1, GH.T @ X, "gw1"), (gb1, GH.sum(0), "gb1"), (gw2, G.T @ H, "gw2"), (gb2, G.sum(0), "gb2")):
            err = (got.cpu().double() - ref).norm().item()
            assert err <= 1e-5 * (ref.norm().item() + 1e-30), (name, n, err, ref.norm().item())


@pytest.mark.parametrize("i,h,o2", SHAPES)
def test_bf16_head_equals_rounded_operand_products(i, h, o2):
    from blackwater.native import ops

    for n in ROWS:
        x, w1, b1, w2, b2, gout, xd = _make(i, h, o2, n, seed=7)
        out, hs, xp = ops.mlp1_forward(xd, w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV), bf16=True)
        assert hs.dtype == torch.bfloat16 and tuple(hs.shape) == (n, 128)
        pre = _bf(x) @ _bf(w1).T + b1.double()
        want_h = torch.relu(pre).to(torch.float32).to(torch.bfloat16)
        got_h = hs[:, :h].cpu()
        # one bf16 ulp where the fp32 sum and the fp64 sum fall on different sides of a rounding boundary (or of zero)
        diff = (got_h.double() - want_h.double()).abs()
        ulp = torch.maximum(want_h.double().abs(), got_h.double().abs()) * 2.0 ** -7 + 1e-6 * pre.abs().max().item()
        assert (diff <= ulp).all()
        assert (diff > 0).double().mean().item() < 0.01
        assert (hs[:, h:].float() == 0).all()
        Hs = got_h.double()                  # everything downstream is exact GIVEN the stash
        want = Hs @ _bf(w2).T + b2.double()
        assert (out.cpu().double() - want).abs().max().item() <= 1e-5 * (want.abs().max().item() + 1e-30)
        gw1, gb1, gw2, gb2 = ops.mlp1_backward(gout.to(DEV), xp, hs, w2.to(DEV), i, h, bf16=True)
        GH = _bf(((_bf(gout) @ _bf(w2)) * (Hs > 0)).float())
        refs = ((gw1, GH.T @ _bf(x), "gw1"), (gb1, GH.sum(0), "gb1"), (gw2, _bf(gout).T @ Hs, "gw2"), (gb2, gout.double().sum(0), "gb2"))
        for got, ref, name in refs:
            err = (got.cpu().double() - ref).norm().item()
            assert err <= 1e-5 * (ref.norm().item() + 1e-30), (name, n, err, ref.norm().item())


def test_mlp1_module_takes_the_fused_path_and_matches_the_oracle():
    """MLP1(58, 64, 4) (the architecture of mlp1_smaller_2.pth) and the demo width (170, 128, 1): forward and every parameter
    gradient against the CPU oracle in fp64; the per-layer path (MLQEM_MLP1_FUSED=0 semantics: x.requires_grad) agrees."""
    from blackwater.nn.mlp import MLP1
    from oracle.models import MLP1 as OracleMLP1

    for (i, h, o2), n in (((58, 64, 4), 300), ((170, 128, 1), 2051)):
        torch.manual_seed(3)
        model = MLP1(i, h, o2)
        ref = OracleMLP1(i, h, o2).double()
        ref.load_state_dict({k: v.double() for k, v in model.state_dict().items()})
        model = model.to(DEV)
        x = torch.randn(n, i)
        y = torch.randn(n, o2)
        out = model(x.to(DEV))
        loss = torch.nn.functional.mse_loss(out, y.to(DEV))
        loss.backward()
        want = ref(x.double())
        torch.nn.functional.mse_loss(want, y.double()).backward()
        assert (out.detach().cpu().double() - want.detach()).abs().max().item() < 1e-5 * want.abs().max().item()
        for (name, p), q in zip(model.named_parameters(), ref.parameters()):
            assert (p.grad.cpu().double() - q.grad).norm().item() <= 1e-5 * q.grad.norm().item() + 1e-12, name
        # the per-layer path (taken when the input needs a gradient) gives the same parameter gradients
        fused = [p.grad.clone() for p in model.parameters()]
        model.zero_grad()
        xg = x.to(DEV).requires_grad_(True)
        torch.nn.functional.mse_loss(model(xg), y.to(DEV)).backward()
        for f, p in zip(fused, model.parameters()):
            assert (f - p.grad).norm().item() <= 2e-5 * f.norm().item() + 1e-12
        assert xg.grad is not None


def test_inference_call_writes_no_stash_and_empty_input_is_fine():
    from blackwater.native import ops
    from blackwater.nn.mlp import MLP1

    torch.manual_seed(0)
    model = MLP1(170, 128, 1).to(DEV).eval()
    x = torch.randn(77, 170, device=DEV)
    with torch.no_grad():
        a = model(x)
    out, hs, _ = ops.mlp1_forward(x, model.fc1.weight, model.fc1.bias, model.fc2.weight, model.fc2.bias, stash=False)
    assert hs is None and torch.equal(a, out)
    with torch.no_grad():
        assert model(torch.empty(0, 170, device=DEV)).shape == (0, 1)
    with pytest.raises(Exception):
        ops.mlp1_forward(torch.randn(4, 300, device=DEV), torch.randn(8, 300, device=DEV), torch.randn(8, device=DEV),
                         torch.randn(1, 8, device=DEV), torch.randn(1, device=DEV))      # I > 175: not this kernel's shape


@pytest.mark.parametrize("kind", ["mlp1", "mlp3"])
def test_rows_trainer_replays_the_step_from_a_graph_bit_for_bit(kind):
    """train.RowsTrainer: the whole MLP step (forward, MSE, backward, Adam) captured in a hipGraph equals the same step
    enqueued eagerly, loss by loss -- with BatchNorm statistics and dropout masks (MLP3) advancing per replay."""
    from blackwater.native import ops
    from blackwater.nn.mlp import MLP1, MLP3
    from blackwater.train import RowsTrainer

    torch.manual_seed(0)
    x, y = torch.randn(4096, 170, device=DEV), torch.randn(4096, 1, device=DEV)
    x2 = torch.randn(4096, 170, device=DEV)
    runs = {}
    for graphs in (False, True):
        torch.manual_seed(1)
        model = (MLP1(170, 128, 1) if kind == "mlp1" else MLP3(170, 125, 1)).to(DEV)
        tr = RowsTrainer(model, lr=1e-3, graphs=graphs)
        losses = []
        for k in range(12):
            losses.append(float(tr.step_rows(x if k % 2 == 0 else x2, y)))
        runs[graphs] = (losses, [p.detach().clone() for p in model.parameters()], [b.detach().clone() for b in model.buffers()])
        ops.set_seed_counter(None)
    assert runs[False][0] == runs[True][0]
    assert runs[False][0][-1] < runs[False][0][0]
    for a, b in zip(runs[False][1], runs[True][1]):
        assert torch.equal(a, b)
    for a, b in zip(runs[False][2], runs[True][2]):
        assert torch.equal(a, b)
    if kind == "mlp3":      # the masks move with the device counter: two consecutive steps on the same rows differ in their dropout
        assert len(set(runs[True][0])) == len(runs[True][0])


@pytest.mark.parametrize("mode,shape", [("f32", (170, 128, 1)), ("bf16", (170, 128, 1)), ("f32", (58, 64, 4)), ("bf16", (33, 17, 3))])
def test_fused_mlp1_step_equals_the_autograd_step(mode, shape, monkeypatch):
    """train.RowsTrainer's five-launch MLP1 step (forward with the MSE loss folded in -> backward -> second stage writing the
    gradients into the flat buffer and the loss -> Adam) against the same step through autograd (MLP1.forward, mse_loss_grad,
    out.backward, gradient filing): the output gradient is formed by the same fp32 expression, so parameters agree BIT FOR
    BIT after every step; the loss is the same sum in another order (1e-6)."""
    from blackwater.nn.mlp import MLP1
    from blackwater.train import RowsTrainer

    i, h, o = shape
    torch.manual_seed(2)
    xs = [torch.randn(n, i, device=DEV) for n in (4099, 4099, 33)]
    ys = [torch.randn(x.shape[0], o, device=DEV) for x in xs]
    runs = {}
    for fused in ("1", "0"):
        torch.manual_seed(3)
        model = MLP1(i, h, o).to(DEV)
        model.mfma = mode
        tr = RowsTrainer(model, lr=1e-3, graphs=False)
        tr.fused_mlp1_step = fused == "1"
        losses = [float(tr.step_rows(x, y)) for _ in range(2) for x, y in zip(xs, ys)]
        runs[fused] = (losses, tr.flat_param.detach().clone(), tr.flat_grad.detach().clone())
    assert torch.equal(runs["1"][1], runs["0"][1]) and torch.equal(runs["1"][2], runs["0"][2])
    for a, b in zip(runs["1"][0], runs["0"][0]):
        assert abs(a - b) <= 1e-6 * abs(b)
    assert runs["1"][0][3] < runs["1"][0][0]          # the same rows, three steps later
