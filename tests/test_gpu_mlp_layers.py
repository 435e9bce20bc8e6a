"""The bf16-storage pipeline of MLP2 / MLP3 (csrc/mlp_layers.hip; reference: docs/tutorials/mlp.py:33-108) -- every building block
against fp64 algebra on the SAME bf16 inputs, then the composed modules against their fp32 path.

Stated tolerances.  A block's output is a bf16 matrix: it equals the exactly computed value rounded to bf16 up to ONE bf16 ulp
(2^-7 relative) where the fp32 and the fp64 sum fall on different sides of a rounding boundary.  fp32 results of a block (batch
statistics, weight gradients, the final outputs) equal the fp64 value on the rounded operands to 1e-5 of their scale.  The
composed MLP2 / MLP3 train step in bf16 mode stays within 3e-2 (outputs) / 10 % Frobenius (parameter gradients) of the fp32
mode: a different arithmetic with 8-bit mantissas at every GEMM operand and every stored activation."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W = 128


def _bf(t):
    return t.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def _act_matrix(n, c, gen, scale=1.0, shift=0.0):
    """A [n, 128] bf16 activation with c live columns (pads zero), and its fp64 value."""
    v = (torch.randn(n, c, generator=gen) * scale + shift).to(torch.bfloat16)
    a = torch.zeros(n, W, dtype=torch.bfloat16)
    a[:, :c] = v
    return a.to(DEV), v.to(torch.float64)


def _close_bf16(got, want, what):
    """got (bf16 tensor) == want (fp64) rounded to bf16, up to one ulp."""
    g = got.cpu().to(torch.float64)
    tol = want.abs() * 2.0 ** -7 + 1e-30 + 2.0 ** -7 * 1e-3 * want.abs().max()
    bad = ((g - want).abs() > tol)
    assert not bad.any(), (what, int(bad.sum()), float((g - want).abs().max()))


@pytest.mark.parametrize("n,k,u", [(1, 170, 125), (33, 170, 125), (1000, 58, 64), (4099, 169, 128), (517, 35, 25)])
def test_layer_gemm_from_fp32_rows(n, k, u):
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + k)
    x, w, b = torch.randn(n, k, generator=g), torch.randn(u, k, generator=g) / k ** 0.5, torch.randn(u, generator=g)
    y = ops.layer_gemm_bf16(x.to(DEV), w.to(DEV), b.to(DEV))
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (n, W)
    want = _bf(x) @ _bf(w).T + b.double()
    _close_bf16(y[:, :u], want, "gemm")
    assert (y[:, u:].float() == 0).all()
    yf = ops.layer_gemm_bf16(x.to(DEV), w.to(DEV), b.to(DEV), out_f32=True)
    assert (yf.cpu().double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()


@pytest.mark.parametrize("n,k,u", [(33, 125, 125), (1000, 64, 64), (4099, 128, 41), (517, 25, 8)])
def test_layer_gemm_from_bf16_activations_and_the_data_gradient_form(n, k, u):
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + k + u)
    xa, xv = _act_matrix(n, k, g)
    w, b = torch.randn(u, k, generator=g) / k ** 0.5, torch.randn(u, generator=g)
    y = ops.layer_gemm_bf16(xa, w.to(DEV), b.to(DEV))
    _close_bf16(y[:, :u], xv @ _bf(w).T + b.double(), "gemm bf16 in")
    # data gradient: gX = dY W (+ add), W [K = u, U = k]
    da, dv = _act_matrix(n, u, g)
    adda, addv = _act_matrix(n, k, g)
    gx = ops.layer_gemm_bf16(da, w.to(DEV), transposed=True, add=adda)
    _close_bf16(gx[:, :k], dv @ _bf(w) + addv, "dgrad")
    gxf = ops.layer_gemm_bf16(da, w.to(DEV), transposed=True, out_f32=True)
    want = dv @ _bf(w)
    assert tuple(gxf.shape) == (n, k) and (gxf.cpu().double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()


@pytest.mark.parametrize("n,c", [(2, 125), (37, 5), (5000, 128), (40000, 41)])
def test_batch_statistics_activation_and_backward_blocks(n, c):
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n * 131 + c)
    ya, yv = _act_matrix(n, c, g, scale=2.0, shift=0.7)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    eps = 1e-5
    mean, var, invstd, scale, shift = ops.layer_colstats_fwd(ya, gamma.to(DEV), beta.to(DEV), eps, n, c)
    m64, v64 = yv.mean(0), yv.var(0, unbiased=False)
    rel = lambda a, b: (a.cpu().double()[:c] - b).abs().max().item() / (b.abs().max().item() + 1e-12)
    assert rel(mean, m64) < 1e-5 and rel(var, v64) < 1e-4
    is64 = 1.0 / torch.sqrt(v64 + eps)
    assert rel(invstd, is64) < 1e-4 and rel(scale, gamma.double() * is64) < 1e-4
    # activation (no dropout): bf16(relu(y scale + shift)) + residual, with the device's own fp32 scale / shift
    sc, sh = scale.cpu().double()[:c], shift.cpu().double()[:c]
    ra, rv = _act_matrix(n, c, g)
    z = ops.layer_act_bf16(ya, scale, shift, n, c, True, 0.0, 0, res=ra)
    u = (yv * sc + sh).relu()
    _close_bf16(z[:, :c], _bf(u) + rv, "act + residual")
    assert (z[:, c:].float() == 0).all()
    # backward of the block: gu = g o (u > 0); sums; dy = gs (gu - k1 - xhat k2)
    ga, gv = _act_matrix(n, c, g)
    dbeta, dgamma, gs, k1, k2 = ops.layer_colstats_bwd(ga, ya, scale, shift, mean, invstd, gamma.to(DEV), True, 0.0, 0, n, c)
    gu = gv * ((yv * sc + sh) > 0)
    xhat = (yv - mean.cpu().double()[:c]) * invstd.cpu().double()[:c]
    assert rel(dbeta, gu.sum(0)) < 1e-5 and rel(dgamma, (gu * xhat).sum(0)) < 1e-5
    dy = ops.layer_bwd_apply_bf16(ga, ya, scale, shift, mean, invstd, gs, k1, k2, n, c, True, 0.0, 0)
    want = gs.cpu().double()[:c] * (gu - k1.cpu().double()[:c] - xhat * k2.cpu().double()[:c])
    _close_bf16(dy[:, :c], want, "bn backward")
    assert (dy[:, c:].float() == 0).all()
    # the same sums from an fp32 gradient (the incoming gradient of a model's last block)
    d2 = ops.layer_colstats_bwd(gv.float().to(DEV), ya, scale, shift, mean, invstd, gamma.to(DEV), True, 0.0, 0, n, c)
    assert rel(d2[0], gu.sum(0)) < 1e-5


def test_dropout_masks_are_recomputed_identically_in_the_backward():
    from blackwater.native import ops

    g = torch.Generator().manual_seed(5)
    n, c, p = 20000, 125, 0.3
    ya, yv = _act_matrix(n, c, g, shift=1.0)
    one, zero = torch.ones(W, device=DEV), torch.zeros(W, device=DEV)
    z = ops.layer_act_bf16(ya, one, zero, n, c, True, p, 1234)
    live = (yv > 0)
    kept = (z[:, :c].cpu().double() != 0)
    frac = 1.0 - kept[live].double().mean().item()
    assert abs(frac - p) < 0.01
    _close_bf16(z[:, :c], torch.where(kept, yv.relu() / (1 - p), torch.zeros_like(yv)), "dropout scaling")
    ga, gv = _act_matrix(n, c, g)
    dy = ops.layer_bwd_apply_bf16(ga, ya, one, zero, zero, one, one, zero, zero, n, c, True, p, 1234)
    assert torch.equal(dy[:, :c].cpu().double() != 0, kept & (gv != 0))          # the same mask, exactly
    z2 = ops.layer_act_bf16(ya, one, zero, n, c, True, p, 1235)
    assert not torch.equal(z2, z)                                                   # another seed, another mask


@pytest.mark.parametrize("n,k,u,bf16_x", [(1, 170, 125, False), (33, 170, 125, False), (4099, 58, 64, False), (1000, 125, 125, True),
                                           (4099, 128, 41, True), (517, 41, 128, True), (7, 125, 125, True), (64, 128, 128, True),
                                           (131104, 125, 125, True), (262144, 64, 125, True)])
def test_layer_weight_gradient(n, k, u, bf16_x):
    """bf16 inputs go through the LDS-DMA / transposing-read kernel: row counts below one 32-row slab (the tail path alone), exact
    multiples of it (no tail), and enough slabs for the LDS ring to wrap many times."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + 7 * k + u)
    da, dv = _act_matrix(n, u, g)
    if bf16_x:
        xa, xv = _act_matrix(n, k, g)
    else:
        x = torch.randn(n, k, generator=g)
        xa, xv = ops.padded_copy(x.to(DEV)), _bf(x)
    gw, gb = ops.layer_wgrad_bf16(da, xa, u, k)
    want_w, want_b = dv.T @ xv, dv.sum(0)
    assert (gw.cpu().double() - want_w).norm().item() <= 1e-5 * (want_w.norm().item() + 1e-30)
    assert (gb.cpu().double() - want_b).norm().item() <= 1e-5 * (want_b.norm().item() + 1e-30)


@pytest.mark.parametrize("n,c,o", [(1, 41, 1), (1000, 125, 4), (40001, 64, 2)])
def test_final_outputs_forward_and_backward(n, c, o):
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + c + o)
    ha, hv = _act_matrix(n, c, g)
    w, b, go = torch.randn(o, c, generator=g), torch.randn(o, generator=g), torch.randn(n, o, generator=g)
    out = ops.layer_rowdot_bf16(ha, w.to(DEV), b.to(DEV), n)
    want = hv @ _bf(w).T + b.double()
    assert (out.cpu().double() - want).abs().max().item() <= 1e-5 * max(want.abs().max().item(), 1.0)
    gh, gw, gb = ops.layer_rowdot_bwd_bf16(go.to(DEV), ha, w.to(DEV), n)
    _close_bf16(gh[:, :c], _bf(go) @ _bf(w), "gh")
    ww, wb = _bf(go).T @ hv, go.double().sum(0)
    assert (gw.cpu().double() - ww).norm().item() <= 1e-5 * (ww.norm().item() + 1e-30)
    assert (gb.cpu().double() - wb).abs().max().item() <= 1e-5 * max(wb.abs().max().item(), 1.0)


@pytest.mark.parametrize("cls,args", [("MLP2", (170, 125, 1)), ("MLP3", (170, 125, 1)), ("MLP3", (58, 64, 4)), ("MLP3", (80, 25, 4))])
def test_train_step_in_bf16_storage_mode_tracks_the_fp32_mode(cls, args):
    """The composed module, dropout off so that the two modes see the same function: loss, outputs, every parameter gradient,
    the BatchNorm running statistics -- and the input gradient where the input needs one (a GNN's head)."""
    import blackwater.nn as bnn

    torch.manual_seed(1)
    n = 3000
    x, y = torch.randn(n, args[0], device=DEV), torch.randn(n, args[2], device=DEV)
    res = {}
    for mode in ("f32", "bf16"):
        torch.manual_seed(2)
        model = getattr(bnn, cls)(*args, dropout_rate=0.0).to(DEV).train()
        model.mfma = mode
        xin = x.clone().requires_grad_(args[0] <= 128)
        out = model(xin)
        loss = torch.nn.functional.mse_loss(out, y)
        loss.backward()
        res[mode] = (out.detach(), [p.grad.clone() for p in model.parameters()], xin.grad, model.bn1.running_var.clone(),
                     model.bn2.running_mean.clone(), int(model.bn1.num_batches_tracked))
    a, b = res["f32"], res["bf16"]
    assert (a[0] - b[0]).abs().max().item() <= 3e-2 * max(a[0].abs().max().item(), 1.0)
    # the bias of a Linear in front of a BatchNorm has an analytically ZERO gradient (the batch mean removes it): both modes return
    # rounding noise there, so the bound is relative to the largest gradient of the model as well
    top = max(g.norm().item() for g in a[1])
    for ga, gb in zip(a[1], b[1]):
        assert (ga - gb).norm().item() <= 0.1 * ga.norm().item() + 1e-3 * top
    if a[2] is not None:
        assert b[2] is not None and (a[2] - b[2]).norm().item() <= 0.1 * a[2].norm().item()
    assert (a[3] - b[3]).abs().max().item() <= 2e-2 * a[3].abs().max().item()
    assert (a[4] - b[4]).abs().max().item() <= 2e-2 * max(a[4].abs().max().item(), 0.1) and a[5] == b[5] == 1


def test_bf16_storage_step_replays_from_a_graph_and_trains():
    """train.RowsTrainer on MLP3 with ``mfma = "bf16"`` and the reference's dropout: eager == replay bit for bit, masks move with
    the device counter, and the loss goes down."""
    from blackwater.native import ops
    from blackwater.nn.mlp import MLP3
    from blackwater.train import RowsTrainer

    torch.manual_seed(0)
    x = torch.randn(8192, 170, device=DEV)
    w = torch.randn(170, 1, device=DEV) / 13.0
    y = (x @ w).tanh()
    runs = {}
    for graphs in (False, True):
        torch.manual_seed(1)
        model = MLP3(170, 125, 1).to(DEV)
        model.mfma = "bf16"
        tr = RowsTrainer(model, lr=1e-3, graphs=graphs)
        runs[graphs] = [float(tr.step_rows(x, y)) for _ in range(30)]
        ops.set_seed_counter(None)
    assert runs[False] == runs[True]
    assert runs[True][-1] < 0.6 * runs[True][0]
