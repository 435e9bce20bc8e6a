"""The fp32-storage twin of the MLP2 / MLP3 layer pipeline (csrc/mlp_layers.hip, ``mfma = "f32"``; reference:
docs/tutorials/mlp.py:33-108 in torch's default dtype) -- every building block against fp64 algebra on the
SAME fp32 inputs, then the composed modules' train step against the CPU oracle in fp64.

Stated tolerance: 1e-5 of the result's scale everywhere (fp32 products and sums against fp64; nothing is rounded to a shorter
format in this mode).  Dropout masks are counter-based; what is checked of them is the keep rate and that the backward regenerates
the forward's mask exactly."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W = 128


def _act_matrix(n, c, gen, scale=1.0, shift=0.0):
    """A [n, 128] fp32 activation with c live columns (pads zero), and its fp64 value."""
    v = (torch.randn(n, c, generator=gen) * scale + shift).float()
    a = torch.zeros(n, W)
    a[:, :c] = v
    return a.to(DEV), v.double()


def _close(got, want, what, tol=1e-5):
    err = (got.cpu().double() - want).abs().max().item()
    assert err <= tol * max(want.abs().max().item(), 1e-30), (what, err)


@pytest.mark.parametrize("n,k,u", [(1, 170, 125), (33, 170, 125), (1000, 58, 64), (4099, 169, 128), (517, 35, 25), (70001, 191, 41)])
def test_layer_gemm_from_fp32_rows(n, k, u):
    """The block's input: [N, k] in padded rows (k not a multiple of 16: the boundary k-group is masked, whatever the pad columns hold)."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + k)
    x, w, b = torch.randn(n, k, generator=g), torch.randn(u, k, generator=g) / k ** 0.5, torch.randn(u, generator=g)
    xp = ops.padded_empty(n, k, DEV)
    xp.copy_(x)
    if xp.stride(0) > k:
        torch.as_strided(xp, (n, xp.stride(0)), (xp.stride(0), 1))[:, k:] = float("nan")      # scratch columns must not matter
    y = ops.layer_gemm_f32(xp, w.to(DEV), b.to(DEV))
    assert y.dtype == torch.float32 and tuple(y.shape) == (n, W)
    want = x.double() @ w.double().T + b.double()
    _close(y[:, :u], want, "gemm")
    assert (y[:, u:] == 0).all()
    yn = ops.layer_gemm_f32(xp, w.to(DEV), b.to(DEV), narrow_out=True)
    assert tuple(yn.shape) == (n, u)
    _close(yn, want, "gemm, narrow output")


@pytest.mark.parametrize("n,k,u", [(33, 125, 125), (1000, 64, 64), (4099, 128, 41), (517, 25, 8), (262144, 125, 125)])
def test_layer_gemm_from_activations_and_the_data_gradient_form(n, k, u):
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + k + u)
    xa, xv = _act_matrix(n, k, g)
    w, b = torch.randn(u, k, generator=g) / k ** 0.5, torch.randn(u, generator=g)
    y = ops.layer_gemm_f32(xa, w.to(DEV), b.to(DEV))
    _close(y[:, :u], xv @ w.double().T + b.double(), "gemm, activation in")
    assert (y[:, u:] == 0).all()
    # data gradient: gX = dY W (+ add), W [K = u, U = k]
    da, dv = _act_matrix(n, u, g)
    adda, addv = _act_matrix(n, k, g)
    gx = ops.layer_gemm_f32(da, w.to(DEV), transposed=True, add=adda)
    _close(gx[:, :k], dv @ w.double() + addv, "dgrad + residual gradient")
    assert (gx[:, k:] == 0).all()
    gxn = ops.layer_gemm_f32(da, w.to(DEV), transposed=True, narrow_out=True)
    assert tuple(gxn.shape) == (n, k)
    _close(gxn, dv @ w.double(), "dgrad, narrow output")


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
    sc, sh = scale.cpu().double()[:c], shift.cpu().double()[:c]
    ra, rv = _act_matrix(n, c, g)
    z = ops.layer_act_bf16(ya, scale, shift, n, c, True, 0.0, 0, res=ra)
    assert z.dtype == torch.float32
    _close(z[:, :c], (yv * sc + sh).relu() + rv, "act + residual")
    assert (z[:, c:] == 0).all()
    ga, gv = _act_matrix(n, c, g)
    dbeta, dgamma, gs, k1, k2 = ops.layer_colstats_bwd(ga, ya, scale, shift, mean, invstd, gamma.to(DEV), True, 0.0, 0, n, c)
    gu = gv * ((yv * sc + sh) > 0)
    xhat = (yv - mean.cpu().double()[:c]) * invstd.cpu().double()[:c]
    assert rel(dbeta, gu.sum(0)) < 1e-5 and rel(dgamma, (gu * xhat).sum(0)) < 1e-5
    dy = ops.layer_bwd_apply_bf16(ga, ya, scale, shift, mean, invstd, gs, k1, k2, n, c, True, 0.0, 0)
    assert dy.dtype == torch.float32
    want = gs.cpu().double()[:c] * (gu - k1.cpu().double()[:c] - xhat * k2.cpu().double()[:c])
    _close(dy[:, :c], want, "bn backward", tol=2e-5)
    assert (dy[:, c:] == 0).all()
    if c != W:       # the same sums from a NARROW fp32 gradient (the incoming gradient of a model's last block)
        d2 = ops.layer_colstats_bwd(gv.float().to(DEV), ya, scale, shift, mean, invstd, gamma.to(DEV), True, 0.0, 0, n, c)
        assert rel(d2[0], gu.sum(0)) < 1e-5


def test_batch_statistics_of_columns_far_from_zero():
    """Columns whose mean is a thousand times their spread: the sums are taken of y - y[0] (as csrc/bn.hip does), so the variance
    does not drown in E[y^2] - E[y]^2 (unshifted fp32 sums lose every digit of it here)."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(3)
    n, c = 100000, 125
    ya, yv = _act_matrix(n, c, g, scale=0.1, shift=100.0)
    gamma, beta = torch.ones(c), torch.zeros(c)
    mean, var, invstd, scale, shift = ops.layer_colstats_fwd(ya, gamma.to(DEV), beta.to(DEV), 1e-5, n, c)
    m64, v64 = yv.mean(0), yv.var(0, unbiased=False)
    assert (mean.cpu().double()[:c] - m64).abs().max().item() <= 1e-6 * 100.0
    assert ((var.cpu().double()[:c] - v64).abs() / v64).max().item() < 1e-4


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
    _close(z[:, :c], torch.where(kept, yv.relu() / (1 - p), torch.zeros_like(yv)), "dropout scaling")
    # every column has its own stream of draws (the two four-column groups of a lane are keyed apart)
    rate = 1.0 - (kept & live).double().sum(0) / live.double().sum(0)
    assert (rate - p).abs().max().item() < 0.03
    ga, gv = _act_matrix(n, c, g)
    dy = ops.layer_bwd_apply_bf16(ga, ya, one, zero, zero, one, one, zero, zero, n, c, True, p, 1234)
    assert torch.equal(dy[:, :c].cpu().double() != 0, kept & (gv != 0))          # the same mask, exactly
    z2 = ops.layer_act_bf16(ya, one, zero, n, c, True, p, 1235)
    assert not torch.equal(z2, z)                                                   # another seed, another mask


@pytest.mark.parametrize("n,k,u,act_x", [(1, 170, 125, False), (33, 170, 125, False), (4099, 58, 64, False), (1000, 125, 125, True),
                                          (4099, 128, 41, True), (517, 41, 128, True), (7, 125, 125, True), (64, 128, 128, True),
                                          (131104, 125, 125, True), (262144, 64, 125, True), (50000, 191, 128, False)])
def test_layer_weight_gradient(n, k, u, act_x):
    """Row counts below one 32-row slab (the tail path alone), exact multiples of it (no tail), many slabs per workgroup; input
    widths that need one, two and three 64-column waves (the ones column that yields the bias gradient sits at index k)."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + 7 * k + u)
    da, dv = _act_matrix(n, u, g)
    if act_x:
        xa, xv = _act_matrix(n, k, g)
    else:
        x = torch.randn(n, k, generator=g)
        xa, xv = ops.padded_copy(x.to(DEV)), x.double()
    gw, gb = ops.layer_wgrad_f32(da, xa, u, k)
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
    _close(out, hv @ w.double().T + b.double(), "final outputs")
    gh, gw, gb = ops.layer_rowdot_bwd_bf16(go.to(DEV), ha, w.to(DEV), n)
    assert gh.dtype == torch.float32
    _close(gh[:, :c], go.double() @ w.double(), "gh")
    assert (gh[:, c:] == 0).all()
    ww, wb = go.double().T @ hv, go.double().sum(0)
    assert (gw.cpu().double() - ww).norm().item() <= 1e-5 * (ww.norm().item() + 1e-30)
    assert (gb.cpu().double() - wb).abs().max().item() <= 1e-5 * max(wb.abs().max().item(), 1.0)
    # gated: h = dropout(relu(u)) is its own gate, gh is the gradient at u
    gg, gw2, _ = ops.layer_rowdot_bwd_bf16(go.to(DEV), ha, w.to(DEV), n, gate_scale=1.25)
    _close(gg[:, :c], torch.where(hv > 0, 1.25 * (go.double() @ w.double()), torch.zeros_like(hv)), "gated gh")
    assert torch.equal(gw2, gw)


def test_a_block_without_batchnorm_in_the_gemm_epilogue():
    """fc3 of MLP3: dropout(relu(x W^T + b)) from the GEMM launch; what is kept is scaled by 1 / (1 - p), the rest is zero, the
    keep rate is 1 - p in every column, another seed draws another mask."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(11)
    n, k, u, p = 30000, 125, 41, 0.3
    xa, xv = _act_matrix(n, k, g)
    w, b = torch.randn(u, k, generator=g) / k ** 0.5, torch.randn(u, generator=g) + 0.5
    pre = xv @ w.double().T + b.double()
    plain = ops.layer_gemm_f32(xa, w.to(DEV), b.to(DEV), relu=True)
    _close(plain[:, :u], pre.relu(), "relu epilogue")
    y = ops.layer_gemm_f32(xa, w.to(DEV), b.to(DEV), relu=True, drop_p=p, seed=77)
    kept = y[:, :u].cpu().double() != 0
    _close(y[:, :u], torch.where(kept, pre.relu() / (1 - p), torch.zeros_like(pre)), "dropout scaling")
    live = pre > 0
    rate = 1.0 - (kept & live).double().sum(0) / live.double().sum(0)
    assert (rate - p).abs().max().item() < 0.03
    assert (y[:, u:] == 0).all()
    assert not torch.equal(ops.layer_gemm_f32(xa, w.to(DEV), b.to(DEV), relu=True, drop_p=p, seed=78), y)


@pytest.mark.parametrize("cls,args", [("MLP2", (170, 125, 1)), ("MLP3", (170, 125, 1)), ("MLP3", (58, 64, 4)), ("MLP3", (80, 25, 4))])
def test_train_step_on_the_layer_kernels_equals_the_oracle_in_fp64(cls, args):
    """The composed module in TRAINING mode (batch statistics, dropout off so that the function is the oracle's): outputs, loss, every
    parameter gradient, the BatchNorm running statistics -- and the input gradient where the input needs one (a GNN's head) --
    within 1e-5 of their scale of the CPU oracle (oracle/models.py: mlp.py:33-108 restated) evaluated in fp64."""
    import blackwater.nn as bnn
    import blackwater.native.functional as F
    import oracle.models as om

    torch.manual_seed(1)
    n = 3000
    x, y = torch.randn(n, args[0]), torch.randn(n, args[2])
    torch.manual_seed(2)
    model = getattr(bnn, cls)(*args, dropout_rate=0.0)
    ref = getattr(om, cls)(*args, dropout_rate=0.0).double().train()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in model.state_dict().items()})
    model = model.to(DEV).train()
    assert model.mfma == "f32"
    needs = args[0] <= 128
    xin = x.to(DEV).requires_grad_(needs)
    calls = []
    real = F._MLPTrunkBf16.apply
    try:
        F._MLPTrunkBf16.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
        out = model(xin)
    finally:
        F._MLPTrunkBf16.apply = real
    assert calls, "the fp32 train step did not take the layer kernels"
    loss = torch.nn.functional.mse_loss(out, y.to(DEV))
    loss.backward()
    xr = x.double().requires_grad_(needs)
    want = ref(xr)
    wl = torch.nn.functional.mse_loss(want, y.double())
    wl.backward()
    _close(out.detach(), want.detach(), "outputs")
    assert abs(loss.item() - wl.item()) <= 1e-5 * max(abs(wl.item()), 1.0)
    top = max(q.grad.abs().max().item() for q in ref.parameters())
    for (name, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        # the bias of a Linear in front of a BatchNorm has an analytically ZERO gradient: rounding noise of a few fp32 ulps of the largest gradient, hence the floor of 5e-7 of it
        err = (p.grad.cpu().double() - q.grad).abs().max().item()
        assert err <= 1e-5 * max(q.grad.abs().max().item(), 5e-2 * top), (name, err)
    if needs:
        _close(xin.grad, xr.grad, "input gradient")
    for bn, rbn in ((model.bn1, ref.bn1), (model.bn2, ref.bn2)):
        _close(bn.running_mean, rbn.running_mean, "running mean")
        _close(bn.running_var, rbn.running_var, "running var")
        assert int(bn.num_batches_tracked) == int(rbn.num_batches_tracked) == 1


def test_fp32_layer_step_replays_from_a_graph_and_trains():
    """train.RowsTrainer on MLP3 in its default mode (fp32) with the reference's dropout: eager == replay bit for bit, masks move
    with the device counter, and the loss goes down."""
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
        assert model.mfma == "f32"
        tr = RowsTrainer(model, lr=1e-3, graphs=graphs)
        runs[graphs] = [float(tr.step_rows(x, y)) for _ in range(30)]
        ops.set_seed_counter(None)
    assert runs[False] == runs[True]
    assert runs[True][-1] < 0.6 * runs[True][0]
