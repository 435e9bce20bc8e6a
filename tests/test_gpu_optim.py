"""The two ends of the train step that are not model layers (csrc/optim.hip) against torch: the reference's loop is
``loss = MSELoss()(out, y); loss.backward(); Adam.step()`` (docs/tutorials/__ml_models.py:100-187).

Stated tolerances: the loss within 1e-6 relative of torch's fp32 value and 1e-6 of the fp64 value's scale (another summation
order); its gradient bit-equal to ``2 (out - y) / numel`` in fp32; Adam's parameters within 2e-6 relative (Frobenius) of
``torch.optim.Adam`` after 50 steps (same formulas, bias corrections in double as torch's fused kernel)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,c", [(1, 1), (7, 4), (1000, 1), (262144, 1), (4099, 3), (300001, 4)])
def test_fused_mse_loss_and_gradient(n, c):
    from blackwater.native import ops

    torch.manual_seed(n + c)
    wide = torch.randn(n, c + 3, device=DEV)
    out = wide[:, :c] if n > 1 else wide[:, :c].contiguous()       # a row-strided view, as a bucket-padded batch's output is
    y = torch.randn(n, c, device=DEV)
    loss, g = ops.mse_loss_grad(out, y)
    ref = out.detach().clone().requires_grad_(True)
    want = torch.nn.functional.mse_loss(ref, y)
    want.backward()
    exact = ((out.double() - y.double()) ** 2).mean().item()
    assert abs(loss.item() - want.item()) <= 1e-6 * abs(want.item())
    assert abs(loss.item() - exact) <= 1e-6 * abs(exact)
    assert torch.equal(g, (out - y) * (2.0 / (n * c)))
    assert (g - ref.grad).abs().max().item() <= 1e-6 * ref.grad.abs().max().item()
    again, g2 = ops.mse_loss_grad(out, y)                              # same bits on every call (fixed summation order)
    assert torch.equal(again, loss) and torch.equal(g2, g)
    only_loss, none = ops.mse_loss_grad(out, y, want_grad=False)
    assert none is None and torch.equal(only_loss, loss)


@pytest.mark.parametrize("n", [1, 5, 1023, 13645, 105418])
def test_flat_adam_equals_torch_adam(n):
    from blackwater.train import FlatAdam

    torch.manual_seed(n)
    p0 = torch.randn(n, device=DEV)
    mine = p0.clone().requires_grad_(True)
    theirs = p0.clone().requires_grad_(True)
    a = FlatAdam([mine], lr=1e-3)
    b = torch.optim.Adam([theirs], lr=1e-3)
    for k in range(50):
        g = torch.randn(n, device=DEV) * (10.0 ** ((k % 7) - 4))     # gradients over seven orders of magnitude
        if k == 20:
            g.zero_()
        mine.grad, theirs.grad = g.clone(), g.clone()
        a.step()
        b.step()
        if k == 30:                                                   # what ReduceLROnPlateau does to a tensor learning rate
            a.param_groups[0]["lr"].fill_(1e-4)
            b.param_groups[0]["lr"] = 1e-4
    st = a.state[mine]
    assert st["step"].item() == 50.0
    assert (mine - theirs).norm().item() <= 2e-6 * theirs.norm().item() + 1e-9
    assert (st["exp_avg"] - b.state[theirs]["exp_avg"]).norm().item() <= 2e-6 * b.state[theirs]["exp_avg"].norm().item() + 1e-12
    assert (st["exp_avg_sq"] - b.state[theirs]["exp_avg_sq"]).norm().item() <= 2e-6 * b.state[theirs]["exp_avg_sq"].norm().item() + 1e-12


def test_reduce_lr_on_plateau_drives_the_device_resident_rate():
    from blackwater.train import FlatAdam

    p = torch.zeros(64, device=DEV, requires_grad=True)
    opt = FlatAdam([p], lr=1e-3)
    lr = opt.param_groups[0]["lr"]
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", factor=0.1, patience=1, min_lr=1e-5)
    for _ in range(4):
        sched.step(1.0)
    assert opt.param_groups[0]["lr"] is lr                           # changed in place: a captured step reads the same address
    assert abs(lr.item() - 1e-4) < 1e-9
    p.grad = torch.ones_like(p)
    opt.step()
    assert torch.allclose(p, torch.full_like(p, -1e-4), rtol=1e-5)   # the first Adam step moves every weight by lr


def test_flat_adam_state_round_trips_through_a_cpu_checkpoint(tmp_path):
    """a state saved with torch.save and loaded with map_location='cpu' (the usual resume) must land IN the
    optimizer's device tensors -- same addresses (captured graphs read them), same values -- and the resumed run must step like
    the one that never stopped."""
    from blackwater.train import FlatAdam

    torch.manual_seed(3)
    n = 4099
    p0 = torch.randn(n, device=DEV)
    grads = [torch.randn(n, device=DEV) for _ in range(12)]
    ref_p = p0.clone().requires_grad_(True)
    ref = FlatAdam([ref_p], lr=1e-3)
    for k, g in enumerate(grads):
        ref_p.grad = g.clone()
        ref.step()
        if k == 5:
            ref.param_groups[0]["lr"].fill_(3e-4)
            torch.save({"opt": ref.state_dict(), "p": ref_p.detach().clone()}, tmp_path / "ck.pt")
    blob = torch.load(tmp_path / "ck.pt", map_location="cpu")
    p = blob["p"].to(DEV).requires_grad_(True)
    opt = FlatAdam([p], lr=1e-3)
    st = opt.state[p]
    addr = (st["step"].data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), opt.param_groups[0]["lr"].data_ptr())
    lr_obj = opt.param_groups[0]["lr"]
    opt.load_state_dict(blob["opt"])
    st = opt.state[p]
    assert addr == (st["step"].data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), opt.param_groups[0]["lr"].data_ptr())
    assert opt.param_groups[0]["lr"] is lr_obj and abs(lr_obj.item() - 3e-4) < 1e-10
    assert st["step"].is_cuda and st["step"].item() == 6.0
    for g in grads[6:]:
        p.grad = g.clone()
        opt.step()
    assert torch.equal(p.detach(), ref_p.detach())
    assert torch.equal(st["exp_avg"], ref.state[ref_p]["exp_avg"]) and torch.equal(st["exp_avg_sq"], ref.state[ref_p]["exp_avg_sq"])


def test_ticket_pool_is_allocated_eagerly_and_shared_by_slots():
    """the counters of the last-workgroup-done kernels come from ONE eagerly allocated per-device pool; a stream
    gets a slot by host bookkeeping, never by an allocation (which a capture would have swallowed)."""
    from blackwater.native import ops

    ops.prepare_device(DEV)
    a = ops._ticket(torch.device(DEV))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        b = ops._ticket(torch.device(DEV))
        again = ops._ticket(torch.device(DEV))
    assert a.data_ptr() != b.data_ptr() and b.data_ptr() == again.data_ptr()
    pool = ops._ticket_pools[torch.device(DEV)][0]
    assert pool.data_ptr() <= a.data_ptr() < pool.data_ptr() + pool.numel() * 4
    assert int(pool.abs().sum().item()) == 0


def test_fused_mse_on_a_padded_batch_equals_slicing_the_output():
    """A bucket-padded batch: the loss sees the first ``rows`` rows, the gradient has the output's shape with zero rows behind
    them -- what ``mse_loss(out[:rows], y[:rows]).backward()`` leaves in ``out.grad``."""
    from blackwater.native import ops

    torch.manual_seed(5)
    for n_pad, rows, c in ((33, 32, 1), (1025, 1024, 4), (40, 7, 3)):
        out = torch.randn(n_pad, c, device=DEV)
        y = torch.randn(n_pad, c, device=DEV)
        loss, g = ops.mse_loss_grad(out, y, rows=rows)
        ref = out.clone().requires_grad_(True)
        want = torch.nn.functional.mse_loss(ref[:rows], y[:rows])
        want.backward()
        assert abs(loss.item() - want.item()) <= 1e-6 * abs(want.item())
        assert tuple(g.shape) == (n_pad, c) and (g[rows:] == 0).all()
        assert (g - ref.grad).abs().max().item() <= 1e-6 * ref.grad.abs().max().item()
