"""GPU parity of the individual kernels, called through the C ABI, against numpy/torch-CPU restatements.
Run on the MI355X box with ``pytest -m gpu``."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _random_graph(n, e, seed, self_loops=0):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    if self_loops:
        loops = torch.randint(0, n, (self_loops,), generator=g)
        src, dst = torch.cat([src, loops]), torch.cat([dst, loops])
    return torch.stack([src, dst])


def _csr_reference(ei, n):
    src, dst = ei[0].numpy(), ei[1].numpy()
    keep = src != dst
    loops = np.bincount(src[~keep], minlength=n)
    s, d = src[keep], dst[keep]
    o_in = np.argsort(d, kind="stable")
    o_out = np.argsort(s, kind="stable")
    in_ptr = np.concatenate([[0], np.cumsum(np.bincount(d, minlength=n))])
    out_ptr = np.concatenate([[0], np.cumsum(np.bincount(s, minlength=n))])
    return in_ptr, s[o_in], out_ptr, d[o_out], loops


@pytest.mark.parametrize("n,e,loops", [(1, 0, 0), (7, 0, 0), (50, 200, 10), (3000, 9000, 500), (5, 40, 40)])
def test_csr_build(n, e, loops):
    from blackwater.native import ops

    ei = _random_graph(n, e, seed=n + e, self_loops=loops)
    got = ops.csr_build(ei.to(DEV), n)
    want = _csr_reference(ei, n)
    m = int(want[0][-1])
    assert np.array_equal(got[0].cpu().numpy()[: n + 1], want[0])
    assert np.array_equal(got[1].cpu().numpy()[:m], want[1])  # stable: edge order kept inside a row
    assert np.array_equal(got[2].cpu().numpy()[: n + 1], want[2])
    assert np.array_equal(got[3].cpu().numpy()[:m], want[3])
    assert np.array_equal(got[4].cpu().numpy()[:n], want[4])


@pytest.mark.parametrize("c", [1, 2, 3, 10, 22, 30, 45, 125])
def test_csr_aggregate_matches_dense(c):
    from blackwater.native import ops

    n, e = 257, 900
    ei = _random_graph(n, e, seed=c)
    in_ptr, in_src, *_ = ops.csr_build(ei.to(DEV), n)
    g = torch.Generator().manual_seed(c)
    x = torch.randn(n, c, generator=g)
    cs, rs, ds = (torch.rand(n, generator=g) + 0.5 for _ in range(3))
    z = torch.randn(n, c, generator=g)
    bias = torch.randn(c, generator=g)
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((ei[1], ei[0]), torch.ones(e, dtype=torch.float64), accumulate=True)
    a.fill_diagonal_(0)  # self-loops are not stored as edges; their term comes through dself
    xd = x.double()
    want = 1.5 * (rs.double()[:, None] * (a @ (cs.double()[:, None] * xd)) + ds.double()[:, None] * xd) \
        - 0.5 * z.double() + bias.double()
    got = ops.csr_aggregate(x.to(DEV), in_ptr, in_src, cscale=cs.to(DEV), rscale=rs.to(DEV), dself=ds.to(DEV),
                            alpha=1.5, z=z.to(DEV), beta=-0.5, bias=bias.to(DEV))
    assert torch.allclose(got.cpu().double(), want, rtol=1e-5, atol=1e-5)
    got_relu = ops.csr_aggregate(x.to(DEV), in_ptr, in_src, cscale=cs.to(DEV), rscale=rs.to(DEV), dself=ds.to(DEV),
                                 alpha=1.5, z=z.to(DEV), beta=-0.5, bias=bias.to(DEV), relu=True)
    assert torch.allclose(got_relu.cpu().double(), want.clamp(min=0), rtol=1e-5, atol=1e-5)
    plain = ops.csr_aggregate(x.to(DEV), in_ptr, in_src)
    assert torch.allclose(plain.cpu().double(), a @ xd, rtol=1e-5, atol=1e-5)
    # the ELL-assisted kernel (first two sources per row in a side table) gives the same rows
    ell = ops.ell_from_csr(in_ptr, in_src, n)
    fast = ops.csr_aggregate(x.to(DEV), in_ptr, in_src, ell=ell, cscale=cs.to(DEV), rscale=rs.to(DEV),
                             dself=ds.to(DEV), alpha=1.5, z=z.to(DEV), beta=-0.5, bias=bias.to(DEV))
    assert torch.allclose(fast.cpu().double(), want, rtol=1e-5, atol=1e-5)
    # strided input (a column slice of a wider matrix) goes through the leading-dimension path
    wide = torch.randn(n, c + 3, generator=g).to(DEV)
    sl = ops.csr_aggregate(wide[:, 1:1 + c], in_ptr, in_src)
    assert torch.allclose(sl.cpu().double(), a @ wide[:, 1:1 + c].cpu().double(), rtol=1e-5, atol=1e-5)


def test_csr_aggregate_empty_rows_and_dropout():
    from blackwater.native import ops

    n, c = 1000, 10
    ei = torch.zeros((2, 0), dtype=torch.long)
    in_ptr, in_src, *_ = ops.csr_build(ei.to(DEV), n)
    x = torch.ones(n, c, device=DEV)
    assert ops.csr_aggregate(x, in_ptr, in_src).abs().max().item() == 0.0
    ones = torch.ones(n, device=DEV)
    y = ops.csr_aggregate(x, in_ptr, in_src, dself=ones, drop_p=0.25, seed=7)
    kept = (y > 0).float().mean().item()
    assert abs(kept - 0.75) < 0.03
    assert torch.allclose(y[y > 0], torch.tensor(1 / 0.75, device=DEV))
    y2 = ops.csr_aggregate(x, in_ptr, in_src, dself=ones, drop_p=0.25, seed=7)
    assert torch.equal(y, y2)  # counter-based generator: same seed, same mask
    g = torch.full((n, c), 2.0, device=DEV)
    gx = ops.relu_dropout_bwd(g, y, 1 / 0.75)
    assert torch.equal(gx > 0, y > 0)


def test_segment_max_includes_self():
    from blackwater.native import ops

    n, c = 300, 45
    ei = _random_graph(n, 700, seed=3)
    in_ptr, in_src, *_ = ops.csr_build(ei.to(DEV), n)
    x = torch.randn(n, c)
    want = x.clone()
    for s, d in ei.t().tolist():
        if s != d:
            want[d] = torch.maximum(want[d], x[s])
    got = ops.csr_segment_max(x.to(DEV), in_ptr, in_src)
    assert torch.equal(got.cpu(), want)
    got = ops.csr_segment_max(x.to(DEV), in_ptr, in_src, ell=ops.ell_from_csr(in_ptr, in_src, n))
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("c", [1, 10, 22])
def test_heavy_rows_barrier_like(c):
    """Rows with hundreds of in-edges (barrier nodes) take the block-cooperative path; mixed with light rows."""
    from blackwater.native import ops

    n = 2500
    g = torch.Generator().manual_seed(11 + c)
    src, dst = [], []
    for hub in (7, 300, 301, 1999):  # hub rows with 100..700 in-edges
        k = 100 + hub % 601
        src.append(torch.randint(0, n, (k,), generator=g)); dst.append(torch.full((k,), hub))
    src.append(torch.arange(0, n - 1)); dst.append(torch.arange(1, n))  # a chain: in-degree 1 elsewhere
    src.append(torch.randint(0, n, (40,), generator=g)); dst.append(torch.full((40,), 55))  # just above 32
    ei = torch.stack([torch.cat(src), torch.cat(dst)])
    in_ptr, in_src, *_ = ops.csr_build(ei.to(DEV), n)
    x = torch.randn(n, c, generator=g)
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((ei[1], ei[0]), torch.ones(ei.shape[1], dtype=torch.float64), accumulate=True)
    a.fill_diagonal_(0)
    want = a @ x.double()
    for ell in (None, ops.ell_from_csr(in_ptr, in_src, n)):
        got = ops.csr_aggregate(x.to(DEV), in_ptr, in_src, ell=ell)
        assert torch.allclose(got.cpu().double(), want, rtol=1e-5, atol=1e-4)
        mx = ops.csr_segment_max(x.to(DEV), in_ptr, in_src, ell=ell)
        ref = x.clone()
        for s_, d_ in ei.t().tolist():
            if s_ != d_:
                ref[d_] = torch.maximum(ref[d_], x[s_])
        assert torch.equal(mx.cpu(), ref)


@pytest.mark.parametrize("n,i,o", [(4096, 22, 180), (70001, 22, 180), (5003, 45, 120), (8192, 45, 180), (4100, 30, 100), (9999, 22, 96), (6000, 48, 192)])
def test_wide_projection_through_lds_writes_whole_rows(n, i, o):
    """The q / k / v / skip projection of a TransformerConv (22 -> 180, 45 -> 120; gnn.py:80-91) takes linear_rows_lds_kernel: a
    wave owns 16 whole rows, the product is transposed through LDS and written as contiguous 1 KB stores, bias added on the way.
    Against fp64 within 1e-5 of the output scale -- with and without a row map of x, row counts that are no multiple of 16, NaN in the
    input's pad columns -- and bit-equal to the general kernel it replaces for these shapes (the general kernel, in a second process, was
    not needed: the same products in the same order)."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + i + o)
    x = torch.randn(n, i, generator=g)
    w = (torch.randn(o, i, generator=g) / i ** 0.5).to(DEV)
    b = torch.randn(o, generator=g).to(DEV)
    xd = ops.padded_copy(x.to(DEV))
    if xd.stride(0) > i:
        torch.as_strided(xd, (n, xd.stride(0)), (xd.stride(0), 1))[:, i:] = float("nan")
    want = x.double() @ w.cpu().double().t() + b.cpu().double()
    scale = want.abs().max().item()
    y = ops.linear(xd, w, b)
    assert y.stride(0) == (o + 3) // 4 * 4
    assert (y.cpu().double() - want).abs().max().item() < 1e-5 * scale
    # through a row map (the batch's rows of the device-resident dataset)
    rows = torch.randperm(n, generator=g).to(torch.int32).to(DEV)
    y2 = ops.linear(ops.RowsOf(xd, rows), w, b)
    assert (y2.cpu().double() - want[rows.cpu().long()]).abs().max().item() < 1e-5 * scale
    # no bias
    y3 = ops.linear(xd, w, None)
    assert (y3.cpu().double() - (want - b.cpu().double())).abs().max().item() < 1e-5 * scale


@pytest.mark.parametrize("n,i,o", [(5003, 128, 45), (70001, 128, 45), (4100, 96, 33), (999, 100, 48), (17, 81, 40), (8192, 112, 36)])
def test_linear_three_output_tiles_with_prefetched_rows(n, i, o):
    """33..48 outputs from 81..128 inputs on padded rows (the pooled rows' projection and its data gradient, gnn.py:85-92) take the
    three-tile form of linear_mfma_v4_kernel, which loads the NEXT tile's operand rows -- unconditionally, from clamped places -- before
    this tile's MFMAs.  Both orientations against fp64 within 1e-5 of the output scale: row counts that are no multiple of 16 (the
    prefetch past the last tile must read nothing it uses), a row map, NaN in the operand's pad columns."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(3 * n + i + o)
    x = torch.randn(n, i, generator=g)
    w = (torch.randn(o, i, generator=g) / i ** 0.5).to(DEV)
    b = torch.randn(o, generator=g).to(DEV)
    xd = ops.padded_copy(x.to(DEV))
    if xd.stride(0) > i:
        torch.as_strided(xd, (n, xd.stride(0)), (xd.stride(0), 1))[:, i:] = float("nan")
    want = x.double() @ w.cpu().double().t() + b.cpu().double()
    scale = want.abs().max().item()
    assert (ops.linear(xd, w, b).cpu().double() - want).abs().max().item() < 1e-5 * scale
    rows = torch.randperm(n, generator=g).to(torch.int32).to(DEV)
    y2 = ops.linear(ops.RowsOf(xd, rows), w, b)
    assert (y2.cpu().double() - want[rows.cpu().long()]).abs().max().item() < 1e-5 * scale
    # the data-gradient orientation: gx [n, i] = gy [n, o] W, o on the operand side (o = 33..48 inputs is NOT this form; i = 81..128
    # inputs to 33..48 outputs is: W^T of a 45 -> 128 layer)
    wt = (torch.randn(i, o, generator=g) / i ** 0.5).to(DEV)            # the weight of an o -> i layer
    want_t = x.double() @ wt.cpu().double()
    yt = ops.linear(xd, wt, None, transposed=True)
    assert (yt.cpu().double() - want_t).abs().max().item() < 1e-5 * want_t.abs().max().item()


@pytest.mark.parametrize("n,i,o", [(5003, 192, 75), (4100, 300, 22), (3000, 180, 45), (2049, 130, 128)])
def test_linear_data_gradient_splits_wide_operands(n, i, o):
    """gx = gy W with more than 128 operand columns (the 192 padded head slots of a three-head TransformerConv's q | k | v | skip gradient,
    gnn.py:178-276) is the sum of matrix-core products over 128-column slices of gy and row slices of W -- not the scalar kernel.  Against
    fp64 within 1e-5 of the output scale, with a gate on the last piece and NaN in the operand's pad columns."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + 5 * i + o)
    gy = torch.randn(n, i, generator=g)
    w = (torch.randn(i, o, generator=g) / i ** 0.5).to(DEV)
    gate = torch.randn(n, o, generator=g)
    gyd = ops.padded_copy(gy.to(DEV))
    if gyd.stride(0) > i:
        torch.as_strided(gyd, (n, gyd.stride(0)), (gyd.stride(0), 1))[:, i:] = float("nan")
    want = gy.double() @ w.cpu().double()
    scale = want.abs().max().item()
    got = ops.linear(gyd, w, None, transposed=True)
    assert (got.cpu().double() - want).abs().max().item() < 1e-5 * scale
    gated = ops.linear(gyd, w, None, transposed=True, gate=ops.padded_copy(gate.to(DEV)), gate_scale=1.25)
    want_g = torch.where(gate.double() > 0, want * 1.25, torch.zeros_like(want))
    assert (gated.cpu().double() - want_g).abs().max().item() < 1e-5 * scale


@pytest.mark.parametrize("n,i,o", [(1, 22, 10), (1000, 22, 45), (777, 45, 30), (64, 35, 15), (5000, 10, 1), (333, 125, 125)])
def test_linear_forward_backward(n, i, o):
    from blackwater.native import functional as F

    g = torch.Generator().manual_seed(n + i + o)
    x = torch.randn(n, i, generator=g)
    w = torch.randn(o, i, generator=g) / i ** 0.5
    b = torch.randn(o, generator=g)
    for relu in (False, True):
        xr, wr, br = (t.clone().double().requires_grad_(True) for t in (x, w, b))
        yr = xr @ wr.t() + br
        if relu:
            yr = yr.relu()
        xg, wg, bg = (t.clone().to(DEV).requires_grad_(True) for t in (x, w, b))
        yg = F.linear(xg, wg, bg, relu=relu)
        assert torch.allclose(yg.detach().cpu().double(), yr.detach(), rtol=1e-5, atol=1e-5)
        go = torch.randn(n, o, generator=g)
        yr.backward(go.double())
        yg.backward(go.to(DEV))
        for a_, b_ in ((xg, xr), (wg, wr), (bg, br)):
            scale = b_.grad.abs().max().item() + 1e-12
            assert (a_.grad.cpu().double() - b_.grad).abs().max().item() / scale < 2e-5


def test_segment_mean_forward_backward():
    from blackwater.native import functional as F
    from blackwater.native.structure import GraphStructure

    sizes = [1, 7, 300, 2, 5000, 31]
    n = sum(sizes)
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32)
    ei = torch.zeros((2, 0), dtype=torch.long, device=DEV)
    s = GraphStructure.from_edge_index(ei, n, graph_ptr=ptr)
    for c in (1, 30, 75):
        x = torch.randn(n, c)
        xr = x.clone().double().requires_grad_(True)
        want = torch.stack([xr[ptr[k]:ptr[k + 1]].mean(0) for k in range(len(sizes))])
        xg = x.clone().to(DEV).requires_grad_(True)
        got = F.segment_mean(xg, s)
        assert torch.allclose(got.detach().cpu().double(), want.detach(), rtol=1e-5, atol=1e-6)
        go = torch.randn(len(sizes), c)
        want.backward(go.double())
        got.backward(go.to(DEV))
        assert torch.allclose(xg.grad.cpu().double(), xr.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("c,padded", [(1, True), (1, False), (10, True), (10, False), (30, True), (125, True)])
def test_segment_pool_weighted_mean_and_backward(c, padded):
    """mlqem_segment_pool_f32 / _bwd: plain and weighted means over row tiles against fp64 loops -- graphs that span
    several 1024-row tiles, several graphs inside one tile, empty graphs (also first and last), one-node graphs; padded
    (16-byte, NaN-poisoned pads) and unpadded rows; the gated backward."""
    from blackwater.native import ops

    sizes = [0, 1, 7, 3000, 0, 0, 2, 5000, 31, 1024, 1, 0]
    n, b = sum(sizes), len(sizes)
    ptr_h = np.concatenate([[0], np.cumsum(sizes)])
    ptr = torch.tensor(ptr_h, dtype=torch.int32, device=DEV)
    gen = torch.Generator().manual_seed(c)
    x, w = torch.randn(n, c, generator=gen), torch.rand(n, generator=gen) * 3 - 1
    xd = _padded(x) if padded else x.to(DEV)
    mean, wmean = ops.segment_pool(xd, ptr, b, weights=w.to(DEV), mean=True, wmean=True)
    only_w = ops.segment_pool(xd, ptr, b, weights=w.to(DEV), mean=False, wmean=True)
    assert only_w[0] is None and torch.equal(only_w[1], wmean)
    x64, w64 = x.double(), w.double()
    for g in range(b):
        seg = slice(ptr_h[g], ptr_h[g + 1])
        want0 = x64[seg].mean(0) if sizes[g] else torch.zeros(c, dtype=torch.float64)
        want1 = (x64[seg] * w64[seg, None]).sum(0) / sizes[g] if sizes[g] else torch.zeros(c, dtype=torch.float64)
        assert torch.allclose(mean[g].cpu().double(), want0, rtol=1e-5, atol=2e-6), g
        assert torch.allclose(wmean[g].cpu().double(), want1, rtol=1e-5, atol=2e-6), g
    # two launches on the same input: bit-identical (fixed summation order, no atomics)
    again = ops.segment_pool(xd, ptr, b, weights=w.to(DEV), mean=True, wmean=True)
    assert torch.equal(again[0], mean) and torch.equal(again[1], wmean)
    # backward, with and without the gate
    g0, g1 = torch.randn(b, c, generator=gen), torch.randn(b, c, generator=gen)
    gate = torch.randn(n, c, generator=gen)
    graph_of = np.repeat(np.arange(b), sizes)
    inv = torch.tensor([1.0 / s if s else 0.0 for s in sizes], dtype=torch.float64)[graph_of]
    base = (g0.double()[graph_of] + w64[:, None] * g1.double()[graph_of]) * inv[:, None]
    gd = _padded(gate) if padded else gate.to(DEV)
    got = ops.segment_pool_bwd(g0.to(DEV), g1.to(DEV), ptr, n, weights=w.to(DEV))
    assert torch.allclose(got.cpu().double(), base, rtol=1e-5, atol=1e-7)
    got = ops.segment_pool_bwd(g0.to(DEV), g1.to(DEV), ptr, n, weights=w.to(DEV), gate=gd, gate_scale=1.25)
    assert torch.allclose(got.cpu().double(), torch.where(gate.double() > 0, base * 1.25, torch.zeros_like(base)), rtol=1e-5, atol=1e-7)
    got = ops.segment_pool_bwd(None, g1.to(DEV), ptr, n, weights=w.to(DEV))
    assert torch.allclose(got.cpu().double(), w64[:, None] * g1.double()[graph_of] * inv[:, None], rtol=1e-5, atol=1e-7)


def test_colsum_scalars_are_the_transposed_propagation_of_ones(g1):
    """GraphStructure.colsum(kind) = P^T 1 against dense algebra, and the arena's copies (gathered by batch assembly)
    equal the ones computed on the batch itself."""
    from blackwater.data.arena import GraphArena
    from blackwater.native.structure import GraphStructure
    from helpers import g1_batch, g1_graph

    batch = g1_batch(g1, [5, 17], self_loops=True)
    n = batch["x"].shape[0]
    ei = batch["edge_index"]
    s = GraphStructure.from_edge_index(ei.to(DEV), n, batch=batch["batch"].to(DEV), num_graphs=2)
    a = torch.zeros(n, n, dtype=torch.float64)
    for src, dst in ei.t().tolist():
        a[dst, src] += 1.0
    off = a - torch.diag(torch.diag(a))
    isq = lambda d: torch.where(d > 0, d.clamp(min=1e-30) ** -0.5, torch.zeros_like(d))
    a_hat = off + torch.eye(n, dtype=torch.float64)
    dm = torch.diag(isq(a_hat.sum(1)))
    want = {"gcn": (dm @ a_hat @ dm).sum(0), "sage": (a / a.sum(1).clamp(min=1)[:, None]).sum(0),
            "cheb": (-torch.diag(isq(off.sum(0))) @ off @ torch.diag(isq(off.sum(0)))).sum(0)}
    for kind, w in want.items():
        assert torch.allclose(s.colsum(kind).cpu().double(), w, rtol=1e-5, atol=1e-6), kind
    xs, eis = [], []
    for i in (5, 17):
        x, e, _ = g1_graph(g1, i)
        loops = np.arange(x.shape[0])
        xs.append(x.astype(np.float32))
        eis.append(np.concatenate([e, np.stack([loops, loops])], axis=1))
    arena = GraphArena.from_arrays(xs, eis, batch["y"].numpy(), batch["noisy"].numpy(), batch["depth"].numpy(),
                                   batch["observable"].numpy(), device=DEV)
    sb = arena.batch([1, 0, 1]).structure
    ref = GraphStructure(sb.num_nodes, sb.in_ptr, sb.in_src, sb.out_ptr, sb.out_dst, sb.loops, sb.graph_ptr, 3)
    for kind in want:
        assert torch.equal(sb.colsum(kind), ref.colsum(kind)), kind


def _padded(t, poison=True):
    """A device copy of ``t`` in the padded row layout, the pad columns holding NaN (they are scratch by contract)."""
    from blackwater.native import ops

    n, c = t.shape
    out = ops.padded_empty(n, c, DEV)
    if poison and out.stride(0) > c and n > 0:
        torch.as_strided(out, (n, out.stride(0)), (out.stride(0), 1)).fill_(float("nan"))
    out.copy_(t.to(DEV))
    return out


@pytest.mark.parametrize("n,i,o,k", [(1, 22, 10, 3), (1000, 22, 10, 3), (777, 10, 1, 2), (4099, 10, 10, 2), (50, 5, 3, 4),
                                     (333, 45, 13, 1)])
def test_linear_over_column_blocks(n, i, o, k):
    """mlqem_linear_parts_f32 / mlqem_linear_wgrad_parts_f32: fan-out (one input, k outputs), fan-in (k inputs summed
    into one output, the data-gradient form) and the k weight gradients in one pass, against fp64 algebra; the pad
    columns of every operand hold NaN and must not leak."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n * 31 + i * 7 + o + k)
    x = torch.randn(n, i, generator=g)
    ws = [torch.randn(o, i, generator=g) / i ** 0.5 for _ in range(k)]
    bs = [torch.randn(o, generator=g) for _ in range(k)]
    ow = (o + 3) // 4 * 4
    wd, bd = [w.to(DEV) for w in ws], [t.to(DEV) for t in bs]
    # fan-out (block 0 with a subtracted weight, as the Clenshaw form of ChebConv uses it; one block without bias)
    xd = _padded(x)
    ys = [_padded(torch.zeros(n, o)) for _ in range(k)]
    minus = [wd[-1]] + [None] * (k - 1)
    ops.linear_parts([xd], wd, ys, w_minus=minus, biases=bd[:-1] + [None])
    for j in range(k):
        wj = ws[j].double() - (ws[-1].double() if j == 0 else 0.0)
        want = x.double() @ wj.t() + (bs[j].double() if j < k - 1 else 0.0)
        assert torch.allclose(ys[j].cpu().double(), want, rtol=1e-5, atol=1e-5), j
    # fan-in, transposed: gx = sum_j g_j @ W_j, gated by a mask matrix
    gs = [torch.randn(n, o, generator=g) for _ in range(k)]
    gd = [_padded(t) for t in gs]
    gx = _padded(torch.zeros(n, i))
    ops.linear_parts(gd, wd, [gx], w_minus=minus, transposed=True)
    want = sum(gs[j].double() @ (ws[j].double() - (ws[-1].double() if j == 0 else 0.0)) for j in range(k))
    assert torch.allclose(gx.cpu().double(), want, rtol=1e-5, atol=1e-5)
    gate = _padded((torch.rand(n, i, generator=g) - 0.4).clamp_min(0.0), poison=False)
    ops.linear_parts(gd, wd, [gx], w_minus=minus, transposed=True, gate=gate, gate_scale=1.25)
    assert torch.allclose(gx.cpu().double(), torch.where(gate.cpu() > 0, want * 1.25, torch.zeros_like(want)), rtol=1e-5, atol=1e-5)
    # weight gradients of all blocks in one pass over x
    gw = torch.empty(k * ow, i, device=DEV)
    gb = torch.empty(k * ow, device=DEV)
    ops.linear_wgrad_parts(gd, xd, gw, gb)
    gw3, gb2 = gw.cpu().double().reshape(k, ow, i), gb.cpu().double().reshape(k, ow)
    for j in range(k):
        want_w, want_b = gs[j].double().t() @ x.double(), gs[j].double().sum(0)
        scale = want_w.abs().max().item() + 1e-12
        assert (gw3[j, :o] - want_w).abs().max().item() / scale < 2e-5
        assert (gb2[j, :o] - want_b).abs().max().item() / (want_b.abs().max().item() + 1e-12) < 2e-5
        assert not gw3[j, o:].any() and not gb2[j, o:].any()          # padding rows: exactly zero, never NaN


def test_linear_over_many_column_blocks_with_row_scale():
    """Six output blocks (96 output columns from one read of x) with a per-block row scale, and the weight gradients of
    seven blocks in one pass -- the widest shapes the column-block kernels are instantiated for."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(9)
    n, i, o = 2500, 22, 10
    x = torch.randn(n, i, generator=g)
    ws = [torch.randn(o, i, generator=g) / i ** 0.5 for _ in range(6)]
    rs = torch.rand(n, generator=g) + 0.5
    xd = _padded(x)
    ys = [_padded(torch.zeros(n, o)) for _ in range(6)]
    ops.linear_parts([xd], [w.to(DEV) for w in ws], ys, rowscales=[None, rs.to(DEV)] + [None] * 4)
    for j in range(6):
        want = x.double() @ ws[j].double().t()
        if j == 1:
            want = want * rs.double()[:, None]
        assert torch.allclose(ys[j].cpu().double(), want, rtol=1e-5, atol=1e-5), j
    # six and seven blocks: the first stage packs the 10 real columns of the 12-wide blocks into four / five MFMA tiles, the
    # second spreads them out again (pad rows zero); "accumulate" adds onto what is there
    for k in (7, 6):
        gs = [torch.randn(n, o, generator=g) for _ in range(k)]
        gw = torch.full((k * 12, i), float("nan"), device=DEV)
        gb = torch.full((k * 12,), float("nan"), device=DEV)
        ops.linear_wgrad_parts([_padded(t) for t in gs], xd, gw, gb)
        gw3 = gw.cpu().double().reshape(k, 12, i)
        gb3 = gb.cpu().double().reshape(k, 12)
        for j in range(k):
            want = gs[j].double().t() @ x.double()
            assert (gw3[j, :o] - want).abs().max().item() / want.abs().max().item() < 2e-5
            assert torch.allclose(gb3[j, :o], gs[j].double().sum(0), rtol=1e-4, atol=1e-4)
        assert torch.equal(gw3[:, o:], torch.zeros_like(gw3[:, o:])) and torch.equal(gb3[:, o:], torch.zeros_like(gb3[:, o:]))
        first = gw.clone()
        ops.linear_wgrad_parts([_padded(t) for t in gs], xd, gw, gb, accumulate=True)
        assert torch.allclose(gw, 2 * first, rtol=1e-6, atol=1e-6)


def test_linear_over_column_blocks_rejects_unpadded_operands():
    from blackwater.native import ops

    x = torch.randn(64, 22, device=DEV)                                # compact rows: stride 22, not a multiple of 4
    y = ops.padded_empty(64, 10, DEV)
    w = torch.zeros(10, 22, device=DEV)
    with pytest.raises(ValueError, match="padded row layout"):
        ops.linear_parts([x], [w], [y])
    with pytest.raises(ValueError, match="shape"):
        ops.linear_parts([ops.padded_empty(64, 22, DEV)], [torch.zeros(12, 24, device=DEV)], [y])
    with pytest.raises(ValueError, match="one side"):
        ops.linear_parts([ops.padded_empty(64, 22, DEV)] * 2, [w, w], [y, y])


def test_mask_handover_between_layers_equals_separate_masking():
    """defer_mask / x_gate_scale (native/functional.py): a two-layer chain of every conv kind gives the same parameter
    gradients whether the hidden ReLU/dropout mask is applied by the producer's backward or by the consumer's
    data-gradient GEMM."""
    from blackwater.nn.conv import ChebConv, GCNConv, SAGEConv
    from blackwater.native.structure import GraphStructure

    g = torch.Generator().manual_seed(5)
    n, e = 3000, 7000
    ei = torch.randint(0, n, (2, e), generator=g)
    s = GraphStructure.from_edge_index(ei.to(DEV), n)
    x = torch.randn(n, 22, generator=g).to(DEV)
    go = torch.randn(n, 3, generator=g).to(DEV)
    for make in (lambda: (GCNConv(22, 10), GCNConv(10, 3)), lambda: (ChebConv(22, 10, K=3), ChebConv(10, 3, K=2)),
                 lambda: (SAGEConv(22, 10), SAGEConv(10, 3)), lambda: (ChebConv(22, 10, K=4), ChebConv(10, 3, K=1))):
        torch.manual_seed(1)
        a, b = (m.to(DEV) for m in make())
        grads = []
        for fused in (False, True):
            for m in (a, b):
                m.zero_grad()
            kw1 = dict(defer_mask=True) if fused else {}
            kw2 = dict(x_gate_scale=1.0 / 0.8) if fused else {}
            h = a(x, s, relu=True, drop_p=0.2, seed=11, **kw1)
            y = b(h, s, **kw2)
            y.backward(go)
            grads.append([p.grad.clone() for m in (a, b) for p in m.parameters()])
        for u, v in zip(*grads):
            assert torch.allclose(u, v, rtol=1e-5, atol=1e-6 * (1 + v.abs().max().item()))


@pytest.mark.parametrize("c", [1, 10, 22])
def test_tiles_full_of_hub_rows(c):
    """More rows above the cooperative-reduction threshold (32 in-edges) in one workgroup tile than its lists hold: a
    complete directed graph on 300 nodes (every row has 299 sources) next to ordinary rows.  Sum and max, with and
    without the ELL side table, against index_add / scatter-max references; two runs must agree bit for bit."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(c)
    hub, n = 300, 1500
    src = torch.arange(hub).repeat_interleave(hub)
    dst = torch.arange(hub).repeat(hub)
    keep = src != dst
    chain = torch.stack([torch.arange(hub, n - 1), torch.arange(hub + 1, n)])
    ei = torch.cat([torch.stack([src[keep], dst[keep]]), chain], dim=1)
    ei = ei[:, torch.randperm(ei.shape[1], generator=g)].to(DEV)
    csr = ops.csr_build(ei, n)
    in_ptr, in_src = csr[0], csr[1]
    ell = ops.ell_from_csr(in_ptr, in_src, n)
    x = ops.padded_empty(n, c, DEV).normal_(generator=torch.Generator(device=DEV).manual_seed(1))
    want_sum = torch.zeros(n, c, device=DEV, dtype=torch.float64).index_add_(0, ei[1], x.double()[ei[0]])
    want_max = x.clone()
    want_max = want_max.scatter_reduce(0, ei[1][:, None].expand(-1, c), x[ei[0]], reduce="amax", include_self=True)
    for side in (ell, None):
        got = ops.csr_aggregate(x, in_ptr, in_src, ell=side)
        again = ops.csr_aggregate(x, in_ptr, in_src, ell=side)
        assert torch.equal(got, again)
        assert torch.allclose(got.double(), want_sum, rtol=1e-5, atol=1e-4)
        mx = ops.csr_segment_max(x, in_ptr, in_src, ell=side)
        assert torch.equal(mx, want_max)


def test_differentiable_aggregation_building_block():
    """functional.csr_aggregate (the generic autograd node: y = act(alpha (R A C x + D x) + beta z + bias)) against the
    same expression written with dense torch ops, forward and the gradients of x, z and bias."""
    from blackwater.native import functional as F
    from blackwater.native.structure import GraphStructure

    g = torch.Generator().manual_seed(11)
    n, e, c = 700, 1600, 7
    ei = torch.randint(0, n, (2, e), generator=g)
    ei = ei[:, ei[0] != ei[1]]
    s = GraphStructure.from_edge_index(ei.to(DEV), n)
    adj = torch.zeros(n, n, dtype=torch.float64).index_put_((ei[1], ei[0]), torch.ones(ei.shape[1], dtype=torch.float64),
                                                            accumulate=True)
    x, z = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)
    bias = torch.randn(c, generator=g)
    cs, rs, ds = (torch.rand(n, generator=g) + 0.5 for _ in range(3))
    ref = [t.clone().double().requires_grad_(True) for t in (x, z, bias)]
    want = (1.5 * (rs.double()[:, None] * (adj @ (cs.double()[:, None] * ref[0])) + ds.double()[:, None] * ref[0])
            - 0.5 * ref[1] + ref[2]).relu()
    dev = [t.clone().to(DEV).requires_grad_(True) for t in (x, z, bias)]
    got = F.csr_aggregate(dev[0], s, cscale=cs.to(DEV), rscale=rs.to(DEV), dself=ds.to(DEV), alpha=1.5, z=dev[1], beta=-0.5,
                          bias=dev[2], relu=True)
    assert torch.allclose(got.detach().cpu().double(), want.detach(), rtol=1e-5, atol=1e-5)
    go = torch.randn(n, c, generator=g)
    want.backward(go.double())
    got.backward(go.to(DEV))
    for a_, b_ in zip(dev, ref):
        scale = b_.grad.abs().max().item() + 1e-12
        assert (a_.grad.cpu().double() - b_.grad).abs().max().item() / scale < 2e-5


def test_relu_dropout_add_forward_backward():
    """mlqem_relu_dropout_f32: y = dropout(relu(x)), sum = y + residual in one launch; keep rate, scaling, mask reuse in
    the backward, and the seed contract (same seed -> same mask, other seed -> other mask)."""
    from blackwater.native import functional as F
    from blackwater.native import ops

    g = torch.Generator().manual_seed(0)
    for n, c, p in ((1, 1, 0.0), (777, 64, 0.5), (1000, 125, 0.3), (33, 10, 0.2)):
        x = torch.randn(n, c, generator=g)
        r = torch.randn(n, c, generator=g)
        xd, rd = x.to(DEV).requires_grad_(True), r.to(DEV).requires_grad_(True)
        s = F.relu_dropout_add(xd, rd, p, 99)
        y = (s - rd).detach()
        kept = ops.relu_dropout(x.to(DEV), p, 99)[0] != 0
        relu = x.clamp(min=0).to(DEV)
        scale = 1.0 / (1.0 - p)
        assert torch.allclose(y[kept], relu[kept] * scale, rtol=1e-6, atol=1e-6)          # kept entries: relu(x) / (1 - p)
        if p > 0 and n * c > 10000:
            frac = (kept & (relu > 0)).sum().item() / (relu > 0).sum().item()
            assert abs(frac - (1 - p)) < 0.02
        go = torch.randn(n, c, generator=g).to(DEV)
        s.backward(go)
        assert torch.equal(rd.grad, go)
        assert torch.allclose(xd.grad, torch.where(kept, go * scale, torch.zeros_like(go)), rtol=1e-6, atol=1e-7)
        again, _ = ops.relu_dropout(x.to(DEV), p, 99)
        other, _ = ops.relu_dropout(x.to(DEV), p, 100)
        assert torch.allclose(again, y, rtol=1e-5, atol=1e-6) and torch.equal(again != 0, kept | (again != 0))
        assert torch.equal(again, ops.relu_dropout(x.to(DEV), p, 99)[0])          # same seed: the same mask, bit for bit
        if p > 0 and n * c > 100:
            assert not torch.equal(other, again)


@pytest.mark.parametrize("n,i,o", [(1, 10, 10), (15, 10, 10), (16, 12, 12), (1000, 10, 10), (4099, 7, 10), (70000, 10, 3)])
def test_linear_bwd_fused_equals_separate_kernels(n, i, o):
    """mlqem_linear_bwd_fused_f32 (gated data gradient + weight gradient + bias gradient in one pass) against fp64 algebra
    and against the two-kernel path it replaces; padded operands with NaN-poisoned pads."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + i)
    gy, gbs, x = torch.randn(n, o, generator=g), torch.randn(n, o, generator=g), torch.randn(n, i, generator=g)
    w = torch.randn(o, i, generator=g)
    gyd, gbd, xd, wd = _padded(gy), _padded(gbs), _padded(x), w.to(DEV)
    gx, gw, gb, cs = ops.linear_bwd_fused(gyd, xd, wd, gb_src=gbd, gate_scale=1.25)
    want_gx = torch.where(x.double() > 0, (gy.double() @ w.double()) * 1.25, torch.zeros(n, i, dtype=torch.float64))
    assert torch.allclose(gx.cpu().double(), want_gx, rtol=1e-5, atol=1e-5)
    want_gw = gy.double().t() @ x.double()
    assert (gw.cpu().double() - want_gw).abs().max().item() < 2e-5 * max(want_gw.abs().max().item(), 1.0)
    want_gb = gbs.double().sum(0)
    assert (gb.cpu().double() - want_gb).abs().max().item() < 2e-5 * max(want_gb.abs().max().item(), 1.0)
    # the column sums of the data gradient (the bias gradient of the layer below)
    want_cs = want_gx.sum(0)
    assert cs.shape == (i,)
    assert (cs.cpu().double() - want_cs).abs().max().item() < 2e-5 * max(want_gx.abs().sum(0).max().item(), 1.0)
    # without a gate and with gb_src = gy
    gx2, _, gb2, cs2 = ops.linear_bwd_fused(gyd, xd, wd)
    assert (cs2.cpu().double() - (gy.double() @ w.double()).sum(0)).abs().max().item() < 2e-5 * max((gy.double() @ w.double()).abs().sum(0).max().item(), 1.0)
    assert torch.allclose(gx2.cpu().double(), gy.double() @ w.double(), rtol=1e-5, atol=1e-5)
    assert (gb2.cpu().double() - gy.double().sum(0)).abs().max().item() < 2e-5 * max(gy.double().sum(0).abs().max().item(), 1.0)
    # the two-kernel path it replaces gives the same data gradient to fp32 rounding
    ref_gx = ops.linear(gyd, wd, transposed=True, gate=xd, gate_scale=1.25)
    assert torch.allclose(ref_gx, gx, rtol=1e-5, atol=1e-6)
    # deterministic
    again = ops.linear_bwd_fused(gyd, xd, wd, gb_src=gbd, gate_scale=1.25)
    assert torch.equal(again[0], gx) and torch.equal(again[1], gw) and torch.equal(again[2], gb) and torch.equal(again[3], cs)


def test_pooled_head_forward_backward():
    """mlqem_pooled_head_f32 / _bwd: the [B, C] x [C] products left of the folded last convs, against torch autograd."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(5)
    for b, c in ((1, 10), (37, 10), (1024, 10), (300, 3)):
        ps = [torch.randn(b, c, generator=g) for _ in range(5)]
        ws = [torch.randn(1, c, generator=g) for _ in range(5)]
        bias = [torch.randn(1, generator=g) for _ in range(3)]
        cols = [0, 1, 1, 2, 2]
        pd = [_padded(p) for p in ps]
        wd = [w.to(DEV) for w in ws]
        bd = [x.to(DEV) for x in bias]
        out = ops.pooled_head([(p, w, k) for p, w, k in zip(pd, wd, cols)], bd)
        pr = [p.double().requires_grad_(True) for p in ps]
        wr = [w.double().requires_grad_(True) for w in ws]
        br = [x.double().requires_grad_(True) for x in bias]
        want = torch.stack([sum((pr[t] @ wr[t].t())[:, 0] for t in range(5) if cols[t] == k) + br[k] for k in range(3)], dim=1)
        assert out.shape == (b, 3) and torch.allclose(out.cpu().double(), want.detach(), rtol=1e-5, atol=1e-5)
        go = torch.randn(b, 3, generator=g)
        want.backward(go.double())
        gps, gw, gb = ops.pooled_head_bwd([(p, w, k) for p, w, k in zip(pd, wd, cols)], bd, go.to(DEV))
        for t in range(5):
            assert torch.allclose(gps[t].cpu().double(), pr[t].grad, rtol=1e-5, atol=1e-6)
            assert (gw[t].cpu().double() - wr[t].grad[0]).abs().max().item() < 2e-5 * max(wr[t].grad.abs().max().item(), 1.0)
        for k in range(3):
            assert abs(gb[k].item() - br[k].grad.item()) < 2e-5 * max(abs(br[k].grad.item()), 1.0)
        again = ops.pooled_head_bwd([(p, w, k) for p, w, k in zip(pd, wd, cols)], bd, go.to(DEV))
        assert torch.equal(again[1], gw) and torch.equal(again[2], gb)


@pytest.mark.parametrize("n,c", [(1000, 125), (37, 5), (5000, 256), (262144 // 8, 64)])
def test_batch_norm_train_matches_torch_in_fp64(n, c):
    """native BatchNorm1d training forward / backward (csrc/bn.hip; the bn1 / bn2 of MLP2 / MLP3, docs/tutorials/mlp.py:45-66)
    against torch.nn.BatchNorm1d evaluated in float64 on the CPU: output, running statistics, all three gradients."""
    from blackwater.native import functional as F

    torch.manual_seed(n + c)
    x = (torch.randn(n, c) * 3.0 + 1.5)
    gy = torch.randn(n, c)
    ref = torch.nn.BatchNorm1d(c).double().train()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.uniform_(-1.0, 1.0)
    bn = torch.nn.BatchNorm1d(c).to(DEV).train()
    bn.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(gy.double())
    xd = x.to(DEV).requires_grad_(True)
    yd = F.batch_norm_train(xd, bn)
    yd.backward(gy.to(DEV))
    rel = lambda a, b: (a.double().cpu() - b).abs().max().item() / (b.abs().max().item() + 1e-12)
    assert rel(yd.detach(), yr.detach()) < 2e-6
    assert rel(xd.grad, xr.grad) < 1e-5
    assert rel(bn.weight.grad, ref.weight.grad) < 1e-5 and rel(bn.bias.grad, ref.bias.grad) < 1e-5
    assert rel(bn.running_mean, ref.running_mean) < 1e-6 and rel(bn.running_var, ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == 1


def test_batch_norm_statistics_of_columns_far_from_zero():
    """Columns with |mean| >> std (un-normalised features: a raw circuit depth of ~300 with a spread of ~1): E[x^2] - mean^2 on
    raw fp32 values loses the variance to cancellation (relative error ~1e-7 (mean/std)^2); the kernels shift every column by
    its first row before squaring, as stable as torch's Welford reduction.  Against float64."""
    from blackwater.native import functional as F

    torch.manual_seed(0)
    n, c = 20000, 16
    mean = torch.tensor([0.0, 1.0, 30.0, 300.0, 3000.0, -3000.0, 1e4, -1e4] * 2)
    std = torch.tensor([1.0] * 8 + [0.01] * 8)
    x = torch.randn(n, c) * std + mean
    from blackwater.native import ops

    ref = torch.nn.BatchNorm1d(c).double().train()
    bn = torch.nn.BatchNorm1d(c).to(DEV).train()
    yr = ref(x.double())
    yd = F.batch_norm_train(x.to(DEV), bn)
    _, got_mean, got_var, _ = ops.batch_norm_train(x.to(DEV), bn.weight.detach(), bn.bias.detach(), bn.eps)
    want_var = x.double().var(0, unbiased=False)
    assert ((got_var.cpu().double() - want_var).abs() / want_var).max().item() < 1e-4      # the naive form is off by 1e-1 ... 1e+3 here
    assert ((got_mean.cpu().double() - x.double().mean(0)).abs() / (mean.abs().double() + 1.0)).max().item() < 1e-6
    # (the normalised VALUES of such columns carry ulp(x) * invstd of fp32 rounding in x - mean whatever the kernel does -- up to
    # 0.1 at |x| = 1e4, std = 0.01 -- so y is compared where x is resolved: the statistics are what the kernels own)
    small = mean.abs() <= 1
    assert (yd.cpu().double() - yr)[:, small].abs().max().item() < 2e-5


@pytest.mark.parametrize("padded", [True, False])
def test_linear_with_more_than_128_inputs_runs_on_the_matrix_cores_in_two_pieces(padded):
    """ops.linear splits inputs wider than 128 columns (the 169/170-wide encode_data_v2_ecr rows of the MLP path) into
    k-pieces for the MFMA kernels, bias on the first, ReLU on the last; against float64."""
    from blackwater.native import ops

    torch.manual_seed(5)
    n, i, o = 3000, 170, 125
    x = torch.randn(n, i, device=DEV)
    if padded:
        x = ops.padded_copy(x)
    w, b = torch.randn(o, i, device=DEV) * 0.1, torch.randn(o, device=DEV)
    want = torch.relu(x.double() @ w.double().t() + b.double())
    got = ops.linear(x, w, b, relu=True)
    assert (got.double() - want).abs().max().item() < 1e-4 * want.abs().max().item()
    got2 = ops.linear(x, w, None, out=got.clone(), accumulate=True)          # accumulate onto an existing matrix
    assert (got2.double() - (want + x.double() @ w.double().t())).abs().max().item() < 1e-4 * want.abs().max().item() * 2


@pytest.mark.parametrize("sizes", [[5, 1, 0, 17, 300], [3000, 9000, 1500, 4096, 2048, 7777], [20000] * 3 + [2089] * 5,
                                   [4097, 1, 8192, 4095, 12288, 0, 5000, 16385], [1025, 1023, 2047, 2049, 1024, 3073, 6000, 1, 513, 511, 512]])
def test_segment_topk_lists_each_graph_by_descending_fitness_ties_to_the_lower_index(sizes):
    """mlqem_segment_topk (PyG ``topk(fitness, ratio, batch)`` of ASAPooling): small graphs go through the segmented sort,
    batches of large graphs (a thousand nodes per graph on average) through the two launches of round 5 -- chunks of 2 048 nodes
    (4 096 until round 6) sorted in LDS, then every node ranked among its graph's chunks (sizes around the chunk boundaries, an empty graph and a graph
    of one node among them) -- both must list, for every graph,
    its ceil(n/2) nodes of largest fitness in descending order with ties broken by the lower index; with and without the
    caller's bound on the graph size.  Fitness values are quantised so that ties are frequent; integers: exact."""
    import numpy as np

    from blackwater.native import ops

    rng = np.random.RandomState(len(sizes))
    sizes = np.asarray(sizes, dtype=np.int64)
    n = int(sizes.sum())
    fit = (rng.randint(0, 50, size=n) / 50.0).astype(np.float32)
    fit[rng.rand(n) < 0.05] *= -1.0
    keep = (sizes + 1) // 2
    gptr = np.zeros(len(sizes) + 1, dtype=np.int32); gptr[1:] = np.cumsum(sizes)
    nptr = np.zeros(len(sizes) + 1, dtype=np.int32); nptr[1:] = np.cumsum(keep)
    want = []
    for g in range(len(sizes)):
        seg = fit[gptr[g]:gptr[g + 1]]
        order = np.lexsort((np.arange(len(seg)), -seg.astype(np.float64)))      # by descending fitness, then ascending index
        want.append(gptr[g] + order[:keep[g]])
    want = np.concatenate(want) if want else np.zeros(0, dtype=np.int64)
    f = torch.from_numpy(fit).to(DEV)
    gp, np_ = torch.from_numpy(gptr).to(DEV), torch.from_numpy(nptr).to(DEV)
    for bound in (int(sizes.max()), 0):
        perm = ops.segment_topk(f, gp, np_, n, len(sizes), int(keep.sum()), max_graph_nodes=bound)
        assert np.array_equal(perm.cpu().numpy().astype(np.int64), want), bound
        # ... and with the slot map of the kept nodes from the same launches (round 6): slot[perm[p]] = p, -1 elsewhere
        perm2, slot = ops.segment_topk(f, gp, np_, n, len(sizes), int(keep.sum()), max_graph_nodes=bound, with_slot=True)
        want_slot = np.full(n, -1, dtype=np.int64); want_slot[want] = np.arange(len(want))
        assert torch.equal(perm2, perm) and np.array_equal(slot[:n].cpu().numpy().astype(np.int64), want_slot), bound


def test_topk_of_more_than_a_million_nodes_replays_from_a_captured_graph():
    """Round 4: above 2^20 keys rocprim's default radix sort switches to Onesweep, whose hipMemsetAsync calls become memset nodes
    of a captured hipGraph -- the second replay of a captured Family B step on 256 100-qubit circuits (2.8 M keys) died inside it.
    mlqem_segment_topk now always takes the merge sort; here 70 graphs of 20 000 nodes (1.4 M keys) are ranked from a captured
    graph, replayed four times on fresh fitness values, and must equal the eager result every time."""
    import numpy as np

    from blackwater.native import ops

    sizes = np.full(70, 20000, dtype=np.int64)
    n = int(sizes.sum())
    keep = (sizes + 1) // 2
    gptr = np.zeros(len(sizes) + 1, dtype=np.int32); gptr[1:] = np.cumsum(sizes)
    nptr = np.zeros(len(sizes) + 1, dtype=np.int32); nptr[1:] = np.cumsum(keep)
    gp, np_ = torch.from_numpy(gptr).to(DEV), torch.from_numpy(nptr).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    f = torch.rand(n, device=DEV, generator=g)
    run = lambda: ops.segment_topk(f, gp, np_, n, len(sizes), int(keep.sum()), max_graph_nodes=20000)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        perm = run()
    for _ in range(4):
        f.copy_(torch.rand(n, device=DEV, generator=g))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(perm, run())


@pytest.mark.parametrize("c", [1, 3, 10, 22, 45])
@pytest.mark.parametrize("sizes", [[700], [3, 0, 170, 1, 171, 0, 0, 900, 12], [40] * 60, [2000, 1, 1, 1, 1500]])
def test_aggregation_with_pooled_means_equals_aggregation_then_pool(c, sizes):
    """mlqem_csr_aggregate_pool_f32: the output AND its per-graph means / weighted means from one launch, against fp64 algebra
    and against the two-launch form -- graphs smaller and larger than a workgroup's tile, empty graphs, hub rows (one node of
    every large graph collects an edge from each of its other nodes), epilogue with z, bias, ReLU; the output itself must be
    bit-equal to the plain launch (the pooled sums are a by-product)."""
    from blackwater.native import ops

    sizes = np.asarray(sizes, dtype=np.int64)
    n, b = int(sizes.sum()), len(sizes)
    gptr = np.zeros(b + 1, dtype=np.int64); gptr[1:] = np.cumsum(sizes)
    rng = np.random.RandomState(c + b)
    src, dst = [], []
    for g in range(b):
        lo, sz = gptr[g], sizes[g]
        if sz < 2:
            continue
        m = int(2 * sz)
        src.append(lo + rng.randint(0, sz, m)); dst.append(lo + rng.randint(0, sz, m))
        if sz >= 100:                                  # a hub row ...
            src.append(lo + np.arange(1, sz)); dst.append(np.full(sz - 1, lo))
            # ... and a hub of the TRANSPOSED structure (a source with an edge to each of the graph's nodes: a barrier has both)
            src.append(np.full(sz - 2, lo + 1)); dst.append(lo + np.arange(2, sz))
    ei = torch.from_numpy(np.stack([np.concatenate(src), np.concatenate(dst)])) if src else torch.zeros((2, 0), dtype=torch.long)
    in_ptr, in_src, *_ = ops.csr_build(ei.to(DEV), n)
    ell = ops.ell_from_csr(in_ptr, in_src, n)
    gen = torch.Generator().manual_seed(c)
    x = ops.padded_copy(torch.randn(n, c, generator=gen).to(DEV))
    z = ops.padded_copy(torch.randn(n, c, generator=gen).to(DEV))
    rs = (torch.rand(n, generator=gen) + 0.5).to(DEV)
    wts = (torch.rand(n, generator=gen) + 0.25).to(DEV)
    bias = torch.randn(c, generator=gen).to(DEV)
    gp = torch.from_numpy(gptr.astype(np.int32)).to(DEV)
    kw = dict(ell=ell, rscale=rs, dself=rs, z=z, beta=0.5, bias=bias, relu=True)
    plain = ops.csr_aggregate(x, in_ptr, in_src, **kw)
    for want_mean in (True, False):
        req = dict(graph_ptr=gp, num_graphs=b, weights=wts, mean=want_mean, wmean=True)
        out = ops.csr_aggregate(x, in_ptr, in_src, pool=req, **kw)
        assert torch.equal(out, plain)
        o64 = plain.double().cpu()
        w64 = wts.double().cpu()
        for g in range(b):
            rows = slice(int(gptr[g]), int(gptr[g + 1]))
            k = max(int(sizes[g]), 1)
            m_want = o64[rows].sum(0) / k
            w_want = (o64[rows] * w64[rows, None]).sum(0) / k
            scale = max(o64.abs().max().item(), 1.0)
            if want_mean:
                assert (req["out_mean"][g].double().cpu() - m_want).abs().max().item() <= 1e-5 * scale
            else:
                assert req["out_mean"] is None
            assert (req["out_wmean"][g].double().cpu() - w_want).abs().max().item() <= 1e-5 * scale
        again = dict(graph_ptr=gp, num_graphs=b, weights=wts, mean=want_mean, wmean=True)
        ops.csr_aggregate(x, in_ptr, in_src, pool=again, **kw)          # a fixed summation order: the same bits every time
        assert torch.equal(again["out_wmean"], req["out_wmean"])
    # the gate bits, and the form that writes nothing but them: bit v of (row, slice) = (out[row, 4 slice + v] > 0); the pool's
    # backward gated by the bits equals the one gated by the activation
    req = dict(graph_ptr=gp, num_graphs=b, weights=wts, mean=True, wmean=True, bits=True, store=False)
    assert ops.csr_aggregate(x, in_ptr, in_src, pool=req, **kw) is None
    cv = (c + 3) // 4
    padded = torch.zeros(n, cv * 4, device=DEV)
    padded[:, :c] = plain
    want_bits = ((padded.reshape(n, cv, 4) > 0).to(torch.int32) * torch.tensor([1, 2, 4, 8], device=DEV, dtype=torch.int32)).sum(-1)
    valid = torch.ones(cv, 4, dtype=torch.bool, device=DEV).reshape(-1)
    valid[c:] = False
    mask_of_valid = (valid.reshape(cv, 4).to(torch.int32) * torch.tensor([1, 2, 4, 8], device=DEV, dtype=torch.int32)).sum(-1)
    got_bits = torch.from_numpy(ops.pool_gate_unpack(req["out_bits"], n, c)).to(DEV) & mask_of_valid    # the pad columns' bits are scratch
    assert torch.equal(got_bits, want_bits.to(torch.int32))
    gm, gwm = torch.randn(b, c, device=DEV), torch.randn(b, c, device=DEV)
    by_act = ops.segment_pool_bwd(gm, gwm, gp, n, weights=wts, gate=plain, gate_scale=1.25)
    by_bits = ops.segment_pool_bwd(gm, gwm, gp, n, weights=wts, gate_bits=req["out_bits"], gate_scale=1.25)
    assert torch.equal(by_act, by_bits)
    if ops.pooled_grad_supported(c):
        # ... the same bits per node, and the transposed aggregation that computes its source from them (csrc/pooled_grad.hip) against
        # the pool's backward written out and gathered: bit-equal, with / without a mean gradient, row scale and self term -- rows of
        # 0, 1, 2, 3-32 and more entries, tiles of many graphs
        got_nodes = torch.from_numpy(ops.pool_node_gates(req["out_bits"], n, c)).to(DEV) & mask_of_valid
        assert torch.equal(got_nodes, want_bits.to(torch.int32))
        _, _, out_ptr, out_dst, _ = ops.csr_build(ei.to(DEV), n)
        oell = ops.ell_from_csr(out_ptr, out_dst, n)
        cs = (torch.rand(n, device=DEV) - 0.5)
        for g0, rs_, ds_, alpha in ((gm, rs, rs * rs, 1.0), (None, None, cs, 1.0), (gm, rs, None, 2.0)):
            written = ops.segment_pool_bwd(g0, gwm, gp, n, weights=wts, gate_bits=req["out_bits"], gate_scale=1.25)
            want = ops.csr_aggregate(written, out_ptr, out_dst, ell=oell, cscale=cs, rscale=rs_, dself=ds_, alpha=alpha)
            pgr = ops.PooledGrad(g0, gwm, gp, n, wts, 1.25, req["out_bits"])
            got, g_rows = pgr.aggregate(out_ptr, out_dst, oell, cs, rscale=rs_, dself=ds_, alpha=alpha)
            assert torch.equal(g_rows[:, :c], written[:, :c])
            assert torch.equal(got[:, :c], want[:, :c])
            assert torch.equal(pgr.materialise(), written)
            # ... without writing the rows at all, and their column sums (a bias gradient) from the bits alone
            got2, none = pgr.aggregate(out_ptr, out_dst, oell, cs, rscale=rs_, dself=ds_, alpha=alpha, want_g=False)
            assert none is None and torch.equal(got2, got)
            sums, want_sums = pgr.colsum().double().cpu(), written[:, :c].double().sum(0).cpu()
            assert (sums - want_sums).abs().max().item() <= 1e-5 * max(written.abs().sum(0).max().item(), 1.0)
    # the switch restores the two-launch form (results agree to fp32 rounding of another summation order)
    with pytest.MonkeyPatch.context() as mp:
        mp.setattr(ops, "_POOL_FUSED", False)
        two = dict(graph_ptr=gp, num_graphs=b, weights=wts, mean=True, wmean=True)
        o2 = ops.csr_aggregate(x, in_ptr, in_src, pool=two, **kw)
        assert "out_mean" not in two                      # switched off: the caller pools
        two["out_mean"], two["out_wmean"] = ops.pooled_means(o2, two)
    assert ops._POOL_FUSED
    one = dict(graph_ptr=gp, num_graphs=b, weights=wts, mean=True, wmean=True)
    ops.csr_aggregate(x, in_ptr, in_src, pool=one, **kw)
    assert (one["out_mean"] - two["out_mean"]).abs().max().item() <= 1e-5 * max(plain.abs().max().item(), 1.0)
    assert (one["out_wmean"] - two["out_wmean"]).abs().max().item() <= 1e-5 * max(plain.abs().max().item(), 1.0)


@pytest.mark.parametrize("n,i,h,o", [(1024, 401, 10, 1), (32, 6, 10, 1), (1, 6, 10, 1), (1000, 22, 15, 4), (77, 600, 16, 8), (65, 257, 3, 2)])
@pytest.mark.parametrize("drop_p", [0.0, 0.3])
def test_two_layer_head_in_one_launch_per_direction(n, i, h, o, drop_p):
    """mlqem_seq2_forward_f32 / _backward_f32 (Linear -> Dropout -> Linear: obs_seq / body_seq of the graph models) against fp64
    algebra with the SAME dropout mask (read back from the op): output, input gradient, all four parameter gradients; the mask
    keeps about 1 - p of the hidden units, differs between seeds and repeats for the same seed; one row, rows that do not fill
    a 64-row chunk, more than 256 input columns (two column chunks)."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + i + h)
    x = torch.randn(n, i, generator=g)
    w1, b1 = torch.randn(h, i, generator=g) / i ** 0.5, torch.randn(h, generator=g)
    w2, b2 = torch.randn(o, h, generator=g) / h ** 0.5, torch.randn(o, generator=g)
    gy = torch.randn(n, o, generator=g)
    xd, w1d, b1d, w2d, b2d, gyd = (t.to(DEV) for t in (x, w1, b1, w2, b2, gy))
    y, hidden, mask = ops.seq2_forward(xd, w1d, b1d, w2d, b2d, drop_p=drop_p, seed=11)
    if drop_p > 0:
        bits = ((mask.cpu().to(torch.int64)[:, None] >> torch.arange(h)) & 1).double()
        if n * h > 2000:
            assert abs(bits.mean().item() - (1 - drop_p)) < 0.03
        _, _, mask2 = ops.seq2_forward(xd, w1d, b1d, w2d, b2d, drop_p=drop_p, seed=11)
        _, _, mask3 = ops.seq2_forward(xd, w1d, b1d, w2d, b2d, drop_p=drop_p, seed=12)
        assert torch.equal(mask, mask2) and (n * h < 64 or not torch.equal(mask, mask3))
        scale = 1.0 / (1.0 - drop_p)
    else:
        assert mask is None
        bits, scale = torch.ones(n, h, dtype=torch.float64), 1.0
    xr, w1r, b1r, w2r, b2r = (t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2))
    hid = (xr @ w1r.t() + b1r) * bits * scale
    want = hid @ w2r.t() + b2r
    want.backward(gy.double())
    tol = lambda ref: 2e-5 * max(1.0, ref.abs().max().item())
    assert (y.cpu().double() - want.detach()).abs().max().item() < tol(want.detach())
    assert (hidden.cpu().double() - hid.detach()).abs().max().item() < tol(hid.detach())
    gx, gw1, gb1, gw2, gb2 = ops.seq2_backward(gyd, xd, w1d, w2d, hidden, mask, drop_p, want_gx=True)
    for got, ref in ((gx, xr.grad), (gw1, w1r.grad), (gb1, b1r.grad), (gw2, w2r.grad), (gb2, b2r.grad)):
        assert (got.cpu().double() - ref).abs().max().item() < tol(ref) * (5 if ref is w1r.grad else 1)
    again = ops.seq2_backward(gyd, xd, w1d, w2d, hidden, mask, drop_p, want_gx=True)
    assert all(torch.equal(a, b) for a, b in zip(again, (gx, gw1, gb1, gw2, gb2)))            # deterministic, ticket back at zero
    nogx = ops.seq2_backward(gyd, xd, w1d, w2d, hidden, mask, drop_p, want_gx=False, want_b1=False)
    assert nogx[0] is None and nogx[2] is None and torch.equal(nogx[1], gw1) and torch.equal(nogx[3], gw2)


@pytest.mark.parametrize("n,d", [(1, 45), (257, 45), (5000, 30), (3333, 125), (40000, 7), (0, 16)])
def test_rank_grad_equals_fp64_weighted_column_sums(n, d):
    """mlqem_rank_grad_f32: ASAPooling's three tiny weight gradients (gpqr^T x' [3, D], g_c^T x, g_a^T segmax, and the weights'
    column sums) from one pass, against fp64; padded rows with NaN in the pad columns of a row-sliced operand; the same bits on a
    second call (fixed-order partial sums)."""
    from blackwater.native import ops

    g = torch.Generator().manual_seed(n + d)
    xs = [ops.padded_copy(torch.randn(n, d, generator=g).to(DEV)) for _ in range(3)]
    ws = [torch.randn(n, 3, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)]
    got = ops.rank_grad(list(zip(ws, xs)))
    again = ops.rank_grad(list(zip(ws, xs)))
    assert [tuple(a.shape) for a, _ in got] == [(3, d), (1, d), (1, d)] and [tuple(b.shape) for _, b in got] == [(3,), (1,), (1,)]
    for (gw, gb), (gw2, gb2), w, x in zip(got, again, ws, xs):
        w2 = (w if w.dim() == 2 else w.unsqueeze(1)).double().cpu()
        want = w2.t() @ x.double().cpu()
        scale = max(1.0, float(want.abs().max())) if n else 1.0
        assert (gw.double().cpu() - want).abs().max().item() <= 2e-5 * scale if n else float(gw.abs().max()) == 0.0
        assert (gb.double().cpu() - w2.sum(0)).abs().max().item() <= 2e-5 * max(1.0, float(w2.sum(0).abs().max())) if n else True
        assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    one, = ops.rank_grad([(ws[1], xs[0])])                     # a single term
    assert torch.equal(one[0], ops.rank_grad([(ws[1], xs[0]), (ws[2], xs[1])])[0][0])


def test_leconv_fitness_and_the_kept_rows_gather_on_short_and_long_rows():
    """mlqem_leconv_fitness_f32 (a thread per row, entries in order; ``long_rows``: a 16-lane group per row), mlqem_gather_scale_rows_f32 and its backward (16-byte slices of padded rows since round 6; compact rows take the per-element forms)
    against their formulas in fp64: f = sigmoid(sum_e (p[src_e] - q_i) + p_i - q_i + r_i); out[k] = x[perm[k]] f[perm[k]];
    g_x[perm[k]] = g_out[k] f, zero rows elsewhere, g_f[perm[k]] = g_out[k] . x[perm[k]]."""
    import numpy as np

    from blackwater.native import ops
    from blackwater.native.structure import GraphStructure

    rng = np.random.RandomState(11)
    n = 3000
    deg = rng.choice([0, 1, 2, 3, 15, 16, 17, 40, 64, 65, 129, 300], size=n)
    src = np.concatenate([rng.choice(n, size=d, replace=False) for d in deg])
    dst = np.repeat(np.arange(n), deg)
    s = GraphStructure.from_edge_index(torch.from_numpy(np.stack([src, dst]).astype(np.int64)).to(DEV), n)
    pqr = torch.from_numpy((rng.standard_normal((n, 3)) * 0.2).astype(np.float32)).to(DEV)
    f = ops.leconv_fitness(pqr, s.in_ptr, s.in_src)
    f_rows = ops.leconv_fitness(pqr, s.in_ptr, s.in_src, long_rows=True)      # the 16-lane-group form of a coarsened graph's rows
    ptr, idx = s.in_ptr.cpu().numpy(), s.in_src.cpu().numpy()
    p64 = pqr.double().cpu().numpy()
    want = np.empty(n)
    for i in range(n):
        e = idx[ptr[i]:ptr[i + 1]]
        want[i] = p64[e, 0].sum() - len(e) * p64[i, 1] + p64[i, 0] - p64[i, 1] + p64[i, 2]
    want = 1.0 / (1.0 + np.exp(-want))
    assert np.abs(f.cpu().numpy() - want).max() < 2e-6 and np.abs(f_rows.cpu().numpy() - want).max() < 2e-6
    for c, padded in ((45, True), (30, True), (7, True), (45, False)):
        x_h = rng.standard_normal((n, c)).astype(np.float32)
        x = torch.from_numpy(x_h).to(DEV)
        if padded:
            base = torch.full((n, (c + 3) // 4 * 4), float("nan"), device=DEV); base[:, :c] = x; x = base[:, :c]      # poisoned pads
        perm_h = rng.permutation(n)[: n // 2].astype(np.int32)
        perm = torch.from_numpy(perm_h).to(DEV)
        out = ops.gather_scale_rows(x, perm, f)
        want_out = x_h[perm_h].astype(np.float64) * want[perm_h][:, None]
        assert np.abs(out[:, :c].cpu().numpy() - want_out).max() < 1e-5
        slot_h = np.full(n, -1, dtype=np.int32); slot_h[perm_h] = np.arange(len(perm_h), dtype=np.int32)
        g_h = rng.standard_normal((len(perm_h), c)).astype(np.float32)
        g = torch.from_numpy(g_h).to(DEV)
        if padded:
            gb = torch.full((len(perm_h), (c + 3) // 4 * 4), float("nan"), device=DEV); gb[:, :c] = g; g = gb[:, :c]
        gx, gf = ops.gather_scale_rows_bwd(g, x, f, torch.from_numpy(slot_h).to(DEV))
        want_gx = np.zeros((n, c)); want_gx[perm_h] = g_h.astype(np.float64) * f.double().cpu().numpy()[perm_h][:, None]
        want_gf = np.zeros(n); want_gf[perm_h] = (g_h.astype(np.float64) * x_h[perm_h]).sum(1)
        assert np.abs(gx[:, :c].cpu().numpy() - want_gx).max() < 1e-5 and np.abs(gf.cpu().numpy() - want_gf).max() < 1e-4
        # ... and the same backward as two launches around the fitness backward (round 6): g_f alone, then g_x' with a rank-3 update
        gf2 = ops.gather_rows_dot(g, x, torch.from_numpy(slot_h).to(DEV))
        assert (gf2 is not None) == padded
        if padded:
            assert torch.equal(gf2, gf)
            g3_h, w3_h = rng.standard_normal((n, 3)).astype(np.float32), rng.standard_normal((3, c)).astype(np.float32)
            gx2 = ops.scatter_scale_rank(g, f, torch.from_numpy(slot_h).to(DEV), torch.from_numpy(g3_h).to(DEV), torch.from_numpy(w3_h).to(DEV), n, c)
            want2 = want_gx + g3_h.astype(np.float64) @ w3_h.astype(np.float64)
            assert np.abs(gx2[:, :c].cpu().numpy() - want2).max() < 2e-5
            base2 = gx2._base if gx2._base is not None else gx2
            assert torch.isfinite(base2).all()                       # the pads of the padded rows: zeros, not the poisoned inputs'
