"""The tiled row walks (csrc/tile_plan.hip, tile_attn.hip, tile_pool.hip; round 5) against the per-edge kernels they replace on
ASAPooling's coarsened graphs (docs/tutorials/gnn.py:80-92,104-112: TransformerConv 2 and ASAPooling 2 of every reference GNN).

The per-edge kernels are themselves pinned to dense fp64 algebra and to the oracle (test_gpu_family_b.py); here the tiled forms must
reproduce them on graphs shaped like the coarsened ones (blocks of rows sharing their sources, rows of 0-200 entries), with a slot
capacity small enough to send entries down the overflow branch, with and without self entries, and with dropout (keyed by
(destination, head, source): the same draws in both forms).  Tolerance: 2e-5 of each result's scale (another fp32 summation order);
integer outputs (the plan itself) exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _blocky_graphs(rng, sizes, loops_p=0.0):
    """Graphs whose rows share sources in blocks, like S^T A S of a circuit with barriers: per graph a few hubs, each with a window of
    ids; a row is short (0-3 sources anywhere near) or long (60-200 sources of one hub's window).  No parallel edges, no stored self
    entries (``loops`` carries those).  Returns (edge_index [2, E], loops [N], sizes)."""
    src, dst, off = [], [], 0
    for n in sizes:
        hubs = max(1, n // 150)
        for i in range(n):
            hub = rng.randint(hubs)
            lo = hub * n // hubs
            win = np.arange(lo, min(n, lo + max(40, min(320, n // hubs + 60))))
            win = win[win != i]
            if rng.rand() < 0.3 and len(win) >= 60:
                d = rng.randint(60, min(200, len(win)) + 1)
            else:
                d = rng.randint(0, 4)
            picks = rng.choice(win, size=min(d, len(win)), replace=False) if len(win) and d else np.zeros(0, np.int64)
            src.append(picks + off)
            dst.append(np.full(len(picks), i + off))
        off += n
    ei = np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64)
    n_total = int(sum(sizes))
    loops = (rng.rand(n_total) < loops_p).astype(np.int64)
    return ei, loops, n_total


def _structure(ei, loops, n, sizes):
    from blackwater.native.structure import GraphStructure

    full = np.concatenate([ei, np.repeat(np.stack([np.arange(n)] * 2), loops, axis=1)], axis=1)
    ptr = np.zeros(len(sizes) + 1, np.int32)
    ptr[1:] = np.cumsum(sizes)
    s = GraphStructure.from_edge_index(torch.from_numpy(full).to(DEV), n, graph_ptr=torch.from_numpy(ptr))
    s.out_eid = None                         # the recomputing backward forms, as on the coarsened graphs
    return s


def _plans(s, sizes, order, tile_rows, cap, max_span=None):
    from blackwater.native import ops

    max_span = 2 * max(sizes) + tile_rows if max_span is None else max_span
    mk = lambda ptr, idx: ops.tile_plan_build(ptr, idx, s.num_nodes, int(idx.shape[0]), order, max_span, tile_rows=tile_rows, cap=cap)
    return mk(s.in_ptr, s.in_src), mk(s.out_ptr, s.out_dst)


@pytest.mark.parametrize("tile_rows,cap,max_span", [(32, 224, None), (16, 40, None), (128, 64, None), (32, 224, 256)])
def test_plan_lists_every_entry_at_its_source(tile_rows, cap, max_span):
    """uni[tile, loc[e]] == idx[e] for every entry with a slot; an entry is without one only when its tile's union outgrew `cap` (or the
    bitset: ``max_span`` = 256 ids); slots ascend with the source id; a tile's row records list its rows, the long ones first, with their
    CSR range and their offset in the tile's entry list."""
    rng = np.random.RandomState(tile_rows)
    sizes = [700, 1, 333, 64, 2]
    ei, loops, n = _blocky_graphs(rng, sizes)
    s = _structure(ei, loops, n, sizes)
    order = torch.cat([torch.from_numpy(rng.permutation(k) + o) for k, o in zip(sizes, np.cumsum([0] + sizes[:-1]))]).to(torch.int32).to(DEV)
    plan, _ = _plans(s, sizes, order, tile_rows, cap, max_span)
    ptr, idx = s.in_ptr.cpu().numpy(), s.in_src.cpu().numpy()
    tinfo, rinfo = plan.tinfo.cpu().numpy(), plan.rinfo.cpu().numpy()
    uni, loc = plan.uni.cpu().numpy(), plan.loc.cpu().numpy().view(np.uint16)
    order_h = order.cpu().numpy()
    seen = np.zeros(n, bool)
    assert plan.num_tiles == (n + tile_rows - 1) // tile_rows
    for t in range(plan.num_tiles):
        rank0, cnt = t * tile_rows, min(tile_rows, n - t * tile_rows)
        assert tinfo[t, 0] == cnt
        rec = rinfo[rank0:rank0 + cnt]
        rows = rec[:, 0]
        assert sorted(rows) == sorted(order_h[rank0:rank0 + cnt])
        deg = ptr[rows + 1] - ptr[rows]
        assert (rec[:, 1] == ptr[rows]).all() and (rec[:, 2] == deg).all()
        assert (rec[:, 3] == np.concatenate([[0], np.cumsum(deg)[:-1]])).all() and tinfo[t, 3] == deg.sum()
        nlong = tinfo[t, 1]
        assert (deg[:nlong] >= 32).all() and (deg[nlong:] < 32).all()
        ent = np.concatenate([idx[ptr[r]:ptr[r + 1]] for r in rows])
        union = np.unique(ent)
        if max_span is not None and len(union):
            union = union[union < union[0] + max_span]                    # ids past the bitset have no slot
        assert tinfo[t, 2] == min(len(union), cap)
        assert (uni[t * cap:t * cap + tinfo[t, 2]] == union[:cap]).all()
        for r in rows:
            seen[r] = True
            for e in range(ptr[r], ptr[r + 1]):
                where = np.searchsorted(union, idx[e])
                has = where < min(len(union), cap) and union[where] == idx[e]
                assert loc[e] == (where if has else 0xFFFF)
    assert seen.all()


def test_order_by_position_is_the_argsort_of_the_kept_nodes():
    from blackwater.native import ops

    rng = np.random.RandomState(5)
    sizes = [1000, 3, 0, 517]
    keep = [(k + 1) // 2 for k in sizes]
    gptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    nptr = np.concatenate([[0], np.cumsum(keep)]).astype(np.int32)
    perm = np.concatenate([np.sort(rng.choice(k, size=kk, replace=False))[rng.permutation(kk)] + o for k, kk, o in zip(sizes, keep, gptr[:-1])])
    perm_d = torch.from_numpy(perm.astype(np.int32)).to(DEV)
    slot = ops.asap_slot_map(perm_d, int(gptr[-1]))
    order = ops.tile_order_by_position(slot, torch.from_numpy(gptr).to(DEV), torch.from_numpy(nptr).to(DEV), len(sizes), int(nptr[-1]))
    want = np.concatenate([np.argsort(perm[a:b], kind="stable") + a for a, b in zip(nptr[:-1], nptr[1:])])
    assert (order.cpu().numpy() == want).all()


@pytest.mark.parametrize("heads,ch,cap,loops_p", [(2, 15, 224, 0.0), (3, 15, 48, 0.5), (5, 13, 128, 0.0), (1, 16, 32, 1.0), (2, 14, 100, 0.3)])
def test_tiled_attention_equals_the_per_edge_kernels(heads, ch, cap, loops_p):
    from blackwater.native import ops

    rng = np.random.RandomState(heads * 100 + ch)
    sizes = [900, 130, 7, 400]
    ei, loops, n = _blocky_graphs(rng, sizes, loops_p)
    s = _structure(ei, loops, n, sizes)
    plan_in, plan_out = _plans(s, sizes, None, 32, cap)
    hc, cp = heads * ch, 16
    g = torch.Generator().manual_seed(ch)
    q_h = torch.zeros(n, 4 * heads, cp)
    q_h[:, :, :ch] = torch.randn(n, 4 * heads, ch, generator=g)
    qkvs = ops.padded_copy(q_h.view(n, 4 * heads * cp).to(DEV))
    gout = ops.padded_copy(torch.randn(n, hc, generator=g).to(DEV))
    e = s.edge_count()
    for drop_p in (0.0, 0.25):
        ref = ops.transformer_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, drop_p, 11, pair_key=True, head_pitch=cp)
        got = ops.tile_attention(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, plan_in, drop_p=drop_p, seed=11, head_pitch=cp)
        scale = max(1.0, ref[0][:, :hc].abs().max().item())
        assert (ref[0][:, :hc] - got[0][:, :hc]).abs().max().item() < 2e-5 * scale
        fin = torch.isfinite(ref[2])                                          # a row without entries keeps m = -inf in both forms
        assert torch.equal(fin, torch.isfinite(got[2]))
        assert (ref[2][fin] - got[2][fin]).abs().max().item() < 1e-5 * max(1.0, ref[2][fin].abs().max().item())
        assert ((ref[3] - got[3]).abs() / ref[3].abs().clamp_min(1.0)).max().item() < 2e-5
        gref = ops.transformer_attention_bwd(qkvs, gout, ref[1], ref[2], ref[3], s, e, heads, ch, drop_p, 11, pair_key=True, head_pitch=cp)
        # the per-edge destination side forms g . attn_out of rows of <= 4 entries itself; the tiled one reads the forward's attn_out
        ggot = ops.tile_attention_bwd(qkvs, gout, got[1], got[2], got[3], s, e, heads, ch, plan_in, plan_out, drop_p=drop_p, seed=11,
                                      head_pitch=cp)
        w = 4 * heads * cp
        gs = max(1.0, gref[:, :w].abs().max().item())
        assert (gref[:, :w] - ggot[:, :w]).abs().max().item() < 5e-5 * gs, drop_p
        assert torch.isfinite(ggot[:, :w]).all()
        if drop_p == 0.0:
            inf = ops.tile_attention(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, plan_in, head_pitch=cp, train=False)
            assert (inf[:, :hc] - got[0][:, :hc]).abs().max().item() < 1e-6 * scale


@pytest.mark.parametrize("d,cap", [(30, 224), (45, 40), (16, 224), (7, 24), (33, 64)])
def test_tiled_pooling_scores_equal_the_per_edge_kernels(d, cap):
    """mlqem_tile_asap_scores_f32 / _bwd_f32 against segment max + composed score + softmax-sum + the LEConv projections and their
    per-edge backward kernels; x carries exact ties (repeated rows and channels) so that the even split of the maximum's gradient is
    exercised; pad columns of x are NaN."""
    from blackwater.native import ops

    rng = np.random.RandomState(d)
    sizes = [800, 90, 3, 350]
    ei, loops, n = _blocky_graphs(rng, sizes)
    s = _structure(ei, loops, n, sizes)
    plan_in, plan_out = _plans(s, sizes, None, 32, cap)
    g = torch.Generator().manual_seed(d)
    x_h = torch.randn(n, d, generator=g)
    x_h[rng.choice(n, n // 3)] = x_h[rng.choice(n, n // 3)]               # repeated rows: ties in every channel
    x_h[:, 0] = torch.round(x_h[:, 0])                                   # a channel of few distinct values
    x = ops.padded_empty(n, d, DEV)
    torch.as_strided(x, (n, x.stride(0)), (x.stride(0), 1)).fill_(float("nan"))
    x.copy_(x_h.to(DEV))
    w_comp = torch.randn(1, d, generator=g).to(DEV)
    b_comp = torch.randn(1, generator=g).to(DEV)
    att_x = torch.randn(1, d, generator=g).to(DEV)
    w3, b3 = torch.randn(3, d, generator=g).to(DEV), torch.randn(3, generator=g).to(DEV)
    slope = 0.2
    c_src = ops.linear(x, att_x)[:, 0].contiguous()
    # per-edge reference
    xmax_r = ops.csr_segment_max(x, s.in_ptr, s.in_src, ell=s.in_ell)
    a_dst = ops.linear(xmax_r, w_comp.contiguous(), b_comp)[:, 0].contiguous()
    xnew_r = ops.csr_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, slope)
    pqr_r = ops.linear(xnew_r, w3, b3).contiguous()
    xnew, xmax, stat, pqr = ops.tile_asap_scores(x, s.in_ptr, s.in_src, c_src, w_comp[0].contiguous(), b_comp, w3, b3, slope, plan_in)
    assert torch.equal(xmax[:, :d], xmax_r[:, :d])
    tol = lambda ref: 2e-5 * max(1.0, ref.abs().max().item())
    assert (stat[:n, 0] - a_dst).abs().max().item() < tol(a_dst)
    assert (xnew[:, :d] - xnew_r[:, :d]).abs().max().item() < tol(xnew_r[:, :d])
    assert (pqr - pqr_r[:, :3]).abs().max().item() < tol(pqr_r[:, :3])
    # backward
    gnew = ops.padded_copy(torch.randn(n, d, generator=g).to(DEV))
    e = s.edge_count()
    gx_r, ga_r, gc_r, ties = ops.csr_softmax_aggregate_bwd(x, xnew_r, gnew, s, e, a_dst, c_src, slope, xmax=xmax_r, gx_rank1=att_x[0])
    ops.csr_segment_max_bwd_(gx_r, x, xmax_r, None, s, ties=ties, gmax_rank1=(ga_r, w_comp[0].contiguous()))
    gx, ga, gc = ops.tile_asap_scores_bwd(x, xnew, gnew, xmax, s, c_src, w_comp[0].contiguous(), att_x[0].contiguous(), slope, plan_in, plan_out, stat)
    assert (ga - ga_r).abs().max().item() < tol(ga_r)
    assert (gc - gc_r).abs().max().item() < tol(gc_r)
    assert (gx[:, :d] - gx_r[:, :d]).abs().max().item() < 2.5 * tol(gx_r[:, :d])
    assert torch.isfinite(gx[:, :d]).all()


def _family_b_on_100q(tiles_on, train, dense_on=False):
    from blackwater.data.arena import GraphArena
    from blackwater.data.synthetic import tfim_corpus
    from blackwater.native import functional as F
    from blackwater.nn import ExpValCircuitGraphModel

    corpus = tfim_corpus(100, [2, 4], 2, seed=42, two_q="ecr", exp_value_size=4)
    arena = GraphArena.from_arrays(corpus["x"], corpus["edge_index"], corpus["y"][:, None, :], corpus["noisy"][:, None, :], corpus["depth"],
                                   corpus["observable"], device=DEV)
    torch.manual_seed(0)
    model = ExpValCircuitGraphModel(22, 15).to(DEV)
    model.train(train)
    was = F._TILES, F._DENSE_BLOCKS
    F._TILES, F._DENSE_BLOCKS = tiles_on, dense_on       # both off: the per-edge kernels on every row
    try:
        batch = arena.batch(np.arange(len(arena)))
        out = model(*batch.model_args())
        grads = None
        if train:
            model.zero_grad()
            (out * torch.linspace(1.0, 2.0, out.numel(), device=DEV).view_as(out)).sum().backward()
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        return out.detach().clone(), grads
    finally:
        F._TILES, F._DENSE_BLOCKS = was


def test_family_b_on_100_qubit_graphs_tiled_equals_per_edge():
    """The whole model (docs/tutorials/gnn.py:70-122) on four 100-qubit circuits, eval mode and train mode with dropout (the same draws in
    both forms), tiled level-1 kernels against the per-edge ones: predictions within 1e-5 of their scale, every parameter gradient
    within 2e-4 of the largest gradient (the two forms sum a row's 200 entries in different orders)."""
    out_t, _ = _family_b_on_100q(True, train=False)
    out_e, _ = _family_b_on_100q(False, train=False)
    assert (out_t - out_e).abs().max().item() < 1e-5 * max(1.0, out_e.abs().max().item())
    out_t, g_t = _family_b_on_100q(True, train=True)
    out_e, g_e = _family_b_on_100q(False, train=True)
    assert (out_t - out_e).abs().max().item() < 2e-5 * max(1.0, out_e.abs().max().item())
    gmax = max(v.abs().max().item() for v in g_e.values())
    for k in g_e:
        assert (g_t[k] - g_e[k]).abs().max().item() < 2e-4 * gmax, k
