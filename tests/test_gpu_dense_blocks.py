"""Dense blocks (csrc/dense_block.hip; round 5): TransformerConv's edge softmax over the long rows of ASAPooling's coarsened graphs on
the f32 matrix cores (docs/tutorials/gnn.py:80-91: the second TransformerConv of every reference GNN), against the per-edge kernels.

The per-edge kernels are pinned to dense fp64 algebra and to the oracle (test_gpu_family_b.py); here the block forms must reproduce
them on graphs shaped like the coarsened ones -- long rows that share their sources, short rows between them, graphs of every size,
with and without self entries, with dropout (keyed by (destination, head, source): the same draws in both forms) -- and the plan
itself must list exactly the structure's entries.  Tolerance: 2e-5 of each result's scale (another fp32 summation order, exp2 on the
hardware's transcendental unit); the plan exactly."""
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


def _family_b_on_100q(train, dense_on=False):
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
    was = F._DENSE_BLOCKS
    F._DENSE_BLOCKS = dense_on       # off: the per-edge kernels on every row
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
        F._DENSE_BLOCKS = was


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


SIZES = [700, 1, 333, 64, 2, 1500]
CAP = 512            # kDbCap of csrc/dense_block.hpp: union slots of a block


def _make(seed, loops_p, sizes=SIZES, permute=True):
    rng = np.random.RandomState(seed)
    ei, loops, n = _blocky_graphs(rng, sizes, loops_p)
    s = _structure(ei, loops, n, sizes)
    # "program order": a graph's rows sorted by where their sources sit (rows around one hub follow each other, as the clusters
    # around one barrier of a circuit do), ties and source-less rows shuffled -- blocks that straddle two hubs outgrow the capacity
    # and stay with the per-edge kernels
    centre = np.full(n, -1.0)
    sums, cnts = np.bincount(ei[1], weights=ei[0], minlength=n), np.bincount(ei[1], minlength=n)
    centre[cnts > 0] = np.floor(sums[cnts > 0] / cnts[cnts > 0] / 40.0)
    centre += rng.rand(n) * (0.5 if permute else 0.0)
    offs = np.cumsum([0] + sizes[:-1])
    centre[offs[-1]:] = rng.rand(sizes[-1])            # ... and the last graph's rows in NO order: its blocks outgrow the capacity
    order = torch.cat([torch.from_numpy(np.argsort(centre[o:o + k], kind="stable") + o) for k, o in zip(sizes, offs)])
    order = order.to(torch.int32).to(DEV)
    s.set_block_order(lambda: (order, max(sizes) + 8))
    return s, order, rng


@pytest.mark.parametrize("direction", ["in", "out"])
@pytest.mark.parametrize("loops_p", [0.0, 0.6])
def test_plan_lists_exactly_the_entries_of_the_long_rows(direction, loops_p):
    """Every row of 32+ entries sits in exactly one block, in program order inside its graph; a block's union is the sorted set of its
    rows' sources and the rows themselves; a cell's bit is set exactly when the cell is an entry (or the row's own self-loop)."""
    from blackwater.native import _lib

    s, order, _ = _make(3, loops_p)
    plan = s.dense_plan(direction)
    ptr, idx = (s.in_ptr, s.in_src) if direction == "in" else (s.out_ptr, s.out_dst)
    ptr, idx, loops = ptr.cpu().numpy(), idx.cpu().numpy(), s.loops.cpu().numpy()
    stride = _lib.load().mlqem_dense_plan_record_ints()
    nblocks = int(plan.counter.item()) // 16
    rec = plan.records.cpu().numpy()[: nblocks * stride].reshape(nblocks, stride)
    flag = plan.row_flag.cpu().numpy()
    deg = np.diff(ptr[: s.num_nodes + 1])
    gid = np.repeat(np.arange(len(SIZES)), SIZES)
    order_h = order.cpu().numpy()
    rank = np.empty(s.num_nodes, np.int64)
    rank[order_h] = np.arange(s.num_nodes)
    seen = np.zeros(s.num_nodes, bool)
    assert nblocks > 10
    usable = 0
    for b in range(nblocks):
        nrows, nu, ok, entries = rec[b, :4]
        rows = rec[b, 4:4 + nrows]
        assert 1 <= nrows <= 16 and (rec[b, 4 + nrows:20] == -1).all()
        assert (deg[rows] >= 32).all() and len(set(gid[rows])) == 1 and (np.diff(rank[rows]) > 0).all()
        assert not seen[rows].any()
        seen[rows] = True
        assert entries == deg[rows].sum()
        union = np.unique(np.concatenate([idx[ptr[r]:ptr[r + 1]] for r in rows] + [rows]))
        assert ok == (len(union) <= CAP) and (flag[rows] == ok).all()
        usable += int(ok)
        if not ok:
            continue
        assert nu == len(union) and (rec[b, 36:36 + nu] == union).all() and (rec[b, 36 + nu:36 + CAP] == rows[0]).all()
        mask = rec[b, 36 + CAP:36 + CAP + CAP // 2].view(np.uint32).reshape(16, CAP // 32)
        bits = ((mask[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(16, CAP).astype(bool)
        for i, r in enumerate(rows):
            want = np.isin(union, idx[ptr[r]:ptr[r + 1]])
            selfs = np.searchsorted(union, r)
            assert rec[b, 20 + i] == selfs
            if loops[r] > 0:
                want[selfs] = True
            assert (bits[i, :nu] == want).all() and not bits[i, nu:].any()
        assert not bits[nrows:].any()
    assert (seen == (deg >= 32)).all() and not flag[deg < 32].any()
    # both kinds of block occur (the out-structure's unions -- the long rows of a few hubs -- are small in every order)
    assert 0 < usable and (usable < nblocks or direction == "out")


def _qkvs(rng, n, heads, ch):
    from blackwater.native import ops

    q = torch.zeros(n, 4 * heads, 16)
    q[:, :, :ch] = torch.from_numpy(rng.standard_normal((n, 4 * heads, ch)).astype(np.float32))
    return ops.padded_copy(q.view(n, -1).to(DEV))


def _close(a, b, what, tol=2e-5):
    scale = max(1.0, b.abs().max().item())
    err = (a - b).abs().max().item()
    assert err < tol * scale, (what, err, scale)


@pytest.mark.parametrize("heads,ch,drop_p,loops_p", [(2, 15, 0.1, 0.0), (2, 15, 0.0, 0.7), (1, 16, 0.25, 0.5), (2, 13, 0.1, 1.0), (3, 15, 0.1, 0.3), (3, 16, 0.0, 0.0)])
def test_attention_on_the_blocks_equals_the_per_edge_kernels(heads, ch, drop_p, loops_p):
    """Forward (out, attn_out, both statistics) and backward (the gradient of [query | key | value | skip]) of the edge softmax."""
    from blackwater.native import ops

    s, _, rng = _make(11 + heads, loops_p)
    n, e = s.num_nodes, s.edge_count()
    assert ops.dense_attention_supported(heads, ch, 16)
    qkvs = _qkvs(rng, n, heads, ch)
    pin, pout = s.dense_plan("in"), s.dense_plan("out")
    assert int(pin.row_flag.sum().item()) > 100 and int(pout.row_flag.sum().item()) > 100
    ref = ops.transformer_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, drop_p, 7, pair_key=True, head_pitch=16)
    got = ops.dense_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, pin, drop_p=drop_p, seed=7)
    deg = (s.in_ptr[1:n + 1] - s.in_ptr[:n]).cpu() + (s.loops.cpu() > 0).int()
    stored = (deg > 4).to(DEV)                                   # rows of at most four entries leave attn_out unwritten
    _close(got[0], ref[0], "out")
    _close(got[1][stored], ref[1][stored], "attn_out")
    fin = torch.isfinite(ref[2])
    _close(got[2][fin], ref[2][fin], "m")
    _close(got[3], ref[3], "den")
    gout = ops.padded_copy(torch.from_numpy(rng.standard_normal((n, heads * ch)).astype(np.float32)).to(DEV))
    gref = ops.transformer_attention_bwd(qkvs, gout, ref[1], ref[2], ref[3], s, e, heads, ch, drop_p, 7, pair_key=True, head_pitch=16)
    ggot = ops.dense_attention_bwd(qkvs, gout, got[1], got[2], got[3], s, e, heads, ch, pin, pout, drop_p=drop_p, seed=7)
    for part, name in enumerate(("query", "key", "value", "skip")):
        w = heads * 16
        _close(ggot[:, part * w:(part + 1) * w], gref[:, part * w:(part + 1) * w], "gradient of " + name)
    # the two launches of a call touch disjoint rows: on two streams (the `parts` of the entry points) they give the same arrays
    side = torch.cuda.Stream()
    two = ops.dense_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, pin, drop_p=drop_p, seed=7, side=side)
    assert all(torch.equal(a[stored] if k == 1 else a, b[stored] if k == 1 else b) for k, (a, b) in enumerate(zip(two, got)))
    gtwo = ops.dense_attention_bwd(qkvs, gout, got[1], got[2], got[3], s, e, heads, ch, pin, pout, drop_p=drop_p, seed=7, side=side)
    assert torch.equal(gtwo, ggot)


def test_a_structure_without_long_rows_has_no_blocks_and_the_same_results():
    from blackwater.native import ops

    rng = np.random.RandomState(2)
    sizes = [40, 9]
    ei = np.stack([rng.randint(0, 40, 90), rng.randint(0, 40, 90)])
    ei = np.unique(ei[:, ei[0] != ei[1]], axis=1)
    s = _structure(ei, np.zeros(49, np.int64), 49, sizes)
    s.set_block_order(lambda: (None, 64))
    pin = s.dense_plan("in")
    assert int(pin.counter.item()) == 0 and int(pin.row_flag.sum().item()) == 0
    qkvs = _qkvs(rng, 49, 2, 15)
    e = s.edge_count()
    ref = ops.transformer_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, 2, 15, 0.1, 3, pair_key=True, head_pitch=16)
    got = ops.dense_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, 2, 15, pin, drop_p=0.1, seed=3)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[3], ref[3])


@pytest.mark.parametrize("d,loops_p", [(30, 0.0), (30, 0.8), (16, 0.3), (7, 0.0), (32, 0.5), (45, 0.3), (48, 0.0), (40, 0.5)])
def test_pooling_cluster_sums_on_the_blocks_equal_the_per_edge_kernels(d, loops_p):
    """ASAPooling's softmax-weighted cluster sum x' (the row itself is ALWAYS an entry there, whatever ``loops`` says)."""
    from blackwater.native import ops

    s, _, rng = _make(29 + d, loops_p)
    n = s.num_nodes
    assert ops.dense_pool_supported(d)
    x = ops.padded_copy(torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(DEV))
    a_dst = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(DEV)
    c_src = torch.from_numpy(rng.standard_normal(n).astype(np.float32) * 2).to(DEV)
    pin = s.dense_plan("in")
    ref = ops.csr_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, 0.2)
    got, stat = ops.dense_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, 0.2, pin)
    _close(got, ref, "x'")
    flag = pin.row_flag.bool()
    assert int(flag.sum().item()) > 100 and torch.equal(got[~flag], ref[~flag])          # the other rows: the same kernel, bit for bit
    assert torch.isfinite(stat[flag]).all() and (stat[flag][:, 1] > 0).all()


@pytest.mark.parametrize("d,loops_p,ties", [(30, 0.0, False), (30, 0.8, True), (32, 0.5, True), (29, 0.2, False), (45, 0.3, True), (48, 0.0, False), (46, 0.6, True)])
def test_pooling_walks_on_the_blocks_equal_the_per_edge_kernels(d, loops_p, ties):
    """ASAPooling's other walks over a coarsened graph: the segment max (the row itself included), the backward of the cluster sum
    (g_x, g_a, g_c, the tie counts of the maximum) and the backward of the maximum.  ``ties``: x drawn from five values, so that maxima
    are attained many times (identical gates on one qubit have identical feature rows: the gradient is split evenly)."""
    from blackwater.native import ops

    s, _, rng = _make(41 + d, loops_p)
    n, e = s.num_nodes, s.edge_count()
    xh = rng.randint(0, 5, size=(n, d)).astype(np.float32) if ties else rng.standard_normal((n, d)).astype(np.float32)
    x = ops.padded_copy(torch.from_numpy(xh).to(DEV))
    assert ops.dense_pool_fits(x)
    pin, pout = s.dense_plan("in"), s.dense_plan("out")
    xmax_ref = ops.csr_segment_max(x, s.in_ptr, s.in_src)
    xmax = ops.dense_segment_max(x, s.in_ptr, s.in_src, pin)
    assert torch.equal(xmax[:, :d], xmax_ref[:, :d])
    a_dst = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(DEV)
    c_src = torch.from_numpy(rng.standard_normal(n).astype(np.float32) * 2).to(DEV)
    w_comp = torch.from_numpy(rng.standard_normal(d).astype(np.float32)).to(DEV)
    att_x = torch.from_numpy(rng.standard_normal(d).astype(np.float32)).to(DEV)
    xnew_ref = ops.csr_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, 0.2)
    xnew, stat = ops.dense_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, 0.2, pin)
    gnew = ops.padded_copy(torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(DEV))
    gx_r, ga_r, gc_r, ties_r = ops.csr_softmax_aggregate_bwd(x, xnew_ref, gnew, s, e, a_dst, c_src, 0.2, xmax=xmax_ref, gx_rank1=att_x)
    gx, ga, gc, ties_d = ops.dense_softmax_aggregate_bwd(x, xnew, gnew, s, e, a_dst, c_src, 0.2, stat, pin, pout, xmax, gx_rank1=att_x)
    assert torch.equal(ties_d[:, :d], ties_r[:, :d])
    _close(ga, ga_r, "g_a")
    _close(gc, gc_r, "g_c")
    _close(gx[:, :d], gx_r[:, :d], "g_x of the cluster sum")
    ops.csr_segment_max_bwd_(gx_r, x, xmax_ref, None, s, ties=ties_r, gmax_rank1=(ga_r, w_comp))
    ops.dense_segment_max_bwd_(gx, x, xmax, s, ties_d, (ga, w_comp), pout)
    _close(gx[:, :d], gx_r[:, :d], "g_x with the maximum's part")


def test_family_b_on_100_qubit_graphs_with_dense_blocks_equals_per_edge():
    """The switch that stays (functional._DENSE_BLOCKS / MLQEM_DENSE_BLOCKS): the whole model (docs/tutorials/gnn.py:70-122) on four
    100-qubit circuits, eval mode and train mode with dropout (the same draws in both forms), level 1 on the dense blocks against the
    per-edge kernels on every row -- predictions within 2e-5 of their scale, every parameter gradient within 2e-4 of the largest
    gradient (the forms sum a row's 200 entries in different orders); and blocks were really used."""
    from blackwater.native import ops

    calls = []
    real = ops.dense_attention_train
    ops.dense_attention_train = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        out_d, _ = _family_b_on_100q(train=False, dense_on=True)
        out_e, _ = _family_b_on_100q(train=False, dense_on=False)
        assert (out_d - out_e).abs().max().item() < 1e-5 * max(1.0, out_e.abs().max().item())
        out_d, g_d = _family_b_on_100q(train=True, dense_on=True)
        out_e, g_e = _family_b_on_100q(train=True, dense_on=False)
    finally:
        ops.dense_attention_train = real
    assert len(calls) == 2                        # the second TransformerConv's forward of both dense runs (parameters want gradients in both)
    assert (out_d - out_e).abs().max().item() < 2e-5 * max(1.0, out_e.abs().max().item())
    gmax = max(v.abs().max().item() for v in g_e.values())
    for k in g_e:
        assert (g_d[k] - g_e[k]).abs().max().item() < 2e-4 * gmax, k


def test_fitness_backward_with_a_wave_per_long_row_equals_the_thread_per_row_kernel():
    from blackwater.native import ops

    s, _, rng = _make(77, 0.3)
    n = s.num_nodes
    gfit = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(DEV)
    fit = torch.from_numpy(rng.rand(n).astype(np.float32)).to(DEV)
    ref = ops.leconv_fitness_bwd(gfit, fit, s.in_ptr, s.out_ptr, s.out_dst)
    got = ops.dense_leconv_fitness_bwd(gfit, fit, s.in_ptr, s.out_ptr, s.out_dst, s.dense_plan("out"))
    assert torch.equal(got[:, 1:], ref[:, 1:])
    _close(got[:, 0], ref[:, 0], "g_p")

