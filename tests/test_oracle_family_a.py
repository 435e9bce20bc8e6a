"""Pins ``oracle.pyg_restatement.{GCNConv, ChebConv, SAGEConv}`` and ``oracle.models.FamilyA`` (SURVEY.md section 8c:
"pin with hand-computed 3-5-node graphs and a dense-matrix cross-check").

The reference holds no checkpoint or printed output for Family A (docs/tutorials/01_ngem.ipynb cell [9] builds it
from ``torch_geometric.nn`` layers), so the oracle's three convolutions are pinned here by two checks that share no
code with the oracle:

* answers worked out BY HAND on 3-node graphs (the arithmetic is in the comments, the expected numbers are literals);
* an independent DENSE formulation in numpy fp64 -- adjacency matrices filled by Python loops over the edge list,
  matrix products only, no scatter / index_add -- on seeded random multigraphs carrying the cases PyG defines
  behaviour for: duplicate edges (counted twice), pre-existing self-loops (GCN replaces them with exactly one,
  Cheb drops them, SAGE counts them), isolated nodes, sources with zero out-degree (inf -> 0 in Cheb's D^-1/2).

PyG definitions followed (torch_geometric >= 2.0, requirements.txt:1):
  GCNConv   gcn_norm: add_remaining_self_loops(fill 1), deg = scatter(w, col=dst), norm = d^-1/2[src] d^-1/2[dst];
            out = sum_j norm_ji (x_j W^T) + b
  ChebConv  get_laplacian(sym) on the edge list WITHOUT self-loops, deg by row=src; scaled by 2/lambda_max with
            lambda_max = 2, diagonal (+1) - 1 = 0: L^[dst,src] = -d^-1/2[src] d^-1/2[dst];
            T_0 = x, T_1 = L^x, T_k = 2 L^ T_{k-1} - T_{k-2}; out = sum_k T_k W_k^T + b
  SAGEConv  out = W_l mean_{j->i} x_j + b_l + W_r x_i, mean over ALL listed in-edges, 0 for a node with none
"""
import math

import numpy as np
import pytest
import torch

from oracle.models import FamilyA
from oracle.pyg_restatement import ChebConv, GCNConv, SAGEConv, global_mean_pool


def _t(a, dtype=torch.float64):
    return torch.tensor(np.asarray(a), dtype=dtype)


def _set(layer_param, value):
    with torch.no_grad():
        layer_param.copy_(_t(value, layer_param.dtype))


# ------------------------------------------------------------------------------------------- hand-computed
def test_gcn_hand_computed_three_nodes():
    # edges 0->1, 0->2, 1->2; x = [1, 2, 3]^T; W = [2]; b = 0.5
    # in-degree incl. the added self-loop: d = [1, 2, 3]  ->  d^-1/2 = [1, 1/sqrt2, 1/sqrt3];  h = xW = [2, 4, 6]
    # out_0 = 1*1*2                                   + .5 = 2.5
    # out_1 = (1/sqrt2)(1*2 + 4/sqrt2)                + .5 = sqrt2 + 2 + .5
    # out_2 = (1/sqrt3)(1*2 + 4/sqrt2 + 6/sqrt3)      + .5 = 2/sqrt3 + 4/sqrt6 + 2 + .5
    conv = GCNConv(1, 1).double()
    _set(conv.lin.weight, [[2.0]])
    _set(conv.bias, [0.5])
    out = conv(_t([[1.0], [2.0], [3.0]]), torch.tensor([[0, 0, 1], [1, 2, 2]]))
    want = [2.5, math.sqrt(2) + 2.5, 2 / math.sqrt(3) + 4 / math.sqrt(6) + 2.5]
    assert want[1] == pytest.approx(3.914213562373095) and want[2] == pytest.approx(5.287693700156686)
    assert out.detach().view(-1).tolist() == pytest.approx(want, abs=1e-12)


def test_cheb_hand_computed_three_nodes():
    # edges 0->1, 1->2, 2->0, 0->2; out-degree by SOURCE: d = [2, 1, 1] -> d^-1/2 = [1/sqrt2, 1, 1]
    # L^[dst,src] = -d^-1/2[src] d^-1/2[dst]:  [1,0] = -1/sqrt2   [2,1] = -1   [0,2] = -1/sqrt2   [2,0] = -1/sqrt2
    # x = [1,2,3]:  T1 = L^x = [-3/sqrt2, -1/sqrt2, -2 - 1/sqrt2]
    #               L^T1 = [(2 + 1/sqrt2)/sqrt2, 3/2, 1/sqrt2 + 3/2] = [sqrt2 + 1/2, 3/2, 1/sqrt2 + 3/2]
    #               T2 = 2 L^T1 - x = [2 sqrt2, 1, sqrt2]
    # W0 = 1, W1 = 2, W2 = -1, b = 0.1:  out = x + 2 T1 - T2 + 0.1 = [1.1 - 5 sqrt2, 1.1 - sqrt2, -0.9 - 2 sqrt2]
    r2 = math.sqrt(2)
    conv = ChebConv(1, 1, K=3).double()
    for lin, w in zip(conv.lins, (1.0, 2.0, -1.0)):
        _set(lin.weight, [[w]])
    _set(conv.bias, [0.1])
    out = conv(_t([[1.0], [2.0], [3.0]]), torch.tensor([[0, 1, 2, 0], [1, 2, 0, 2]]))
    want = [1.1 - 5 * r2, 1.1 - r2, -0.9 - 2 * r2]
    assert want == pytest.approx([-5.971067811865475, -0.314213562373095, -3.728427124746190])
    assert out.detach().view(-1).tolist() == pytest.approx(want, abs=1e-12)


def test_cheb_hand_computed_sink_and_self_loop():
    # edges 0->1, 0->2, 1->2, 1->1 (self-loop: DROPPED); out-degree d = [2, 1, 0] -> d^-1/2 = [1/sqrt2, 1, inf -> 0]
    # L^[1,0] = -1/sqrt2; L^[2,0] = -(1/sqrt2)*0 = 0; L^[2,1] = -1*0 = 0   (a sink's row of L^ vanishes)
    # x = [1,2,3]:  T1 = [0, -1/sqrt2, 0];  K = 2, W0 = 1, W1 = 3, b = 0:  out = [1, 2 - 3/sqrt2, 3]
    conv = ChebConv(1, 1, K=2).double()
    _set(conv.lins[0].weight, [[1.0]])
    _set(conv.lins[1].weight, [[3.0]])
    out = conv(_t([[1.0], [2.0], [3.0]]), torch.tensor([[0, 0, 1, 1], [1, 2, 2, 1]]))
    assert out.detach().view(-1).tolist() == pytest.approx([1.0, 2 - 3 / math.sqrt(2), 3.0], abs=1e-12)


def test_sage_hand_computed_three_nodes():
    # edges 0->1, 0->2, 1->2, 2->2 (self-loop: COUNTED); node 0 has no in-edge -> mean = 0
    # x = [[1,0],[0,2],[3,3]]; W_l = [1,-1], b_l = .5; W_r = [2, .5]
    # mean_in = [[0,0], [1,0], [(1+0+3)/3, (0+2+3)/3]] -> lin_l = [.5, 1.5, 4/3 - 5/3 + .5];  lin_r = [2, 1, 7.5]
    conv = SAGEConv(2, 1).double()
    _set(conv.lin_l.weight, [[1.0, -1.0]])
    _set(conv.lin_l.bias, [0.5])
    _set(conv.lin_r.weight, [[2.0, 0.5]])
    out = conv(_t([[1.0, 0.0], [0.0, 2.0], [3.0, 3.0]]), torch.tensor([[0, 0, 1, 2], [1, 2, 2, 2]]))
    assert out.detach().view(-1).tolist() == pytest.approx([2.5, 2.5, 7.5 + 1 / 6], abs=1e-12)


def test_gcn_hand_computed_existing_and_duplicate_self_loops():
    # edges 0->1, 1->1, 1->1 (two self-loops on node 1: REPLACED by exactly one), node 2 isolated (gets its own loop)
    # d = [1, 2, 1]; x = [1, 1, 4]; W = 1, b = 0:  out = [1, (1/sqrt2)(1 + 1/sqrt2), 4] = [1, 1/sqrt2 + .5, 4]
    conv = GCNConv(1, 1).double()
    _set(conv.lin.weight, [[1.0]])
    out = conv(_t([[1.0], [1.0], [4.0]]), torch.tensor([[0, 1, 1], [1, 1, 1]]))
    assert out.detach().view(-1).tolist() == pytest.approx([1.0, 1 / math.sqrt(2) + 0.5, 4.0], abs=1e-12)


# ---------------------------------------------------------------------------- independent dense formulation
def _random_multigraph(rng, n, e, n_loops, n_dups, isolated):
    """[2,E] edge list on n nodes: e random edges among the non-isolated nodes (the LAST non-isolated node never a
    source -> zero out-degree), plus self-loops and exact duplicates of existing edges."""
    live = [v for v in range(n) if v not in isolated]
    sources = live[:-1]
    src = rng.choice(sources, size=e)
    dst = rng.choice(live, size=e)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    loops = rng.choice(live, size=n_loops)
    dup = rng.integers(0, len(src), size=n_dups)
    src = np.concatenate([src, loops, src[dup]])
    dst = np.concatenate([dst, loops, dst[dup]])
    order = rng.permutation(len(src))
    return np.stack([src[order], dst[order]]).astype(np.int64)


def _adjacency(ei, n, with_loops):
    a = np.zeros((n, n), dtype=np.float64)        # a[dst, src] = multiplicity of the edge src -> dst
    for s, d in zip(ei[0].tolist(), ei[1].tolist()):
        if s == d and not with_loops:
            continue
        a[d, s] += 1.0
    return a


def _inv_sqrt(d):
    out = np.zeros_like(d)
    out[d > 0] = d[d > 0] ** -0.5
    return out


def dense_gcn(x, ei, w, b):
    n = x.shape[0]
    a_hat = _adjacency(ei, n, with_loops=False) + np.eye(n)      # exactly one self-loop per node
    dm = np.diag(_inv_sqrt(a_hat.sum(axis=1)))                   # degree by destination (row of a[dst, src])
    return dm @ a_hat @ dm @ x @ w.T + b


def dense_cheb(x, ei, ws, b):
    n = x.shape[0]
    a = _adjacency(ei, n, with_loops=False)
    dm = np.diag(_inv_sqrt(a.sum(axis=0)))                       # degree by SOURCE (column of a[dst, src])
    lap = -dm @ a @ dm
    terms = [x]
    if len(ws) > 1:
        terms.append(lap @ x)
    for _ in range(2, len(ws)):
        terms.append(2.0 * lap @ terms[-1] - terms[-2])
    return sum(t @ w.T for t, w in zip(terms, ws)) + b


def dense_sage(x, ei, wl, bl, wr):
    n = x.shape[0]
    a = _adjacency(ei, n, with_loops=True)
    deg = np.maximum(a.sum(axis=1), 1.0)
    return (a @ x / deg[:, None]) @ wl.T + bl + x @ wr.T


CASES = [  # (seed, nodes, edges, self-loops, duplicates, isolated nodes)
    (0, 5, 8, 0, 0, ()), (1, 4, 6, 2, 2, ()), (2, 12, 30, 3, 5, (4, 9)), (3, 40, 90, 10, 20, (0, 17, 39)),
    (4, 7, 3, 7, 1, (2,)),
]


@pytest.mark.parametrize("seed,n,e,n_loops,n_dups,isolated", CASES)
def test_convs_equal_independent_dense_algebra(seed, n, e, n_loops, n_dups, isolated):
    rng = np.random.default_rng(seed)
    ei = _random_multigraph(rng, n, e, n_loops, n_dups, isolated)
    x = rng.normal(size=(n, 6))
    ei_t, x_t = torch.from_numpy(ei), _t(x)
    # the cases the docstring promises are really present
    if n_loops:
        assert (ei[0] == ei[1]).any()
    if n_dups:
        assert len({(s, d) for s, d in ei.T.tolist()}) < ei.shape[1]
    zero_out = [v for v in range(n) if v not in isolated and not (ei[0][ei[0] != ei[1]] == v).any()]
    assert zero_out, "a non-isolated node with zero out-degree must exist"

    torch.manual_seed(seed)
    gcn = GCNConv(6, 4).double()
    _set(gcn.bias, rng.normal(size=4))
    want = dense_gcn(x, ei, gcn.lin.weight.detach().numpy(), gcn.bias.detach().numpy())
    assert np.abs(gcn(x_t, ei_t).detach().numpy() - want).max() < 1e-12

    for k in (1, 2, 3, 4):
        cheb = ChebConv(6, 3, K=k).double()
        _set(cheb.bias, rng.normal(size=3))
        want = dense_cheb(x, ei, [l.weight.detach().numpy() for l in cheb.lins], cheb.bias.detach().numpy())
        assert np.abs(cheb(x_t, ei_t).detach().numpy() - want).max() < 1e-11, k

    sage = SAGEConv(6, 5).double()
    want = dense_sage(x, ei, sage.lin_l.weight.detach().numpy(), sage.lin_l.bias.detach().numpy(),
                      sage.lin_r.weight.detach().numpy())
    assert np.abs(sage(x_t, ei_t).detach().numpy() - want).max() < 1e-12


def test_train_time_add_self_loops_semantics():
    """The training path appends one (i, i) per node (AddSelfLoops, loaders/exp_val.py:33) before the convs see the
    graph: GCN must give the SAME result as without them (it replaces them), Cheb the same (it drops them), and SAGE
    must count them as one more in-neighbour."""
    rng = np.random.default_rng(7)
    n = 9
    ei = _random_multigraph(rng, n, 20, 0, 3, ())
    loops = np.arange(n)
    ei_l = np.concatenate([ei, np.stack([loops, loops])], axis=1)
    x = _t(rng.normal(size=(n, 4)))
    torch.manual_seed(0)
    gcn, cheb, sage = GCNConv(4, 3).double(), ChebConv(4, 3, K=3).double(), SAGEConv(4, 3).double()
    a, b = torch.from_numpy(ei), torch.from_numpy(ei_l)
    assert torch.equal(gcn(x, a), gcn(x, b))
    assert torch.equal(cheb(x, a), cheb(x, b))
    want = dense_sage(x.numpy(), ei_l, sage.lin_l.weight.detach().numpy(), sage.lin_l.bias.detach().numpy(),
                      sage.lin_r.weight.detach().numpy())
    assert np.abs(sage(x, b).detach().numpy() - want).max() < 1e-12
    assert (sage(x, a) - sage(x, b)).abs().max() > 1e-3


def test_family_a_model_equals_dense_wiring():
    """oracle.models.FamilyA in eval mode against the notebook's wiring (01_ngem.ipynb cell [9]) written with the dense
    layer forms above: relu between convs, per-graph mean pools, mean over observable terms, concat order
    [gcn, cheb, sage, obs, depth, noisy]."""
    rng = np.random.default_rng(11)
    sizes, nq = [5, 9, 3], 2
    eis, xs, off = [], [], 0
    for k, n in enumerate(sizes):
        ei = _random_multigraph(rng, n, 2 * n, 1, 1, ())
        loops = np.arange(n)
        eis.append(np.concatenate([ei, np.stack([loops, loops])], axis=1) + off)
        xs.append(rng.normal(size=(n, 22)))
        off += n
    ei, x = np.concatenate(eis, axis=1), np.concatenate(xs)
    batch = np.repeat(np.arange(len(sizes)), sizes)
    noisy, depth = rng.normal(size=(3, 1)), rng.uniform(1, 9, size=(3, 1))
    obs = rng.normal(size=(3, 2, 4 * nq + 1))
    torch.manual_seed(3)
    model = FamilyA(nq, 22, 10).double().eval()
    with torch.no_grad():
        for p in model.parameters():       # biases start at zero in PyG: make every term visible
            if p.dim() == 1:
                p.copy_(_t(rng.normal(size=p.shape)))
    sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    relu = lambda a: np.maximum(a, 0.0)
    pool = lambda h: np.stack([h[batch == g].mean(axis=0) for g in range(len(sizes))])
    g = relu(dense_gcn(x, ei, sd["conv1.lin.weight"], sd["conv1.bias"]))
    g = relu(dense_gcn(g, ei, sd["conv2.lin.weight"], sd["conv2.bias"]))
    g = pool(dense_gcn(g, ei, sd["conv3.lin.weight"], sd["conv3.bias"]))
    c = relu(dense_cheb(x, ei, [sd[f"cheb_conv1.lins.{k}.weight"] for k in range(3)], sd["cheb_conv1.bias"]))
    c = pool(dense_cheb(c, ei, [sd[f"cheb_conv2.lins.{k}.weight"] for k in range(2)], sd["cheb_conv2.bias"]))
    s = relu(dense_sage(x, ei, sd["sage_conv1.lin_l.weight"], sd["sage_conv1.lin_l.bias"], sd["sage_conv1.lin_r.weight"]))
    s = pool(dense_sage(s, ei, sd["sage_conv2.lin_l.weight"], sd["sage_conv2.lin_l.bias"], sd["sage_conv2.lin_r.weight"]))
    o = (obs @ sd["obs_seq.0.weight"].T + sd["obs_seq.0.bias"]) @ sd["obs_seq.2.weight"].T + sd["obs_seq.2.bias"]
    merged = np.concatenate([g, c, s, o.mean(axis=1), depth, noisy], axis=1)
    want = (merged @ sd["body_seq.0.weight"].T + sd["body_seq.0.bias"]) @ sd["body_seq.1.weight"].T + sd["body_seq.1.bias"]
    got = model(_t(noisy), _t(obs), _t(depth), _t(x), torch.from_numpy(ei), torch.from_numpy(batch))
    assert got.shape == (3, 1)
    assert np.abs(got.detach().numpy() - want).max() < 1e-11


def test_global_mean_pool_against_loops():
    x = _t(np.arange(24.0).reshape(8, 3))
    batch = torch.tensor([0, 0, 0, 1, 2, 2, 2, 2])
    want = np.stack([x.numpy()[batch.numpy() == g].mean(axis=0) for g in range(3)])
    assert np.abs(global_mean_pool(x, batch, 3).numpy() - want).max() < 1e-14
    assert torch.equal(global_mean_pool(x, None), x.mean(dim=0, keepdim=True))
