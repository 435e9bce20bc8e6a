"""Edge cases of the graph inputs, GPU vs oracle: single-node graphs, graphs without edges, duplicate edges (two
wires between the same two ops), isolated nodes, a batch of one, self-loops already present in the edge list."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graphs():
    g = torch.Generator().manual_seed(5)
    feats = lambda n: torch.randn(n, 22, generator=g) * 0.5
    out = []
    out.append((feats(1), torch.zeros((2, 0), dtype=torch.long)))                              # one node, no edge
    out.append((feats(4), torch.zeros((2, 0), dtype=torch.long)))                              # no edges at all
    out.append((feats(3), torch.tensor([[0, 0, 1], [1, 1, 2]])))                               # duplicate edge 0->1
    out.append((feats(6), torch.tensor([[0, 1, 2, 2, 4], [1, 2, 3, 3, 4]])))                   # node 5 isolated, 4->4 loop
    out.append((feats(9), torch.stack([torch.arange(0, 8), torch.arange(1, 9)])))              # a chain
    out.append((feats(40), torch.stack([torch.randint(0, 40, (150,), generator=g),
                                        torch.randint(0, 40, (150,), generator=g)])))          # dense-ish, loops, repeats
    return out


def _collate(graphs, add_loops):
    xs, eis, bs, off = [], [], [], 0
    for b, (x, ei) in enumerate(graphs):
        n = x.shape[0]
        if add_loops:
            ei = torch.cat([ei, torch.arange(n).repeat(2, 1)], dim=1)
        xs.append(x)
        eis.append(ei + off)
        bs.append(torch.full((n,), b, dtype=torch.long))
        off += n
    return torch.cat(xs), torch.cat(eis, 1), torch.cat(bs)


@pytest.mark.parametrize("add_loops", [False, True])
@pytest.mark.parametrize("which", ["all", "single"])
def test_family_a_edge_graphs(add_loops, which):
    from blackwater.nn import ExpValCircuitGraphModelA
    from oracle.models import FamilyA

    graphs = _graphs() if which == "all" else _graphs()[:1]
    x, ei, batch = _collate(graphs, add_loops)
    b = len(graphs)
    torch.manual_seed(21)
    model = ExpValCircuitGraphModelA(2, 22, 10)
    ref = FamilyA(2, 22, 10).double().eval()
    ref.load_state_dict(model.state_dict())
    model = model.to(DEV).eval()
    g = torch.Generator().manual_seed(1)
    noisy, depth, obs = torch.randn(b, 1, generator=g), torch.rand(b, 1, generator=g) * 10, torch.randn(b, 2, 9, generator=g)
    out = model(noisy.to(DEV), obs.to(DEV), depth.to(DEV), x.to(DEV), ei.to(DEV), batch.to(DEV))
    want = ref(noisy.double(), obs.double(), depth.double(), x.double(), ei, batch)
    assert (out.detach().cpu().double() - want.detach()).abs().max().item() < 1e-5
    out.sum().backward()
    want.sum().backward()
    grads = {k: p.grad for k, p in ref.named_parameters()}
    overall = max(v.abs().max().item() for v in grads.values())
    for name, p in model.named_parameters():
        scale = max(grads[name].abs().max().item(), 1e-2 * overall)
        assert (p.grad.cpu().double() - grads[name]).abs().max().item() / scale < 2e-4, name


@pytest.mark.parametrize("add_loops", [False, True])
def test_family_b_edge_graphs(golden_dir, add_loops):
    import os

    from blackwater.nn import family_b_from_state_dict
    from oracle.models import family_b_from_state_dict as oracle_from_sd

    sd = torch.load(os.path.join(golden_dir, "ckpt", "gnn1.pth"), weights_only=True)
    model = family_b_from_state_dict(sd).to(DEV).eval()
    ref = oracle_from_sd(sd).double().eval()
    graphs = _graphs()
    x, ei, batch = _collate(graphs, add_loops)
    b = len(graphs)
    g = torch.Generator().manual_seed(2)
    noisy, depth = torch.randn(b, 1, 4, generator=g) * 0.3, torch.rand(b, 1, generator=g) * 10
    out = model(noisy.to(DEV), None, depth.to(DEV), x.to(DEV), ei.to(DEV), batch.to(DEV))
    want = ref(noisy.double(), None, depth.double(), x.double(), ei, batch)
    assert out.shape == (b, 4)
    assert (out.detach().cpu().double() - want.detach()).abs().max().item() < 2e-5
    out.square().sum().backward()
    want.square().sum().backward()
    grads = {k: p.grad for k, p in ref.named_parameters()}
    overall = max(v.abs().max().item() for v in grads.values())
    for name, p in model.named_parameters():
        scale = max(grads[name].abs().max().item(), 1e-2 * overall)
        assert (p.grad.cpu().double() - grads[name]).abs().max().item() / scale < 1e-3, name
    # unbatched single tiny graph through the inference convention (batch=None)
    x1, e1 = graphs[2]
    with torch.no_grad():
        o1 = model(noisy[:1].to(DEV), None, depth[:1].to(DEV), x1.to(DEV), e1.to(DEV), None)
        w1 = ref(noisy[:1].double(), None, depth[:1].double(), x1.double(), e1, None)
    assert (o1.cpu().double() - w1).abs().max().item() < 2e-5
