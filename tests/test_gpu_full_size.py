"""Size-independent properties at the benchmark's full size (1024 circuits of the 100-qubit TFIM corpus, 11 M nodes),
where the CPU oracle is too slow to be the checker: linearity of the aggregation, the adjoint identity that ties the
forward (in-CSR) and backward (out-CSR) kernels together for every normalisation, exact row sums, pooling of constants,
and a checksum of the assembled batch against the arena it was gathered from."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def full_batch():
    sys.path.insert(0, ROOT)
    import bench

    corpus = bench.build_corpus(103)          # 1 030 circuits, replicated from the ten templates on the device
    arena = corpus.arena(DEV)
    ids = bench.fixed_ids(len(corpus))
    return arena, ids, arena.batch(ids)


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


def test_aggregation_is_linear_and_adjoint_consistent(full_batch):
    from blackwater.native import ops

    _, _, b = full_batch
    s, n = b.structure, b.structure.num_nodes
    assert n > 10_000_000
    g = torch.Generator(device=DEV).manual_seed(0)
    x = ops.padded_empty(n, 10, DEV).normal_(generator=g)
    y = ops.padded_empty(n, 10, DEV).normal_(generator=g)
    norms = {"gcn": dict(fwd=dict(rscale=s.gcn_dinv, dself=s.gcn_dinv, cscale=s.gcn_dinv),
                         bwd=dict(rscale=s.gcn_dinv, dself=s.derived("gcn_dself"), cscale=s.gcn_dinv)),
             "mean": dict(fwd=dict(rscale=s.sage_rinv, dself=s.derived("sage_dself")),
                          bwd=dict(cscale=s.sage_rinv, dself=s.derived("sage_dself"))),
             "lap": dict(fwd=dict(cscale=s.cheb_dinv, rscale=s.derived("cheb_neg")),
                         bwd=dict(cscale=s.derived("cheb_neg"), rscale=s.cheb_dinv))}
    for name, kw in norms.items():
        fwd = lambda t: ops.csr_aggregate(t, s.in_ptr, s.in_src, ell=s.in_ell, **kw["fwd"])
        bwd = lambda t: ops.csr_aggregate(t, s.out_ptr, s.out_dst, ell=s.out_ell, **kw["bwd"])
        if name == "gcn":   # the layer's forward form takes pre-scaled rows: A_hat x = dinv * (sum dinv_j x_j + dinv_i x_i)
            fwd = lambda t: ops.csr_aggregate(t * s.gcn_dinv[:, None], s.in_ptr, s.in_src, ell=s.in_ell,
                                              rscale=s.gcn_dinv, dself=s.gcn_dinv)
        ax, ay = fwd(x), fwd(y)
        lin = fwd(2.0 * x - 3.0 * y)
        assert torch.allclose(lin, 2.0 * ax - 3.0 * ay, rtol=1e-4, atol=1e-4), name
        lhs, rhs = _dot(ax, y), _dot(x, bwd(y))            # <A x, y> = <x, A^T y>
        assert abs(lhs - rhs) < 1e-6 * (abs(lhs) + np.sqrt(_dot(ax, ax) * _dot(y, y))), (name, lhs, rhs)


def test_row_sums_and_pooling_of_constants(full_batch):
    from blackwater.native import functional as F, ops

    _, ids, b = full_batch
    bench_batch = len(ids)
    s, n = b.structure, b.structure.num_nodes
    ones = ops.padded_empty(n, 4, DEV).fill_(1.0)
    mean = ops.csr_aggregate(ones, s.in_ptr, s.in_src, ell=s.in_ell, rscale=s.sage_rinv, dself=s.derived("sage_dself"))
    indeg = (s.in_ptr[1:n + 1] - s.in_ptr[:n]) + s.loops[:n]
    want = (indeg > 0).float()                              # the mean of ones over a non-empty neighbourhood is one
    assert torch.allclose(mean[:, 0], want, rtol=0, atol=1e-6) and torch.equal(mean[:, 0], mean[:, 3])
    deg = ops.csr_aggregate(ones, s.in_ptr, s.in_src, ell=s.in_ell)      # plain sum: the in-degree, exactly
    assert torch.equal(deg[:, 0], (s.in_ptr[1:n + 1] - s.in_ptr[:n]).float())
    assert int(deg[:, 0].sum().item()) == s.num_edges
    pooled = F.segment_mean(ones, s)
    assert pooled.shape == (bench_batch, 4) and torch.allclose(pooled, torch.ones_like(pooled), rtol=0, atol=1e-6)
    mx = ops.csr_segment_max(ones * 3.0, s.in_ptr, s.in_src, ell=s.in_ell)
    assert torch.equal(mx, ones * 3.0)                      # idempotence of max over equal entries


def test_assembled_batch_checksums_match_the_arena(full_batch):
    arena, ids, b = full_batch
    s = b.structure
    starts = arena.gptr.cpu().numpy()[ids]
    counts = arena.node_counts[ids]
    want_n, want_e = int(counts.sum()), int(arena.edge_counts[ids].sum())
    assert s.num_nodes == want_n and s.num_edges == want_e
    # checksum of checksums: per-graph feature sums of the batch equal those of the arena rows they came from
    x64 = arena.x.double().sum(1).cpu().numpy()
    want = np.array([x64[a:a + c].sum() for a, c in zip(starts, counts)])
    rows = b.x.double().sum(1).cpu().numpy()
    ptr = np.concatenate([[0], np.cumsum(counts)])
    got = np.array([rows[ptr[k]:ptr[k + 1]].sum() for k in range(len(ids))])
    assert np.allclose(got, want, rtol=1e-12, atol=1e-9)
    # every edge stays inside its graph and the two CSR views hold the same multiset of edges
    in_src, out_dst = s.in_src[:want_e].long(), s.out_dst[:want_e].long()
    dst_of_in = torch.repeat_interleave(torch.arange(want_n, device=DEV), (s.in_ptr[1:want_n + 1] - s.in_ptr[:want_n]).long())
    src_of_out = torch.repeat_interleave(torch.arange(want_n, device=DEV), (s.out_ptr[1:want_n + 1] - s.out_ptr[:want_n]).long())
    gid = torch.bucketize(torch.arange(want_n, device=DEV), s.graph_ptr[1:].long(), right=True)
    assert torch.equal(gid[in_src], gid[dst_of_in]) and torch.equal(gid[out_dst], gid[src_of_out])
    key_in = torch.sort(in_src * want_n + dst_of_in).values
    key_out = torch.sort(src_of_out * want_n + out_dst).values
    assert torch.equal(key_in, key_out)
