"""how large is the union of source rows of a tile of level-1 destination rows?

CPU only (the oracle's ASAPooling on one synthetic 100-qubit TFIM circuit per step count): the level-1 graph is what
ASAPooling's coarsening S^T A S makes of a circuit DAG.  Rows are ordered (a) as PyG leaves them (fitness order) and
(b) by the program position of their centres; tiles of T consecutive rows; per tile the number of distinct source rows
of its in-entries (forward walks) and of distinct destination rows of its out-entries (source-side backward walks).

    python scripts/tile_union_probe.py [steps ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-qem_amd"))


def level1(steps):
    from blackwater.data.synthetic import tfim_corpus
    from oracle.models import FamilyB

    corpus = tfim_corpus(100, [steps], 1, seed=42, two_q="ecr", exp_value_size=1)
    torch.manual_seed(0)
    ref = FamilyB(22, 15, exp_value_size=1).eval()
    x = torch.from_numpy(corpus["x"][0]).float()
    ei = torch.from_numpy(corpus["edge_index"][0])
    n = x.shape[0]
    loops = torch.arange(n)
    ei = torch.cat([ei, torch.stack([loops, loops])], 1)       # the loader's AddSelfLoops (train-time graphs)
    with torch.no_grad():
        g = ref.transformer1(x, ei)
        g, ei1, _, _, perm = ref.pooling1(g, ei, batch=None)
    return n, perm.numpy(), ei1.numpy()


def unions(order, ptr, col, tile):
    out = []
    for a in range(0, len(order), tile):
        rows = order[a:a + tile]
        ent = np.concatenate([col[ptr[r]:ptr[r + 1]] for r in rows]) if len(rows) else np.zeros(0, np.int64)
        out.append((len(np.unique(ent)), len(ent)))
    return np.array(out)


def csr(rows, cols, k):
    o = np.lexsort((cols, rows))
    rows, cols = rows[o], cols[o]
    ptr = np.zeros(k + 1, np.int64)
    np.add.at(ptr, rows + 1, 1)
    return np.cumsum(ptr), cols


def main():
    for steps in [int(a) for a in sys.argv[1:]] or [5]:
        n, perm, ei1 = level1(steps)
        k = len(perm)
        src, dst = ei1[0], ei1[1]
        in_ptr, in_src = csr(dst, src, k)
        out_ptr, out_dst = csr(src, dst, k)
        deg = np.diff(in_ptr)
        print(f"steps {steps}: level-0 nodes {n}, level-1 nodes {k}, level-1 edges {len(src)}, in-degree mean {deg.mean():.1f} "
              f"p50 {np.median(deg):.0f} p95 {np.percentile(deg, 95):.0f} max {deg.max()}")
        orders = {"fitness order (PyG's)": np.arange(k), "program position of the centre": np.argsort(perm, kind="stable")}
        for name, order in orders.items():
            for tile in (16, 32, 64, 128):
                for side, (p, c) in (("in ", (in_ptr, in_src)), ("out", (out_ptr, out_dst))):
                    u = unions(order, p, c, tile)
                    un, en = u[:, 0], u[:, 1]
                    print(f"  {name:32s} T={tile:4d} {side}: union mean {un.mean():7.1f} p50 {np.median(un):6.0f} p95 {np.percentile(un, 95):6.0f} "
                          f"max {un.max():5d} | entries/tile mean {en.mean():8.0f} | reuse (entries/union) {en.sum() / max(un.sum(), 1):6.1f}")
        # the span of a tile's union in program-position rank: is the union a contiguous window?
        rank = np.empty(k, np.int64)
        rank[orders["program position of the centre"]] = np.arange(k)
        order = orders["program position of the centre"]
        for tile in (32, 64, 128):
            spans, uns = [], []
            for a in range(0, k, tile):
                rows = order[a:a + tile]
                ent = np.concatenate([in_src[in_ptr[r]:in_ptr[r + 1]] for r in rows])
                if len(ent):
                    r = rank[ent]
                    spans.append(r.max() - r.min() + 1)
                    uns.append(len(np.unique(ent)))
            spans, uns = np.array(spans), np.array(uns)
            print(f"  window of a tile's sources in position rank, T={tile}: span mean {spans.mean():.0f} p95 {np.percentile(spans, 95):.0f} max {spans.max()} "
                  f"(union mean {uns.mean():.0f}: fill {uns.sum() / spans.sum():.2f})")


if __name__ == "__main__":
    main()
