"""Row-degree statistics of the coarsened (level 1) graph of one batch, per circuit family of bench.py's mixed corpus (cfg5) -- what the
list coarsening (coarsen_gather / coarsen_unique) sees: python scripts/coarse_degree_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
import bench
from blackwater.data.synthetic import encode_corpus, pauli_twirl, random_circuit, tfim_circuit
from blackwater.nn import ExpValCircuitGraphModel, ExpValCircuitGraphModel_3
from blackwater.nn.models import as_structure

dev = "cuda:0"
fams = {"tfim4": [tfim_circuit(4, st, J=0.3 + 0.01 * st, two_q="cx") for st in range(15)],
        "random20": [random_circuit(20, 40, seed=s, two_q="cx") for s in range(12)],
        "twirled100": [pauli_twirl(tfim_circuit(100, st, J=0.5, two_q="cx"), seed=100 + st, two_q=("cx",)) for st in range(1, 11)],
        "tfim100": [tfim_circuit(100, st, J=0.5, two_q="cx") for st in range(1, 11)]}
torch.manual_seed(0)
model = (ExpValCircuitGraphModel if os.environ.get("MODEL", "3") == "1" else ExpValCircuitGraphModel_3)(22, 15, 4).to(dev)
model.train(os.environ.get("TRAIN", "0") == "1")
only = os.environ.get("FAM")
fams["mixed"] = fams["tfim4"][:8] + fams["random20"][:5] + fams["twirled100"][:3]
from blackwater.data.synthetic import TfimCorpus
fams["corpus100 (cfg4)"] = None
for name, circs in fams.items():
    if only and only not in name:
        continue
    if circs is None:
        arena = TfimCorpus(100, list(range(1, 11)), 2, seed=42, exp_value_size=4).arena(dev)
    else:
        enc = encode_corpus(circs, 100, two_q="cx", exp_value_size=4)
        arena, _ = bench.replicated_arena(enc, np.full(len(circs), 2), dev, scalar_labels=False)
    b = arena.batch(np.arange(len(arena)))
    ev, obs, depth, nodes, ei, bt = b.model_args()
    s = as_structure(ei, nodes.shape[0], bt, ev.shape[0])
    with torch.no_grad():
        g = model.transformer1(nodes, s)
        times = []
        for _ in range(3):                       # the coarsening is deferred until a layer reads the structure: time that read
            g1, s1, perm = model.pooling1(g, s)
            torch.cuda.synchronize()
            if os.environ.get("BUSY"):           # keep the GPU busy right up to the timed call (clock state)
                a_ = torch.randn(8192, 8192, device=dev)
                for _k in range(int(os.environ["BUSY"])):
                    a_ = (a_ @ a_).clamp_(-1, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _ = s1.in_ptr; e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        deg = (s1.in_ptr[1:] - s1.in_ptr[:-1]).cpu().numpy().astype(np.int64)
        from blackwater.native import _lib, ops
        lib = _lib.load()
        k = int(perm.numel())
        need = lib.mlqem_asap_coarsen_lists_workspace_bytes(s.num_nodes, k, 0, 0)
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
        totals = torch.empty(4, dtype=torch.int64, device=dev)
        lib.mlqem_asap_coarsen_lists_caps(ops._p(s.in_ptr), ops._p(s.in_src), ops._p(s.out_ptr), ops._p(s.out_dst), ops._p(s1.graph_ptr), ops._p(perm.to(torch.int32)),
                                          s.num_nodes, k, s.num_graphs, ops._p(totals), ops._p(ws), need, ops._stream())
        tot = totals.tolist()
        # what the list passes walk: |C(v)| = [v kept] + kept out-neighbours of v; |R(u)| = sum of |C(v)| over v in N+[u] and u itself;
        # candidates of cluster c = sum of |R(u)| over u in N-[c] and c itself
        n0 = s.num_nodes
        kept = torch.zeros(n0, dtype=torch.int64, device=dev); kept[perm.long()] = 1
        optr, iptr = s.out_ptr[:n0 + 1].long(), s.in_ptr[:n0 + 1].long()
        rows_o = torch.repeat_interleave(torch.arange(n0, device=dev), optr[1:] - optr[:-1])
        rows_i = torch.repeat_interleave(torch.arange(n0, device=dev), iptr[1:] - iptr[:-1])
        odst, isrc = s.out_dst[:rows_o.numel()].long(), s.in_src[:rows_i.numel()].long()
        csz = kept.clone().index_add_(0, rows_o, kept[odst])
        rsz = csz.clone().index_add_(0, rows_o, csz[odst])
        cand_all = rsz.clone().index_add_(0, rows_i, rsz[isrc])
        cand = cand_all[perm.long()].float()
        members = (iptr[1:] - iptr[:-1])[perm.long()] + 1
        q = lambda t, f: float(torch.quantile(t.float()[:: max(1, t.numel() // 1000000)], f))
        print(f"    |C| max {int(csz.max())}  |R| mean {float(rsz.float().mean()):.1f} p99 {q(rsz, 0.99):.0f} max {int(rsz.max())}   members max {int(members.max())}   candidates per cluster: "
              f"sum {int(cand.sum())} mean {float(cand.mean()):.0f} p50 {q(cand, 0.5):.0f} p99 {q(cand, 0.99):.0f} max {int(cand.max())}  clusters > 64: {int((cand > 64).sum())} > 4096: {int((cand > 4096).sum())} > 65536: {int((cand > 65536).sum())}", flush=True)
        cap_ = getattr(s, 'coarse_capacity', None)
    d0 = (s.in_ptr[1:] - s.in_ptr[:-1]).cpu().numpy()
    print(f"{name:16s} circuits {len(arena):3d}  level 0: nodes {len(d0):7d} max in-degree {d0.max():4d}   level 1: rows {len(deg):7d} entries {deg.sum():9d} "
          f"mean {deg.mean():7.1f} max {deg.max():6d}  rows > 64: {(deg > 64).sum():6d}  > 1024: {(deg > 1024).sum():6d}  entries in rows > 1024: {deg[deg > 1024].sum():9d}  coarsening {min(times):8.0f} us  list totals {tot}  capacity {cap_}  edges {int(s.in_src.numel())} train={model.training}", flush=True)
