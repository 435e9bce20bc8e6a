"""One-process sweep of the aggregation kernel over channel widths / layouts (interleaved rounds, rule 24)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from bench import build_corpus
from blackwater.data.arena import GraphArena
from blackwater.native import ops
corpus = build_corpus(50)
arena = corpus.arena("cuda:0")
n_graphs = len(corpus)
s = arena.batch(np.arange(256) * n_graphs // 256).structure
n = s.num_nodes
dev = torch.device("cuda:0")
cfgs = [(8, False), (10, False), (10, True), (12, False), (16, False), (20, False), (22, False), (22, True), (24, False)]
bufs = {}
for c, pad in cfgs:
    mk = (lambda: ops.padded_empty(n, c, dev)) if pad else (lambda: torch.empty(n, c, device=dev))
    bufs[(c, pad)] = ([mk().normal_() for _ in range(3)], [mk() for _ in range(3)])
ell = s.in_ell
res = {k: [] for k in cfgs}
for rnd in range(5):
    for key in cfgs:
        hs, outs = bufs[key]
        for k in range(3):
            ops.csr_aggregate(hs[k], s.in_ptr, s.in_src, ell=ell, rscale=s.gcn_dinv, dself=s.gcn_dinv, out=outs[k])
        beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        beg.record()
        for k in range(12):
            ops.csr_aggregate(hs[k % 3], s.in_ptr, s.in_src, ell=ell, rscale=s.gcn_dinv, dself=s.gcn_dinv, out=outs[k % 3])
        end.record(); torch.cuda.synchronize()
        res[key].append(beg.elapsed_time(end) * 1e3 / 12)
e = s.num_edges + n
for (c, pad), v in res.items():
    alg = 4 * (n + 1) + 4 * e + 4 * n + 4 * c * (e + n)
    print(f"C={c:3d} padded={pad!s:5s} median {np.median(v):7.1f} us  min {min(v):7.1f}  alg {alg / np.median(v) / 1e3:6.0f} GB/s")
