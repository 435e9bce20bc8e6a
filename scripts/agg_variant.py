"""Times (and checks against a torch index_add reference) the aggregation kernel variant selected by the MLQEM_AGG_*
environment knobs, on the benchmark batch.  One process per variant: the knobs are read once."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from bench import build_corpus, fixed_ids as bench_fixed_ids
from blackwater.data.arena import GraphArena
from blackwater.native import ops
corpus = build_corpus(50)
arena = corpus.arena("cuda:0")
n_graphs = len(corpus)
s = arena.batch(bench_fixed_ids(n_graphs)).structure
n, dev = s.num_nodes, torch.device("cuda:0")
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MLQEM_AGG"))
dst = torch.repeat_interleave(torch.arange(n, device=dev), (s.in_ptr[1:n + 1] - s.in_ptr[:n]).long())
src = s.in_src[: s.num_edges].long()
for c in (10, 22, 1):
    hs = [ops.padded_empty(n, c, dev).normal_() for _ in range(4)]
    outs = [ops.padded_empty(n, c, dev) for _ in range(4)]
    run = lambda k: ops.csr_aggregate(hs[k % 4], s.in_ptr, s.in_src, ell=s.in_ell, rscale=s.gcn_dinv, dself=s.gcn_dinv, out=outs[k % 4])
    run(0)
    ref = torch.zeros(n, c, device=dev).index_add_(0, dst, hs[0][src]) * s.gcn_dinv[:, None] + hs[0] * s.gcn_dinv[:, None]
    err = (outs[0] - ref).abs().max().item()
    ts = []
    for rnd in range(5):
        for k in range(4):
            run(k)
        beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        beg.record()
        for k in range(20):
            run(k)
        end.record(); end.synchronize()
        ts.append(beg.elapsed_time(end) * 1e3 / 20)
    e = s.num_edges + n
    alg = 4 * (n + 1) + 4 * e + 4 * n + 4 * c * (e + n)
    print(f"[{tag or 'default'}] C={c:2d}: median {np.median(ts):6.1f} us  min {min(ts):6.1f}  {alg / np.median(ts) / 1e3:5.0f} GB/s alg  max|err| {err:.2e}")
