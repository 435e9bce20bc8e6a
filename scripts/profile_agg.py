"""Runs only the CSR aggregation kernel on the benchmark's representative batch (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from bench import build_corpus, fixed_ids as bench_fixed_ids
from blackwater.data.arena import GraphArena
from blackwater.native import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
c = int(sys.argv[2]) if len(sys.argv) > 2 else 10
corpus = build_corpus(50)
arena = corpus.arena("cuda:0")
n_graphs = len(corpus)
b = arena.batch(bench_fixed_ids(n_graphs))
s = b.structure
n = s.num_nodes
PAD = os.environ.get("MLQEM_PAD", "1") == "1"
mk = (lambda: ops.padded_empty(n, c, torch.device("cuda:0"))) if PAD else (lambda: torch.empty(n, c, device="cuda:0"))
hs = [mk().normal_() for _ in range(4)]
outs = [mk() for _ in range(4)]
ELL = s.in_ell if os.environ.get('MLQEM_USE_ELL', '1') == '1' else None
torch.cuda.synchronize()
beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for k in range(4):
    ops.csr_aggregate(hs[k % 4], s.in_ptr, s.in_src, ell=ELL, rscale=s.gcn_dinv, dself=s.gcn_dinv, out=outs[k % 4])
beg.record()
for k in range(reps):
    ops.csr_aggregate(hs[k % 4], s.in_ptr, s.in_src, ell=ELL, rscale=s.gcn_dinv, dself=s.gcn_dinv, out=outs[k % 4])
end.record()
torch.cuda.synchronize()
us = beg.elapsed_time(end) * 1e3 / reps
e = s.num_edges + n
alg = 4 * (n + 1) + 4 * e + 4 * n + 4 * c * (e + n)
print("nodes", n, "edges", s.num_edges, "C", c, "IPT", os.environ.get("MLQEM_AGG_IPT"), "ell", ELL is not None, "padded", PAD,  "us/launch %.1f" % us,
      "alg GB/s %.0f" % (alg / us / 1e3))
