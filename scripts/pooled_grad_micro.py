"""The pooled gradient written and gathered (segment_pool_bwd + csr_aggregate) against computed inside the aggregation
(ops.PooledGrad.aggregate, csrc/pooled_grad.hip) on the benchmark's batch: python scripts/pooled_grad_micro.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import torch
from bench import build_corpus, fixed_ids as bench_fixed_ids
from blackwater.native import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
c = 10
corpus = build_corpus(50)
arena = corpus.arena("cuda:0")
b = arena.batch(bench_fixed_ids(len(corpus)))
s = b.structure
n, nb, gptr = s.num_nodes, s.num_graphs, s.graph_ptr
dev = torch.device("cuda:0")
x = ops.padded_empty(n, c, dev).normal_()
kinds = {"gcn": dict(rscale=s.gcn_dinv, dself=s.derived("gcn_dself"), cscale=s.gcn_dinv, mean=False),
         "cheb": dict(rscale=s.cheb_dinv, dself=None, cscale=s.derived("cheb_neg"), mean=True),
         "sage": dict(rscale=None, dself=s.derived("sage_dself"), cscale=s.sage_rinv, mean=True)}


def timed(fn):
    for _ in range(3):
        fn()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for _ in range(reps):
        fn()
    end.record()
    torch.cuda.synchronize()
    return beg.elapsed_time(end) * 1e3 / reps


for kind, kw in kinds.items():
    wts = s.colsum(kind)
    req = dict(graph_ptr=gptr, num_graphs=nb, weights=wts, mean=kw["mean"], wmean=True, bits=True, store=False)
    ops.csr_aggregate(x, s.in_ptr, s.in_src, ell=s.in_ell, rscale=s.gcn_dinv, dself=s.gcn_dinv, relu=True, pool=req)
    bits = req["out_bits"]
    gm = ops.padded_empty(nb, c, dev).normal_() if kw["mean"] else None
    gw = ops.padded_empty(nb, c, dev).normal_()
    pg = ops.PooledGrad(gm, gw, gptr, n, wts, 1.25, bits)
    form = os.environ.get("PG_FORM", "")      # one form only (counter passes: scripts/pmc_micro.sh averages a kernel's launches)
    agg = lambda **k: pg.aggregate(s.out_ptr, s.out_dst, s.out_ell, kw["cscale"], rscale=kw["rscale"], dself=kw["dself"], **k)
    if form == "computed":
        timed(lambda: agg()); continue
    if form == "computed_not_written":
        timed(lambda: agg(want_g=False)); timed(lambda: pg.colsum()); continue
    t_w = timed(lambda: ops.segment_pool_bwd(gm, gw, gptr, n, weights=wts, gate_scale=1.25, gate_bits=bits))
    g = ops.segment_pool_bwd(gm, gw, gptr, n, weights=wts, gate_scale=1.25, gate_bits=bits)
    t_a = timed(lambda: ops.csr_aggregate(g, s.out_ptr, s.out_dst, ell=s.out_ell, cscale=kw["cscale"], rscale=kw["rscale"], dself=kw["dself"]))
    if form == "written":
        continue
    t_p = timed(lambda: agg())
    t_n = timed(lambda: agg(want_g=False))
    t_c = timed(lambda: pg.colsum())
    want = ops.csr_aggregate(g, s.out_ptr, s.out_dst, ell=s.out_ell, cscale=kw["cscale"], rscale=kw["rscale"], dself=kw["dself"])
    got, rows = agg()
    print(f"{kind}: written {t_w:.1f} + gathered {t_a:.1f} = {t_w + t_a:.1f} us; computed in the aggregation: and written {t_p:.1f} us, "
          f"not written {t_n:.1f} us + column sums {t_c:.1f} us; equal {torch.equal(got, want) and torch.equal(rows, g)}", flush=True)
