"""Hypothesis probe (round 4): do the level-1 row kernels of Family B lose their time to rows of mixed length inside a wave?
Times attention forward / backward, the segment max and ASAPooling's softmax-weighted sum on the graph ASAPooling makes of
64 100-qubit circuits, as it stands and with its nodes RELABELLED by descending in-degree (same graph, same work).
    python scripts/deg_sort_probe.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.arena import GraphArena
from blackwater.data.synthetic import TfimCorpus
from blackwater.native import _lib, ops
if os.environ.get('MLQEM_LIB'):
    _lib.LIB_PATH = os.environ['MLQEM_LIB']
from blackwater.native.structure import GraphStructure
from blackwater.nn import ExpValCircuitGraphModel

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"


def arena_of(corpus):
    h = corpus.host_graphs()
    return GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"], device=dev)


def timed(fn, reps):
    for _ in range(3):
        fn()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for _ in range(reps):
        fn()
    end.record()
    end.synchronize()
    return beg.elapsed_time(end) / reps * 1e3


def run(tag, s, heads, ch):
    n, e = s.num_nodes, s.edge_count()
    hc = heads * ch
    torch.manual_seed(1)
    qkvs = ops.padded_empty(n, 4 * hc, dev).normal_()
    g = ops.padded_empty(n, hc, dev).normal_()
    pk = s.out_eid is None          # the recomputed backward forms (what the step runs on coarsened graphs)
    fwd = lambda: ops.transformer_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, 0.1, 1234, pair_key=pk, ell=None if pk else s.in_ell)
    out, attn, m, den = fwd()
    bwd = lambda: ops.transformer_attention_bwd(qkvs, g, attn, m, den, s, e, heads, ch, 0.1, 1234, pair_key=pk)
    x = ops.padded_empty(n, hc, dev).normal_()
    smax = lambda: ops.csr_segment_max(x, s.in_ptr, s.in_src, ell=s.in_ell)
    a_dst = torch.randn(n, device=dev)
    c_src = torch.randn(n, device=dev)
    sagg = lambda: ops.csr_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, 0.2)
    xnew = sagg()
    xq = smax()
    saggb = lambda: ops.csr_softmax_aggregate_bwd(x, xnew, g, s, e, a_dst, c_src, 0.2, xmax=xq)
    print(f"{tag}: N = {n}, E = {e}: attn fwd {timed(fwd, reps):.1f} us, attn bwd {timed(bwd, reps):.1f} us, "
          f"segmax {timed(smax, reps):.1f} us, softagg fwd {timed(sagg, reps):.1f} us, softagg bwd {timed(saggb, reps):.1f} us", flush=True)


def relabel(s, key):
    """The same graph with node i renamed rank(i) under descending `key` (stable)."""
    n = s.num_nodes
    e = int(s.in_ptr[n].item())          # edge_count() of a pooled structure is its capacity bound
    in_ptr = s.in_ptr.long()
    deg = in_ptr[1:n + 1] - in_ptr[:n]
    dst = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    src = s.in_src[:e].long()
    order = torch.sort(key, descending=True, stable=True).indices
    new_id = torch.empty(n, dtype=torch.long, device=dev)
    new_id[order] = torch.arange(n, device=dev)
    ei = torch.stack([new_id[src], new_id[dst]])
    r = GraphStructure.from_edge_index(ei, n)
    if s.out_eid is None:
        r.out_eid = None
    return r


rng = np.random.RandomState(0)
torch.manual_seed(0)
model = ExpValCircuitGraphModel(22, 15, 4).to(dev).train()
a4 = arena_of(TfimCorpus(100, list(range(1, 11)), 7, seed=42, exp_value_size=4))
b4 = a4.batch(rng.randint(0, len(a4), size=64))
with torch.no_grad():
    g = model.transformer1(b4.nodes.materialize() if hasattr(b4.nodes, "materialize") else b4.nodes, b4.structure)
    g, s1, perm1 = model.pooling1(g, b4.structure)
n = s1.num_nodes
indeg = (s1.in_ptr[1:n + 1] - s1.in_ptr[:n]).long()
outdeg = (s1.out_ptr[1:n + 1] - s1.out_ptr[:n]).long()
d = indeg.cpu().numpy()
print("level-1 in-degree quantiles 50/83/90/99/max:", [int(np.quantile(d, q)) for q in (0.5, 0.83, 0.9, 0.99, 1.0)],
      "share of edges in rows >= 32:", float(d[d >= 32].sum()) / float(d.sum()), flush=True)
run("level 1 as it stands   ", s1, 2, 15)
if os.environ.get("PROBE_ALIGNED", "0") == "1":      # what 128-byte-aligned q / k / v parts would buy: the same graph with 16 channels per head
    run("level 1, 16 channels   ", s1, 2, 16)
    run("level 0, 16 channels   ", b4.structure, 3, 16)
if os.environ.get("PROBE_ONLY_L1", "0") == "1":
    sys.exit(0)
if os.environ.get("PROBE_FULL", "0") == "1":
    run("level 1 by in-degree   ", relabel(s1, indeg), 2, 15)
    run("level 1 by out-degree  ", relabel(s1, outdeg), 2, 15)
    run("level 1 random order   ", relabel(s1, torch.rand(n, device=dev)), 2, 15)
gid = torch.repeat_interleave(torch.arange(s1.num_graphs, device=dev), (s1.graph_ptr[1:] - s1.graph_ptr[:-1]).long())
run("level 1 by (graph, in-degree)", relabel(s1, indeg - gid * 4096), 2, 15)
# the kept nodes in the order of the circuit (their level-0 index: program order of the ops) instead of the order of their scores
run("level 1 by level-0 position  ", relabel(s1, -perm1.long()), 2, 15)
s0 = b4.structure
n0 = s0.num_nodes
indeg0 = (s0.in_ptr[1:n0 + 1] - s0.in_ptr[:n0]).long()
run("level 0 as it stands   ", s0, 3, 15)
if os.environ.get("PROBE_FULL", "0") == "1":
    run("level 0 by in-degree   ", relabel(s0, indeg0), 3, 15)
