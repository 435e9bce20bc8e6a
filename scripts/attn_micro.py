"""Times the TransformerConv attention kernels alone (forward-train, backward) on the structures Family B meets:
cfg2 level 0 (4-qubit circuits, batch 1024), and levels 0 / 1 of a 64-circuit batch of 100-qubit circuits (level 1 = the
graph ASAPooling makes of it: rows of 100-500 in-edges).   python scripts/attn_micro.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.arena import GraphArena
from blackwater.data.synthetic import TfimCorpus
from blackwater.native import _lib, ops
if os.environ.get('MLQEM_LIB'):
    _lib.LIB_PATH = os.environ['MLQEM_LIB']
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
    qkvs = ops.padded_empty(n, 4 * hc, dev).normal_()
    g = ops.padded_empty(n, hc, dev).normal_()
    fwd = lambda: ops.transformer_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, 0.1, 1234)
    out, attn, m, den = fwd()
    bwd = lambda: ops.transformer_attention_bwd(qkvs, g, attn, m, den, s, e, heads, ch, 0.1, 1234)
    e1 = e + n
    by = 4 * (n + 1) + 4 * e1 + 4 * hc * (n + e1 + e1 + n)
    tf, tb = timed(fwd, reps), timed(bwd, reps)
    print(f"{tag}: N = {n}, E = {e}, H = {heads}: forward {tf:.1f} us ({by / tf / 1e3:.0f} GB/s algorithmic), backward {tb:.1f} us", flush=True)


rng = np.random.RandomState(0)
a2 = arena_of(TfimCorpus(4, list(range(15)), 70, seed=42, two_q="cx", exp_value_size=4))
b2 = a2.batch(np.arange(1024) * len(a2) // 1024)
run("cfg2 level 0", b2.structure, 3, 15)
torch.manual_seed(0)
model = ExpValCircuitGraphModel(22, 15, 4).to(dev).train()
with torch.no_grad():
    g = model.transformer1(b2.nodes.materialize() if hasattr(b2.nodes, "materialize") else b2.nodes, b2.structure)
    g, s1, _ = model.pooling1(g, b2.structure)
run("cfg2 level 1", s1, 2, 15)
a4 = arena_of(TfimCorpus(100, list(range(1, 11)), 7, seed=42, exp_value_size=4))
b4 = a4.batch(rng.randint(0, len(a4), size=64))
run("100q level 0", b4.structure, 3, 15)
with torch.no_grad():
    g = model.transformer1(b4.nodes.materialize() if hasattr(b4.nodes, "materialize") else b4.nodes, b4.structure)
    g, s1, _ = model.pooling1(g, b4.structure)
run("100q level 1", s1, 2, 15)
