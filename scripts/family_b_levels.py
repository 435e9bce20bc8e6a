"""Family B on the headline graphs, level by level: node / edge counts and degree extremes of the input graph and of the two
coarsened graphs, and the per-call durations of one train step's kernels (torch profiler off: HIP events around each
autograd node would change the launch sequence, so the durations come from `rocprofv3 --kernel-trace` of this script).

    python scripts/family_b_levels.py [batch] [qubits]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.arena import GraphArena
from blackwater.data.synthetic import TfimCorpus
from blackwater.nn import ExpValCircuitGraphModel
from blackwater.native import functional as F

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 100
corpus = (TfimCorpus(4, list(range(15)), 70, seed=42, two_q="cx", exp_value_size=4) if nq == 4
          else TfimCorpus(nq, list(range(1, 11)), 7, seed=42, exp_value_size=4))
h = corpus.host_graphs()
arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"], device="cuda:0")
torch.manual_seed(0)
model = ExpValCircuitGraphModel(22, 15, 4).to("cuda:0").train()
rng = np.random.RandomState(0)
b = arena.batch(rng.randint(0, len(arena), size=batch))


def describe(tag, s):
    ip, op = s.in_ptr.cpu().numpy().astype(np.int64), s.out_ptr.cpu().numpy().astype(np.int64)
    din, dout = np.diff(ip), np.diff(op)
    e = int(ip[-1])
    print(f"{tag}: N = {s.num_nodes}, E (no self-loops) = {e}, E/N = {e / max(1, s.num_nodes):.2f}, in-degree max {din.max()} "
          f"p99 {np.percentile(din, 99):.0f} p999 {np.percentile(din, 99.9):.0f}, out-degree max {dout.max()}, rows with in-degree > 64: "
          f"{(din > 64).sum()} holding {din[din > 64].sum() / max(1, e):.1%} of the edges; > 16: {(din > 16).sum()} holding "
          f"{din[din > 16].sum() / max(1, e):.1%}", flush=True)


nodes = b.nodes.materialize() if hasattr(b.nodes, "materialize") else b.nodes
s = b.structure
describe("level 0", s)
g = model.transformer1(nodes, s)
g, s, _ = model.pooling1(g, s)
describe("level 1", s)
g = model.transformer2(g, s)
g, s, _ = model.pooling2(g, s)
describe("level 2", s)
torch.cuda.synchronize()
