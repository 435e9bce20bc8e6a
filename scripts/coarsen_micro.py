"""Times the coarsening of the first pooling alone on 64 100-qubit circuits (list form), kernel by kernel under rocprofv3:
    rocprofv3 --kernel-trace --stats -d DIR -- python3 scripts/coarsen_micro.py [reps]"""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda:0"
h = TfimCorpus(100, list(range(1, 11)), 7, seed=42, exp_value_size=4).host_graphs()
arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"], device=dev)
rng = np.random.RandomState(0)
b = arena.batch(rng.randint(0, len(arena), size=64))
s = b.structure
torch.manual_seed(0)
model = ExpValCircuitGraphModel(22, 15, 4).to(dev).train()
with torch.no_grad():
    g = model.transformer1(b.nodes.materialize() if hasattr(b.nodes, "materialize") else b.nodes, s)
    _, s1, perm = model.pooling1(g, s)
keep = np.asarray(s1.graph_sizes)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
for mode in ("lists",):
    for _ in range(2):
        ops.asap_coarsen_lists(s.in_ptr, s.in_src, s.out_ptr, s.out_dst, s.graph_ptr, s1.graph_ptr, perm, s.num_nodes, s.edge_count(), keep, capacity=s.coarse_capacity)
    t0.record()
    for _ in range(reps):
        ops.asap_coarsen_lists(s.in_ptr, s.in_src, s.out_ptr, s.out_dst, s.graph_ptr, s1.graph_ptr, perm, s.num_nodes, s.edge_count(), keep, capacity=s.coarse_capacity)
    t1.record(); t1.synchronize()
    print(f"{mode}: {t0.elapsed_time(t1) / reps * 1e3:.1f} us per coarsening (skip = {os.environ.get('MLQEM_LISTS_SKIP', '0')})", flush=True)
