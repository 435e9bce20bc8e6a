"""Times ASAPooling's forward on the circuit DAGs of 64 100-qubit circuits (level 0 of Family B): the chain of kernels against the fused
pass (csrc/attn.hip asap_scores_fused_kernel).   python scripts/pool0_micro.py [reps] [circuits]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.arena import GraphArena
from blackwater.data.synthetic import TfimCorpus
from blackwater.native import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
circuits = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = "cuda:0"


def timed(fn):
    for _ in range(2):
        fn()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    for _ in range(reps):
        fn()
    end.record()
    end.synchronize()
    return beg.elapsed_time(end) / reps * 1e3


h = TfimCorpus(100, list(range(1, 11)), 7, seed=42, exp_value_size=4).host_graphs()
arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"], device=dev)
s = arena.batch(np.random.RandomState(0).randint(0, len(arena), size=circuits)).structure
n, d = s.num_nodes, 45
deg = (s.in_ptr[1:n + 1] - s.in_ptr[:n]).cpu().numpy()
print(f"level 0: N = {n}, E = {int(deg.sum())}; rows of more than two entries: {int((deg > 2).sum())} holding {100.0 * deg[deg > 2].sum() / deg.sum():.1f} % "
      f"of the entries (max {deg.max()})", flush=True)
torch.manual_seed(0)
x = ops.padded_copy(torch.randn(n, d, device=dev))
w_comp, b_comp, att_x = torch.randn(1, d, device=dev), torch.randn(1, device=dev), torch.randn(1, d, device=dev)
w3, b3 = torch.randn(3, d, device=dev), torch.randn(3, device=dev)
one = lambda k: torch.empty((n, k), dtype=torch.float32, device=dev)


def chain():
    xmax = ops.csr_segment_max(x, s.in_ptr, s.in_src, ell=s.in_ell)
    a = ops.linear(xmax, w_comp, b_comp, out=one(1))[:, 0]
    c = ops.linear(x, att_x, out=one(1))[:, 0]
    xn = ops.csr_softmax_aggregate(x, s.in_ptr, s.in_src, a, c, 0.2)
    return xn, ops.linear(xn, w3, b3, out=one(3))


fused = lambda: ops.asap_scores_fused(x, s.in_ptr, s.in_src, w_comp, b_comp, att_x, w3, b3, 0.2)
print("pooling forward on the circuit DAGs: chain %.1f us, fused %.1f us (max diff of x' %.2e)" % (
    timed(chain), timed(fused), (chain()[0] - fused()[3]).abs().max().item()), flush=True)
