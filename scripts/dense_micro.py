"""Times TransformerConv's edge softmax at level 1 of Family B -- the per-edge kernels alone, and with the long rows on the matrix
cores (csrc/dense_block.hip) -- on the graph ASAPooling makes of a batch of 100-qubit circuits.

    python scripts/dense_micro.py [reps] [circuits]          (under rocprofv3 --kernel-trace --stats for per-kernel times)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.arena import GraphArena
from blackwater.data.synthetic import TfimCorpus
from blackwater.native import _lib, ops
from blackwater.nn import ExpValCircuitGraphModel

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
rng = np.random.RandomState(0)
b = arena.batch(rng.randint(0, len(arena), size=circuits))
torch.manual_seed(0)
model = ExpValCircuitGraphModel(22, 15, 4).to(dev).train()
with torch.no_grad():
    g = model.transformer1(b.nodes, b.structure)
    g, s, _ = model.pooling1(g, b.structure)
n, e = s.num_nodes, s.edge_count()
real_e = int(s.in_ptr[n].item())
print(f"level 1: N = {n}, E = {real_e} (capacity {e})", flush=True)
stride = _lib.load().mlqem_dense_plan_record_ints()
plans = {}
for name in ("in", "out"):
    p = plans[name] = s.dense_plan(name)
    nb = int(p.counter.item()) // 16
    rec = p.records.cpu().numpy()[: nb * stride].reshape(nb, stride)
    ok = rec[:, 2] == 1
    ptr = (s.in_ptr if name == "in" else s.out_ptr)[: n + 1].cpu().numpy().astype(np.int64)
    deg = np.diff(ptr)
    flag = p.row_flag.cpu().numpy().astype(bool)
    cells = (rec[ok, 0] * ((rec[ok, 1] + 15) // 16 * 16)).sum()
    print(f"plan {name}: {nb} blocks ({100.0 * ok.mean():.1f} % usable), union mean {rec[ok, 1].mean():.0f} max {rec[:, 1].max()}, "
          f"rows in blocks {flag.sum()} ({100.0 * flag.mean():.1f} %) holding {100.0 * deg[flag].sum() / deg.sum():.1f} % of the entries, "
          f"cells {cells} = {cells / max(deg[flag].sum(), 1):.2f} per entry", flush=True)
spec = s._block_order
t_build = timed(lambda: ops.dense_plan_build(s.in_ptr, s.in_src, s.loops, n, s.graph_ptr, s.num_graphs, spec[0], spec[1]))
print(f"plan build (one direction): {t_build:.1f} us")

heads, ch, cp = 2, 15, 16
qh = torch.zeros(n, 4 * heads, cp)
qh[:, :, :ch] = torch.randn(n, 4 * heads, ch)
qkvs = ops.padded_copy(qh.view(n, -1).to(dev))
gout = ops.padded_copy(torch.randn(n, heads * ch).to(dev))
pin, pout = plans["in"], plans["out"]
edge_f = lambda: ops.transformer_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, 0.1, 7, pair_key=True, head_pitch=cp)
side = torch.cuda.Stream()
dense_f = lambda side=None: ops.dense_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, pin, drop_p=0.1, seed=7, side=side)
ref, got = edge_f(), dense_f()
print("attention forward: per-edge %.1f us, with dense blocks %.1f us, the two kernels on two streams %.1f us (max diff %.2e)" % (
    timed(edge_f), timed(dense_f), timed(lambda: dense_f(side)), (ref[0] - dense_f(side)[0]).abs().max().item()), flush=True)
edge_b = lambda: ops.transformer_attention_bwd(qkvs, gout, ref[1], ref[2], ref[3], s, e, heads, ch, 0.1, 7, pair_key=True, head_pitch=cp)
dense_b = lambda side=None: ops.dense_attention_bwd(qkvs, gout, got[1], got[2], got[3], s, e, heads, ch, pin, pout, drop_p=0.1, seed=7, side=side)
print("attention backward (both sides): per-edge %.1f us, with dense blocks %.1f us, on two streams %.1f us (max diff %.2e)" % (
    timed(edge_b), timed(dense_b), timed(lambda: dense_b(side)), (edge_b() - dense_b(side)).abs().max().item()), flush=True)
