"""Times the level-1 row kernels of Family B alone -- tiled (csrc/tile_*.hip) and per-edge -- on the graph ASAPooling makes of a
64-circuit batch of 100-qubit circuits.  Tile shape from the environment (MLQEM_TILE_ROWS, MLQEM_TILE_CAP, read by ops at import).

    python scripts/tile_micro.py [reps] [circuits]          (under rocprofv3 --kernel-trace --stats for per-kernel times)
"""
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
print(f"level 1: N = {n}, E = {real_e} (capacity {e}), tiled = {s.tiled}, tile rows {ops.TILE_ROWS}, cap {ops.TILE_CAP}", flush=True)
for name, ptr in (("in", s.in_ptr), ("out", s.out_ptr)):
    # how the quad walk of the per-edge kernels (4 lanes per (row, head), 8 rows per wave, 4 entries per trip) fills its waves:
    # a wave makes as many trips as its LONGEST row needs
    deg = np.diff(ptr[: n + 1].cpu().numpy().astype(np.int64)) + 1
    trips = -(-deg // 4)
    pad = (-len(trips)) % 8
    per_wave = np.concatenate([trips, np.zeros(pad, np.int64)]).reshape(-1, 8)
    long_rows = deg >= 32
    short_trips = np.where(long_rows, 0, trips)
    coop = np.concatenate([short_trips, np.zeros(pad, np.int64)]).reshape(-1, 8).max(1).sum() + 2 * (-(-deg[long_rows] // 64)).sum()
    print(f"rows {name}: degree median {int(np.median(deg))}, p90 {int(np.percentile(deg, 90))}, p99 {int(np.percentile(deg, 99))}, max {deg.max()}; "
          f"rows of 32+ entries: {100.0 * long_rows.mean():.2f} % of the rows, {100.0 * deg[long_rows].sum() / deg.sum():.1f} % of the entries; "
          f"wave trips: {per_wave.max(1).sum()} as walked, {int(np.ceil(trips.sum() / 8))} if every quad were busy, {coop} with long rows shared by the wave",
          flush=True)
pin, pout = s.tile_plan("in"), s.tile_plan("out")
for name, p in (("in", pin), ("out", pout)):
    ti = p.tinfo.cpu().numpy()
    loc = p.loc.cpu().numpy().view(np.uint16)[:real_e]
    print(f"plan {name}: {p.num_tiles} tiles, union mean {ti[:, 2].mean():.0f} max {ti[:, 2].max()}, long rows per tile {ti[:, 1].mean():.1f}, "
          f"entries per tile mean {ti[:, 3].mean():.0f} max {ti[:, 3].max()}, entries without a slot {100.0 * (loc == 0xFFFF).mean():.2f} %", flush=True)
t_plan = timed(lambda: ops.tile_plan_build(s.in_ptr, s.in_src, n, int(s.in_src.shape[0]), s._tile_spec[0], s._tile_spec[1]))
print(f"plan build (one direction): {t_plan:.1f} us")

heads, ch, cp = 2, 15, 16
qh = torch.zeros(n, 4 * heads, cp)
qh[:, :, :ch] = torch.randn(n, 4 * heads, ch)
qkvs = ops.padded_copy(qh.view(n, -1).to(dev))
gout = ops.padded_copy(torch.randn(n, heads * ch).to(dev))
ref = ops.transformer_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, 0.1, 7, pair_key=True, head_pitch=cp)
got = ops.tile_attention(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, pin, drop_p=0.1, seed=7, head_pitch=cp)
print("attention forward: per-edge %.1f us, tiled %.1f us (max diff %.2e)" % (
    timed(lambda: ops.transformer_attention_train(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, 0.1, 7, pair_key=True, head_pitch=cp)),
    timed(lambda: ops.tile_attention(qkvs, s.in_ptr, s.in_src, s.loops, e, heads, ch, pin, drop_p=0.1, seed=7, head_pitch=cp)),
    (ref[0] - got[0]).abs().max().item()), flush=True)
print("attention backward (both sides): per-edge %.1f us, tiled %.1f us" % (
    timed(lambda: ops.transformer_attention_bwd(qkvs, gout, ref[1], ref[2], ref[3], s, e, heads, ch, 0.1, 7, pair_key=True, head_pitch=cp)),
    timed(lambda: ops.tile_attention_bwd(qkvs, gout, got[1], got[2], got[3], s, e, heads, ch, pin, pout, drop_p=0.1, seed=7, head_pitch=cp))), flush=True)

d = heads * ch
x = ops.padded_copy(torch.randn(n, d).to(dev))
w_comp, b_comp, att_x = torch.randn(1, d, device=dev), torch.randn(1, device=dev), torch.randn(1, d, device=dev)
w3, b3 = torch.randn(3, d, device=dev), torch.randn(3, device=dev)
c_src = ops.linear(x, att_x)[:, 0].contiguous()


def edge_fwd():
    xmax = ops.csr_segment_max(x, s.in_ptr, s.in_src, ell=s.in_ell)
    a_dst = ops.linear(xmax, w_comp, b_comp)[:, 0].contiguous()
    xnew = ops.csr_softmax_aggregate(x, s.in_ptr, s.in_src, a_dst, c_src, 0.2)
    return xmax, a_dst, xnew, ops.linear(xnew, w3, b3)


xmax_r, a_dst, xnew_r, _ = edge_fwd()
tile_fwd = lambda: ops.tile_asap_scores(x, s.in_ptr, s.in_src, c_src, w_comp[0].contiguous(), b_comp, w3, b3, 0.2, pin)
xnew, xmax, stat, pqr = tile_fwd()
print("pooling forward (max + score + softmax-sum + projections): per-edge %.1f us, tiled %.1f us (max diff %.2e)" % (
    timed(edge_fwd), timed(tile_fwd), (xnew - xnew_r).abs().max().item()), flush=True)
gnew = ops.padded_copy(torch.randn(n, d).to(dev))


def edge_bwd():
    gx, ga, gc, ties = ops.csr_softmax_aggregate_bwd(x, xnew_r, gnew, s, e, a_dst, c_src, 0.2, xmax=xmax_r, gx_rank1=att_x[0])
    ops.csr_segment_max_bwd_(gx, x, xmax_r, None, s, ties=ties, gmax_rank1=(ga, w_comp[0].contiguous()))
    return gx


tile_bwd = lambda: ops.tile_asap_scores_bwd(x, xnew, gnew, xmax, s, c_src, w_comp[0].contiguous(), att_x[0].contiguous(), 0.2, pin, pout, stat)[0]
print("pooling backward: per-edge %.1f us, tiled %.1f us (max diff %.2e)" % (timed(edge_bwd), timed(tile_bwd),
                                                                            (edge_bwd() - tile_bwd()).abs().max().item()), flush=True)
