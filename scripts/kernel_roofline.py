"""Per-kernel roofline table of the Family A train step (profiles/rNN_kernel_roofline.json).

Records every native call (``blackwater.native.ops``) of ONE real train step on the benchmark's representative batch,
single stream, then replays each recorded call in isolation: per-launch HIP events on the launch stream, the 256 MiB
Infinity Cache flushed between launches (an untimed READ of a 1 GiB buffer: the caches are left full of clean lines of
foreign data, so the timed launch neither finds its operands cached nor pays for someone else's write-backs), median of
``--reps`` launches.  Algorithmic bytes per
call follow DESIGN.md section 3 / SURVEY.md section 8d (no cache credit); fraction = bytes / time / 8 TB/s.
Rows are aggregated by call signature (op, shapes, flags); ``share`` = the signature's part of the summed native time.

    python scripts/kernel_roofline.py [--batch 1024] [--reps 7] [--out gpurun_out/kernel_roofline.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
os.environ["MLQEM_SINGLE_STREAM"] = "1"

import numpy as np
import torch

import bench
from blackwater.native import ops
from blackwater.nn import ExpValCircuitGraphModelA
from blackwater.train import Trainer

PEAK = 8000.0  # GB/s


def shape_of(t):
    if t is None:
        return None
    if isinstance(t, ops.RowsOf):
        return ("rows",) + tuple(t.shape)
    return tuple(t.shape)


class Recorder:
    """Wraps the ops entry points the Family A step uses; keeps (name, args, kwargs) so the call can be replayed."""

    NAMES = ("csr_aggregate", "linear", "linear_parts", "linear_wgrad", "linear_wgrad_parts", "segment_mean",
             "segment_mean_bwd", "relu_dropout_bwd", "segment_pool", "segment_pool_bwd", "linear_bwd_fused", "pooled_head",
             "pooled_head_bwd", "pooled_grad_aggregate", "pooled_grad_colsum")

    def __init__(self):
        self.calls, self.orig = [], {}

    def __enter__(self):
        for name in self.NAMES:
            if not hasattr(ops, name):
                continue
            self.orig[name] = getattr(ops, name)
            setattr(ops, name, self._wrap(name, self.orig[name]))
        return self

    def __exit__(self, *exc):
        for name, fn in self.orig.items():
            setattr(ops, name, fn)

    def _wrap(self, name, fn):
        def inner(*a, **k):
            out = fn(*a, **k)
            self.calls.append((name, fn, a, dict(k)))
            return out
        return inner


def describe(name, a, k, struct):
    """(signature string, algorithmic bytes) of one recorded call."""
    n_nodes, e = struct.num_nodes, struct.num_edges
    if name == "csr_aggregate":
        x = a[0]
        n, c = x.shape
        e_eff = e + (n if k.get("dself") is not None else 0)     # the self term is one more source row per node
        by = bench.agg_bytes(n, e_eff, c) + (4 * n * c if k.get("z") is not None else 0)
        flags = "+".join(f for f in ("z" if k.get("z") is not None else "", "bias" if k.get("bias") is not None else "",
                                     "relu" if k.get("relu") else "", "drop" if k.get("drop_p", 0) > 0 else "",
                                     "self" if k.get("dself") is not None else "",
                                     "T" if a[1].data_ptr() == struct.out_ptr.data_ptr() else "") if f)
        return f"csr_aggregate N={n} C={c} [{flags}]", by
    if name == "pooled_grad_aggregate":      # a branch's first backward aggregation, its source computed (csrc/pooled_grad.hip)
        pg = a[0]
        n = pg.num_nodes
        c = (pg.g_wmean if pg.g_wmean is not None else pg.g_mean).shape[1]
        e_eff = e + n                             # the row itself is always computed (its g, or its self term)
        want_g = k.get("want_g", True)
        by = 4 * (n + 1) + 4 * e_eff + 4 * n + 10 * (e_eff + n) + 4 * c * n * (2 if want_g else 1)
        flags = "+".join(f for f in ("self" if k.get("dself") is not None else "", "g" if want_g else "", "T") if f)
        return f"pooled_grad_aggregate N={n} C={c} [{flags}]", by
    if name == "pooled_grad_colsum":
        pg = a[0]
        return f"pooled_grad_colsum N={pg.num_nodes}", 6 * pg.num_nodes
    if name == "segment_pool":
        x = a[0]
        n, c = x.shape
        b = a[2]
        outs = int(bool(k.get("mean", True))) + int(bool(k.get("wmean", False)))
        return f"segment_pool N={n} C={c} outs={outs}", 4 * c * (n + b * outs) + 4 * (b + 1) + (4 * n if k.get("wmean") else 0)
    if name == "segment_pool_bwd":
        ref = a[0] if a[0] is not None else a[1]
        b, c = ref.shape
        n = a[3]
        gated = k.get("gate") is not None
        return (f"segment_pool_bwd N={n} C={c}{' gate' if gated else ''}",
                4 * c * n * (2 if gated else 1) + 4 * n + 4 * c * b * 2 + 4 * (b + 1))
    if name == "linear":
        x, w = a[0], a[1]
        n, i = x.shape
        o = w.shape[1] if k.get("transposed") else w.shape[0]
        by = 4 * (n * (i + o) + i * o) + (4 * n * o if k.get("gate") is not None else 0) + (4 * n * o if k.get("accumulate") else 0)
        flags = "+".join(f for f in ("T" if k.get("transposed") else "", "rows" if isinstance(x, ops.RowsOf) else "",
                                     "rowscale" if k.get("rowscale") is not None else "", "gate" if k.get("gate") is not None else "",
                                     "relu" if k.get("relu") else "") if f)
        return f"linear N={n} {i}->{o} [{flags}]", by
    if name == "linear_parts":
        xs, ws, ys = a[0], a[1], a[2]
        n = xs[0].shape[0]
        i_tot, o_tot = sum(t.shape[1] for t in xs), sum(t.shape[1] for t in ys)
        by = 4 * n * (i_tot + o_tot) + 4 * sum(w.numel() for w in ws) + (4 * n * o_tot if k.get("gate") is not None else 0)
        flags = "+".join(f for f in ("T" if k.get("transposed") else "", "rows" if isinstance(xs[0], ops.RowsOf) else "",
                                     "gate" if k.get("gate") is not None else "") if f)
        return f"linear_parts N={n} {'+'.join(str(t.shape[1]) for t in xs)}->{'+'.join(str(t.shape[1]) for t in ys)} [{flags}]", by
    if name == "linear_wgrad":
        gy, x = a[0], a[1]
        n, o = gy.shape
        i = x.shape[1]
        return f"wgrad N={n} x[{i}]^T gy[{o}]{' rows' if isinstance(x, ops.RowsOf) else ''}", 4 * n * (i + o)
    if name == "linear_wgrad_parts":
        gys, x = a[0], a[1]
        n, i = gys[0].shape[0], x.shape[1]
        o_tot = sum(t.shape[1] for t in gys)
        return (f"wgrad_parts N={n} x[{i}]^T gy[{'+'.join(str(t.shape[1]) for t in gys)}]"
                f"{' rows' if isinstance(x, ops.RowsOf) else ''}"), 4 * n * (i + o_tot)
    if name == "segment_mean":
        x = a[0]
        n, c = x.shape
        b = a[2]
        return f"segment_mean N={n} C={c}", 4 * c * (n + b) + 4 * (b + 1)
    if name == "segment_mean_bwd":
        g = a[0]
        b, c = g.shape
        n = a[2]
        return f"segment_mean_bwd N={n} C={c}", 4 * c * (n + b) + 4 * (b + 1)
    if name == "linear_bwd_fused":
        gy, x, w = a[0], a[1], a[2]
        n, o = gy.shape
        i = x.shape[1]
        extra = 4 * n * o if (len(a) > 3 and a[3] is not None) or k.get("gb_src") is not None else 0
        return f"linear_bwd_fused N={n} gy[{o}] x[{i}] (gx + gW + gb, one pass)", 4 * n * (o + 2 * i) + extra
    if name in ("pooled_head", "pooled_head_bwd"):
        terms = a[0]
        b, c = terms[0][0].shape
        return f"{name} B={b} C={c} terms={len(terms)}", 4 * b * c * len(terms) * (2 if name.endswith("bwd") else 1)
    if name == "relu_dropout_bwd":
        n, c = a[0].shape
        return f"relu_dropout_bwd N={n} C={c}", 12 * n * c
    return name, 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=bench.DEFAULT_BATCH)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "kernel_roofline.json"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n_j = -(-args.batch // len(bench.STEPS_LIST)) + 1
    corpus = bench.build_corpus(n_j)
    arena = corpus.arena(dev)
    ids = bench.fixed_ids(len(arena), args.batch)
    torch.manual_seed(0)
    model = ExpValCircuitGraphModelA(100, 22, 10).to(dev)
    trainer = Trainer(model, lr=1e-3)
    batch = arena.batch(ids)
    trainer.step(batch)                         # warm: allocator, side tables
    torch.cuda.synchronize()
    with Recorder() as rec:
        trainer.step(batch)
    torch.cuda.synchronize()
    struct = batch.structure
    flush = torch.zeros(256 << 20, dtype=torch.float32, device=dev)   # 1 GiB
    stream = torch.cuda.current_stream()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rows = {}
    for name, fn, a, k in rec.calls:
        sig, by = describe(name, a, k, struct)
        times = []
        for _ in range(args.reps):
            flush.sum()
            beg.record(stream)
            fn(*a, **k)
            end.record(stream)
            end.synchronize()
            times.append(beg.elapsed_time(end) * 1e3)
        us = float(np.median(times))
        r = rows.setdefault(sig, {"op": name, "signature": sig, "calls_per_step": 0, "algorithmic_bytes": int(by), "us": []})
        r["calls_per_step"] += 1
        r["us"].append(us)
    # batch assembly (called by arena.batch, outside ops)
    times = []
    for _ in range(args.reps):
        flush.sum()
        beg.record(stream)
        arena.batch(ids)
        end.record(stream)
        end.synchronize()
        times.append(beg.elapsed_time(end) * 1e3)
    n, e, f = struct.num_nodes, struct.num_edges, 22
    rows["batch_assemble"] = {"op": "batch_assemble", "signature": f"batch_assemble B={args.batch} N={n} E={e} (structure + scalars + row map; x not copied)",
                              "calls_per_step": 1, "algorithmic_bytes": int(2 * 4 * 3 * n + 5 * 4 * e + 6 * 4 * n + 2 * 8 * n * 2),
                              "us": [float(np.median(times))]}
    table, total = [], 0.0
    for r in rows.values():
        us = float(np.mean(r["us"]))
        total += us * r["calls_per_step"]
        gbps = r["algorithmic_bytes"] / us / 1e3 if us > 0 else 0.0
        table.append({"op": r["op"], "signature": r["signature"], "calls_per_step": r["calls_per_step"],
                      "algorithmic_bytes": r["algorithmic_bytes"], "avg_us": round(us, 1), "GBps": round(gbps, 1),
                      "frac_of_8TBps": round(gbps / PEAK, 3)})
    for t in table:
        t["share_of_native_time"] = round(t["avg_us"] * t["calls_per_step"] / total, 4)
    table.sort(key=lambda t: -t["share_of_native_time"])
    out = {"what": "every native call of one Family A train step on the bench's representative batch, replayed in isolation "
                   "(single stream, Infinity Cache flushed between launches, median of %d); algorithmic bytes per "
                   "DESIGN.md section 3; regenerate with scripts/kernel_roofline.py" % args.reps,
           "batch": args.batch, "nodes": n, "edges": e, "sum_native_us_per_step": round(total, 1),
           "below_50pct": [t["signature"] for t in table if t["frac_of_8TBps"] < 0.5 and t["share_of_native_time"] >= 0.03],
           "rows": table}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"{'signature':78s} {'n':>2s} {'us':>8s} {'GB/s':>7s} {'frac':>5s} {'share':>6s}")
    for t in table:
        print(f"{t['signature'][:78]:78s} {t['calls_per_step']:2d} {t['avg_us']:8.1f} {t['GBps']:7.0f} {t['frac_of_8TBps']:5.2f} {t['share_of_native_time']:6.3f}")
    print(f"sum of native time per step: {total:.0f} us")


if __name__ == "__main__":
    main()
