"""Per-call roofline table of the Family B train step on 100-qubit circuits (profiles/rNN_family_b_kernel_roofline.json).

One eager, single-stream, bucketed (size-stable) train step of the reference's model (docs/tutorials/gnn.py:70-122: TransformerConv x2,
ASAPooling x2, mean pool, head; hidden 15, 4 outputs) on 64 size-stratified 100-qubit circuits -- bench.py's cfg4 Family B point -- with
every ``blackwater.native.ops`` call wrapped in HIP events on its stream.  Per call signature: launches per step, median microseconds,
ALGORITHMIC bytes and the fraction of the 8 TB/s HBM peak they amount to.

Algorithmic bytes (no cache credit): every dense operand once (inputs read, outputs written) plus one row per
ENTRY and gathered operand for the walks over a graph (entries = stored in-edges + one self entry per row, E'):
  attention forward      k and v rows per entry                           2 E' 4 H C
  attention backward     k, v (destination side) and q, g (source side)   4 E' 4 H C
  ASAPooling scores      x for the segment max and x for the cluster sum  2 E' 4 D   (+ 4 E' for c[src])
  cluster-sum backward   x (g . x per entry), g_new and the max's xmax    3 E' 4 D
  segment max / its backward / softmax aggregate alone                    1 E' 4 D
  LEConv fitness / backward                                               4 E'
Structural passes (top-k, coarsening, plans, boundaries) move index arrays only: their row gives the arrays read and written once, and
is marked ``structural`` -- integer work bound by instruction issue and dependent loads, not by bandwidth.
Level-1 launches (the coarsened graph: rows of hundreds of entries that share their sources) run on DENSE BLOCKS (csrc/dense_block.hpp):
16 rows x the union of their sources, every source row read ONCE per block.  One row per entry overstates what such a launch has to
move (fractions of 2-3.5 "of peak"), so for calls that take a block plan the table's ``algorithmic_bytes`` charges one row per (block,
distinct source) for the entries inside usable blocks -- read from the plan's records -- and one per entry for the rest;
``per_entry_model_bytes`` keeps the other figure.  Counted HBM bytes per kernel name: profiles/rNN_family_b_100q_pmc.json
(scripts/make_pmc_step.sh with PMC_SCRIPT="scripts/profile_family_b.py 64 6 100").

    python scripts/family_b_roofline.py [--batch 64] [--steps 3] [--out gpurun_out/family_b_kernel_roofline.json]
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

from blackwater.data.synthetic import TfimCorpus
from blackwater.native import ops
from blackwater.native.structure import GraphStructure
from blackwater.nn import ExpValCircuitGraphModel
from blackwater.train import BucketedTrainer, StratifiedBatches

PEAK = 8000.0  # GB/s

# op -> (gathered row operands per entry, row width from the call, extra scalar bytes per entry)
GATHERS = {
    "transformer_attention_train": lambda a, k: (2, a[5] * a[6], 0),
    "dense_attention_train": lambda a, k: (2, a[5] * a[6], 0),
    "transformer_attention_bwd": lambda a, k: (4, a[7] * a[8], 0),
    "dense_attention_bwd": lambda a, k: (4, a[7] * a[8], 0),
    "asap_scores_fused": lambda a, k: (2, a[0].shape[1], 4),
    "csr_segment_max": lambda a, k: (1, a[0].shape[1], 0),
    "dense_segment_max": lambda a, k: (1, a[0].shape[1], 0),
    "csr_softmax_aggregate": lambda a, k: (1, a[0].shape[1], 4),
    "dense_softmax_aggregate": lambda a, k: (1, a[0].shape[1], 4),
    "csr_softmax_aggregate_bwd": lambda a, k: (3, a[0].shape[1], 8),
    "dense_softmax_aggregate_bwd": lambda a, k: (3, a[0].shape[1], 8),
    "csr_segment_max_bwd_": lambda a, k: (1, a[0].shape[1], 0),
    "dense_segment_max_bwd_": lambda a, k: (1, a[0].shape[1], 0),
    "leconv_fitness": lambda a, k: (0, 0, 4),
    "leconv_fitness_bwd": lambda a, k: (0, 0, 4),
    "dense_leconv_fitness_bwd": lambda a, k: (0, 0, 4),
}
# (dense_plan_build returns a plan whose buffers it allocates at capacity: counted as the blocks it wrote)
STRUCTURAL = {"segment_topk", "asap_coarsen_lists", "asap_coarsen_dense", "asap_coarsen_rows", "asap_coarsen", "dense_plan_build",
              "tile_order_by_position", "pool_keep_ptr", "asap_slot_map", "csr_build", "ell_from_csr"}
SKIP = {"padded_empty", "padded_copy", "rowmajor", "set_seed_counter", "prepare_device", "reset_tickets", "check_overflow_flags",
        "pool_gate_unpack", "pool_node_gates", "layer_identity_vectors", "pooled_means", "dense_attention_supported",
        "dense_pool_supported", "dense_pool_fits", "seq2_fits", "mlp1_fits",
        "pooled_grad_supported", "asap_lists_max_k", "asap_rows_max_bits", "asap_dense_max_k"}


def tensors_in(obj, seen):
    """Every distinct tensor reachable from a call's arguments / results (tuples, lists, dicts, RowsOf, plans, structures excluded)."""
    if obj is None or isinstance(obj, (int, float, bool, str, bytes, np.ndarray, GraphStructure)):
        return
    if torch.is_tensor(obj):
        if obj.is_cuda:
            seen.setdefault((obj.data_ptr(), tuple(obj.shape)), obj)
        return
    if isinstance(obj, ops.RowsOf):
        seen.setdefault(("rows", obj.rows.data_ptr()), obj)
        return
    if isinstance(obj, dict):
        for v in obj.values():
            tensors_in(v, seen)
        return
    if isinstance(obj, (tuple, list)):
        for v in obj:
            tensors_in(v, seen)
        return
    if hasattr(obj, "records") and hasattr(obj, "counter"):            # a DensePlan: its buffers are sized for the worst case --
        seen.setdefault(("plan", obj.records.data_ptr()), obj)           # only the blocks in use are read (nbytes below)


def nbytes(t):
    if hasattr(t, "records") and hasattr(t, "counter"):
        blocks = int(t.counter[0].item()) // 16
        return blocks * (832 * 4 + 17 * 4) + int(t.row_flag.numel())     # records + the blocks' rows + the per-row flag
    if isinstance(t, ops.RowsOf):
        return int(t.shape[0]) * int(t.shape[1]) * 4 + int(t.shape[0]) * 4          # the gathered rows and the row map
    return int(t.numel()) * t.element_size()


class Tracer:
    def __init__(self):
        self.records, self.orig, self.depth = [], {}, 0
        self.entries = {}                       # in_ptr address -> stored entries (read once per structure)
        self.blocks = {}                        # plan records address -> (sum of union sizes, entries, usable blocks)

    def names(self):
        out = []
        for name, fn in vars(ops).items():
            if name.startswith("_") or name in SKIP or not callable(fn) or isinstance(fn, type):
                continue
            if getattr(fn, "__module__", None) == ops.__name__:
                out.append(name)
        return out

    def __enter__(self):
        for name in self.names():
            self.orig[name] = getattr(ops, name)
            setattr(ops, name, self._wrap(name, self.orig[name]))
        return self

    def __exit__(self, *exc):
        for name, fn in self.orig.items():
            setattr(ops, name, fn)

    def _entries_of(self, a, k):
        """(rows, stored entries) of the graph a walk runs over: from a structure argument or from (x, in_ptr, ...)."""
        for v in list(a) + list(k.values()):
            if isinstance(v, GraphStructure):
                key = v.in_ptr.data_ptr()
                if key not in self.entries:
                    self.entries[key] = int(v.in_ptr[v.num_nodes].item())
                return v.num_nodes, self.entries[key]
        n = int(a[0].shape[0])
        for v in a[1:4]:
            if torch.is_tensor(v) and v.dtype == torch.int32 and v.dim() == 1 and v.numel() >= n + 1:
                key = v.data_ptr()
                if key not in self.entries:
                    self.entries[key] = int(v[n].item())
                return n, self.entries[key]
        return n, 0

    def _wrap(self, name, fn):
        def inner(*a, **k):
            if self.depth:                      # an op that calls another op: the outer call is the row
                return fn(*a, **k)
            self.depth += 1
            beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            beg.record()
            try:
                out = fn(*a, **k)
            finally:
                self.depth -= 1
            end.record()
            seen = {}
            tensors_in(a, seen)
            tensors_in(k, seen)
            tensors_in(out, seen)
            dense = sum(nbytes(t) for t in seen.values())
            gathered, per_entry, sig_extra = 0, 0, ""
            if name in GATHERS:
                rows, width, scalars = GATHERS[name](a, k)
                n, e = self._entries_of(a, k)
                ep = e + n
                # index arrays of a coarsened graph are allocated at a structural CAPACITY (data/arena.py coarse_caps): the walk reads
                # the stored entries only
                dense = sum(min(nbytes(t), 4 * max(e, 1)) if (torch.is_tensor(t) and t.dtype == torch.int32 and t.dim() == 1 and t.numel() > ep + 1)
                            else nbytes(t) for t in seen.values())
                per_entry = gathered = ep * (rows * width * 4 + scalars)
                sig_extra = f" N={n} E'={ep} rows/entry={rows}x{width}"
                plans = [v for v in list(a) + list(k.values()) if hasattr(v, "records") and hasattr(v, "counter")]
                if plans:          # dense blocks: a source row once per (block, distinct source) instead of once per entry
                    eff = []
                    for plan in plans:
                        key = plan.records.data_ptr()
                        if key not in self.blocks:
                            nb = int(plan.counter[0].item()) // 16
                            rec = plan.records[: nb * 832].view(nb, 832)[:, :4].cpu().numpy() if nb else np.zeros((0, 4), np.int64)
                            ok = rec[:, 2] != 0
                            self.blocks[key] = (int(rec[ok, 1].sum()), int(rec[ok, 3].sum()), int(ok.sum()))
                        uni, ent, nblk = self.blocks[key]
                        eff.append(uni + max(ep - ent, 0))
                    rows_eff = sum(eff) / len(eff)
                    gathered = int(rows_eff * (rows * width * 4) + ep * scalars)
                    sig_extra += f" blocks={self.blocks[plans[0].records.data_ptr()][2]} rows-read={int(rows_eff)}"
            shapes = "x".join(str(tuple(t.shape)) for t in list(seen.values())[:2])
            self.records.append((name, name + sig_extra + " " + shapes, dense, gathered, beg, end, per_entry))
            return out
        return inner


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "family_b_kernel_roofline.json"))
    args = ap.parse_args()
    dev = "cuda:0"
    corpus = TfimCorpus(100, list(range(1, 11)), 104, seed=42, exp_value_size=4)
    arena = corpus.arena(dev, filler_nodes=1024)
    n = len(arena)
    torch.manual_seed(0)
    sampler = StratifiedBatches(arena.node_counts[:n], arena.edge_counts[:n], args.batch, seed=13)
    tr = BucketedTrainer(ExpValCircuitGraphModel(22, 15, 4).to(dev), arena, lr=1e-3, graphs=False, node_quantum=1024, edge_quantum=4096)
    for _ in range(4):
        tr.step_ids(sampler.draw())
    torch.cuda.synchronize()
    per_step = []
    with Tracer() as t:
        for _ in range(args.steps):
            t.records = []
            torch.cuda.synchronize()
            tr.step_ids(sampler.draw())
            torch.cuda.synchronize()
            per_step.append([(r[0], r[1], r[2], r[3], r[4].elapsed_time(r[5]) * 1e3, r[6]) for r in t.records])
    ops.set_seed_counter(None)
    # aggregate by signature over the steps (median per position is overkill: calls of a signature are alike)
    agg = {}
    for step in per_step:
        for name, sig, dense, gathered, us, per_entry in step:
            row = agg.setdefault(sig, {"op": name, "signature": sig, "calls": 0, "us": [], "dense_bytes": dense, "gathered_bytes": gathered,
                                       "per_entry": per_entry})
            row["calls"] += 1
            row["us"].append(us)
    rows = []
    total_us = sum(float(np.median(r["us"])) * r["calls"] / args.steps for r in agg.values())
    for r in agg.values():
        us = float(np.median(r["us"]))
        by = r["dense_bytes"] + r["gathered_bytes"]
        rows.append({"op": r["op"], "signature": r["signature"][:160], "calls_per_step": round(r["calls"] / args.steps, 2),
                     "avg_us": round(us, 1), "algorithmic_bytes": int(by), "dense_operand_bytes": int(r["dense_bytes"]),
                     "gathered_row_bytes": int(r["gathered_bytes"]),
                     "per_entry_model_bytes": int(r["dense_bytes"] + r["per_entry"]) if r["per_entry"] != r["gathered_bytes"] else None,
                     "GBps": round(by / us / 1e3, 1) if us > 0 else None,
                     "frac_of_8TBps": round(by / us / 1e3 / PEAK, 3) if us > 0 else None,
                     "structural": r["op"] in STRUCTURAL,
                     "share_of_native_time": round(us * r["calls"] / args.steps / total_us, 3)})
    rows.sort(key=lambda r: -r["share_of_native_time"])
    out = {"what": "every native call (blackwater.native.ops) of one Family B train step on %d size-stratified 100-qubit circuits, timed IN an "
                   "eager single-stream bucketed step with HIP events (median of %d steps); algorithmic bytes = dense operands once + one "
                   "row per entry and gathered operand -- per (dense block, distinct source) where the call runs on a block plan "
                   "(per_entry_model_bytes keeps the one-row-per-entry figure) (scripts/family_b_roofline.py)"
                   % (args.batch, args.steps),
           "batch": args.batch, "nodes_per_step": int(sampler.nodes_per_batch), "sum_native_us_per_step": round(total_us, 1),
           "calls_per_step": round(sum(r["calls_per_step"] for r in rows), 1), "rows": rows}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print("family B train step: %d calls, %.1f us of native time per step" % (out["calls_per_step"], total_us))
    for r in rows[:24]:
        print("%6.1f us x%-4s %5.3f of 8 TB/s  share %5.3f  %s" % (r["avg_us"], r["calls_per_step"], r["frac_of_8TBps"] or 0, r["share_of_native_time"],
                                                                  r["signature"][:110]))


if __name__ == "__main__":
    main()
