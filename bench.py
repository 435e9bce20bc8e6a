"""Benchmark of the hot path: circuits/sec of one GNN train step on synthetic 100-qubit TFIM-Trotter graphs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = device batch assembly -> forward -> MSE -> backward -> (gradient all-reduce) -> Adam, nothing skipped.
Rank 0 prints ONE JSON line (contract in the task statement) carrying ``roofline`` (the CSR aggregation kernel,
timed live with HIP events on the stream it runs on) and, at N=1, ``cpu_baseline`` (the CPU oracle on a bounded
sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "ml-qem_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch


# Circuits per step per GPU.  Sized for 288 GB of HBM rather than for the reference's host-collated batches of 32: one step
# then moves 11 M graph nodes (~20 GB of activations), launches are long enough for their tails not to matter
# (aggregation kernel: 72 % of the HBM peak at 256 circuits, 78 % at 1024) and the host has 11 ms to enqueue 2 ms of work.
# Measured on one MI355X: 256 -> 88 k, 1024 -> 99-101 k, 2048 -> 101 k circuits/s.
DEFAULT_BATCH = 1024


def fixed_ids(n_graphs, batch=DEFAULT_BATCH):
    """The representative batch the roofline leg (and the profiling scripts) use: every step count, evenly."""
    return np.arange(batch) * n_graphs // batch


def build_corpus(n_j, seed=42):
    from blackwater.data.synthetic import tfim_corpus

    return tfim_corpus(100, list(range(1, 11)), n_j, seed=seed, two_q="ecr", exp_value_size=1)


def agg_bytes(n, e_with_loops, c):
    """Algorithmic bytes of one CSR aggregation: rowptr + col + norm scalar + one source row per
    edge + one output row per node, fp32 data / int32 indices, no cache credit."""
    return 4 * (n + 1) + 4 * e_with_loops + 4 * n + 4 * c * (e_with_loops + n)


def roofline_leg(batch, reps=20):
    """Times the dominant kernel -- the GCN-normalised CSR aggregation at C = 10 (the hidden width of the model) --
    on the benchmark batch with HIP events on the launch stream."""
    from blackwater.native import ops

    s = batch.structure
    n = s.num_nodes
    e_loops = s.num_edges + n  # the self-loop of every node is one more source row (E')
    c = 10
    # the GCN layer's forward aggregation exactly as the model launches it: input pre-scaled by the projection,
    # one norm scalar per node.  Four input/output buffer pairs are rotated so that no launch finds its operands
    # in the 256 MiB Infinity Cache left there by the previous one.
    nbuf = 4
    hs = [ops.padded_empty(n, c, batch.structure.in_ptr.device).normal_() for _ in range(nbuf)]   # the layout the model uses
    outs = [ops.padded_empty(n, c, batch.structure.in_ptr.device) for _ in range(nbuf)]
    dinv = s.gcn_dinv
    run = lambda k: ops.csr_aggregate(hs[k % nbuf], s.in_ptr, s.in_src, ell=s.in_ell, rscale=dinv, dself=dinv, out=outs[k % nbuf])
    for k in range(nbuf):
        run(k)
    stream = torch.cuda.current_stream()
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record(stream)
    for k in range(reps):
        run(k)
    end.record(stream)
    end.synchronize()
    sec = beg.elapsed_time(end) * 1e-3 / reps
    b = agg_bytes(n, e_loops, c)
    # the box's own stream-copy rate, for context (SURVEY section 8d asks for the measured peak next to the vendor one)
    src_buf = torch.empty(256 << 20, dtype=torch.float32, device=batch.structure.in_ptr.device).normal_()
    dst_bufs = [torch.empty_like(src_buf) for _ in range(2)]
    for k in range(2):
        dst_bufs[k].copy_(src_buf)
    beg.record(stream)
    for k in range(6):
        dst_bufs[k % 2].copy_(src_buf)
    end.record(stream)
    end.synchronize()
    copy_gbps = 6 * 2 * src_buf.numel() * 4 / (beg.elapsed_time(end) * 1e-3) / 1e9
    del src_buf, dst_bufs
    peak = 8000.0  # GB/s, MI355X HBM3E (guide: MI355X_MICROARCH.md chip table)
    ach = b / sec / 1e9
    # HBM traffic per launch comes from rocprofv3 PMC passes (they cannot run inside this process); the committed
    # summary applies only if it was taken on exactly this batch.
    traffic, src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_aggregate_pmc.json")) as fh:
            pmc = json.load(fh)
        if pmc["nodes"] == n and pmc["edges_with_loops"] == e_loops and pmc["C"] == c:
            traffic, src = pmc["traffic_bytes_per_launch"], "profiles/r01_aggregate_pmc.json (FETCH_SIZE x2 + WRITE_SIZE)"
    except (OSError, KeyError, ValueError):
        pass
    return {"bound": "hbm", "kernel": "csr_aggregate_ell_kernel<4,false,2> (GCN forward aggregation, C=10)",
            "achieved": round(ach, 1), "peak": peak, "unit": "GB/s", "frac": round(ach / peak, 4), "traffic": traffic,
            "traffic_source": src, "bytes_per_launch": int(b), "us_per_launch": round(sec * 1e6, 2), "nodes": n,
            "edges_with_loops": e_loops, "measured_copy_GBps": round(copy_gbps, 1)}


def mae_leg(model, batch):
    """Mean absolute error of the model after the benchmark's few dozen steps from random initialisation, and of the
    unmitigated noisy values, against the SYNTHETIC ideal values on the fixed representative batch.  It only shows that the
    loss plumbing is live (600 steps bring the MSE from ~8 to 0.09, scripts/soak.py); the expectation-value accuracy claim
    of this build -- the "exp-val MAE" half of BASELINE.json's metric -- is the `parity` object: device predictions vs the
    CPU reference arithmetic on identical inputs and weights."""
    was_training = model.training
    model.eval()
    with torch.no_grad():
        pred = model(*batch.model_args())
    model.train(was_training)
    y = batch.y.reshape(pred.shape)
    noisy = batch.noisy_0.reshape(pred.shape)
    return {"mitigated": round(float((pred - y).abs().mean()), 6), "noisy": round(float((noisy - y).abs().mean()), 6),
            "circuits": int(y.shape[0]), "labels": "synthetic"}


def parity_leg(model, arena, corpus, n_qubits, n_check=10):
    """Predictions of the trained device model vs the CPU oracle carrying the same weights, on one circuit per
    Trotter step count (eval mode, fp32 oracle = the reference's CPU arithmetic, fp64 oracle = the exact value)."""
    from oracle.models import FamilyA

    n_graphs = len(corpus["x"])
    sel = np.arange(n_check) * n_graphs // n_check
    was_training = model.training
    model.eval()
    with torch.no_grad():
        b = arena.batch(sel)
        got = model(*b.model_args()).double().cpu()
    model.train(was_training)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    out = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        ref = FamilyA(n_qubits, 22, 10).eval()
        ref.load_state_dict(state)
        ref = ref.to(dt)
        want = []
        with torch.no_grad():
            for g in sel:
                x = torch.from_numpy(corpus["x"][g]).to(dt)
                t = lambda k: torch.from_numpy(corpus[k][g:g + 1]).to(dt)
                want.append(ref(t("noisy"), t("observable"), t("depth"), x, torch.from_numpy(corpus["edge_index"][g]),
                                torch.zeros(x.shape[0], dtype=torch.long)).double())
        err = (got - torch.cat(want)).abs()
        out[name] = {"mae": float(err.mean()), "max": float(err.max())}
    return {"circuits": int(n_check), "tolerance": 1e-5, "exp_val_mae_vs_cpu_f32": out["f32"]["mae"],
            "max_abs_err_vs_cpu_f32": out["f32"]["max"], "exp_val_mae_vs_cpu_f64": out["f64"]["mae"],
            "max_abs_err_vs_cpu_f64": out["f64"]["max"], "prediction_scale": float(got.abs().mean())}


def cpu_baseline_leg(corpus, ids, n_qubits, budget_s=20.0):
    """The CPU oracle (pure-torch restatement of the reference's PyG math) doing the same train step on a bounded
    sample: batches of the same graphs, all host cores."""
    from oracle.models import FamilyA

    torch.manual_seed(0)
    model = FamilyA(n_qubits, 22, 10).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    def collate(sel):
        xs, eis, bs, off = [], [], [], 0
        for b, g in enumerate(sel):
            x = torch.from_numpy(corpus["x"][g])
            xs.append(x)
            eis.append(torch.from_numpy(corpus["edge_index"][g]) + off)
            bs.append(torch.full((x.shape[0],), b, dtype=torch.long))
            off += x.shape[0]
        t = lambda k: torch.from_numpy(corpus[k][sel])
        return (t("noisy"), t("observable"), t("depth"), torch.cat(xs), torch.cat(eis, 1), torch.cat(bs)), t("y")

    def one_step(sel):
        t0 = time.perf_counter()
        args, y = collate(sel)
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(model(*args), y)
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    bsz = 8
    # torch's intra-op pool does not scale on these scatter/gather ops (256 threads is ~400x slower than 8 on the
    # GPU box's host): pick the fastest of a few thread counts on one batch each, then time with that setting.
    best_t, best_n = None, 1
    for nt in sorted({1, 4, 8, 16, min(32, os.cpu_count())}):
        if nt > os.cpu_count():
            continue
        torch.set_num_threads(nt)
        one_step(ids[:bsz])
        dt = one_step(ids[:bsz])
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nt
    torch.set_num_threads(best_n)
    done, t_total, pos = 0, 0.0, 0
    while t_total < budget_s and pos + bsz <= len(ids):
        t_total += one_step(ids[pos:pos + bsz])
        pos += bsz
        done += bsz
    return {"value": round(done / max(t_total, 1e-9), 2), "unit": "circuits/s", "cores": best_n, "kind": "port",
            "sample": f"{done} circuits of the same corpus in batches of {bsz} (oracle/models.py FamilyA, fp32, "
                      f"torch {best_n} threads = fastest of 1/4/8/16/32 on this {os.cpu_count()}-core host, "
                      f"full train step: collate + forward + MSE + backward + Adam)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH, help="circuits per step per GPU")
    ap.add_argument("--n-j", type=int, default=50, help="J values per Trotter step count (corpus = 10 x n_j circuits)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    # MLQEM_BENCH_BACKEND=gloo lets two ranks share one GPU to rehearse the multi-rank control flow on a 1-GPU box;
    # the driver's runs use the default: one GPU per rank, RCCL ("nccl") over xGMI.
    backend = os.environ.get("MLQEM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)

    from blackwater.data.arena import GraphArena
    from blackwater.nn import ExpValCircuitGraphModelA
    from blackwater.train import Trainer

    corpus = build_corpus(args.n_j)
    n_graphs = len(corpus["x"])
    arena = GraphArena.from_arrays(corpus["x"], corpus["edge_index"], corpus["y"], corpus["noisy"], corpus["depth"],
                                   corpus["observable"], device=dev)
    torch.manual_seed(0)
    model = ExpValCircuitGraphModelA(100, 22, 10).to(dev)
    trainer = Trainer(model, lr=1e-3, distributed=distributed)

    # weak scaling: every rank draws its own `batch` circuits per step from the (replicated) corpus
    rng = np.random.RandomState(1000 + rank)
    draw = lambda: rng.randint(0, n_graphs, size=args.batch)

    # Batches differ in node count (circuits vary 13x), so torch's caching allocator keeps growing -- each growth is a
    # hipMalloc that drains the queue -- until it has seen the largest batch.  Show it that batch once, untimed.
    sizes = np.asarray([x.shape[0] for x in corpus["x"]])
    trainer.step(arena.batch(np.argsort(sizes)[::-1][np.arange(args.batch) % max(1, min(args.batch // 4, n_graphs))]))
    for _ in range(args.warmup):
        trainer.step(arena.batch(draw()))
    if distributed:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = None
    for _ in range(args.steps):
        loss = trainer.step(arena.batch(draw()))
    if distributed:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        total = args.batch * world * args.steps
        fixed = arena.batch(fixed_ids(n_graphs, args.batch))
        line = {
            "metric": "circuits/sec (GNN train step), 100q TFIM Trotter",
            "value": round(total / elapsed, 2), "unit": "circuits/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg4: 100-qubit TFIM Trotter steps 1-10 x %d J values, GNN family A "
                                   "(GCNx3 || Chebx2 || SAGEx2, hidden 10, F=22), full train step" % args.n_j,
                       "circuits_per_step_per_gpu": args.batch, "corpus_circuits": n_graphs,
                       "mean_nodes_per_circuit": round(arena.num_nodes / n_graphs, 1), "parallelism": f"dp{world}"},
            "final_loss": round(float(loss.item()), 6),
            "train_mae_synthetic": dict(mae_leg(model, fixed), after_steps=args.warmup + args.steps + 1),
            "roofline": roofline_leg(fixed),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_leg(corpus, np.arange(n_graphs), 100)
            line["parity"] = parity_leg(model, arena, corpus, 100)   # the oracle as the checker, outside the timed region
        print(json.dumps(line), flush=True)
    if distributed:
        torch.distributed.barrier()  # rank 0 is still in its roofline leg: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
