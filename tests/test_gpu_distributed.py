"""The RCCL leg of the data-parallel path, as far as one GPU can exercise it: a world-size-1 ``nccl`` process group
carries the flat-gradient all-reduce of ``Trainer`` (the multi-rank logic is covered on CPU with gloo in
tests/test_distributed_cpu.py; the driver runs 2/4/8 ranks)."""
import socket

import numpy as np
import pytest
import torch

from helpers import g1_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_trainer_step_over_rccl_world1(g1):
    from blackwater.data.arena import GraphArena
    from blackwater.nn import ExpValCircuitGraphModelA
    from blackwater.train import Trainer
    from helpers import g1_batch

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                         device_id=torch.device(DEV))
    try:
        xs, eis = [], []
        for i in range(64):
            x, ei, _ = g1_graph(g1, i)
            loops = np.arange(x.shape[0])
            xs.append(x.astype(np.float32))
            eis.append(np.concatenate([ei, np.stack([loops, loops])], axis=1))
        host = g1_batch(g1, range(64))
        arena = GraphArena.from_arrays(xs, eis, host["y"].numpy(), host["noisy"].numpy(), host["depth"].numpy(),
                                       host["observable"].numpy(), device=DEV)
        results = []
        for distributed in (True, False):
            torch.manual_seed(0)
            model = ExpValCircuitGraphModelA(5, 22, 10).to(DEV)
            model.eval()
            model.train = lambda *a, **k: model  # dropout off: the two runs must agree bit for bit
            tr = Trainer(model, lr=1e-3, distributed=distributed)
            assert tr.distributed == distributed
            for step in range(3):
                loss = tr.step(arena.batch(list(range(step * 8, step * 8 + 32))))
            results.append((loss.item(), tr.flat_param.detach().clone()))
        assert results[0][0] == results[1][0]
        assert torch.equal(results[0][1], results[1][1])
    finally:
        torch.distributed.destroy_process_group()


def test_all_reduce_captured_inside_the_step_graph_equals_the_eager_step(g1):
    """``BucketedTrainer(capture_collective=True)`` captures assembly, forward, backward, the flat-gradient all-reduce
    and Adam as ONE hipGraph (a world-size-1 ``nccl`` group exercises RCCL's capture path on the one GPU); replays of it equal the eager
    bucketed step and the two-graph form (eager collective between two replays) bit for bit, dropout on.  Should this RCCL refuse the
    capture, the trainer must say so and still produce the same numbers from the two-graph form."""
    from blackwater.data.arena import GraphArena
    from blackwater.native import ops
    from blackwater.nn import ExpValCircuitGraphModelA
    from blackwater.train import BucketedTrainer
    from helpers import g1_batch

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        xs, eis = [], []
        for i in range(64):
            x, ei, _ = g1_graph(g1, i)
            loops = np.arange(x.shape[0])
            xs.append(x.astype(np.float32))
            eis.append(np.concatenate([ei, np.stack([loops, loops])], axis=1))
        host = g1_batch(g1, range(64))
        arena = GraphArena.from_arrays(xs, eis, host["y"].numpy(), host["noisy"].numpy(), host["depth"].numpy(), host["observable"].numpy(),
                                       device=DEV, filler_nodes=1024)
        plans = [list(range(k, k + 32)) for k in (0, 8, 16, 24, 32, 0, 8, 16)]
        runs = {}
        for mode in ("eager", "two_graphs", "one_graph"):
            torch.manual_seed(0)
            model = ExpValCircuitGraphModelA(5, 22, 10).to(DEV)
            tr = BucketedTrainer(model, arena, lr=1e-3, graphs=mode != "eager", node_quantum=1024, distributed=True,
                                 capture_collective=mode == "one_graph")
            assert tr.distributed
            losses = [float(tr.step_ids(ids).item()) for ids in plans]
            runs[mode] = (losses, tr.flat_param.detach().clone())
            if mode == "one_graph":
                assert tr.collective_in_graph is not None
                assert tr.collective_in_graph or tr.collective_capture_error        # the form that ran is on record
                one_graph_form = tr.collective_in_graph
            ops.set_seed_counter(None)
        for mode in ("two_graphs", "one_graph"):
            assert runs[mode][0] == runs["eager"][0], mode
            assert torch.equal(runs[mode][1], runs["eager"][1]), mode
        print("collective captured inside the step graph:", one_graph_form)
    finally:
        torch.distributed.destroy_process_group()


def test_bench_two_ranks_on_one_gpu_rehearsal():
    """Plain ``python bench.py --gpus 2`` (the shape of the driver's command): bench.py starts the two ranks itself before
    touching the GPU, each rank builds ITS shard of the corpus, and rank 0 prints the one line.  The two ranks share this
    box's single GPU over gloo (MLQEM_BENCH_BACKEND) -- the driver's real runs use one GPU per rank over RCCL, which one
    GPU cannot host."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MLQEM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1",
               MLQEM_BENCH_FULL_RECORD=os.path.join("gpurun_out", "bench_full_rehearsal.json"))     # not over a real run's record
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1                                    # rank 0 only
    # the headline record is the LAST stdout line, parses, and is small enough for the driver's stdout window
    assert out.stdout.strip().splitlines()[-1] == lines[0] and len(lines[0]) < 8192
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["value"] > 0
    cfg = rec["config"]
    assert cfg["parallelism"] == "dp2" and cfg["ranks_joined"] == 2 and "cpu_baseline" not in rec and rec["roofline"]["frac"] > 0
    # the step is replayed from two captured graphs with the gradient all-reduce enqueued between them
    assert cfg["step_mode"].startswith("hipgraph replay")
    # corpus = 8 x batch x ranks circuits (rounded up to whole J grids), each rank holds half of it
    assert cfg["corpus_circuits"] >= 8 * 32 * 2 and cfg["corpus_circuits_per_gpu"] == cfg["corpus_circuits"] // 2


def test_family_b_two_ranks_on_one_gpu_equal_one_process_on_the_whole_batch(tmp_path):
    """The reference's model (Family B, hidden 32: a flat gradient buffer of 50 314 floats) trained by two ranks that share
    this box's GPU over gloo, each on its half of every batch, against ONE process on the whole batch: replicas start from
    rank 0's parameters (they are built from different seeds), stay in lock-step, and all-reduce to the single process's
    gradient up to fp32 rounding (pre-scaled by 1/world before a SUM all-reduce).  Parameters are compared through the
    gradient and the losses, not element by element: Adam turns a gradient that is zero in exact arithmetic (the key bias of
    an attention layer: softmax is shift-invariant) into +-lr steps whose sign is rounding noise.  tests/dp_family_b_worker.py
    is one rank."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(here, "dp_family_b_worker.py")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    outs = [str(tmp_path / f"rank{r}.pt") for r in (0, 1)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), outs[r]], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=600)[0])
    finally:
        for p in procs:          # exactly the two children this test started
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(log[-1500:] for log in logs)
    single = str(tmp_path / "single.pt")
    one = subprocess.run([sys.executable, worker, "0", "1", "0", single], env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-1500:]
    r0, r1, ref = (torch.load(p, weights_only=False) for p in (outs[0], outs[1], single))
    assert r0["floats"] == 50314
    assert torch.equal(r0["param"], r1["param"])                          # lock-step, bit for bit
    assert r0["losses"] != r1["losses"]                                   # ... on different halves
    assert torch.equal(r0["grad0"], r1["grad0"])
    gap = (ref["grad0"] - r0["grad0"]).norm().item() / ref["grad0"].norm().item()
    assert gap <= 1e-5, gap                                               # mean of the two half-batch gradients = the batch's
    # all but the parameters whose gradient is rounding noise (2 % of them here: key biases, dead units) end where the single
    # process's do
    assert ((ref["param"] - r0["param"]).abs() > 1e-5).float().mean().item() < 0.05
    mean = [(a + b) / 2 for a, b in zip(r0["losses"], r1["losses"])]     # equal halves: the whole batch's MSE is their mean
    assert np.allclose(mean, ref["losses"], rtol=1e-4, atol=1e-7)
    assert ref["losses"][-1] < ref["losses"][0]
