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
    env = dict(os.environ, MLQEM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1                                    # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["value"] > 0
    cfg = rec["config"]
    assert cfg["parallelism"] == "dp2" and cfg["ranks_joined"] == 2 and "cpu_baseline" not in rec and rec["roofline"]["frac"] > 0
    # the step is replayed from two captured graphs with the gradient all-reduce enqueued between them
    assert cfg["step_mode"].startswith("hipgraph replay")
    # corpus = 8 x batch x ranks circuits (rounded up to whole J grids), each rank holds half of it
    assert cfg["corpus_circuits"] >= 8 * 32 * 2 and cfg["corpus_circuits_per_gpu"] == cfg["corpus_circuits"] // 2
