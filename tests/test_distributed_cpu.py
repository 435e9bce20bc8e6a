"""Data-parallel host logic on CPU: world_size-2 gloo runs of the flat-buffer gradient all-reduce and the
node-balanced sharding (the N>1 path of bench.py / Trainer; RCCL replaces gloo on the GPU box)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from blackwater.train import DataParallelShard, Trainer, flatten_parameters


class _ToyBatch:
    def __init__(self, x, y):
        self.x, self.y = x, y

    def model_args(self):
        return (self.x,)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)  # identical replicas
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2))
    trainer = Trainer(model, lr=1e-2, distributed=True)
    g = torch.Generator().manual_seed(100)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 2, generator=g)
    shard = slice(rank * 4, rank * 4 + 4)  # each rank sees its half
    for _ in range(3):
        trainer.step(_ToyBatch(X[shard], Y[shard]))
    torch.save(trainer.flat_param.detach().clone(), os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_flat_allreduce_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = (torch.load(tmp_path / f"rank{r}.pt") for r in (0, 1))
    assert torch.equal(p0, p1)  # replicas stay in lock-step
    # one process on the full batch takes the same steps (mean of the two half-batch MSE gradients)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2))
    trainer = Trainer(model, lr=1e-2, distributed=False)
    g = torch.Generator().manual_seed(100)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 2, generator=g)
    for _ in range(3):
        trainer.step(_ToyBatch(X, Y))
    assert torch.allclose(trainer.flat_param.detach(), p0, rtol=1e-5, atol=1e-6)


def test_flatten_parameters_keeps_views_and_state_dict():
    model = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 1))
    before = {k: v.clone() for k, v in model.state_dict().items()}
    flat, grad = flatten_parameters(model)
    assert flat.numel() == 3 * 4 + 4 + 4 + 1
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k])
    model(torch.ones(2, 3)).sum().backward()
    assert grad.abs().sum() > 0  # autograd accumulated straight into the flat gradient buffer
    with torch.no_grad():
        flat.add_(1.0)
    assert torch.equal(model[0].weight, before["0.weight"] + 1.0)  # parameters are views of the flat buffer


def test_node_balanced_sharding():
    rng = np.random.RandomState(0)
    counts = rng.choice([1584, 8999, 20711], size=97)  # 100-qubit graphs vary 13x in size
    shards = DataParallelShard.split(counts, 8)
    assert len({len(s) for s in shards}) == 1
    flat = np.concatenate(shards)
    assert len(set(flat.tolist())) == len(flat)  # disjoint
    loads = np.array([counts[s].sum() for s in shards], dtype=float)
    assert loads.max() / loads.mean() < 1.15


def test_trainer_step_equals_plain_adam_and_handles_unused_parameters():
    """Trainer.step files autograd's gradients into the flat buffer with one multi-tensor copy: same updates as a plain
    per-parameter Adam loop; a parameter the loss does not reach keeps a zero gradient slot; flat=False agrees too."""
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = torch.nn.Linear(6, 5), torch.nn.Linear(5, 2)
            self.unused = torch.nn.Parameter(torch.ones(3))

        def forward(self, x):
            return self.b(torch.relu(self.a(x)))

    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(16, 6, generator=g), torch.randn(16, 2, generator=g)
    finals = []
    for mode in ("flat", "per-parameter", "reference"):
        torch.manual_seed(1)
        net = Net()
        if mode == "reference":
            opt = torch.optim.Adam(net.parameters(), lr=1e-2)
            for _ in range(4):
                opt.zero_grad()
                torch.nn.functional.mse_loss(net(X), Y).backward()
                opt.step()
        else:
            trainer = Trainer(net, lr=1e-2, flat=mode == "flat")
            for _ in range(4):
                trainer.step(_ToyBatch(X, Y))
            if mode == "flat":
                assert net.unused.grad is not None and not net.unused.grad.any()
                for prm, slot in zip(trainer._params, trainer._grad_slots):            # .grad fields: views of the flat buffer
                    assert prm.grad.data_ptr() == slot.data_ptr()
        finals.append(torch.cat([p.detach().reshape(-1) for p in net.parameters()]))
    assert torch.allclose(finals[0], finals[2], rtol=1e-6, atol=1e-7)
    assert torch.allclose(finals[1], finals[2], rtol=1e-6, atol=1e-7)
    assert torch.equal(finals[0][:3], torch.ones(3))         # the unused parameter (first in parameters()) never moved


class _ToyArena:
    """arena.batch(ids) over in-memory rows (the Trainer.fit protocol)."""

    def __init__(self, X, Y):
        self.X, self.Y = X, Y

    def batch(self, ids):
        ids = torch.as_tensor(np.asarray(ids), dtype=torch.long)
        return _ToyBatch(self.X[ids], self.Y[ids])


def _fit_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(rank)          # DIFFERENT seeds: the Trainer must bring every replica to rank 0's parameters
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2))
    trainer = Trainer(model, lr=1e-2, distributed=True)
    torch.save(trainer.flat_param.detach().clone(), os.path.join(out_dir, f"init{rank}.pt"))
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(80, 6, generator=g), torch.randn(80, 2, generator=g)
    # round-robin shards of an ODD corpus: 33 vs 32 training graphs at batch 32 -> 2 vs 1 batches per epoch
    mine = np.arange(rank, 65, world)
    steps = []
    orig_step = trainer.step
    trainer.step = lambda b: (steps.append(1), orig_step(b))[1]
    hist = trainer.fit(_ToyArena(X, Y), mine, np.arange(70 + rank, 80, world), epochs=3, batch_size=32)
    torch.save({"steps": len(steps), "param": trainer.flat_param.detach().clone(), "hist": hist},
               os.path.join(out_dir, f"fit{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_fit_agrees_on_step_count_and_initial_parameters(tmp_path):
    """An odd corpus gives the two ranks different batch counts; every step ends in an all-reduce, so an unmatched
    step would hang the job (round-1 advisor finding).  Also: replicas built from different seeds start from rank 0's
    parameters."""
    port = _free_port()
    mp.spawn(_fit_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    i0, i1 = (torch.load(tmp_path / f"init{r}.pt") for r in (0, 1))
    assert torch.equal(i0, i1)
    torch.manual_seed(0)
    want = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2))
    assert torch.equal(i0, torch.cat([p.detach().reshape(-1) for p in want.parameters()]))
    f0, f1 = (torch.load(tmp_path / f"fit{r}.pt", weights_only=False) for r in (0, 1))
    assert f0["steps"] == f1["steps"] == 3                    # min(2, 1) batches x 3 epochs on BOTH ranks
    assert torch.equal(f0["param"], f1["param"])
    assert f0["hist"]["val_losses"] == f1["hist"]["val_losses"]
