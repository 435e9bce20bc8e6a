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
