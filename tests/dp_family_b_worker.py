"""One rank of the data-parallel Family B rehearsal (tests/test_gpu_distributed.py starts two of these, both on cuda:0, over
gloo): the reference's model (docs/tutorials/gnn.py:70-116) with a flat gradient buffer of 50 314 floats (hidden 32: the attention kernels serve up to 32 channels per head), each rank stepping on
ITS half of every batch through ``Trainer(distributed=True)``.  ``world == 1`` is the single-process run on the whole batch
that the two ranks must reproduce.  Writes the final flat parameters and the losses to ``out``."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "ml-qem_amd")]


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from helpers import g1_graph

    from blackwater.data.arena import GraphArena
    from blackwater.nn import ExpValCircuitGraphModel
    from blackwater.train import Trainer

    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda:0"
    z = dict(np.load(os.path.join(HERE, "golden", "g1_dataset.npz")))
    xs, eis = [], []
    for i in range(128):
        x, ei, _ = g1_graph(z, i)
        loops = np.arange(x.shape[0])
        xs.append(x.astype(np.float32))
        eis.append(np.concatenate([ei, np.stack([loops, loops])], axis=1))
    arena = GraphArena.from_arrays(xs, eis, z["ideal"][:128, None, :].astype(np.float32), z["noisy"][:128, None, :].astype(np.float32),
                                   z["depth"][:128, None].astype(np.float32), np.zeros((128, 1, 1), np.float32), device=dev)
    torch.manual_seed(rank)                 # different seeds: the Trainer brings every replica to rank 0's parameters
    model = ExpValCircuitGraphModel(22, 32, 4).to(dev)
    model.eval()
    model.train = lambda *a, **k: model     # dropout off: masks are keyed by the rank, the comparison is with ONE process
    trainer = Trainer(model, lr=1e-3, distributed=world > 1)
    losses = []
    for step in range(4):
        ids = np.arange(step * 16, step * 16 + 64)
        mine = ids[rank::world]             # equal halves, so the mean of the two MSE gradients is the whole batch's
        losses.append(float(trainer.step(arena.batch(mine))))
        if step == 0:
            grad0 = trainer.flat_grad.detach().cpu().clone()     # after the all-reduce: the mean over the ranks
    torch.save({"param": trainer.flat_param.detach().cpu(), "losses": losses, "grad0": grad0, "floats": trainer.flat_param.numel(),
                "prescaled": bool(getattr(trainer, "_prescaled", False))}, out)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
