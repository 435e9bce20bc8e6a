"""The reference's own model in the reference's own training regime, every step a replayed hipGraph capture.

    python examples/train_family_b.py --qubits 4 --epochs 10

Model: ``ExpValCircuitGraphModel`` (docs/tutorials/gnn.py:70-122: TransformerConv x2, ASAPooling x2, mean pool, head) -- the
architecture of all 42 GNN checkpoints the reference ships.  Loop: ``BucketedTrainer.fit`` = the reference's loop
(docs/tutorials/__ml_models.py:100-187: shuffled epochs of 32 circuits, Adam(1e-3), the summed validation loss into
ReduceLROnPlateau) with the whole step -- device batch assembly, forward, loss, backward, Adam -- captured once per size-stable
bucket and replayed: ~0.4 ms per step of 32 four-qubit circuits on one MI355X against ~2 ms when the same
launches are enqueued from Python one by one.  A reference checkpoint (``--checkpoint gnn1.pth``) loads with ``strict=True``.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ml-qem_amd")]

from blackwater.data.arena import GraphArena                      # noqa: E402
from blackwater.data.synthetic import TfimCorpus                  # noqa: E402
from blackwater.metrics.improvement_factor import mitigation_report  # noqa: E402
from blackwater.nn import ExpValCircuitGraphModel, family_b_from_state_dict   # noqa: E402
from blackwater.train import BucketedTrainer                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qubits", type=int, default=4)
    ap.add_argument("--steps", type=int, default=15, help="Trotter step counts 0..steps-1")
    ap.add_argument("--n-j", type=int, default=40, help="J values per step count")
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--checkpoint", default=None, help="a reference .pth (e.g. tests/golden/ckpt/gnn1.pth) to start from")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)

    # 1. a synthetic corpus in the reference's wire format, encoded once and kept on the device (filler graph: bucket padding)
    h = TfimCorpus(args.qubits, list(range(args.steps)), args.n_j, seed=0, two_q="cx", exp_value_size=4).host_graphs()
    arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"],
                                   device=dev, filler_nodes=1024)
    ids = np.random.RandomState(0).permutation(len(arena))
    n_val = max(len(ids) // 5, 1)
    val_ids, train_ids = ids[:n_val], ids[n_val:]

    # 2. the model: fresh, or one of the reference's checkpoints
    torch.manual_seed(0)
    if args.checkpoint:
        model = family_b_from_state_dict(torch.load(args.checkpoint, map_location="cpu", weights_only=True)).to(dev)
    else:
        model = ExpValCircuitGraphModel(num_node_features=arena.x.shape[1], hidden_channels=15, exp_value_size=4).to(dev)

    # 3. the reference's loop, replayed
    trainer = BucketedTrainer(model, arena, lr=1e-3, graphs=True, node_quantum=1024, edge_quantum=4096)
    t0 = time.perf_counter()
    hist = trainer.fit(arena, train_ids, val_ids, epochs=args.epochs, batch_size=args.batch,
                       log=lambda e, hh: print(f"epoch {e}: " + ", ".join(f"{k} {v[-1]:.5f}" for k, v in hh.items() if v)))
    torch.cuda.synchronize()
    steps = args.epochs * (-(-len(train_ids) // args.batch))
    print(f"{steps} steps in {time.perf_counter() - t0:.2f} s ({len(trainer._entries)} buckets captured)")

    # 4. the reference's evaluation cell: noisy vs mitigated against the ideal values
    pred = trainer.predict(arena, val_ids).cpu().numpy()
    rep = mitigation_report(arena.y[val_ids].cpu().numpy()[:, 0], arena.noisy[val_ids].cpu().numpy()[:, 0], pred)
    print({k: round(v, 5) for k, v in rep.items() if k in ("RMSE_noisy", "RMSE_mitigated", "MAE_noisy", "MAE_mitigated")})
    trainer.save(os.path.join("gpurun_out", "family_b_example.pth"), hist)      # the reference's .pth + .pk


if __name__ == "__main__":
    main()
