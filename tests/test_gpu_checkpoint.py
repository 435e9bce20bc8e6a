"""Checkpoint round trip on the device: a model TRAINED here is written by ``Trainer.save`` in the reference's form
(docs/tutorials/__ml_models.py:196-205: ``model.pth`` + ``{'train_losses','val_losses'}.pk``), reloads ``strict=True`` into the
CPU oracle's ``FamilyB`` / ``MLP1`` (the restatement of the reference's modules, oracle/models.py) and reproduces the device's
predictions within 1e-5; and a fresh device trainer that loads the files continues from the same parameters."""
import pickle

import numpy as np
import pytest
import torch

from helpers import g1_graph
from test_gpu_arena import _arena

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_family_b_trained_on_the_device_reloads_into_the_oracle(g1, tmp_path):
    from blackwater.data.arena import GraphArena
    from blackwater.nn import ExpValCircuitGraphModel
    from blackwater.train import Trainer
    from oracle.models import FamilyB

    count = 96
    xs, eis = [], []
    for i in range(count):
        x, ei, _ = g1_graph(g1, i)
        loops = np.arange(x.shape[0])
        xs.append(x.astype(np.float32))
        eis.append(np.concatenate([ei, np.stack([loops, loops])], axis=1))
    y = g1["ideal"][:count].astype(np.float32)[:, None, :]
    noisy = g1["noisy"][:count].astype(np.float32)[:, None, :]
    depth = g1["depth"][:count].astype(np.float32)[:, None]
    arena = GraphArena.from_arrays(xs, eis, y, noisy, depth, np.zeros((count, 1, 1), np.float32), device=DEV)
    torch.manual_seed(0)
    model = ExpValCircuitGraphModel(22, 15, 4).to(DEV)
    tr = Trainer(model, lr=1e-3)
    hist = tr.fit(arena, np.arange(64), np.arange(64, 96), epochs=3, batch_size=32, seed=0)
    assert len(hist["train_losses"]) == 2          # the reference drops epoch 0 from its curves (__ml_models.py:182)
    path = tr.save(str(tmp_path / "gnn1.pth"))
    with open(str(tmp_path / "gnn1.pk"), "rb") as fh:
        curves = pickle.load(fh)
    assert curves == {"train_losses": hist["train_losses"], "val_losses": hist["val_losses"]}
    ref = FamilyB(22, 15, 4).double().eval()
    res = ref.load_state_dict({k: v.double() for k, v in torch.load(path, weights_only=True).items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.eval()
    worst = 0.0
    with torch.no_grad():
        for i in range(64, 96):
            x, ei, _ = g1_graph(g1, i)      # inference as the decorator runs it: one circuit, no self-loops (ngem/estimator.py:75-82)
            args = (torch.tensor(g1["noisy"][i], dtype=torch.float32).view(1, 1, -1), None, torch.tensor([[float(g1["depth"][i])]]),
                    torch.tensor(x, dtype=torch.float32), torch.tensor(ei, dtype=torch.long), None)
            got = model(*[a.to(DEV) if a is not None else None for a in args]).cpu().double()
            want = ref(*[(a.double() if a.is_floating_point() else a) if a is not None else None for a in args])
            worst = max(worst, (got - want).abs().max().item())
    assert worst < 1e-5, worst
    # a fresh trainer continues from the files
    torch.manual_seed(1)
    other = Trainer(ExpValCircuitGraphModel(22, 15, 4).to(DEV), lr=1e-3)
    assert other.load(path) == curves
    assert torch.equal(other.flat_param.detach(), tr.flat_param.detach())


def test_mlp1_trained_on_the_device_reloads_into_the_oracle(tmp_path):
    from blackwater.nn.mlp import MLP1
    from blackwater.train import Trainer
    from oracle.models import MLP1 as OracleMLP1

    class _Rows:
        def __init__(self, x, y):
            self.x, self.y = x, y

        def model_args(self):
            return (self.x,)

    torch.manual_seed(0)
    x, y = torch.randn(512, 58), torch.randn(512, 4)
    for mfma in ("f32", "bf16"):
        model = MLP1(58, 64, 4).to(DEV)
        model.mfma = mfma
        tr = Trainer(model, lr=1e-3)
        for _ in range(10):
            tr.step(_Rows(x.to(DEV), y.to(DEV)))
        path = tr.save(str(tmp_path / f"mlp1_{mfma}"), history={"train_losses": [1.0], "val_losses": [2.0]})
        assert path.endswith(".pth")
        ref = OracleMLP1(58, 64, 4).double()
        ref.load_state_dict({k: v.double() for k, v in torch.load(path, weights_only=True).items()}, strict=True)
        model.mfma = "f32"            # the saved PARAMETERS are fp32 in either mode: evaluate them exactly
        model.eval()
        with torch.no_grad():
            got = model(x.to(DEV)).cpu().double()
            want = ref(x.double())
        assert (got - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())
