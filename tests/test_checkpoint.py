"""``Trainer.save`` writes the reference's artefacts (docs/tutorials/__ml_models.py:196-205): ``model.pth`` = the state dict,
``model.pk`` = ``{'train_losses', 'val_losses'}`` pickled -- the form of the 63 checkpoints and loss curves the reference ships
(SURVEY.md section 2.3 / Appendix D).  CPU part: file format and round trip through a plain torch module."""
import pickle

import torch

from blackwater.train import Trainer
from oracle.models import MLP1


class _Rows:
    def __init__(self, x, y):
        self.x, self.y = x, y

    def model_args(self):
        return (self.x,)


def test_save_writes_state_dict_and_loss_curves_in_the_reference_form(tmp_path):
    torch.manual_seed(0)
    model = MLP1(58, 64, 4)
    tr = Trainer(model, lr=1e-3)
    x, y = torch.randn(64, 58), torch.randn(64, 4)
    losses = [float(tr.step(_Rows(x, y))) for _ in range(5)]
    assert losses[-1] < losses[0]
    tr.history = {"train_losses": losses, "val_losses": [l * 2 for l in losses]}
    path = tr.save(str(tmp_path / "run" / "mlp1_smaller_2.pth"))
    assert path.endswith("mlp1_smaller_2.pth")
    state = torch.load(path, map_location="cpu", weights_only=True)          # plain tensors: loads with weights_only
    assert list(state) == ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]   # the reference's key names, in order
    assert all(v.is_contiguous() and v.storage_offset() == 0 for v in state.values())   # not views into the flat buffer
    with open(str(tmp_path / "run" / "mlp1_smaller_2.pk"), "rb") as fh:
        curves = pickle.load(fh)
    assert set(curves) == {"train_losses", "val_losses"} and curves["train_losses"] == losses
    fresh = MLP1(58, 64, 4)
    fresh.load_state_dict(state, strict=True)
    assert torch.equal(fresh(x), model(x))
    # and back into a trainer: parameters land in the flat buffer the optimizer steps
    other = Trainer(MLP1(58, 64, 4), lr=1e-3)
    hist = other.load(path)
    assert hist["val_losses"] == curves["val_losses"]
    assert torch.equal(other.model(x), model(x))
    assert torch.equal(other.flat_param.detach()[:58 * 64].view(64, 58), model.fc1.weight.detach())


def test_g6_every_reference_checkpoint_loads_strictly_into_the_product_modules(golden_dir):
    """SURVEY section 8c, golden G6, on the BUILD's modules (tests/test_oracle_goldens.py checks the oracle's classes): the shapes
    of all 63 checkpoints the reference ships (42 GNN + 21 MLP; tests/golden/ckpt_manifest.json) strict-load into ``blackwater.nn``,
    and the seven checkpoints committed in full load with their real weights and keep them bit for bit."""
    import json
    import os

    from blackwater.nn import family_b_from_state_dict
    from blackwater.nn.mlp import MLP1 as PMLP1, MLP2 as PMLP2, MLP3 as PMLP3

    def product_module(shapes, sd):
        if "transformer1.lin_key.weight" in shapes:
            return family_b_from_state_dict(sd)                    # strict=True inside
        i, h = shapes["fc1.weight"][1], shapes["fc1.weight"][0]
        if "fc4.weight" in shapes:
            m = PMLP3(i, h, shapes["fc4.weight"][0])
        elif "fc3.weight" in shapes:
            m = PMLP2(i, h, shapes["fc3.weight"][0])
        else:
            m = PMLP1(i, h, shapes["fc2.weight"][0])
        m.load_state_dict(sd, strict=True)
        return m

    with open(os.path.join(golden_dir, "ckpt_manifest.json")) as fh:
        manifest = json.load(fh)
    assert len(manifest) == 63
    n_gnn = n_mlp = 0
    for name, shapes in manifest.items():
        sd = {k: torch.zeros(s) for k, s in shapes.items()}
        m = product_module(shapes, sd)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(s) for k, s in shapes.items()}, name
        n_gnn += "transformer1.lin_key.weight" in shapes
        n_mlp += "fc1.weight" in shapes
    assert (n_gnn, n_mlp) == (42, 21)
    ckpts = sorted(f for f in os.listdir(os.path.join(golden_dir, "ckpt")) if f.endswith(".pth"))
    assert len(ckpts) == 7 and "iskandar.pth" in ckpts
    for f in ckpts:
        sd = torch.load(os.path.join(golden_dir, "ckpt", f), weights_only=True)
        m = product_module({k: list(v.shape) for k, v in sd.items()}, sd)
        back = m.state_dict()
        assert list(back) == list(sd) and all(torch.equal(back[k], sd[k]) for k in sd), f
