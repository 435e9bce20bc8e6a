"""cfg4-size parity under ``-m gpu``: one synthetic 100-qubit TFIM-Trotter circuit per step count 1 / 5 / 10
(2 089 / 10 041 / 19 981 nodes), Family A (the bench model) AND Family B (the reference's gnn.py model), device
predictions against the fp64 oracle with the north_star tolerance 1e-5 -- and, next to it, how far the reference's own
fp32 CPU arithmetic (the oracle in fp32, at 1 and at 8 threads) lands from the same fp64 values.  The measured gaps are
written to ``gpurun_out/parity_cfg4.json`` (copied to ``profiles/`` by the builder) so the statement "on graphs this
large the fp32 CPU path is itself not reproducible to 1e-5" is a recorded measurement, not a sentence.

Model under test: docs/tutorials/01_ngem.ipynb cell [9] (Family A), docs/tutorials/gnn.py:70-122 (Family B).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5          # BASELINE.json north_star: "within 1e-5 fp32 on identical inputs"
STEPS = (1, 5, 10)


def _record(key, value):
    path = os.path.join(ROOT, "gpurun_out", "parity_cfg4.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        with open(path) as fh:
            data = json.load(fh)
    data[key] = value
    with open(path, "w") as fh:
        json.dump(data, fh, indent=1, sort_keys=True)


def _corpus(exp_value_size):
    from blackwater.data.synthetic import tfim_corpus

    return tfim_corpus(100, list(STEPS), 1, seed=42, two_q="ecr", exp_value_size=exp_value_size)


def _oracle_per_graph(ref, corpus, dtype, noisy_of):
    outs = []
    with torch.no_grad():
        for g in range(len(corpus["x"])):
            x = torch.from_numpy(corpus["x"][g]).to(dtype)
            t = lambda k: torch.from_numpy(corpus[k][g:g + 1]).to(dtype)
            outs.append(ref(noisy_of(g).to(dtype), t("observable"), t("depth"), x,
                            torch.from_numpy(corpus["edge_index"][g]), torch.zeros(x.shape[0], dtype=torch.long)).double())
    return torch.cat(outs)


def _cpu_f32_at(threads, make_ref, corpus, noisy_of):
    was = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        return _oracle_per_graph(make_ref(torch.float32), corpus, torch.float32, noisy_of)
    finally:
        torch.set_num_threads(was)


def _gaps(got, f64, f32_1, f32_8):
    gap = lambda a, b: float((a - b).abs().max())
    return {"gpu_vs_f64": gap(got, f64), "cpu_f32_1thread_vs_f64": gap(f32_1, f64), "cpu_f32_8threads_vs_f64": gap(f32_8, f64),
            "cpu_f32_1thread_vs_8threads": gap(f32_1, f32_8), "gpu_vs_cpu_f32_1thread": gap(got, f32_1),
            "prediction_scale": float(f64.abs().max()), "nodes": None}


def _trained_family_a(corpus, steps):
    """The bench's regime: Family A after a few dozen Adam steps from seed 0 on the same kind of batch."""
    from blackwater.data.arena import GraphArena
    from blackwater.nn import ExpValCircuitGraphModelA
    from blackwater.train import Trainer

    arena = GraphArena.from_arrays(corpus["x"], corpus["edge_index"], corpus["y"], corpus["noisy"], corpus["depth"],
                                   corpus["observable"], device=DEV)
    torch.manual_seed(0)
    model = ExpValCircuitGraphModelA(100, 22, 10).to(DEV)
    trainer = Trainer(model, lr=1e-3)
    ids = np.arange(len(arena))
    for _ in range(steps):
        trainer.step(arena.batch(ids))
    return model, arena


@pytest.mark.parametrize("train_steps", [0, 30])
def test_family_a_100q_predictions_within_1e5_of_fp64_oracle(train_steps):
    from oracle.models import FamilyA

    corpus = _corpus(1)
    model, arena = _trained_family_a(corpus, train_steps)
    model.eval()
    with torch.no_grad():
        got = model(*arena.batch(np.arange(len(arena))).model_args()).double().cpu()
        one_by_one = torch.cat([model(*arena.batch([g]).model_args()).double().cpu() for g in range(len(arena))])
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}

    def make_ref(dtype):
        ref = FamilyA(100, 22, 10).eval()
        ref.load_state_dict(state)
        return ref.to(dtype)

    noisy_of = lambda g: torch.from_numpy(corpus["noisy"][g:g + 1])
    f64 = _oracle_per_graph(make_ref(torch.float64), corpus, torch.float64, noisy_of)
    f32_1 = _cpu_f32_at(1, make_ref, corpus, noisy_of)
    f32_8 = _cpu_f32_at(8, make_ref, corpus, noisy_of)
    rec = _gaps(got, f64, f32_1, f32_8)
    rec["nodes"] = [int(x.shape[0]) for x in corpus["x"]]
    rec["batched_vs_one_by_one"] = float((got - one_by_one).abs().max())
    _record(f"family_a_after_{train_steps}_steps", rec)
    assert rec["gpu_vs_f64"] < TOL, rec
    # a graph's prediction does not depend on what else is in the batch (rows of other graphs never mix)
    assert rec["batched_vs_one_by_one"] < TOL, rec
    # the device result is at least as close to the exact value as the reference's fp32 CPU arithmetic is (up to one
    # fp32 ulp of the prediction: both are fp32 computations of the same expression)
    ulp = float(np.spacing(np.float32(rec["prediction_scale"])))
    assert rec["gpu_vs_f64"] <= rec["cpu_f32_1thread_vs_f64"] + 2 * ulp, rec


def test_family_b_100q_predictions_within_1e5_of_fp64_oracle():
    """Family B (TransformerConv / ASAPooling x2 / mean pool / head) on the same three circuits, exp_value_size 4.
    The pooling's top-k is a discrete choice: the test first checks that the device and the fp64 oracle keep the same
    clusters at both poolings -- up to near-ties at the k-th place, which fp32 may legitimately resolve the other way
    (circuit graphs hold many nodes with almost equal neighbourhoods): at most 0.2 % of a pooling's clusters may
    differ, and the prediction must not notice -- then the 1e-5 tolerance on the predictions (absolute, although the predictions are ~20 in magnitude here because the raw
    circuit depth, up to 331, enters the head un-normalised, gnn.py:118-120)."""
    from blackwater.native.structure import GraphStructure
    from blackwater.nn import ExpValCircuitGraphModel
    from oracle.models import FamilyB

    corpus = _corpus(4)
    torch.manual_seed(0)
    model = ExpValCircuitGraphModel(22, 15).to(DEV).eval()
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}

    def make_ref(dtype):
        ref = FamilyB(22, 15).eval()
        ref.load_state_dict(state, strict=True)
        return ref.to(dtype)

    noisy_of = lambda g: torch.from_numpy(corpus["noisy"][g:g + 1, None, :])
    got, same_clusters = [], []
    ref64 = make_ref(torch.float64)
    with torch.no_grad():
        for g in range(len(corpus["x"])):
            x = torch.from_numpy(corpus["x"][g]).to(DEV)
            ei = torch.from_numpy(corpus["edge_index"][g]).to(DEV)
            s = GraphStructure.from_edge_index(ei, x.shape[0])
            got.append(model(noisy_of(g).to(DEV), None, torch.from_numpy(corpus["depth"][g:g + 1]).to(DEV), x, s, None)
                       .double().cpu())
            # discrete part: the kept clusters of both poolings
            h = model.transformer1(x, s)
            h, s1, perm1 = model.pooling1(h, s)
            _, _, perm2 = model.pooling2(model.transformer2(h, s1), s1)
            xr, eir = torch.from_numpy(corpus["x"][g]).double(), torch.from_numpy(corpus["edge_index"][g])
            hr = ref64.transformer1(xr, eir)
            hr, ei1, _, _, p1 = ref64.pooling1(hr, eir)
            _, _, _, _, p2 = ref64.pooling2(ref64.transformer2(hr, ei1), ei1)
            if sorted(perm1.cpu().tolist()) == sorted(p1.tolist()):
                swapped = len(set(perm2.cpu().tolist()) ^ set(p2.tolist())) // 2
                same_clusters.append(swapped / max(1, len(p2)))
            else:       # the second pooling then runs on other clusters: only the first can be compared
                same_clusters.append(len(set(perm1.cpu().tolist()) ^ set(p1.tolist())) // 2 / max(1, len(p1)))
    got = torch.cat(got)
    f64 = _oracle_per_graph(ref64, corpus, torch.float64, noisy_of)
    f32_1 = _cpu_f32_at(1, make_ref, corpus, noisy_of)
    f32_8 = _cpu_f32_at(8, make_ref, corpus, noisy_of)
    rec = _gaps(got, f64, f32_1, f32_8)
    rec["nodes"] = [int(x.shape[0]) for x in corpus["x"]]
    rec["fraction_of_clusters_kept_differently_from_fp64_oracle"] = same_clusters
    rec["relative_gpu_vs_f64"] = rec["gpu_vs_f64"] / max(1.0, rec["prediction_scale"])
    _record("family_b_seed0", rec)
    assert all(f <= 2e-3 for f in same_clusters), rec
    assert rec["gpu_vs_f64"] < TOL, rec
