"""bench.py's launcher logic that needs no GPU: ``--gpus N`` must never silently run fewer ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        if k not in env:
            e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, cwd=ROOT, capture_output=True,
                          text=True, timeout=600)


def test_more_ranks_than_gpus_is_an_error_not_a_single_rank_run():
    import torch

    have = torch.cuda.device_count()
    out = _run(["--gpus", str(have + 2)])
    assert out.returncode != 0
    assert f"--gpus {have + 2}" in out.stderr and "GPU" in out.stderr
    assert '{"metric"' not in out.stdout


def test_world_size_must_match_gpus():
    out = _run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr


def test_corpus_is_sharded_in_equal_node_balanced_parts():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
    import numpy as np

    from blackwater.data.synthetic import TfimCorpus
    from blackwater.train import DataParallelShard

    corpus = TfimCorpus(6, [1, 2, 3, 4], 10, two_q="ecr")
    shards = DataParallelShard.split(corpus.node_counts, 4)
    assert [len(s) for s in shards] == [10] * 4
    assert sorted(np.concatenate(shards).tolist()) == list(range(40))
    loads = [int(corpus.node_counts[s].sum()) for s in shards]
    assert max(loads) - min(loads) <= corpus.node_counts.max()
    for s in shards:
        assert (np.diff(s) > 0).all()          # ascending ids: what TfimCorpus.arena takes
    h = corpus.host_graphs(shards[1][:3])
    assert len(h["x"]) == 3 and h["observable"].shape == (3, 1, 25) and h["y"].shape == (3, 1)


def test_accuracy_fixture_and_pooled_split(golden_dir):
    """The training-to-accuracy set (tests/golden/ising_trainval.npz): 300 + 3 x 100 circuits of the reference's
    ising_init_from_qasm_no_readout folders, the recorded loss curves, and the seeded pooled split."""
    import numpy as np

    from blackwater.metrics.accuracy import load_trainval, pooled_split

    z = load_trainval(golden_dir)
    assert len(z["qasm"]) == 600 and z["split"].tolist().count(0) == 300
    assert [int((z["split"] == k).sum()) for k in (1, 2, 3)] == [100, 100, 100]
    assert z["x"].shape[1] == 22 and z["node_ptr"][-1] == z["x"].shape[0] == 25837
    assert len(z["ref_curves"]["gnn1"]["val_losses"]) == 99            # epochs 1-99, as the reference records them
    assert round(z["ref_curves"]["gnn1"]["val_losses"][0], 4) == 0.0808 and round(z["ref_curves"]["gnn1"]["val_losses"][-1], 5) == 0.00687
    tr, va = pooled_split(z["split"], seed=0)
    assert len(tr) == 510 and len(va) == 90 and not set(tr.tolist()) & set(va.tolist())
    assert set(np.flatnonzero(z["split"] == 0).tolist()) <= set(tr.tolist())
    tr2, va2 = pooled_split(z["split"], seed=0)
    assert np.array_equal(tr, tr2) and np.array_equal(va, va2)
    # every circuit text parses and has as many ops as its graph has nodes
    from blackwater.data.circuit import Circuit

    for i in (0, 299, 300, 450, 599):
        assert len(Circuit.from_qasm_str(z["qasm"][i]).ops) == z["node_ptr"][i + 1] - z["node_ptr"][i]
