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


def _canned_full_record():
    """A full bench record as the legs would return it, with every leg fatter than the real one (long notes, nested lists):
    what grew the r05 line to 22.8 KB and left BENCH_r05.json unparsed."""
    pad = "x" * 600
    return {
        "metric": "circuits/sec (GNN train step), 100q TFIM Trotter", "value": 166486.7, "unit": "circuits/s", "n_gpus": 1, "steps": 20,
        "warmup": 5, "ms_per_step": 6.151, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "cfg4: 100-qubit TFIM Trotter steps 1-10 x 820 J values, GNN family A " + pad, "circuits_per_step_per_gpu": 1024,
                   "corpus_circuits": 8200, "parallelism": "dp1", "nodes_per_step_per_gpu": 11299840, "step_mode": "hipgraph replay " + pad,
                   "sampling": pad, "backend": "nccl", "ranks_joined": 1, "host_ms_between_graph_replays_per_rank": [0.11, 0.12, 0.13],
                   "gradient_floats_all_reduced": 5625, "global_circuits_per_step": 1024, "collective": "eager all-reduce " + pad,
                   "rccl_version": "2.26.6"},
        "ms_per_step_percentiles": {"note": pad}, "ms_per_step_p50": 6.148, "final_loss": 1.05, "host_enqueue_ms_per_step": 0.231,
        "roofline": {"bound": "hbm", "kernel": "csr_aggregate_ell_kernel<4,false,2,false,8> " + pad, "achieved": 6761.1, "peak": 8000.0,
                     "unit": "GB/s", "frac": 0.8451, "traffic": 1340704992, "bytes_per_launch": 1658229556, "us_per_launch": 245.26,
                     "variants": [{"launch": pad}] * 5, "in_step": [{"launch": pad}] * 9, "note": pad,
                     "in_step_all_aggregations": {"bytes_per_step": 15405725504, "us_per_step": 3437.9, "frac": 0.5602},
                     "step_dominant": {"kernel": "csr_aggregate_ell_kernel<4,false,2,true,7,true>", "launches_per_step": 3,
                                       "bytes_per_launch": 1793732212, "us_per_launch": 410.2, "achieved": 4372.8, "frac": 0.5466},
                     "hbm_frac": 0.6833, "measured_copy_GBps": 4712.6, "measured_add_GBps": 6178.8},
        "cpu_baseline": {"value": 79.56, "unit": "circuits/s", "cores": 1, "kind": "port", "sample": "batch 32 " + pad,
                         "batch32_ms_per_step": 402.2, "all_cores": {"note": pad}, "large_batch": {"note": pad}},
        "parity": {"circuits": 10, "tolerance": 1e-5, "exp_val_mae_vs_cpu_f64": 4.4e-7, "max_abs_err_vs_cpu_f64": 1.5e-6,
                   "within_tolerance_of_exact": True, "max_abs_err_vs_cpu_f32": 1.43e-5, "cpu_f32_max_abs_err_vs_cpu_f64": 1.5e-5, "note": pad},
        "accuracy": {"data": pad * 3}, "inference": {"note": pad * 8},
        "family_b": {"batch32_stratified_hipgraph": {"circuits_per_s": 68283.3}, "batch32_shuffled_hipgraph": {"circuits_per_s": 51000.0},
                     "cfg4_100q": {"best_circuits_per_s": 16566.3, "ms_per_step": 5.77, "note": pad * 6}},
        "small_batch": {"hipgraph": {"circuits_per_s": 115865.4}},
        "mlp_head": {"mlp3_170_125_1_f32": {"ms_per_step": 1.27}, "mlp3_170_125_1_bf16": {"ms_per_step": 0.717}, "note": pad * 4},
        "configs": {"cfg1_mlp1_169": {"circuits_per_s": 9715953.1}, "cfg3_random_20q": {"family_a": {"circuits_per_s": 1517531.3},
                                                                                       "family_b": {"circuits_per_s": 335139.8}},
                    "cfg5_mixed": {"family_a_f32": {"circuits_per_s": 680101.8}, "family_b_mlp3_head_bf16": {"circuits_per_s": 14461.3}}},
    }


def test_headline_record_is_small_and_carries_the_contract_keys(capsys, tmp_path, monkeypatch):
    """The LAST stdout line of bench.py is what the driver parses.  r05's grew to 22.8 KB and came back `parsed: null`: the
    headline record is built from the legs' output, stays under 8 KB whatever the legs return, and carries the contract's keys,
    `roofline` (re-based on the step-dominant kernel in the step) and `cpu_baseline`."""
    import json

    sys.path[:0] = [ROOT]
    import bench

    full = _canned_full_record()
    assert len(json.dumps(full)) > 20000
    rec = bench.headline_record(full)
    text = json.dumps(rec)
    assert len(text) < bench.HEADLINE_MAX_BYTES == 8192
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in rec, key
    assert rec["config"]["workload"].startswith("cfg4") and "model" not in rec["config"]
    assert rec["config"]["host_ms_between_graph_replays_per_rank"] == 0.13          # a scalar: the slowest rank
    rf = rec["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")) <= set(rf)
    assert rf["kernel"] == "csr_aggregate_ell_kernel<4,false,2,true,7,true>" and rf["frac"] == 0.5466       # in-step, dominant
    assert rf["frac_isolated_plain"] == 0.8451 and rf["in_step_all_aggregations_frac"] == 0.5602
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-3 and rf["frac"] <= 1.0
    cb = rec["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] in ("port", "reference")
    assert rec["parity"]["criterion"].startswith("max |device - fp64 oracle| < 1e-5")
    assert rec["parity"]["max_abs_err_vs_cpu_f32"] == 1.43e-5 and rec["parity"]["cpu_f32_own_gap_vs_f64"] == 1.5e-5
    assert "Family A" in rec["parity"]["oracle_pin"]
    for key in bench.FLAT_KEYS:
        assert isinstance(rec[key], float), key
    assert all(not isinstance(v, (list, dict)) for k, v in rec.items() if k not in ("config", "roofline", "cpu_baseline", "parity"))
    # every string in the record is bounded: a leg that grows a note cannot grow the line
    assert max(len(v) for d in (rec, rec["config"], rec["roofline"], rec["cpu_baseline"], rec["parity"]) for v in d.values()
               if isinstance(v, str)) <= 220

    # emit(): exactly ONE stdout line, the headline; the full record goes to the side file
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.delenv("MLQEM_BENCH_FULL_STDOUT", raising=False)
    bench.emit(full)
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and json.loads(out) == rec
    with open(tmp_path / bench.FULL_RECORD) as fh:
        assert json.load(fh)["inference"] == full["inference"]
    # a rank-0 record of an N > 1 run: no cpu_baseline / parity / in-step leg; still the contract keys, still small
    multi = {k: v for k, v in full.items() if k not in ("cpu_baseline", "parity", "family_b", "configs")}
    multi["roofline"] = {k: v for k, v in full["roofline"].items() if k not in ("step_dominant", "in_step", "in_step_all_aggregations")}
    multi.update(n_gpus=8)
    rec8 = bench.headline_record(multi)
    assert rec8["roofline"]["frac"] == 0.8451 and "alone" in rec8["roofline"]["basis"] and "cpu_baseline" not in rec8
    assert len(json.dumps(rec8)) < 8192
