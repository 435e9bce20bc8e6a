"""Binary shard format: round trips, header/size validation, conversion from the
reference's dataset format and from QASM text through the native encoder."""
import json
import os

import numpy as np
import pytest

from helpers import G1_GATES_ORDER, g1_graph


def _g1_lists(g1, count):
    xs, eis = [], []
    for i in range(count):
        x, ei, _ = g1_graph(g1, i)
        xs.append(x.astype(np.float32))
        eis.append(ei.astype(np.int64))
    idx = list(range(count))
    obs = np.zeros((count, 1, 21), np.float32)
    obs[:, 0, 0] = 1.0
    return xs, eis, g1["ideal"][idx], g1["noisy"][idx], g1["depth"][idx].reshape(-1, 1), obs


def test_round_trip_is_bit_exact(g1, tmp_path):
    from blackwater.data.shards import MAGIC, pack_graphs, read_shard, write_shard

    xs, eis, y, noisy, depth, obs = _g1_lists(g1, 40)
    shard = pack_graphs(xs, eis, y, noisy, depth, obs, meta={"source": "g1", "gates_order": G1_GATES_ORDER})
    path = str(tmp_path / "g1.mlqs")
    write_shard(path, shard)
    with open(path, "rb") as fh:
        assert fh.read(8) == MAGIC
        hlen = int(np.frombuffer(fh.read(8), "<u8")[0])
        head = json.loads(fh.read(hlen))
    assert all(ent["offset"] % 64 == 0 for ent in head["arrays"].values())
    assert head["graphs"] == 40 and head["nodes"] == sum(a.shape[0] for a in xs)
    for mmap in (True, False):
        back = read_shard(path, mmap=mmap)
        assert len(back) == 40 and back.meta["gates_order"] == G1_GATES_ORDER
        for k, a in shard.arrays.items():
            assert back.arrays[k].dtype == a.dtype and np.array_equal(np.asarray(back.arrays[k]), a), k
        for i in (0, 17, 39):
            x, ei, yi, ni, di, oi = back.graph(i)
            assert np.array_equal(x, xs[i]) and np.array_equal(ei, eis[i])
            assert np.array_equal(yi, np.asarray(y[i], np.float32)) and di[0] == np.float32(depth[i, 0])


def test_empty_shard_and_edgeless_graph(tmp_path):
    from blackwater.data.shards import pack_graphs, read_shard, write_shard

    empty = pack_graphs([], [], np.zeros((0, 1)), np.zeros((0, 1)), np.zeros((0, 1)), np.zeros((0, 1, 5)))
    write_shard(str(tmp_path / "e.mlqs"), empty)
    assert len(read_shard(str(tmp_path / "e.mlqs"))) == 0
    one = pack_graphs([np.ones((1, 6), np.float32)], [np.zeros((2, 0), np.int64)], [[0.5]], [[0.4]], [[1.0]],
                      np.zeros((1, 1, 5)))
    write_shard(str(tmp_path / "o.mlqs"), one)
    back = read_shard(str(tmp_path / "o.mlqs"))
    assert len(back) == 1 and back.num_edges == 0 and back.graph(0)[0].shape == (1, 6)


def test_corrupt_files_are_refused(g1, tmp_path):
    from blackwater.data.shards import ShardFormatError, pack_graphs, read_shard, write_shard

    xs, eis, y, noisy, depth, obs = _g1_lists(g1, 8)
    path = str(tmp_path / "s.mlqs")
    write_shard(path, pack_graphs(xs, eis, y, noisy, depth, obs))
    blob = open(path, "rb").read()
    bad = str(tmp_path / "bad.mlqs")
    open(bad, "wb").write(b"NOTASHRD" + blob[8:])
    with pytest.raises(ShardFormatError, match="magic"):
        read_shard(bad)
    open(bad, "wb").write(blob[: len(blob) - 100])            # truncated tail
    with pytest.raises(ShardFormatError, match="truncated"):
        read_shard(bad)
    open(bad, "wb").write(blob[:12])                           # truncated header
    with pytest.raises(ShardFormatError):
        read_shard(bad)
    # an edge that leaves its graph must never reach the device
    eis_bad = [e.copy() for e in eis]
    eis_bad[3][0, 0] = xs[3].shape[0]
    with pytest.raises(ShardFormatError, match="outside its graph"):
        pack_graphs(xs, eis_bad, y, noisy, depth, obs)
    with pytest.raises(ValueError, match="feature width"):
        pack_graphs([xs[0], xs[1][:, :5]], eis[:2], y[:2], noisy[:2], depth[:2], obs[:2])


def test_shard_from_reference_json_dataset(golden_dir, tmp_path):
    """The reference's wire format (a .json list of ExpValueEntry dicts; tests/golden/encoder_goldens.json holds 30 taken
    from its data files) -> host dataset -> shard: same node features, edges and labels as the dataset's entries."""
    from blackwater.data.loaders.exp_val import CircuitGraphExpValMitigationDataset
    from blackwater.data.shards import read_shard, shard_from_dataset, write_shard

    goldens = json.load(open(os.path.join(golden_dir, "encoder_goldens.json")))
    widths = {}
    for g in goldens:   # one shard holds one feature width / label shape: keep the most common kind
        key = (len(g["circuit_graph"]["nodes"]["DAGOpNode"][0]), np.asarray(g["observable"]).shape,
               np.asarray(g["ideal_exp_value"]).shape, np.asarray(g["noisy_exp_values"]).shape)
        widths.setdefault(key, []).append(g)
    records = [{k: v for k, v in g.items() if k != "source_file"} for g in max(widths.values(), key=len)]
    assert len(records) >= 5
    src = str(tmp_path / "ds.json")
    json.dump(records, open(src, "w"))
    ds = CircuitGraphExpValMitigationDataset(src)
    shard = shard_from_dataset(ds)
    write_shard(str(tmp_path / "ds.mlqs"), shard)
    back = read_shard(str(tmp_path / "ds.mlqs"))
    assert len(back) == len(ds) > 0
    for i, entry in enumerate(ds):
        x, ei, y, noisy, depth, obs = back.graph(i)
        assert np.array_equal(x, entry.x.numpy()) and np.array_equal(ei, entry.edge_index.numpy())
        assert np.array_equal(y, entry.y.numpy().reshape(-1)) and depth[0] == entry.circuit_depth.item()
        assert np.array_equal(noisy, entry.noisy_0.numpy().reshape(-1))
        assert np.array_equal(obs, entry.observable.numpy().reshape(obs.shape))


def test_shard_from_qasm_matches_python_encoder(g1, lima_props):
    from blackwater.data.shards import shard_from_qasm
    from blackwater.data.utils import circuit_to_graph_data_json

    qasms = g1["qasm"][:6]
    props = dict(lima_props, gates_set=G1_GATES_ORDER)
    shard = shard_from_qasm(qasms, props, y=np.zeros((6, 1)), noisy=np.zeros((6, 1)), observable=np.zeros((6, 1, 21)),
                            add_self_loops=False)
    for i, text in enumerate(qasms):
        want = circuit_to_graph_data_json(text, props, use_gate_features=True, use_qubit_features=True)
        x, ei = shard.graph(i)[:2]
        assert np.array_equal(x, np.asarray(want["nodes"]["DAGOpNode"], np.float32))
        assert np.array_equal(ei, np.asarray(want["edges"]["DAGOpNode_wire_DAGOpNode"]["edge_index"]))
        assert np.array_equal(x, g1_graph(g1, i)[0].astype(np.float32)) and shard.depth[i, 0] == g1["depth"][i]
