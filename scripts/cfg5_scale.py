"""cfg5 at scale (BASELINE.json: "Mixed 1M-circuit corpus (TFIM + random + Pauli-twirled), GNN, 8xMI355X, bf16 MFMA MLP
head"): builds ONE rank's shard of the 8-way data-parallel split -- 125 000 circuits: 50 % 4-qubit TFIM (cfg2-like), 30 %
random 20-qubit depth-40 (cfg3-like), 20 % 100-qubit TFIM with Pauli twirling (cfg4-like) -- as a device-resident arena,
then runs train steps of Family A and of Family B with the MLP3 head on the bf16 matrix cores over mixed batches.
Distinct circuits are encoded once on the host (templates) and replicated on the device.  Writes
gpurun_out/cfg5_scale.json.   python scripts/cfg5_scale.py [circuits_per_rank]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.arena import GraphArena
from blackwater.data.synthetic import TfimCorpus, encode_corpus, pauli_twirl, random_circuit, tfim_circuit
from blackwater.nn import ExpValCircuitGraphModelA, ExpValCircuitGraphModel_3
from blackwater.train import Trainer

total = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
# templates: every distinct STRUCTURE is encoded once by the product encoder (all three families on one 100-qubit table, F = 22)
n2, n3, n4 = total // 2, total * 3 // 10, total - total // 2 - total * 3 // 10
tf = TfimCorpus(100, list(range(1, 11)), 1, seed=1, two_q="cx")                      # only used for its backend table shape
small = [tfim_circuit(4, s, J=0.3 + 0.01 * s, two_q="cx") for s in range(15)]
rand = [random_circuit(20, 40, seed=k, two_q="cx") for k in range(24)]
twirled = [pauli_twirl(tfim_circuit(100, s, J=0.5, two_q="cx"), seed=100 + s, two_q=("cx",)) for s in range(1, 11)]
enc = encode_corpus(small + rand + twirled, 100, two_q="cx", exp_value_size=4)
copies = np.concatenate([np.full(15, -(-n2 // 15)), np.full(24, -(-n3 // 24)), np.full(10, -(-n4 // 10))]).astype(np.int64)
sizes = np.array([x.shape[0] for x in enc["x"]])
f = enc["x"][0].shape[1]
f4 = (f + 3) // 4 * 4
n_total = int((sizes * copies).sum())
x = torch.zeros((n_total, f4), dtype=torch.float32, device=dev)
eis, base, tmpl_of = [], 0, []
for t, (xt, et, c) in enumerate(zip(enc["x"], enc["edge_index"], copies)):
    n_t = xt.shape[0]
    x[base:base + c * n_t].view(c, n_t, f4)[:, :, :f] = torch.from_numpy(xt).to(dev).unsqueeze(0)
    offs = base + torch.arange(c, device=dev, dtype=torch.int64) * n_t
    eis.append((torch.from_numpy(et).to(dev).unsqueeze(1) + offs.view(1, c, 1)).reshape(2, -1))
    base += c * n_t
    tmpl_of.append(np.full(c, t))
tmpl_of = np.concatenate(tmpl_of)
g = len(tmpl_of)
rng = np.random.default_rng(0)
y = rng.uniform(-1, 1, size=(g, 1, 4)).astype(np.float32)
noisy = (y * 0.9 + rng.normal(0, 0.01, size=y.shape)).astype(np.float32)
arena = GraphArena.from_device(x[:, :f], sizes[tmpl_of], torch.cat(eis, dim=1), y, noisy, enc["depth"][tmpl_of],
                               enc["observable"][tmpl_of])
del eis
torch.cuda.synchronize()
build_s = time.perf_counter() - t0
rec = {"what": "one rank's shard (1/8) of BASELINE.json's cfg5 corpus, device-resident", "circuits": g, "nodes": int(arena.num_nodes),
       "edges": int(arena.edge_counts.sum()), "mix": {"tfim_4q": int(copies[:15].sum()), "random_20q_d40": int(copies[15:39].sum()),
                                                       "tfim_100q_twirled": int(copies[39:].sum())},
       "arena_build_s": round(build_s, 1), "hbm_after_build_GB": round(torch.cuda.memory_allocated() / 1e9, 2)}
rs = np.random.RandomState(0)
for name, make, batch, steps in (("family_a", lambda: ExpValCircuitGraphModelA(100, f, 10), 1024, 10),
                                 ("family_b_mlp3_head_bf16", lambda: ExpValCircuitGraphModel_3(f, 15, 4), 64, 20)):
    torch.manual_seed(0)
    model = make().to(dev)
    if name.startswith("family_b"):
        model.body_seq.mfma = "bf16"
    tr = Trainer(model, lr=1e-3)
    if name == "family_a":       # scalar labels for the 1-output family
        arena.y, arena.noisy = arena.y[:, :, :1].reshape(g, 1).contiguous(), arena.noisy[:, :, :1].reshape(g, 1).contiguous()
    draw = lambda: rs.randint(0, g, size=batch)
    for _ in range(2):
        tr.step(arena.batch(draw()))
    torch.cuda.synchronize(); t1 = time.perf_counter(); nodes = 0
    for _ in range(steps):
        ids = draw(); nodes += int(arena.node_counts[ids].sum())
        loss = tr.step(arena.batch(ids))
    torch.cuda.synchronize(); dt = time.perf_counter() - t1
    rec[name] = {"circuits_per_step": batch, "ms_per_step": round(dt / steps * 1e3, 2), "circuits_per_s": round(batch * steps / dt, 1),
                 "mean_nodes_per_step": nodes // steps, "loss": float(loss.item()), "hbm_peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}
    if name == "family_a":
        arena.y, arena.noisy = torch.as_tensor(y).to(dev), torch.as_tensor(noisy).to(dev)
    del tr, model
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "cfg5_scale.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
