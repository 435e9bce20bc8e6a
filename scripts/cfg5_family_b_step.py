"""bench.py's cfg5 Family B leg alone (64 random circuits of the mixed corpus per eager step, ExpValCircuitGraphModel_3 with the bf16 MLP3
head): python scripts/cfg5_family_b_step.py [steps]   (MLQEM_ROOT=<tree> runs another checkout's package and bench helpers)"""
import os, sys, time
ROOT = os.environ.get("MLQEM_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
import bench
from blackwater.data.synthetic import encode_corpus, pauli_twirl, random_circuit, tfim_circuit
from blackwater.nn import ExpValCircuitGraphModel_3
from blackwater.train import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = "cuda:0"
total = 15_625
n2, n3 = total // 2, total * 3 // 10
n4 = total - n2 - n3
small = [tfim_circuit(4, st, J=0.3 + 0.01 * st, two_q="cx") for st in range(15)]
rand = [random_circuit(20, 40, seed=s, two_q="cx") for s in range(12)]
twirled = [pauli_twirl(tfim_circuit(100, st, J=0.5, two_q="cx"), seed=100 + st, two_q=("cx",)) for st in range(1, 11)]
enc5 = encode_corpus(small + rand + twirled, 100, two_q="cx", exp_value_size=4)
copies = np.concatenate([np.full(15, -(-n2 // 15)), np.full(12, -(-n3 // 12)), np.full(10, -(-n4 // 10))])
arena, _ = bench.replicated_arena(enc5, copies, dev, scalar_labels=False)
rs = np.random.RandomState(0)
torch.manual_seed(0)
model = ExpValCircuitGraphModel_3(22, 15, 4).to(dev)
model.body_seq.mfma = os.environ.get("HEAD", "bf16")
tr = Trainer(model, lr=1e-3)
draw = lambda: rs.randint(0, len(arena), size=64)
warm = int(os.environ.get("WARM", "2"))
sec, loss = bench._timed_steps(lambda: tr.step(arena.batch(draw())), warm, steps)
print("cfg5 family B step (%s, head %s): %.3f ms/step, %.0f circuits/s, loss %.6f" % (ROOT, model.body_seq.mfma, sec * 1e3, 64 / sec, float(loss.item())), flush=True)
