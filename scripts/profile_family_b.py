"""Family B train steps for rocprofv3: `python scripts/profile_family_b.py [batch] [steps] [qubits]` -- cfg2 (4-qubit TFIM graphs) by
default, qubits = 100 for the headline graphs (cfg4: 100-qubit circuits, Trotter steps 1-10)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.arena import GraphArena
from blackwater.data.synthetic import TfimCorpus
from blackwater.native import _lib
if os.environ.get('MLQEM_LIB'):          # A/B builds of the library (scripts/ab_*.sh)
    _lib.LIB_PATH = os.environ['MLQEM_LIB']
from blackwater.nn import ExpValCircuitGraphModel
from blackwater.train import Trainer
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 4
corpus = (TfimCorpus(4, list(range(15)), 70, seed=42, two_q="cx", exp_value_size=4) if nq == 4
          else TfimCorpus(nq, list(range(1, 11)), 7, seed=42, exp_value_size=4))
h = corpus.host_graphs()
arena = GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"], device="cuda:0")
torch.manual_seed(0)
model = ExpValCircuitGraphModel(22, 15, 4).to("cuda:0")
tr = Trainer(model, lr=1e-3)
rng = np.random.RandomState(0)
draw = lambda: rng.randint(0, len(arena), size=batch)
for _ in range(5):
    tr.step(arena.batch(draw()))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    loss = tr.step(arena.batch(draw()))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"family B train step: batch {batch}, {dt / steps * 1e3:.2f} ms/step, {batch * steps / dt:.0f} circuits/s, loss {loss.item():.4f}")
