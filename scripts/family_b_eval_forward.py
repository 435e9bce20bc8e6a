"""The eval-mode forward of Family B on 64 100-qubit circuits (what the estimator decorators run), for rocprofv3 --kernel-trace --stats:
python scripts/family_b_eval_forward.py [runs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.data.synthetic import TfimCorpus
from blackwater.nn import ExpValCircuitGraphModel

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
arena = TfimCorpus(100, list(range(1, 11)), 7, seed=42, exp_value_size=4).arena(dev)
torch.manual_seed(0)
model = ExpValCircuitGraphModel(22, 15, 4).to(dev).eval()
ids = np.arange(64) * len(arena) // 64
with torch.no_grad():
    for _ in range(3):
        out = model(*arena.batch(ids).model_args())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(runs):
        out = model(*arena.batch(ids).model_args())
    torch.cuda.synchronize()
print("family B eval forward: 64 circuits, %.3f ms per forward" % ((time.perf_counter() - t0) / runs * 1e3), flush=True)
