"""bench.py's attention_roofline on the 64-circuit 100-qubit structure, alone: python scripts/attn_roofline_probe.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
import bench
from blackwater.data.synthetic import TfimCorpus
dev = torch.device("cuda:0")
arena = TfimCorpus(100, list(range(1, 11)), 104, seed=42, exp_value_size=4).arena(dev, filler_nodes=1024)
n = len(arena)
r = bench.attention_roofline(arena.batch(np.arange(64) * n // 64).structure, dev, "64 100-qubit circuits")
print(json.dumps({k: r[k] for k in ("frac", "frac_r02_model", "us_per_launch", "bytes_per_launch", "kernel")}))
