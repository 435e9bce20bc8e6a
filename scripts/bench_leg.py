"""One leg of bench.py on its own: python scripts/bench_leg.py inference_leg | mlp_head_leg | family_b_leg | small_batch_leg"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import torch
import bench
torch.cuda.set_device(0)
print(json.dumps(getattr(bench, sys.argv[1])(torch.device("cuda:0")), indent=1))
