"""bench.py's small_batch leg on its own: eager Trainer vs hipGraph-replayed BucketedTrainer at 32 circuits per step."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import torch
import bench
print(json.dumps(bench.small_batch_leg(torch.device("cuda", 0)), indent=1))
