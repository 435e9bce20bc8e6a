import sys, json
sys.path[:0] = ['.', 'ml-qem_amd']
import torch, bench
print(json.dumps(bench.mlp_head_leg(torch.device("cuda:0")), indent=1))
