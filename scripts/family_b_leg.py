import sys, os, json
sys.path[:0] = ['.', 'ml-qem_amd']
import torch, numpy as np, time
import bench
out = bench.family_b_leg(torch.device("cuda:0"), steps=30)
for k in [k for k in out if k.startswith(("batch", "cfg4"))]:
    print(k, out[k])
