import sys, os, json
sys.path[:0] = ['.', 'ml-qem_amd']
import torch, numpy as np, time
import bench
out = bench.family_b_leg(torch.device("cuda:0"), steps=30)
for k in ("batch1024", "batch32", "batch32_stratified_eager", "batch32_stratified_hipgraph", "cfg4_100q_batch64"):
    print(k, out[k])
