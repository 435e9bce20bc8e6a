import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
from blackwater.native import ops
n=262144
dy=torch.randn(n,128,device="cuda").to(torch.bfloat16); x=torch.randn(n,128,device="cuda").to(torch.bfloat16)
for _ in range(5): ops.layer_wgrad_bf16(dy,x,125,125)
torch.cuda.synchronize()
ev=[torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(50): ops.layer_wgrad_bf16(dy,x,125,125)
ev[1].record(); torch.cuda.synchronize()
print(os.environ.get("MLQEM_LAYER_WGRAD_LDS","default"), "us per call (kernel + reduce):", ev[0].elapsed_time(ev[1])/50*1e3)
