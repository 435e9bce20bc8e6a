import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import numpy as np, torch
from blackwater.native import ops
dev = "cuda:0"
for per in (7360, 3680):
    sizes = np.full(96, per, dtype=np.int64)
    n = int(sizes.sum()); keep = (sizes + 1) // 2
    gptr = np.zeros(len(sizes) + 1, dtype=np.int32); gptr[1:] = np.cumsum(sizes)
    nptr = np.zeros(len(sizes) + 1, dtype=np.int32); nptr[1:] = np.cumsum(keep)
    gp, np_ = torch.from_numpy(gptr).to(dev), torch.from_numpy(nptr).to(dev)
    f = torch.sigmoid(torch.randn(n, device=dev))
    for _ in range(3):
        ops.segment_topk(f, gp, np_, n, len(sizes), int(keep.sum()), max_graph_nodes=per)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.segment_topk(f, gp, np_, n, len(sizes), int(keep.sum()), max_graph_nodes=per)
    b.record(); torch.cuda.synchronize()
    print(per, "us per call", a.elapsed_time(b) / 20 * 1e3)
