"""Host cost of the Python wrappers, piece by piece (tiny tensors, so the GPU is never the bottleneck)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-qem_amd")]
import torch
from blackwater.native import _lib, ops
from blackwater.native.structure import GraphStructure
dev = "cuda:0"
n = 512
ei = torch.randint(0, n, (2, 1500), device=dev)
s = GraphStructure.from_edge_index(ei, n)
x = ops.padded_empty(n, 10, dev).normal_(); out = ops.padded_empty(n, 10, dev)
w = torch.randn(10, 10, device=dev); dinv = s.gcn_dinv; ell = s.in_ell
def t(name, fn, reps=3000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(reps):
        fn()
        if k % 256 == 255: torch.cuda.synchronize()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps * 1e6
    print(f"{name:44s} {dt:6.2f} us")
lib = _lib.load()
t("padded_empty(n, 10)", lambda: ops.padded_empty(n, 10, dev))
t("torch.empty((n,12))", lambda: torch.empty((n, 12), dtype=torch.float32, device=dev))
t("torch.empty((n,12)).narrow(1,0,10)", lambda: torch.empty((n, 12), dtype=torch.float32, device=dev).narrow(1, 0, 10))
t("ops._stream()", ops._stream)
t("_mat(x)", lambda: ops._mat(x, "x"))
t("_vec(dinv)", lambda: ops._vec(dinv, "d", n))
t("x.data_ptr()", x.data_ptr)
args = (x.data_ptr(), 12, s.in_ptr.data_ptr(), s.in_src.data_ptr(), ell.data_ptr(), None, dinv.data_ptr(), dinv.data_ptr(), 1.0, 0.0,
        None, 0, None, 0, 0.0, 0, None, out.data_ptr(), 12, n, 10, ops._stream())
t("raw ctypes mlqem_csr_aggregate_f32 (22 args)", lambda: lib.mlqem_csr_aggregate_f32(*args))
t("ops.csr_aggregate(out=...)", lambda: ops.csr_aggregate(x, s.in_ptr, s.in_src, ell=ell, rscale=dinv, dself=dinv, out=out))
t("ops.csr_aggregate()  (allocating)", lambda: ops.csr_aggregate(x, s.in_ptr, s.in_src, ell=ell, rscale=dinv, dself=dinv))
t("ops.linear(out=...)", lambda: ops.linear(x, w, out=out))
t("ops.linear_parts 1->2", lambda: ops.linear_parts([x], [w, w], [out, out]))
from blackwater.native import functional as F
xg = x.detach().requires_grad_(True)
t("F.gcn_layer forward (autograd node)", lambda: F.gcn_layer(xg, w, None, s))
t("torch.add tiny (reference point)", lambda: torch.add(w, w))
