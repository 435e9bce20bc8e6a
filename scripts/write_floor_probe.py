"""What a kernel that only WRITES an [N, 12] fp32 matrix (the pool backward's output on the headline batch: 0.54 GB) can reach:
torch's fill and a copy of the same bytes, for the pool backward's 167 us to be read against.  python scripts/write_floor_probe.py"""
import torch
n = 11291888
dev = "cuda:0"
a = torch.empty(n, 12, device=dev); b = torch.randn(n, 12, device=dev)
def timed(fn, reps=20):
    for _ in range(3): fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps): fn()
    t1.record(); t1.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3
gb = n * 48 / 1e9
t = timed(lambda: a.fill_(1.0)); print(f"fill  {gb:.2f} GB: {t:.0f} us, {gb / t * 1e3:.0f} GB/s written")
t = timed(lambda: a.copy_(b)); print(f"copy  {gb:.2f} GB: {t:.0f} us, {2 * gb / t * 1e3:.0f} GB/s moved")
t = timed(lambda: torch.mul(b, 2.0, out=a)); print(f"scale {gb:.2f} GB: {t:.0f} us, {2 * gb / t * 1e3:.0f} GB/s moved")
