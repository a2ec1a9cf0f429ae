#!/usr/bin/env python3
"""Developer probe: achievable HBM write / read / copy bandwidth on this box (torch kernels), for roofline calibration."""
import torch, time
dev = torch.device("cuda:0")
n = 2 * 1024 ** 3 // 4          # 2 GiB of fp32
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e))
    return best
gb = n * 4 / 1e9
print(f"fill  (write only): {gb / t(lambda: a.fill_(1.0)) :.2f} TB/s" .replace("TB/s", "GB/ms = TB/s"))
print(f"sum   (read only) : {gb / t(lambda: a.sum()):.2f} TB/s")
print(f"copy  (read+write): {2 * gb / t(lambda: b.copy_(a)):.2f} TB/s total")
