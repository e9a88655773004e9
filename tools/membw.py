#!/usr/bin/env python3
"""Achievable HBM bandwidth on this box by access type (the ceilings the corr kernels are priced against):
pure write (fill), pure read (sum), copy.  8 GB buffers (>> 256 MB Infinity Cache)."""
import torch
dev = torch.device("cuda:0")
n = 2 * 1024 ** 3                     # 2 Gi floats = 8 GB
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
b = n * 4
print(f"fill  (write only): {b / t(lambda: x.fill_(1.5)) / 1e12:.2f} TB/s")
print(f"zero  (memset)    : {b / t(lambda: x.zero_()) / 1e12:.2f} TB/s")
print(f"sum   (read only) : {b / t(lambda: x.sum()) / 1e12:.2f} TB/s")
print(f"copy  (read+write): {2 * b / t(lambda: y.copy_(x)) / 1e12:.2f} TB/s (both directions counted)")
h = torch.empty(n, dtype=torch.float16, device=dev)
print(f"fill fp16 4 GB    : {n * 2 / t(lambda: h.fill_(1.5)) / 1e12:.2f} TB/s")
