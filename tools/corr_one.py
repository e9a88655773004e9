#!/usr/bin/env python3
"""Correlation-volume build alone at the Sintel shape (timing / rocprofv3 runs).  argv: reps [h w]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops, synthetic as syn
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (55, 128)
B, T, D = 1, 4, 256
fm, _ = syn.make_features(1, B, T, h, w)
fm = fm.to(dev).contiguous()
P = h * w
dims = [(h >> l, w >> l) for l in range(4)]
strides = [B * P * a * b for a, b in dims]
lv = [torch.empty((T - 1) * s, dtype=torch.float32, device=dev) for s in strides]
ws = torch.empty(ops.corr_build_ws_bytes(B, T - 1, D, h, w), dtype=torch.uint8, device=dev)
def run():
    ops.corr_build(fm.data_ptr(), fm.data_ptr() + 4 * D * P, T * D * P, D * P, lv, strides, B, T - 1, D, h, w, ws=ws)
for _ in range(3):
    run()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    run()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / reps
cells = sum(a * b for a, b in dims)
nbytes = (T - 1) * (2.0 * P * D * 4 + 4.0 * P * cells)
print(f"corr_build {us:.1f} us  {nbytes / us / 1e3:.0f} GB/s algorithmic  {2.0 * P * P * D * (T - 1) / us / 1e6:.1f} TF")

