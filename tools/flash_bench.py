#!/usr/bin/env python3
"""Fused GMA aggregation alone (sf_gma_flash_pack_qk once, sf_gma_flash_aggregate timed) at the bench shape: 24 images of 7040 px.
usage: flash_bench.py [n_img] [P]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
P = int(sys.argv[2]) if len(sys.argv) > 2 else 7040
dev = torch.device("cuda:0")
cx = ops.Ctx(precision=ops.PRECISION_F16X2)
qk = torch.randn(n, 256, P, device=dev)
v16 = torch.randn(n, 128, P, device=dev).half().contiguous()
V = Planes(v16.view(-1).view(torch.float32), 0, 128 * P, n, 128, P, f16=True)
mf = torch.randn(n, 128, P, device=dev)
out = torch.empty(n, 128, P, device=dev)
gamma = torch.tensor([0.5], device=dev)
ws = torch.empty(ops.gma_flash_ws_bytes(n, P), dtype=torch.uint8, device=dev)
ops.gma_flash_pack_qk(Planes.of(qk), ws, 128 ** -0.5, stats_qk_products=1, cx=cx)
for _ in range(3):
    ops.gma_flash_aggregate(ws, V, Planes.of(mf), gamma, Planes.of(out), 1, use_stats=True, cx=cx)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    ops.gma_flash_aggregate(ws, V, Planes.of(mf), gamma, Planes.of(out), 1, use_stats=True, cx=cx)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
print(f"gma_flash n={n} P={P}: {us:.1f} us per call (pack_v + kernel), {4.0 * n * P * P * 128 / us / 1e6:.0f} TF")
