#!/usr/bin/env python3
"""Depthwise K x K layer of the SK blocks at the update block's shapes (24 images of 55 x 128): fp32-input and fp16-input
(DMA-staged, double-buffered) forms, fp16 output.  usage: dw_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes
dev = torch.device("cuda:0")
h, w = 55, 128
P = h * w
for (k, C, n, single) in [(15, 256, 24, True), (15, 324, 24, True), (15, 128, 24, True), (15, 640, 24, True), (15, 384, 8, True), (7, 640, 24, False), (15, 256, 24, False)]:
    wgt = (torch.randn(C, k * k) / k).to(dev)
    b = (torch.randn(C) * 0.1).to(dev)
    x = torch.randn(n, C, P, device=dev)
    x16 = x.half().contiguous()
    Xf = Planes.of(x)
    Xh = Planes(x16.view(torch.float32).view(-1), 0, C * P, n, C, P, f16=True)
    y = torch.zeros(n, C, P, dtype=torch.float16, device=dev)
    Y = Planes(y.view(torch.float32).view(-1), 0, C * P, n, C, P, f16=True)
    cx = ops.Ctx(precision=ops.PRECISION_F16X2)
    res = []
    for X in (Xf, Xh):
        for _ in range(3):
            ops.dwconv_res_gelu(X, wgt, b, Y, h, w, k, single=single, cx=cx)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.dwconv_res_gelu(X, wgt, b, Y, h, w, k, single=single, cx=cx)
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) * 50)
    by = n * C * P
    print(f"k{k} C{C} n{n} {'1p' if single else '2p'}: fp32-in {res[0]:7.1f} us ({6 * by / res[0] / 1e6:5.2f} TB/s)   fp16-in {res[1]:7.1f} us ({4 * by / res[1] / 1e6:5.2f} TB/s)", flush=True)
