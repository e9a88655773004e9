#!/usr/bin/env python3
"""The SK block's pw layer (C -> C, GELU, k-octet output, activation-stationary kernel): B operand as fp16 ROWS (what the
depthwise kernel writes today) against fp16 k-octets.  usage: pw_b_format.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
dev = torch.device("cuda:0")
P = 7040
cx = ops.Ctx(precision=ops.PRECISION_F16X2)
def t(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for (C, n, single) in [(640, 24, False), (256, 24, True), (324, 24, True), (128, 24, True), (384, 8, False)]:
    W = PackedLinear(torch.randn(C, C, 1, 1) / C ** 0.5, torch.randn(C) * 0.1, dev)
    W.single = single
    Ca = (C + 7) // 8 * 8
    x = torch.randn(n, Ca, P, device=dev)
    rows16 = x[:, :C].half().contiguous()
    Xr = Planes(rows16.view(-1).view(torch.float32), 0, C * P, n, C, P, f16=True)
    Xk = Planes(torch.zeros(n * Ca * P // 2, device=dev), 0, Ca * P, n, C, P, f16=True, koct=True)
    ops.pack_koct(Planes.of(x[:, :C].contiguous()), Xk)
    Y = Planes(torch.zeros(n * Ca * P // 2, device=dev), 0, Ca * P, n, C, P, f16=True, koct=True)
    tr = t(lambda: ops.gemm(W, Xr, Y, ops.EPI_GELU, cx=cx))
    tk = t(lambda: ops.gemm(W, Xk, Y, ops.EPI_GELU, cx=cx))
    print(f"pw C={C} n={n} {'1p' if single else '2p'}: rows {tr:7.1f} us   k-octets {tk:7.1f} us   ({100 * (tr - tk) / tr:4.1f} % )", flush=True)
