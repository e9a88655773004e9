#!/usr/bin/env python3
"""Micro-benchmark of sf_gemm shapes from the update block (run on the GPU box).
Prints us and algorithmic TFLOP/s per (shape, epilogue, precision); K=32 rows isolate prologue+epilogue cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear

dev = torch.device("cuda:0")
P, n = 7040, 3
shapes = [(486, 324), (324, 486), (256, 384), (960, 640), (128, 960), (384, 128), (486, 32), (128, 32)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
epis = [("none", ops.EPI_NONE), ("gelu", ops.EPI_GELU), ("res_gelu_dw1", ops.EPI_RES_GELU_DW1)]
for M, K in shapes:
    W = PackedLinear(torch.randn(M, K) / K ** 0.5, torch.randn(M) * 0.1, dev)
    X = Planes.of(torch.randn(n, K, P, device=dev))
    Y = Planes.of(torch.empty(n, M, P, device=dev))
    R = Planes.of(torch.randn(n, M, P, device=dev))
    dw = torch.randn(M, device=dev)
    for ename, e in epis:
        for prec in ("fp32", "f16x3"):
            ops.set_precision(prec)
            kw = dict(R=R, dw_w=dw, dw_b=dw) if e == ops.EPI_RES_GELU_DW1 else {}
            for _ in range(3):
                ops.gemm(W, X, Y, e, **kw)
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            s.record()
            for _ in range(reps):
                ops.gemm(W, X, Y, e, **kw)
            t.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(t) * 1e3 / reps
            print(f"M={M:4d} K={K:4d} {ename:13s} {prec:6s} {us:8.1f} us  {2.0 * M * K * P * n / us / 1e6:7.1f} TF", flush=True)
