#!/usr/bin/env python3
"""Does HBM traffic bound the fused GEMM?  Times one sf_gemm shape over n images (a) with distinct operand images,
(b) input images aliased (img_stride 0: B tiles stay in L2 / Infinity Cache), (c) output images aliased, (d) both.
usage: gemm_alias.py M K [epi] [n_img]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear

M, K = int(sys.argv[1]), int(sys.argv[2])
epi = {"none": ops.EPI_NONE, "gelu": ops.EPI_GELU}[sys.argv[3] if len(sys.argv) > 3 else "gelu"]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 24
ops.set_precision(os.environ.get("SF_PREC", "f16x3"))
dev = torch.device("cuda:0")
P = 7040
W = PackedLinear(torch.randn(M, K) / K ** 0.5, torch.randn(M) * 0.1, dev)
xb = torch.randn(n, K, P, device=dev)
yb = torch.empty(n, M, P, device=dev)


def run(alias_in, alias_out, reps=20):
    X = Planes(xb.view(-1), 0, 0 if alias_in else K * P, n, K, P)
    Y = Planes(yb.view(-1), 0, 0 if alias_out else M * P, n, M, P)
    for _ in range(3):
        ops.gemm(W, X, Y, epi)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.gemm(W, X, Y, epi)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    return us, 2.0 * M * K * P * n / us / 1e6


for rnd in range(2):
    for name, ai, ao in (("distinct", False, False), ("in-aliased", True, False), ("out-aliased", False, True),
                         ("both-aliased", True, True)):
        us, tf = run(ai, ao)
        print(f"M{M} K{K} n{n} {name:13s} {us:8.1f} us {tf:7.1f} TF", flush=True)
