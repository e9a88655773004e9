#!/usr/bin/env python3
"""Run one sf_gemm shape N times (for rocprofv3 --pmc runs).  usage: gemm_one.py M K [epi] [precision]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
M, K = int(sys.argv[1]), int(sys.argv[2])
epi = {"none": ops.EPI_NONE, "gelu": ops.EPI_GELU}[sys.argv[3] if len(sys.argv) > 3 else "none"]
ops.set_precision(sys.argv[4] if len(sys.argv) > 4 else "f16x3")
dev = torch.device("cuda:0")
P, n = 7040, 3
W = PackedLinear(torch.randn(M, K) / K ** 0.5, torch.randn(M) * 0.1, dev)
X = Planes.of(torch.randn(n, K, P, device=dev))
Y = Planes.of(torch.empty(n, M, P, device=dev))
for _ in range(10):
    ops.gemm(W, X, Y, epi)
torch.cuda.synchronize()
