#!/usr/bin/env python3
"""Is the K = 640 GELU GEMM power-bound?  Same launch, same counters, four operand sets: random / zero activations x two / one product.
The guide (MI355X_MICROARCH.md, DVFS give-back) measured +19 % on zero-filled inputs for a kernel at its power budget.
usage: power_probe.py [seconds per arm]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
dev = torch.device("cuda:0"); ops.set_precision("f16x2")
M, K, n, P = 960, 640, 24, 7040
W = PackedLinear(torch.randn(M, K, 1, 1) / K ** 0.5, torch.randn(M) * 0.1, dev)
xbuf = torch.zeros(n * K * P // 2, device=dev)
X = Planes(xbuf, 0, K * P, n, K, P, f16=True, koct=True)
Y = Planes(torch.zeros(n * M * P // 2, device=dev), 0, M * P, n, M, P, f16=True, koct=True)
for single in (False, True):
    W.single = single
    for zero in (False, True):
        if zero: xbuf.zero_()
        else: ops.pack_koct(Planes.of(torch.randn(n, K, P, device=dev)), X)
        torch.cuda.synchronize(); t0 = time.time(); first = None
        while time.time() - t0 < secs:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(100): ops.gemm(W, X, Y, ops.EPI_GELU)
            e.record(); torch.cuda.synchronize()
            last = s.elapsed_time(e) * 10
            first = first or last
        print(f"products={1 if single else 2} zero_activations={zero}: first 100 launches {first:.1f} us each, after {secs:.0f} s {last:.1f} us", flush=True)
        time.sleep(3)
