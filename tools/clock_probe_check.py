#!/usr/bin/env python3
"""Does sf_clock_probe run BESIDE the load, and what does it read?  (a) probe alone; (b) probe beside a loop of the two-product K = 640
GEMM; (c) the GEMM's own in-kernel clock needs the timers build (tools/gemm_bs_timers.py).  Prints wall times so that serialisation
(probe first, load after) would show."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
dev = torch.device("cuda:0"); ops.set_precision("f16x2")
M, K, n, P = 960, 640, 24, 7040
W = PackedLinear(torch.randn(M, K, 1, 1) / K ** 0.5, torch.randn(M) * 0.1, dev)
X = Planes(torch.zeros(n * K * P // 2, device=dev), 0, K * P, n, K, P, f16=True, koct=True)
ops.pack_koct(Planes.of(torch.randn(n, K, P, device=dev)), X)
Y = Planes(torch.zeros(n * M * P // 2, device=dev), 0, M * P, n, M, P, f16=True, koct=True)
side, out = torch.cuda.Stream(device=dev), torch.zeros(2, dtype=torch.int64, device=dev)
def probe(us):
    with torch.cuda.stream(side):
        ops.clock_probe(out, us)
for _ in range(20): ops.gemm(W, X, Y, ops.EPI_GELU)
torch.cuda.synchronize()
t0 = time.time(); probe(100000); torch.cuda.synchronize(); t1 = time.time()
c = out.tolist(); print(f"probe alone: {100.0 * c[0] / c[1]:.1f} MHz, wall {1e3 * (t1 - t0):.1f} ms (100 ms of spin)")
for reps in (300, 600):
    torch.cuda.synchronize(); t0 = time.time()
    probe(100000)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): ops.gemm(W, X, Y, ops.EPI_GELU)
    e.record(); torch.cuda.synchronize(); t1 = time.time()
    c = out.tolist()
    print(f"probe beside {reps} GEMM launches: {100.0 * c[0] / c[1]:.1f} MHz; GEMMs {s.elapsed_time(e):.1f} ms ({1e3 * s.elapsed_time(e) / reps:.1f} us each), wall {1e3 * (t1 - t0):.1f} ms")
