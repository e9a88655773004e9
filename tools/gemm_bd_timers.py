#!/usr/bin/env python3
"""Per-wave phase timers of the B-direct GEMM kernel (needs tools/build_variant.sh bdt gemm_split.hip -DSF_GEMM_TIMERS and
SF_HIP_LIB=.../variant_bdt.so).  usage: gemm_bd_timers.py M K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
M, K = int(sys.argv[1]), int(sys.argv[2])
n, P = 24, 7040
dev = torch.device("cuda:0")
ops.set_precision("f16x2")
W = PackedLinear(torch.randn(M, K, 1, 1) / K ** 0.5, torch.randn(M) * 0.1, dev)
W.single = os.environ.get("SF_SINGLE", "0") == "1"
X = Planes(torch.zeros(n * K * P // 2, device=dev), 0, K * P, n, K, P, f16=True, koct=True)
ops.pack_koct(Planes.of(torch.randn(n, K, P, device=dev)), X)
Y = Planes.of(torch.empty(n, M, P, device=dev))
for _ in range(3):
    ops.gemm(W, X, Y, ops.EPI_NONE)
torch.cuda.synchronize()
ts = torch.zeros(8192 * 64, dtype=torch.int64, device=dev)
os.environ["SF_GEMM_TS_BUF"] = str(ts.data_ptr())
ops.gemm(W, X, Y, ops.EPI_NONE); torch.cuda.synchronize()
t = ts.view(-1, 8).cpu().double()
t = t[t[:, 0] > 0]
nk = (K + 31) // 32
nk += nk & 1
print(f"M{M} K{K}: {t.shape[0]} waves, {nk} stages each")
for name, col in (("vmcnt wait", 1), ("barrier", 2), ("issue B+DMA", 3), ("mfma block", 4)):
    v = t[:, col] / nk
    print(f"  {name:12s} per stage: mean {v.mean().item():7.0f} median {v.median().item():7.0f} min {v.min().item():7.0f} max {v.max().item():7.0f} cycles")
for name, col in (("prologue+k-loop", 5), ("epilogue", 6)):
    v = t[:, col]
    print(f"  {name:15s}: mean {v.mean().item():8.0f} median {v.median().item():8.0f} min {v.min().item():8.0f} max {v.max().item():8.0f} cycles")
print(f"  clock {((t[:, 5] + t[:, 6]).sum() / t[:, 7].sum()).item() * 100:.0f} MHz")
