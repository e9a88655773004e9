#!/usr/bin/env python3
"""Per-wave phase timers of the activation-stationary GEMM (needs tools/build_variant.sh bst gemm_bstat.hip -DSF_BSTAT_TIMERS and
SF_HIP_LIB=.../variant_bst.so).  usage: gemm_bs_timers.py M K   (SF_SINGLE=1: one product)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import _lib, ops
from streamflow_amd.ops import Planes, PackedLinear
M, K = int(sys.argv[1]), int(sys.argv[2])
EPI = {"none": 0, "gelu": 1, "relu": 2}[sys.argv[3] if len(sys.argv) > 3 else "gelu"]
CF = int(sys.argv[4]) if len(sys.argv) > 4 else 2
n, P = 24, 7040
dev = torch.device("cuda:0")
ops.set_precision("f16x2")
W = PackedLinear(torch.randn(M, K, 1, 1) / K ** 0.5, torch.randn(M) * 0.1, dev)
W.single = os.environ.get("SF_SINGLE", "0") == "1"
X = Planes(torch.zeros(n * K * P // 2, device=dev), 0, K * P, n, K, P, f16=True, koct=True)
ops.pack_koct(Planes.of(torch.randn(n, K, P, device=dev)), X)
Mo = (M + 7) // 8 * 8
Y = (Planes(torch.zeros(n * Mo * P // 2, device=dev), 0, Mo * P, n, M, P, f16=True, koct=True) if CF == 2
     else Planes.of(torch.empty(n, M, P, device=dev)))
for _ in range(3):
    ops.gemm(W, X, Y, EPI, algo=_lib.ALGO_BSTAT)
torch.cuda.synchronize()
ts = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device=dev)
os.environ["SF_GEMM_TS_BUF"] = str(ts.data_ptr())
ops.gemm(W, X, Y, EPI, algo=_lib.ALGO_BSTAT); torch.cuda.synchronize()
t = ts.view(-1, 8).cpu().double()
t = t[t[:, 6] > 0]
sk = 128 if W.single else 64
nstage = ((K + sk - 1) // sk) * ((M + 63) // 64)
if EPI == 1 and CF == 2:                            # the pipelined kernel: fragment-granular stages over 32-row tiles
    nks = {1: 8, 2: 16, 3: 24}.get((K + 127) // 128, 40)
    nf = nks * (1 if W.single else 2)
    S = 16 if nf % 16 == 0 else 20 if nf % 20 == 0 else 12 if nf % 12 == 0 else 8
    nstage = (nf // S) * ((M + 31) // 32)
print(f"M{M} K{K} single={W.single} epi={EPI} cf={CF}: {t.shape[0]} waves, {nstage} stages each ({(M + 63) // 64} m-steps)")
for name, col in (("vmcnt wait", 1), ("barrier", 2), ("issue DMA", 3), ("mfma block", 4)):
    v = t[:, col] / nstage
    print(f"  {name:12s} per stage: mean {v.mean().item():7.0f} median {v.median().item():7.0f} min {v.min().item():7.0f} max {v.max().item():7.0f} cycles")
v = t[:, 5] / ((M + 63) // 64)
print(f"  epilogue per m-step    : mean {v.mean().item():7.0f} median {v.median().item():7.0f} min {v.min().item():7.0f} max {v.max().item():7.0f} cycles")
for name, col in (("prologue (B loads)", 0), ("whole wave", 6)):
    v = t[:, col]
    print(f"  {name:18s}: mean {v.mean().item():8.0f} median {v.median().item():8.0f} min {v.min().item():8.0f} max {v.max().item():8.0f} cycles")
print(f"  clock {(t[:, 6].sum() / t[:, 7].sum()).item() * 100:.0f} MHz")
if EPI == 1 and CF == 2:                            # round structure: when do the workgroups start / end (100 MHz real time)
    t0 = t[:, 5].min()
    st, en = (t[:, 5] - t0) / 100.0, (t[:, 5] - t0 + t[:, 7]) / 100.0
    print(f"  kernel span {en.max().item():.1f} us; wave life mean {(en - st).mean().item():.1f} us")
    edges = [0, 5, 10, 20, 40, 60, 80, 100, 120, 140, 160, 200, 250, 300, 400, 1000]
    print("  starts per window [us]:", [(edges[i], int(((st >= edges[i]) & (st < edges[i + 1])).sum().item()) // 4) for i in range(len(edges) - 1)])
    print("  ends   per window [us]:", [(edges[i], int(((en >= edges[i]) & (en < edges[i + 1])).sum().item()) // 4) for i in range(len(edges) - 1)])
