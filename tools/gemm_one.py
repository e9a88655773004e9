#!/usr/bin/env python3
"""Run one sf_gemm shape N times (timing / rocprofv3 --pmc runs).  usage: gemm_one.py M K [epi] [precision]
SF_N_IMG=24 sets the batch (images); SF_GEMM_TS=1 prints per-workgroup phase timers and the sustained clock, which
needs a library built with the timers: tools/build_variant.sh timers gemm_split.hip -DSF_GEMM_TIMERS, then
SF_HIP_LIB=streamflow_amd/csrc/build/variant_timers.so."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
M, K = int(sys.argv[1]), int(sys.argv[2])
epi = {"none": ops.EPI_NONE, "gelu": ops.EPI_GELU}[sys.argv[3] if len(sys.argv) > 3 else "none"]
ops.set_precision(sys.argv[4] if len(sys.argv) > 4 else "f16x3")
dev = torch.device("cuda:0")
P, n = 7040, int(os.environ.get('SF_N_IMG', '3'))
W = PackedLinear(torch.randn(M, K) / K ** 0.5, torch.randn(M) * 0.1, dev)
X = Planes.of(torch.randn(n, K, P, device=dev))
Y = Planes.of(torch.empty(n, M, P, device=dev))
for _ in range(10):
    ops.gemm(W, X, Y, epi)
torch.cuda.synchronize()

if os.environ.get("SF_GEMM_TS"):
    ts = torch.zeros(65536 * 8, dtype=torch.int64, device=dev)
    os.environ["SF_GEMM_TS_BUF"] = str(ts.data_ptr())
    ops.gemm(W, X, Y, epi); torch.cuda.synchronize()
    t = ts.view(-1, 8).cpu().double()
    t = t[t[:, 0] > 0]
    print("workgroups", t.shape[0])
    for name, v in (("prologue", t[:, 1] - t[:, 0]), ("k-loop", t[:, 2] - t[:, 1]), ("epilogue", t[:, 3] - t[:, 2])):
        print(f"{name:9s} mean {v.mean().item():9.0f} median {v.median().item():9.0f} min {v.min().item():9.0f} max {v.max().item():9.0f}")
    dt_c, dt_r = t[:, 3] - t[:, 0], t[:, 5] - t[:, 4]
    print(f"clock: {(dt_c.sum() / dt_r.sum()).item() * 100:.0f} MHz (s_memtime ticks per 100 MHz s_memrealtime tick)")
    span = (t[:, 5].max() - t[:, 4].min()).item() / 100.0
    print(f"kernel span by realtime: {span:.1f} us; sum of workgroup times / span = {dt_r.sum().item() / 100.0 / span:.1f} resident workgroups on average (max {256 * 3})")
    starts = (t[:, 4] - t[:, 4].min()) / 100.0
    q = [starts.kthvalue(max(1, int(f * t.shape[0]))).values.item() for f in (0.05, 0.25, 0.5, 0.75, 0.95, 1.0)]
    print("workgroup start times (us) at 5/25/50/75/95/100 %:", " ".join(f"{v:.1f}" for v in q))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    ops.gemm(W, X, Y, epi)
e.record(); torch.cuda.synchronize()
print(f"M{M} K{K}: {s.elapsed_time(e) * 50:.1f} us  {2.0 * M * K * P * n / (s.elapsed_time(e) * 50) / 1e6:.1f} TF")
