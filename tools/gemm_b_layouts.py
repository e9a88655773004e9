#!/usr/bin/env python3
"""f16x2 GEMM pair timing by hand-over format of the intermediate: fp32 planes, fp16 rows (c_f16 = 1 ->
SF_LAYOUT_F16_K_MAJOR), fp16 k-octet planes (c_f16 = 2 -> SF_LAYOUT_F16_KOCT, DMA-fed).  Producer (C -> H, GELU) and
consumer (H -> Cout) are timed separately.  usage: gemm_b_layouts.py C H Cout [n_img]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
C, H, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 24
P = 7040
dev = torch.device("cuda:0")
ops.set_precision("f16x2")
W1 = PackedLinear(torch.randn(H, C, 1, 1) / C ** 0.5, torch.randn(H) * 0.1, dev)
W2 = PackedLinear(torch.randn(Cout, H, 1, 1) / H ** 0.5, torch.randn(Cout) * 0.1, dev)
X = Planes.of(torch.randn(n, C, P, device=dev))
Y = Planes.of(torch.empty(n, Cout, P, device=dev))
Ha = (H + 7) // 8 * 8
store = torch.zeros(n, Ha, P, device=dev)
fmts = {"fp32": Planes.of(store[:, :H].contiguous()) if Ha != H else Planes.of(store),
        "f16 rows": Planes(store.view(-1), 0, H * P, n, H, P, f16=True),
        "f16 k-octets": Planes(store.view(-1), 0, Ha * P, n, H, P, f16=True, koct=True)}
def t(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for rnd in range(2):
    for name, hid in fmts.items():
        if name == "f16 k-octets" and not ops.uses_dma_tile(Cout):
            continue
        tp = t(lambda: ops.gemm(W1, X, hid, ops.EPI_GELU))
        tc = t(lambda: ops.gemm(W2, hid, Y, ops.EPI_NONE))
        print(f"{name:13s} producer {C}->{H}: {tp:7.1f} us {2.0*C*H*P*n/tp/1e6:6.1f} TF | consumer {H}->{Cout}: {tc:7.1f} us {2.0*H*Cout*P*n/tc/1e6:6.1f} TF", flush=True)
