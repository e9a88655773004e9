#!/usr/bin/env python3
"""Phase timers of sf_ffn_pair (a -DSF_PAIR_TIMERS build: tools/build_variant.sh pair_timers ffn_pair.hip -DSF_PAIR_TIMERS; run with
SF_HIP_LIB=streamflow_amd/csrc/build/variant_pair_timers.so): mean cycles per wave of prologue / DMA issue / MFMA + GELU block /
LDS + DMA drain / barrier / whole loop / epilogue.   usage: ffn_pair_timers.py [mode K1 H M2 pm1 pm2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
buf = torch.zeros(8192 * 8 * 8, dtype=torch.int64, device=dev)
os.environ["SF_PAIR_TS_BUF"] = str(buf.data_ptr())
from streamflow_amd import ops
from streamflow_amd.ops import PackedLinear, PackedPair, Planes
a = [int(x) for x in sys.argv[1:]] or [1, 256, 384, 256, 1, 1]
mode, K1, H, M2, pm1, pm2 = a
n, P = 24, 7040
cx = ops.Ctx(precision=ops.PRECISION_F16X2)
A1 = PackedLinear(torch.randn(H, K1, 1, 1) / K1 ** 0.5, torch.randn(H) * 0.1, dev)
A2 = PackedLinear(torch.randn(M2, H, 1, 1) / H ** 0.5, torch.randn(M2) * 0.1, dev)
A1.single, A2.single = pm1 == 1, pm2 == 1
pair = PackedPair(A1, A2)
ra = (K1 + 7) // 8 * 8
X = Planes(torch.zeros(n * ra * P // 2 + 8, device=dev), 0, ra * P, n, K1, P, f16=True, koct=True)
X.base.view(torch.float16)[: n * ra * P].copy_(torch.randn(n * ra * P, device=dev).half())
if mode == 1:
    Y = Planes(torch.zeros(n * M2 * P // 2 + 8, device=dev), 0, M2 * P, n, M2, P, f16=True)
    kw = dict(dw_w=torch.randn(M2, device=dev), dw_b=torch.randn(M2, device=dev))
else:
    Y = Planes(torch.zeros(n * ((M2 + 7) // 8 * 8) * P // 2 + 8, device=dev), 0, (M2 + 7) // 8 * 8 * P, n, M2, P, f16=True, koct=True)
    kw = {}
for _ in range(3):
    ops.ffn_pair(pair, X, Y, mode, cx=cx, **kw)
torch.cuda.synchronize()
buf.zero_()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); ops.ffn_pair(pair, X, Y, mode, cx=cx, **kw); e.record(); torch.cuda.synchronize()
t = buf.view(-1, 8).double()
t = t[t[:, 7] > 0]
names = ["prologue", "dma_issue", "mfma+gelu", "drain", "barrier", "loop", "epilogue", "stages"]
m = t.mean(dim=0)
print(f"mode {mode} {K1}->{H}->{M2} pm=({pm1},{pm2}): {s.elapsed_time(e) * 1e3:.1f} us (timer build), {len(t)} waves sampled")
print("  " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, m.tolist())))
st = m[7].item()
print("  per stage: " + ", ".join(f"{nm} {v / st:.0f}" for nm, v in zip(names[1:5], m[1:5].tolist())))
