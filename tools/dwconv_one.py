#!/usr/bin/env python3
"""Depthwise conv + residual + GELU alone (timing).  argv: C n_img k [precision] [f16out] [single] [f16in]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes
C, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ops.set_precision(sys.argv[4] if len(sys.argv) > 4 else "f16x3")
h, w = 55, 128
dev = torch.device("cuda:0")
X = Planes.of(torch.randn(n, C, h * w, device=dev)); Y = Planes.of(torch.empty(n, C, h * w, device=dev))
single = "single" in sys.argv
if "f16in" in sys.argv:                 # x as fp16 rows (sf_dwconv_res_gelu_f16in; needs f16out and f16x2)
    xs = torch.randn(n, C, h * w, device=dev).half().contiguous()
    X = Planes(xs.view(-1).view(torch.float32), 0, C * h * w, n, C, h * w, f16=True)
if "f16out" in sys.argv:
    Y = Planes(torch.empty(n * C * h * w // 2, device=dev), 0, C * h * w, n, C, h * w, f16=True)
wgt = (torch.randn(C, k, k, device=dev) / k).contiguous(); b = torch.randn(C, device=dev) * 0.1
for _ in range(3):
    ops.dwconv_res_gelu(X, wgt, b, Y, h, w, k, single=single)
if os.environ.get("SF_DW_TS"):          # needs tools/build_variant.sh dwt conv.hip -DSF_DW_TIMERS + SF_HIP_LIB=.../variant_dwt.so
    torch.cuda.synchronize()
    ts = torch.zeros(65536 * 4, dtype=torch.int64, device=dev)
    os.environ["SF_DW_TS_BUF"] = str(ts.data_ptr())
    ops.dwconv_res_gelu(X, wgt, b, Y, h, w, k, single=single); torch.cuda.synchronize()
    os.environ.pop("SF_DW_TS_BUF")
    t = ts.view(-1, 4).cpu().double(); t = t[t[:, 3] > 0]
    print(f"workgroups {t.shape[0]}, images per workgroup {t[:, 3].mean().item():.1f}")
    print(f"total {t[:, 0].mean().item():.0f} cycles per workgroup; per image: stage {(t[:, 1] / t[:, 3]).mean().item():.0f}, "
          f"compute+store {(t[:, 2] / t[:, 3]).mean().item():.0f}; setup {(t[:, 0] - t[:, 1] - t[:, 2]).mean().item():.0f}")
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
reps = 20
for _ in range(reps):
    ops.dwconv_res_gelu(X, wgt, b, Y, h, w, k, single=single)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / reps
print(f"dwconv{k} C={C} n={n} {ops.precision_name()}: {us:.1f} us  {2.0 * k * k * n * C * h * w / us / 1e6:.1f} TF-equivalent  {((2.0 if X.f16 else 4.0) + (2.0 if Y.f16 else 4.0)) * n * C * h * w / us / 1e3:.0f} GB/s")

