#!/usr/bin/env python3
"""f16x2 GEMMs with a k-octet fp16 B operand (the hand-over format of the conv stack) at the shapes that hold the GEMM time,
24 images of 7040 pixels.  Prints us and algorithmic TF per shape; SF_GEMM_BDIRECT=0/1 selects the kernel, SF_SINGLE=1 single-product weights, SF_GEMM_BD256=0/1 the 256-row tile.
usage: gemm_koct_bench.py [epi: none|gelu|koct]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
ALGO = int(os.environ.get("SF_ALGO", "0"))          # 0 auto, 1 tiled kernels, 2 activation-stationary kernel
n, P = int(os.environ.get("SF_N_IMG", "24")), 7040
dev = torch.device("cuda:0")
ops.set_precision("f16x2")
shapes = [(960, 640), (640, 960), (640, 640), (128, 960), (486, 324), (324, 486), (384, 256), (256, 384), (256, 256), (192, 128), (128, 192), (128, 128)]
if os.environ.get("SF_SHAPES"):                                 # e.g. SF_SHAPES=960x640,384x256
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["SF_SHAPES"].split(",")]
tot = 0.0
for M, K in shapes:
    W = PackedLinear(torch.randn(M, K, 1, 1) / K ** 0.5, torch.randn(M) * 0.1, dev)
    W.single = os.environ.get("SF_SINGLE", "0") == "1"          # single-product weights (a layer of the mixed preset)
    R = int(os.environ.get("SF_REPL", "1"))                     # experiment builds with -DSF_BSTAT_REPL=R: R copies of the planes
    if R > 1:
        W.hi, W.lo = torch.cat([W.hi] * R).contiguous(), torch.cat([W.lo] * R).contiguous()
    Ka = (K + 7) // 8 * 8
    xs = torch.randn(n, Ka, P, device=dev)
    X = Planes(torch.zeros(n * Ka * P // 2, device=dev), 0, Ka * P, n, K, P, f16=True, koct=True)
    ops.pack_koct(Planes.of(xs[:, :K].contiguous()) if Ka != K else Planes.of(xs), X)
    if mode == "koct":
        Ma = (M + 7) // 8 * 8
        Y = Planes(torch.zeros(n * Ma * P // 2, device=dev), 0, Ma * P, n, M, P, f16=True, koct=True)
        epi = ops.EPI_GELU
    elif mode == "dw1":                                         # an SK block's ffn*.2: residual + GELU + depthwise 1 x 1 + GELU, fp32 planes out
        Y = Planes.of(torch.empty(n, M, P, device=dev))
        epi = ops.EPI_RES_GELU_DW1
    else:
        Y = Planes.of(torch.empty(n, M, P, device=dev))
        epi = ops.EPI_GELU if mode == "gelu" else ops.EPI_NONE
    if mode == "koct" and not (ops.uses_dma_tile(M) if ALGO == 1 else ops.takes_koct(M, K)):
        continue
    if ALGO == 2 and (K > 640 or (mode == "dw1" and K > 384)):
        continue
    kw = {}
    if mode == "dw1":
        kw = dict(R=Planes.of(torch.randn(n, M, P, device=dev)), dw_w=torch.randn(M, device=dev) * 0.3, dw_b=torch.randn(M, device=dev) * 0.1)
    for _ in range(3):
        ops.gemm(W, X, Y, epi, algo=ALGO, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.gemm(W, X, Y, epi, algo=ALGO, **kw)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    tot += us
    print(f"M{M:4d} K{K:4d} {mode:5s}: {us:7.1f} us  {2.0 * M * K * P * n / us / 1e6:6.1f} TF", flush=True)
print(f"sum {tot:.1f} us")
