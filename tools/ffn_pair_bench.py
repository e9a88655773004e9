#!/usr/bin/env python3
"""sf_ffn_pair against the two sf_gemm launches it replaces, at the bench shape (24 images of 7040 pixels; flow head: 8).
usage: ffn_pair_bench.py [pm1 pm2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import PackedLinear, PackedPair, Planes
dev = torch.device("cuda:0")
pm = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 1)
n, P = 24, 7040
cx = ops.Ctx(precision=ops.PRECISION_F16X2)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def koct(rows):
    ra = (rows + 7) // 8 * 8
    return Planes(torch.randn(n * ra * P // 2 + 8, device=dev).mul_(0.01), 0, ra * P, n, rows, P, f16=True, koct=True)


for mode, K1, H, M2 in ((1, 128, 192, 128), (1, 256, 384, 256), (1, 324, 486, 324), (0, 128, 192, 64), (0, 256, 384, 192),
                        (0, 256, 384, 126), (0, 324, 486, 256)):
    A1 = PackedLinear(torch.randn(H, K1, 1, 1) / K1 ** 0.5, torch.randn(H) * 0.1, dev)
    A2 = PackedLinear(torch.randn(M2, H, 1, 1) / H ** 0.5, torch.randn(M2) * 0.1, dev)
    A1.single, A2.single = pm[0] == 1, pm[1] == 1
    pair = PackedPair(A1, A2)
    X = koct(K1)
    X.base.view(torch.float16)[: n * ((K1 + 7) // 8 * 8) * P].copy_(torch.randn(n * ((K1 + 7) // 8 * 8) * P, device=dev).half())
    hid = koct(H)
    dw_w, dw_b = torch.randn(M2, device=dev) * 0.5, torch.randn(M2, device=dev) * 0.1
    if mode == 1:
        Y = Planes(torch.zeros(n * M2 * P // 2 + 8, device=dev), 0, M2 * P, n, M2, P, f16=True)
        two = lambda: (ops.gemm(A1, X, hid, ops.EPI_GELU, cx=cx),
                       ops.gemm(A2, hid, Y, ops.EPI_RES_GELU_DW1, R=X, dw_w=dw_w, dw_b=dw_b, cx=cx))
        one = lambda: ops.ffn_pair(pair, X, Y, 1, dw_w=dw_w, dw_b=dw_b, cx=cx)
    elif M2 % 8 == 0:
        Y = koct(M2)
        two = lambda: (ops.gemm(A1, X, hid, ops.EPI_GELU, cx=cx), ops.gemm(A2, hid, Y, ops.EPI_NONE, cx=cx))
        one = lambda: ops.ffn_pair(pair, X, Y, 0, cx=cx)
    else:
        from dataclasses import replace
        y32 = torch.zeros(n, M2, P, device=dev)
        Y = replace(Planes.of(y32), shadow=ops.new_shadow(Planes.of(y32), dev))
        two = lambda: (ops.gemm(A1, X, hid, ops.EPI_GELU, cx=cx), ops.gemm(A2, hid, Y, ops.EPI_NONE, cx=cx))
        one = lambda: ops.ffn_pair(pair, X, Y, 0, cx=cx)
    t2, t1 = timed(two), timed(one)
    fl = 2.0 * n * P * (K1 * H * pm[0] + H * M2 * pm[1])
    print(f"mode {mode} {K1}->{H}->{M2} pm={pm}: two launches {t2:7.1f} us, pair {t1:7.1f} us ({fl / t1 / 1e6:6.0f} TF issued)", flush=True)
