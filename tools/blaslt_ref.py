#!/usr/bin/env python3
"""What does the vendor library reach on the update block's GEMM shapes?  (a reference ceiling for csrc/gemm_split.hip under
the same power limit; fp16 operands, fp32 accumulate, no epilogue)  usage: blaslt_ref.py"""
import torch, time
dev = torch.device("cuda:0")
shapes = [(960, 640), (640, 960), (640, 640), (384, 256), (486, 324), (256, 384), (128, 960)]
N = 24 * 7040
for M, K in shapes:
    W = torch.randn(M, K, device=dev, dtype=torch.float16) / K ** 0.5
    for layout in ("KxN", "NxK"):
        X = torch.randn(K, N, device=dev, dtype=torch.float16) if layout == "KxN" else torch.randn(N, K, device=dev, dtype=torch.float16)
        f = (lambda: W @ X) if layout == "KxN" else (lambda: X @ W.t())
        for _ in range(3): f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 20 * 1e6
        print(f"M{M} K{K} N{N} activations {layout}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF")
