#!/usr/bin/env python3
"""Numerical check of sf_gemm against torch on the GPU for a few shapes/epilogues (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes, PackedLinear
dev = torch.device("cuda:0")
torch.manual_seed(0)
for prec in ("f16x3", "fp32"):
    ops.set_precision(prec)
    for (M, K, P, n) in [(96, 64, 192, 2), (486, 324, 256, 1), (6, 576, 128, 1)]:
        w = torch.randn(M, K) / K ** 0.5; b = torch.randn(M) * 0.1
        W = PackedLinear(w, b, dev)
        x = torch.randn(n, K, P, device=dev); r = torch.randn(n, M, P, device=dev)
        y = torch.empty(n, M, P, device=dev)
        for name, e, kw in [("none", ops.EPI_NONE, {}), ("res", ops.EPI_RES, dict(R=Planes.of(r)))]:
            y.fill_(float("nan"))
            ops.gemm(W, Planes.of(x), Planes.of(y), e, **kw)
            ref = torch.einsum("mk,nkp->nmp", w.to(dev), x) + b.to(dev)[None, :, None]
            if e == ops.EPI_RES: ref = ref + r
            err = (y - ref).abs()
            bad = (err > 1e-3) | torch.isnan(y)
            print(prec, (M, K, P, n), name, "max err", float(err[~torch.isnan(err)].max()) if (~torch.isnan(err)).any() else None,
                  "bad", int(bad.sum()), "of", bad.numel())
            if bad.any():
                idx = bad.nonzero()[:6].tolist()
                print("   first bad (img,m,p):", idx, " nan:", int(torch.isnan(y).sum()))
                # is it a transposition inside 32x32 tiles?
                t = y[0, :32, :32]; rt = ref[0, :32, :32]
                print("   tile00 err", float((t - rt).abs().max()), " vs transposed", float((t - rt.t()).abs().max()))
