#!/usr/bin/env python3
"""Diagnostic: error of the matrix-core sub-sample attention against float64 under operand rescalings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes
dev = torch.device("cuda:0")
heads, N, M, n = 8, 65, 100, 3
C = heads * 32
g = torch.Generator().manual_seed(0)
q0 = torch.randn(n, C, N, generator=g); kv0 = torch.randn(n, 2 * C, M, generator=g)
def run(prec, qs=1.0, ks=1.0, vs=1.0):
    q = q0 * qs; kv = kv0.clone(); kv[:, :C] *= ks; kv[:, C:] *= vs
    out = torch.full((n, C, N), float("nan"), device=dev)
    prev = ops.set_precision(prec)
    ops.subsample_attn(Planes.of(q.to(dev)), Planes.of(kv.to(dev)), Planes.of(out), heads)
    ops.set_precision(prev)
    qd = q.double().view(n, heads, 32, N); kd = kv[:, :C].double().view(n, heads, 32, M); vd = kv[:, C:].double().view(n, heads, 32, M)
    ref = torch.einsum("bhnm,bhdm->bhdn", torch.softmax(torch.einsum("bhdn,bhdm->bhnm", qd, kd) * 32 ** -0.5, -1), vd).reshape(n, C, N)
    return ((out.cpu().double() - ref).abs().max() / vs).item()
for prec in ("fp32", "f16x3", "f16x2"):
    print(prec, "base %.3e" % run(prec), "q*16,k/16 %.3e" % run(prec, 16, 1 / 16), "q/16,k*16 %.3e" % run(prec, 1 / 16, 16),
          "v*64 %.3e" % run(prec, vs=64), "q*16,k*4,/64.. %.3e" % run(prec, 64, 1 / 64))
