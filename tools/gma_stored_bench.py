#!/usr/bin/env python3
"""GMA aggregation per iteration at the bench shape (24 images of 7040 px), kernels alone: the fused recompute kernel
(sf_gma_flash_aggregate, stored statistics) against the stored-weights kernel (sf_gma_stored_aggregate).  v is packed once
(sf_gma_flash_aggregate's own pack), then both run with v == NULL.   usage: gma_stored_bench.py [n_img] [P] [qk_products]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops
from streamflow_amd.ops import Planes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
P = int(sys.argv[2]) if len(sys.argv) > 2 else 7040
qkp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
cx = ops.Ctx(precision=ops.PRECISION_F16X2)
qk = torch.randn(n, 256, P, device=dev)
v = torch.randn(n, 128, P, device=dev)
mf = torch.randn(n, 128, P, device=dev)
out = torch.empty(n, 128, P, device=dev)
gamma = torch.tensor([0.5], device=dev)
ws = torch.empty(ops.gma_flash_ws_bytes(n, P), dtype=torch.uint8, device=dev)
pbuf = torch.empty(ops.gma_stored_p_bytes(n, P), dtype=torch.uint8, device=dev)
ops.gma_flash_pack_qk(Planes.of(qk), ws, 128 ** -0.5, stats_qk_products=qkp, cx=cx)
ops.gma_flash_aggregate(ws, Planes.of(v), Planes.of(mf), gamma, Planes.of(out), qkp, use_stats=True, cx=cx)   # packs v
ref = out.clone()


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


t_store = timed(lambda: ops.gma_flash_store_p(ws, pbuf, n, P, qkp, cx=cx), 3)
t_flash = timed(lambda: ops.gma_flash_aggregate(ws, None, Planes.of(mf), gamma, Planes.of(out), qkp, use_stats=True, cx=cx))
t_pv = timed(lambda: ops.gma_stored_aggregate(ws, pbuf, None, Planes.of(mf), gamma, Planes.of(out), cx=cx))
same = bool(torch.equal(out, ref))
gb = ops.gma_stored_p_bytes(n, P) / 1e9
print(f"n={n} P={P} qkp={qkp} lib={os.environ.get('SF_HIP_LIB', 'default')}: store_p {t_store:.1f} us | flash {t_flash:.1f} us | "
      f"stored {t_pv:.1f} us = {gb / t_pv * 1e6:.0f} GB/s of weights ({gb:.2f} GB) | bit-identical {same}")
