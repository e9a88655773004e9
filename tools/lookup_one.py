#!/usr/bin/env python3
"""Correlation build + N lookups at the Sintel shape (for rocprofv3 --pmc runs / timing)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops, synthetic as syn
from streamflow_amd.engine import HotPathEngine
dev = torch.device("cuda:0")
B, T, h, w = 1, 4, 55, 128
P = syn.make_params(0, T)
fm, cn = syn.make_features(1, B, T, h, w)
eng = HotPathEngine(P, device=dev, T=T)
ups, low = eng.forward(fm.to(dev), cn.to(dev), iters=2)        # realistic coords1
pl = eng.plan(B, h, w, 256)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.corr_lookup(pl.lvls, pl.lvl_pair_stride, pl.coords1, pl.corr, B, T - 1, h, w)
s.record()
for _ in range(reps):
    ops.corr_lookup(pl.lvls, pl.lvl_pair_stride, pl.coords1, pl.corr, B, T - 1, h, w)
e.record(); torch.cuda.synchronize()
print("lookup us", s.elapsed_time(e) * 1e3 / reps)
