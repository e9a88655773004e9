#!/usr/bin/env python3
"""Twins_CSC encoder (fnet on T frames + cnet on T-1 frames) at the Sintel shape: ms per clip and the per-kernel table.
usage: encoder_bench.py [clips] [shapes]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import ops, synthetic as syn
from streamflow_amd.encoders import Twins_CSC
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T, H, W = 4, 440, 1024
fnet, cnet = Twins_CSC().to(dev), Twins_CSC().to(dev)
fnet.svt.load_state_dict({k[4:]: v for k, v in syn.make_twins_params(1).items()}, strict=True)
cnet.svt.load_state_dict({k[4:]: v for k, v in syn.make_twins_params(2).items()}, strict=True)
x = (torch.rand(B, T, 3, H, W, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
for prec in ("f16x3", "f16x2"):
    ops.set_precision(prec)
    for _ in range(2):
        fnet(x); cnet(x[:, :-1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fnet(x); cnet(x[:, :-1])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    ops.PROFILER = ops.Profiler()
    ops.PROFILE_SHAPES = len(sys.argv) > 2 and sys.argv[2] == "shapes"
    fnet(x); cnet(x[:, :-1])
    summ = ops.PROFILER.summary()
    ops.PROFILER = None
    ops.PROFILE_SHAPES = False
    rows = sorted(((k, round(v["ms"], 3), v["launches"], round(v["bytes"] / v["ms"] / 1e6), round(v["flops"] / v["ms"] / 1e9, 1))
                   for k, v in summ.items()), key=lambda r: -r[1])       # name, ms, launches, algorithmic GB/s, TFLOP/s
    print(json.dumps({"precision": prec, "clips": B, "encoder_ms_per_clip": round(ms / B, 3), "kernels": rows}))
