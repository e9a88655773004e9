#!/usr/bin/env python3
"""The ROUND-4 single-product layer set on the round-5 kernels (like-for-like kernel delta of the round; the set itself is retired:
it reached 1.4e-3 of the flow on a hard seed).  8 clips, Sintel shape, graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from streamflow_amd import presets, synthetic as syn
from streamflow_amd.engine import HotPathEngine, HotPathWeights
dev = torch.device("cuda:0")
T, B, h, w = 4, 8, 55, 128
keep4 = ("gru.pw", "gru.ffn2_0", "gru.ffn2_2", "qkv", "proj", "fc1", "flow_head.pw", "flow_head.ffn2_0", "flow_head.ffn2_2")
names = list(HotPathWeights.PLAIN_LAYERS) + [f"{b}.{l}" for b in HotPathWeights.SK_BLOCKS for l in HotPathWeights.SK_LAYERS]
single4 = tuple(n for n in names if n not in keep4) + ("convc1.dw", "convc2.dw", "convf2.dw", "conv.dw")
P = syn.make_params(0, T)
fm, cn = syn.make_features(1000, B, T, h, w)
fm, cn = fm.to(dev), cn.to(dev)
for tag, kw in (("round-4 set", dict(presets.engine_kwargs("config2_fp16"), single_layers=single4)), ("round-5 set", presets.engine_kwargs("config2_mixed"))):
    eng = HotPathEngine(P, device=dev, T=T, use_graph=True, **kw)
    for _ in range(3):
        eng.forward(fm, cn, iters=15)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.forward(fm, cn, iters=15)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{tag}: {B * 3 / dt:.1f} ff/s ({1e3 * dt:.2f} ms/step)", flush=True)
    del eng
