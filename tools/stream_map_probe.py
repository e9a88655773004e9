#!/usr/bin/env python3
"""HIP maps streams onto a few hardware queues in creation order.  How much does the mapping of the engine's streams matter?
k dummy streams are created before the engine (shifting the mapping); graph replay, 8 clips, bench preset.  usage: stream_map_probe.py k [k ...]
(run one k per process: GPU_MAX_HW_QUEUES is read when the runtime starts)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamflow_amd import presets, synthetic as syn
from streamflow_amd.engine import HotPathEngine
dev = torch.device("cuda:0")
k = int(sys.argv[1])
dummies = [torch.cuda.Stream(device=dev) for _ in range(k)]
for d in dummies:
    with torch.cuda.stream(d):
        torch.zeros(16, device=dev).add_(1)          # (a stream gets its queue when it is first used)
torch.cuda.synchronize()
T, B, h, w = 4, 8, 55, 128
params = syn.make_params(0, T)
fm, cn = (t.to(dev) for t in syn.make_features(1000, B, T, h, w))
eng = HotPathEngine(params, device=dev, T=T, use_graph=True, **presets.engine_kwargs(presets.BENCH_PRESET))
for _ in range(3): eng.forward(fm, cn, iters=15)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(15): eng.forward(fm, cn, iters=15)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 15
print(f"dummy streams {k}, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: {1e3 * dt:.2f} ms/step = {24 / dt:.1f} ff/s", flush=True)
