#!/usr/bin/env python3
"""Which part of an engine forward keeps sf_clock_probe from running beside it?  wall time of [probe(150 ms) || work] per variant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from streamflow_amd import ops, presets, synthetic as syn
from streamflow_amd.engine import HotPathEngine, EngineOptions
dev = torch.device("cuda:0")
T, B, h, w = 4, 8, 55, 128
params = syn.make_params(0, T)
cfg = presets.engine_kwargs(presets.BENCH_PRESET)
fm, cn = (t.to(dev) for t in syn.make_features(1000, B, T, h, w))
side, out = torch.cuda.Stream(device=dev), torch.zeros(2, dtype=torch.int64, device=dev)
def run(name, fn, us=150000):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t_alone = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.cuda.stream(side): ops.clock_probe(out, us)
    fn(); torch.cuda.synchronize(); t_both = time.perf_counter() - t0
    c = out.tolist()
    print(f"{name:44s} alone {1e3 * t_alone:7.1f} ms, beside a {us // 1000} ms probe {1e3 * t_both:7.1f} ms, probe read {100.0 * c[0] / c[1]:.0f} MHz", flush=True)
for opts in (dict(parallel_branches=False, split_solo=0), dict(parallel_branches=True, split_solo=0), dict()):
    eng = HotPathEngine(params, device=dev, T=T, use_graph=False, options=EngineOptions(**opts), **cfg)
    run(f"eager forward x2 {opts}", lambda: [eng.forward(fm, cn, iters=15) for _ in range(2)])
eng = HotPathEngine(params, device=dev, T=T, use_graph=True, **cfg)
run("graph replay x2", lambda: [eng.forward(fm, cn, iters=15) for _ in range(2)])
a = torch.randn(64, 1 << 20, device=dev); b = torch.empty_like(a)
run("torch copy_ x400", lambda: [b.copy_(a) for _ in range(400)])
run("torch tanh x400", lambda: [torch.tanh(a, out=b) for _ in range(400)])
