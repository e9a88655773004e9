#!/usr/bin/env python3
"""Probe for an OFFSET two-chain schedule: two engines of 4 clips each (one chain per engine: split_solo = 0), each replayed from its
own host thread (graph launches do not serialise behind one another), the second thread started `offset` ms late so that the engines
run different phases of the iteration side by side -- against the shipped schedule (one engine, 8 clips, two lockstep half-batch
chains).   usage: offset_probe.py [steps]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dataclasses import replace
from streamflow_amd import presets, synthetic as syn
from streamflow_amd.engine import EngineOptions, HotPathEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
T, h, w, iters = 4, 55, 128, 15
params = syn.make_params(0, T)
kw = presets.engine_kwargs(presets.BENCH_PRESET)
fm8, cn8 = (t.to(dev) for t in syn.make_features(1000, 8, T, h, w))


def run_single():
    eng = HotPathEngine(params, device=dev, T=T, use_graph=True, **kw)
    for _ in range(3):
        eng.forward(fm8, cn8, iters=iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward(fm8, cn8, iters=iters)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def run_two(offset_ms, split_solo):
    engs = [HotPathEngine(params, device=dev, T=T, use_graph=True, options=EngineOptions(split_solo=split_solo), **kw) for _ in range(2)]
    parts = [(fm8[:4].contiguous(), cn8[:4].contiguous()), (fm8[4:].contiguous(), cn8[4:].contiguous())]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for e, (f, c), s in zip(engs, parts, streams):
        with torch.cuda.stream(s):
            for _ in range(3):
                e.forward(f, c, iters=iters)
    torch.cuda.synchronize()
    go = threading.Barrier(3)

    def worker(i):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[i]):
            go.wait()
            if i == 1 and offset_ms > 0:
                time.sleep(offset_ms / 1e3)
            for _ in range(steps):
                engs[i].forward(*parts[i], iters=iters)
            streams[i].synchronize()

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0 - offset_ms / 1e3 * 0) / steps


base = run_single()
print(f"one engine, 8 clips, two lockstep chains: {1e3 * base:.2f} ms per 8 clips = {24 / base:.1f} ff/s")
for ss in (0, 2):
    for off in (0, 1, 2, 8, 16, 31):
        t = run_two(off, ss)
        print(f"two engines x 4 clips (split_solo={ss}), two host threads, offset {off:2d} ms: {1e3 * t:.2f} ms per 8 clips = {24 / t:.1f} ff/s "
              f"(incl. the offset once over {steps} steps)")
