#!/usr/bin/env python3
"""Single-product depthwise weights on top of the config2_mixed preset: 15-iteration EPE at the headline shape (one clip,
vs the fp32 CPU oracle) with one block's K x K depthwise layer at a time, then cumulatively.  usage: dw_ablation.py [seed ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import streamflow_oracle as orc
from streamflow_amd import presets, synthetic as syn
from streamflow_amd.engine import HotPathEngine, HotPathWeights
dev = torch.device("cuda:0")
B, T, h, w, iters = 1, 4, 55, 128, 15
kw = presets.engine_kwargs("config2_mixed")
base_single = tuple(kw.pop("single_layers"))
dws = [b + ".dw" for b in HotPathWeights.SK_BLOCKS]
for seed in [int(a) for a in sys.argv[1:]] or [0]:
    P = syn.make_params(seed, T)
    fmaps, cnets = syn.make_features(1000 + seed, B, T, h, w)
    ups_o, _ = orc.hotpath_forward(fmaps, cnets, P, iters)
    fd, cd = fmaps.to(dev), cnets.to(dev)

    def epe(extra):
        eng = HotPathEngine(P, device=dev, T=T, single_layers=base_single + tuple(extra), **kw)
        ups, _ = eng.forward(fd, cd, iters=iters)
        return max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))

    print(json.dumps({"seed": seed, "extra": [], "epe": epe(())}), flush=True)
    res = {}
    for n in dws:
        res[n] = epe((n,))
        print(json.dumps({"seed": seed, "extra": [n], "epe": res[n]}), flush=True)
    cum = []
    for n in sorted(dws, key=lambda n: res[n]):
        cum.append(n)
        print(json.dumps({"seed": seed, "cumulative": list(cum), "epe": epe(cum)}), flush=True)
