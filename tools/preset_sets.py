#!/usr/bin/env python3
"""EPE (15 iterations, headline shape, one clip, vs the fp32 CPU oracle) and step time of candidate single-product layer sets
on several seeds.  usage: preset_sets.py seed [seed ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import streamflow_oracle as orc
from streamflow_amd import presets, synthetic as syn
from streamflow_amd.engine import HotPathEngine
dev = torch.device("cuda:0")
B, T, h, w, iters = 1, 4, 55, 128, 15
KEEP = {"k3": ["gru.ffn2_2", "flow_head.ffn2_0", "flow_head.ffn2_2"],
        "k5": ["gru.ffn2_0", "gru.ffn2_2", "flow_head.pw", "flow_head.ffn2_0", "flow_head.ffn2_2"],
        "k6": ["gru.pw", "gru.ffn2_0", "gru.ffn2_2", "flow_head.pw", "flow_head.ffn2_0", "flow_head.ffn2_2"],
        "k9": ["gru.pw", "gru.ffn2_0", "gru.ffn2_2", "flow_head.pw", "flow_head.ffn2_0", "flow_head.ffn2_2", "qkv", "proj", "fc1"]}
kw = presets.engine_kwargs("config2_fp16")
for seed in [int(a) for a in sys.argv[1:]] or [1]:
    P = syn.make_params(seed, T)
    fmaps, cnets = syn.make_features(2000 + seed, B, T, h, w)
    ups_o, _ = orc.hotpath_forward(fmaps, cnets, P, iters)
    fd, cd = fmaps.to(dev), cnets.to(dev)
    names = sorted(HotPathEngine(P, device=dev, T=T, **kw).W.layers())
    for tag, keep in [("none", None)] + list(KEEP.items()):
        single = () if keep is None else tuple(n for n in names if n not in keep)
        eng = HotPathEngine(P, device=dev, T=T, single_layers=single, **kw)
        ups, _ = eng.forward(fd, cd, iters=iters)
        e = max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))
        print(json.dumps({"seed": seed, "set": tag, "single_layers": len(single), "epe": e}), flush=True)
