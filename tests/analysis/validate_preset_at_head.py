#!/usr/bin/env python3
"""The shipped presets at HEAD on every case of the round-5 selection (tests/analysis/preset_select_v2.py: headline seeds 11 - 13,
hard seeds 11 - 13, 31 - 37) and on its held-out validation seeds (21, 38 - 41), every case against the CPU oracle -- a re-check after
the kernels changed under the selection (temporal block, mask head and FFN pairs in one launch each, depthwise residual in the
centre tap).  Caps as in the selection: headline <= 1.25 x config2_fp16, hard <= 0.6e-3 of the mean flow (selection) / reported
(validation; the test bound is 1e-3).   usage: validate_preset_at_head.py > profiles/r05_preset_validation_at_head.jsonl"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import streamflow_oracle as orc, twins_oracle as two
from streamflow_amd import presets, synthetic as syn
from streamflow_amd.engine import HotPathEngine

dev = torch.device("cuda:0")
T = 4
torch.set_num_threads(min(64, len(os.sched_getaffinity(0))))


def make_case(kind, seed):
    if kind == "headline":
        P = syn.make_params(seed, T)
        fm, cn = syn.make_features(3000 + seed, 1, T, 55, 128)
        iters = 15
    else:
        ps, fs, a, b = (21, 24, 22, 23) if seed == 21 else (seed, 100 + seed, 200 + seed, 300 + seed)
        P = syn.make_params(ps, T)
        frames = torch.stack([(syn.randn(fs, f"frame{t}", (1, 3, 128, 192)).sigmoid() * 255.0) for t in range(T)], dim=1)
        imgs = 2 * (frames / 255.0) - 1.0
        fm = two.twins_csc_forward(imgs, syn.make_twins_params(a))
        cn = two.twins_csc_forward(imgs[:, :-1], syn.make_twins_params(b))
        iters = 4
    ref, _ = orc.hotpath_forward(fm, cn, P, iters)
    mag = float(torch.stack([r.norm(dim=1).mean() for r in ref]).mean())
    return dict(kind=kind, seed=seed, P=P, fd=fm.to(dev).contiguous(), cd=cn.to(dev).contiguous(), iters=iters, ref=ref, mag=mag)


def epe(case, preset):
    eng = HotPathEngine(case["P"], device=dev, T=T, **presets.engine_kwargs(preset))
    ups, _ = eng.forward(case["fd"], case["cd"], iters=case["iters"])
    return max(orc.epe(u.cpu(), r) for u, r in zip(ups, case["ref"]))


worst = {"headline_ratio": 0.0, "hard_selection_rel": 0.0, "hard_validation_rel": 0.0}
for role, kind, seeds in (("selection", "headline", (11, 12, 13)), ("selection", "hard", (11, 12, 13, 31, 32, 33, 34, 35, 36, 37)),
                          ("validation", "hard", (21, 38, 39, 40, 41))):
    for s in seeds:
        c = make_case(kind, s)
        e16, emix = epe(c, "config2_fp16"), epe(c, "config2_mixed")
        rec = {"role": role, "case": kind, "seed": s, "mean_flow_px": round(c["mag"], 2), "epe_config2_fp16": e16, "epe_config2_mixed": emix,
               "mixed_over_fp16": round(emix / e16, 3), "mixed_relative_to_flow": emix / max(1.0, c["mag"])}
        print(json.dumps(rec), flush=True)
        if kind == "headline":
            worst["headline_ratio"] = max(worst["headline_ratio"], emix / e16)
        else:
            k = "hard_selection_rel" if role == "selection" else "hard_validation_rel"
            worst[k] = max(worst[k], emix / max(1.0, c["mag"]))
print(json.dumps({"worst": worst, "caps": {"headline_ratio": 1.25, "hard_selection_rel": 0.6e-3, "hard_test_bound_rel": 1e-3}}), flush=True)
