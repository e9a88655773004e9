#!/usr/bin/env python3
"""Which layers of the config-2 class may use ONE fp16 weight (single MFMA product) instead of hi + lo?  Round-5 redo of
tools/preset_select.py with every case referenced to the CPU ORACLE (round 4's run was taken on a library with a defect in the
k-octet epilogue -- profiles/r05_hard_case_at_88f9fb4.jsonl -- and against the fp32-class engine).

Selection cases (none of them is a bench or test input):
  * headline: seeded unit-scale features at the Sintel grid (55 x 128, T = 4, 15 iterations, one clip), seeds 11 12 13;
  * hard: frames -> exact Twins_CSC features of a random-weight encoder -> 128 x 192, 4 iterations, seeds 11 12 13 31 .. 37
    (a first pass on six hard seeds, profiles/r05_preset_select_pass1.jsonl, left two of three validation seeds at 0.9e-3 of the
    flow: the selection set was widened to ten).
Caps: headline cases may lose 25 % against the all-split preset (config2_fp16); hard cases must stay under 0.6e-3 of the mean flow
magnitude (the test bound is 1e-3 of it; the reference's own fp16-autocast arithmetic sits at 0.75 - 1.2e-3 of the flow on these
inputs, tests/analysis/autocast_emulation.py).  Layers are ranked by the worst share of the cap's head-room their single-product
form uses up alone, then admitted greedily in that order while EVERY case stays under its cap (a layer that breaks a cap is
skipped, the scan goes on).  Depthwise layers ('<block>.dw') go through the same procedure after the GEMM layers.
Validation cases (reported only): hard seeds 21 38 39 40 41.   usage: preset_select_v2.py > profiles/r05_preset_select.jsonl"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import streamflow_oracle as orc, twins_oracle as two
from streamflow_amd import presets, synthetic as syn
from streamflow_amd.engine import HotPathEngine, HotPathWeights

dev = torch.device("cuda:0")
T = 4
kw16 = presets.engine_kwargs("config2_fp16")
HEAD_SEEDS, HARD_SEEDS, VALID_HARD = (11, 12, 13), (11, 12, 13, 31, 32, 33, 34, 35, 36, 37), (21, 38, 39, 40, 41)
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


def epe(case, single):
    eng = HotPathEngine(case["P"], device=dev, T=T, **dict(kw16, single_layers=tuple(single)))
    ups, _ = eng.forward(case["fd"], case["cd"], iters=case["iters"])
    return max(orc.epe(u.cpu(), r) for u, r in zip(ups, case["ref"]))


cases = [make_case("headline", s) for s in HEAD_SEEDS] + [make_case("hard", s) for s in HARD_SEEDS]
valid = [make_case("hard", s) for s in VALID_HARD]
base = [epe(c, ()) for c in cases]
cap = [1.25 * b if c["kind"] == "headline" else max(1.25 * b, 0.6e-3 * max(1.0, c["mag"])) for c, b in zip(cases, base)]
for c, b, k in zip(cases, base, cap):
    print(json.dumps({"case": c["kind"], "seed": c["seed"], "mean_flow_px": round(c["mag"], 2), "epe_config2_fp16": b, "cap": k}), flush=True)
gemm_layers = sorted(HotPathEngine(cases[0]["P"], device=dev, T=T, **kw16).W.layers())
dw_layers = [b + ".dw" for b in HotPathWeights.SK_BLOCKS]
use = {}
for n in gemm_layers + dw_layers:
    e = [epe(c, (n,)) for c in cases]
    share = [(x - b) / (k - b) for x, b, k in zip(e, base, cap)]
    use[n] = max(share)
    print(json.dumps({"layer": n, "worst_headroom_share": round(use[n], 4), "epe_alone": e}), flush=True)
chosen = []
for group in (gemm_layers, dw_layers):
    for n in sorted(group, key=lambda n: use[n]):
        trial = chosen + [n]
        e = [epe(c, trial) for c in cases]
        worst = max(x / k for x, k in zip(e, cap))
        ok = worst <= 1.0
        print(json.dumps({"try": n, "n_single": len(trial), "worst_epe_over_cap": round(worst, 4), "accepted": ok}), flush=True)
        if ok:
            chosen = trial
final = [epe(c, chosen) for c in cases]
vfinal = [{"seed": c["seed"], "mean_flow_px": round(c["mag"], 2), "epe_selected": epe(c, chosen), "epe_config2_fp16": epe(c, ())} for c in valid]
print(json.dumps({"single": chosen, "keep_split": [n for n in gemm_layers if n not in chosen],
                  "single_depthwise": [n for n in chosen if n.endswith(".dw")],
                  "epe_selected": final, "epe_config2_fp16": base, "cap": cap,
                  "mean_flow_px": [round(c["mag"], 2) for c in cases], "validation_hard": vfinal}), flush=True)
