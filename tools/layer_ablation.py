#!/usr/bin/env python3
"""Per-layer sensitivity of the final flows to SINGLE-product weights (weights rounded once to fp16) in the f16x2 arithmetic:
headline shape (55 x 128 grid, T = 4, 15 iterations, one clip), EPE vs the fp32 CPU oracle, one layer at a time, then the
cumulative set in order of increasing damage.  Writes JSON lines.  usage: layer_ablation.py [seed | hard]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import streamflow_oracle as orc
from streamflow_amd import presets, synthetic as syn
from streamflow_amd.engine import HotPathEngine
seed = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "hard" else 0
dev = torch.device("cuda:0")
if "hard" in sys.argv:
    # the ill-conditioned case of DESIGN.md 5c: frames -> random-init Twins_CSC features (|f| up to 19) at 128 x 192, 4
    # iterations, flows of 22-61 px (what tests/test_gpu_parity.py::test_real_frames_end_to_end_with_twins_encoder runs)
    from oracle import twins_oracle as two
    B, T, iters = 1, 4, 4
    P = syn.make_params(21, T)
    frames = torch.stack([(syn.randn(24, f"frame{t}", (B, 3, 128, 192)).sigmoid() * 255.0) for t in range(T)], dim=1)
    imgs = 2 * (frames / 255.0) - 1.0
    fmaps = two.twins_csc_forward(imgs, syn.make_twins_params(22))
    cnets = two.twins_csc_forward(imgs[:, :-1], syn.make_twins_params(23))
else:
    B, T, h, w, iters = 1, 4, 55, 128, 15
    P = syn.make_params(seed, T)
    fmaps, cnets = syn.make_features(1000 + seed, B, T, h, w)
t0 = time.time()
ups_o, _ = orc.hotpath_forward(fmaps, cnets, P, iters)
print(json.dumps({"oracle_s": round(time.time() - t0, 1)}), flush=True)
fd, cd = fmaps.to(dev), cnets.to(dev)
kw = presets.engine_kwargs("config2_fp16")


def epe(single):
    eng = HotPathEngine(P, device=dev, T=T, single_layers=single, **kw)
    ups, _ = eng.forward(fd, cd, iters=iters)
    return max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))


base = epe(())
print(json.dumps({"layers": [], "epe": base}), flush=True)
names = sorted(HotPathEngine(P, device=dev, T=T, **kw).W.layers())
res = {}
for n in names:
    res[n] = epe((n,))
    print(json.dumps({"layers": [n], "epe": res[n]}), flush=True)
order = sorted(names, key=lambda n: res[n])
cum = []
for n in order:
    cum.append(n)
    e = epe(tuple(cum))
    print(json.dumps({"cumulative": len(cum), "added": n, "epe": e}), flush=True)
print(json.dumps({"all": epe(("all",))}), flush=True)
