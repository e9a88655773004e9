#!/usr/bin/env python3
"""Frames -> flows at a small shape: EPE against the chained CPU oracles for combinations of encoder precision class and
hot-path preset (which side carries an end-to-end deviation?).  usage: e2e_diag.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import streamflow_oracle as orc, twins_oracle as two
from streamflow_amd import ops, presets, synthetic as syn
from streamflow_amd.encoders import Twins_CSC
from streamflow_amd.engine import HotPathEngine
dev = torch.device("cuda:0")
B, T, H, W, iters = 1, 4, 128, 192, 4
hot, ef, ec = syn.make_params(21, T), syn.make_twins_params(22), syn.make_twins_params(23)
frames = torch.stack([(syn.randn(24, f"frame{t}", (B, 3, H, W)).sigmoid() * 255.0) for t in range(T)], dim=1)
imgs = 2 * (frames / 255.0) - 1.0
fm_o = two.twins_csc_forward(imgs, ef); cn_o = two.twins_csc_forward(imgs[:, :-1], ec)
ups_o, _ = orc.hotpath_forward(fm_o, cn_o, hot, iters)
fnet, cnet = Twins_CSC().to(dev), Twins_CSC().to(dev)
fnet.load_state_dict({k: v for k, v in ef.items()}, strict=True); cnet.load_state_dict({k: v for k, v in ec.items()}, strict=True)
feats = {}
for enc_prec, koct in (("f16x3", "1"), ("f16x2", "1"), ("f16x2", "0"), ("fp32", "1")):
    fnet.koct_handover = cnet.koct_handover = koct == "1"
    fm, cn = fnet(imgs.to(dev), precision=enc_prec), cnet(imgs[:, :-1].to(dev), precision=enc_prec)
    rel = lambda a, b: ((a.cpu() - b).abs().max() / b.abs().max()).item()
    print(f"encoder {enc_prec} koct={koct}: fmap max err / max |f| = {rel(fm, fm_o):.2e} (max |f| {fm_o.abs().max():.2f}), cnet {rel(cn, cn_o):.2e}")
    feats[(enc_prec, koct)] = (fm, cn)
for (enc_prec, koct), (fm, cn) in feats.items():
    for preset in ("fp32_class", "config2_mixed"):
        eng = HotPathEngine(hot, device=dev, T=T, **presets.engine_kwargs(preset))
        ups, _ = eng.forward(fm.float().contiguous(), cn.float().contiguous(), iters=iters)
        e = max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))
        print(f"  encoder {enc_prec} koct={koct} + loop {preset}: max EPE {e:.2e}")
fm, cn = feats[("f16x3", "1")]
print("which part of the config-2 arithmetic carries it (exact features):")
for kw in (dict(precision="f16x3", corr_dtype="f16", gma_mode="auto", flash_qk_products=3),
           dict(precision="f16x2", corr_dtype="f32", gma_mode="auto", flash_qk_products=3),
           dict(precision="f16x3", corr_dtype="f32", gma_mode="flash", flash_qk_products=1),
           dict(precision="f16x2", corr_dtype="f16", gma_mode="flash", flash_qk_products=1)):
    eng = HotPathEngine(hot, device=dev, T=T, **kw)
    ups, _ = eng.forward(fm.float().contiguous(), cn.float().contiguous(), iters=iters)
    e = max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))
    print(f"  {kw}: max EPE {e:.2e}")
for scale in (1.0, 0.25):
    f2, c2 = fm_o * scale, cn_o
    uo, _ = orc.hotpath_forward(f2, c2, hot, iters)
    eng = HotPathEngine(hot, device=dev, T=T, **presets.engine_kwargs("config2_fp16"))
    ups, _ = eng.forward(f2.to(dev).contiguous(), c2.to(dev).contiguous(), iters=iters)
    print(f"  fmaps x {scale}: config2_fp16 max EPE {max(orc.epe(u.cpu(), o) for u, o in zip(ups, uo)):.2e}, mean |flow| {uo[0].norm(dim=1).mean():.2f} px")
