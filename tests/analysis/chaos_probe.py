#!/usr/bin/env python3
"""Is the ill-conditioned input class chaotic?  (VERDICT r4 #1b)  CPU only: the CPU oracle's own flows under perturbations of its
inputs -- 1 ulp (random sign) and 2^-20 / 2^-11 relative (uniform) of the encoder features `fmaps`, 2^-20 / 2^-11 relative of the
context features `cnets`, and both rounded to fp16 -- on the hard seeds (frames -> exact Twins_CSC features of a random-weight
encoder -> 128 x 192, 4 iterations).  If the oracle moved by O(1 px) under 2^-11 the cases would be ill-posed; it moves by <= 1.1e-3.
Writes JSON lines (profiles/r05_chaos_test.jsonl is a copy).   usage: chaos_probe.py [seed ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import streamflow_oracle as orc, twins_oracle as two
from streamflow_amd import synthetic as syn

torch.set_num_threads(min(16, os.cpu_count() or 1))
T = 4


def case(seed):
    ps, fs, a, b = (21, 24, 22, 23) if seed == 21 else (seed, 100 + seed, 200 + seed, 300 + seed)
    P = syn.make_params(ps, T)
    frames = torch.stack([(syn.randn(fs, f"frame{t}", (1, 3, 128, 192)).sigmoid() * 255.0) for t in range(T)], dim=1)
    imgs = 2 * (frames / 255.0) - 1.0
    return P, two.twins_csc_forward(imgs, syn.make_twins_params(a)), two.twins_csc_forward(imgs[:, :-1], syn.make_twins_params(b))


run = lambda P, fm, cn: orc.hotpath_forward(fm, cn, P, 4)[0]
epe = lambda a, b: max(orc.epe(x, y) for x, y in zip(a, b))
for seed in [int(a) for a in sys.argv[1:]] or [21, 11, 12, 13, 31, 32]:
    P, fm, cn = case(seed)
    base = run(P, fm, cn)
    g = torch.Generator().manual_seed(5)
    out = {"seed": seed, "mean_flow_px": round(float(torch.stack([o.norm(dim=1).mean() for o in base]).mean()), 2),
           "fmaps_absmax": round(float(fm.abs().max()), 2), "cnets_absmax": round(float(cn.abs().max()), 2)}
    sgn = (torch.randint(0, 2, fm.shape, generator=g) * 2 - 1).to(torch.int32)
    out["epe_fmaps_1ulp"] = epe(run(P, (fm.view(torch.int32) + sgn).view(torch.float32), cn), base)
    for e in (20, 11):
        out[f"epe_fmaps_rel_2^-{e}"] = epe(run(P, fm * (1 + torch.empty_like(fm).uniform_(-1, 1, generator=g) * 2.0 ** -e), cn), base)
        out[f"epe_cnets_rel_2^-{e}"] = epe(run(P, fm, cn * (1 + torch.empty_like(cn).uniform_(-1, 1, generator=g) * 2.0 ** -e)), base)
    out["epe_fmaps_fp16_rounded"] = epe(run(P, fm.half().float(), cn), base)
    out["epe_cnets_fp16_rounded"] = epe(run(P, fm, cn.half().float()), base)
    print(json.dumps(out), flush=True)
