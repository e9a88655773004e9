#!/usr/bin/env python3
"""How far does the REFERENCE's own deployment arithmetic (fp16 autocast around the update block and the GMA attention,
core/models/streamflow.py:106-137, evaluate_mf.py:1106) move its flows away from its own fp32 result on the inputs this build is
judged on?  CPU only: the CPU oracle is run twice -- in fp32, and with every contraction of the autocast regions emulated
(operands and weights rounded to fp16, fp32 accumulation as the matrix cores do, result rounded to fp16; GELU output rounded to
fp16; LayerNorm / softmax in fp32 as autocast keeps them; the correlation volume stays fp32: the reference never autocasts it).
The fp16 rounding of the residual additions is NOT emulated, so the figure is a LOWER bound of the reference's deviation.
Writes JSON lines (profiles/r05_reference_autocast_deviation.jsonl is a copy).
usage: autocast_emulation.py [hard seeds ...]  [--headline]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle import streamflow_oracle as orc, twins_oracle as two
from streamflow_amd import synthetic as syn

h16 = lambda t: t.half().float()
_conv2d, _gelu, _einsum, _tblock = F.conv2d, orc.gelu, torch.einsum, orc.temporal_block


class autocast_emulated:
    def __enter__(self):
        def conv2d(x, w, b=None, **kw):
            return h16(_conv2d(h16(x), h16(w), None if b is None else h16(b), **kw))
        def einsum(eq, *ops):
            return h16(_einsum(eq, *[h16(o) for o in ops]))
        def temporal_block(tokens, p, prefix):
            # Linear layers in fp16 (weights and operands rounded, fp32 accumulate, fp16 result); LayerNorm and softmax in fp32
            g = lambda name: p[prefix + "." + name]
            S, T, C = tokens.shape
            lin = lambda x, w, b=None: h16(h16(x) @ h16(w).t() + (0 if b is None else h16(b)))
            hh = orc.layer_norm(tokens, g("norm1.weight"), g("norm1.bias"))
            qkv = lin(hh, g("attn.qkv.weight"))
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
            a = torch.softmax(h16(h16(q * C ** -0.5) @ k.transpose(1, 2)), dim=-1)
            o = lin(h16(h16(a) @ v), g("attn.proj.weight"), g("attn.proj.bias"))
            x = tokens + o
            hh = orc.layer_norm(x, g("norm2.weight"), g("norm2.bias"))
            hh = h16(_gelu(lin(hh, g("mlp.fc1.weight"), g("mlp.fc1.bias"))))
            return x + lin(hh, g("mlp.fc2.weight"), g("mlp.fc2.bias"))
        orc.F.conv2d = conv2d
        orc.gelu = lambda x: h16(_gelu(x))
        orc.temporal_block = temporal_block
        self._ga, self._gg = orc.gma_attention, orc.gma_aggregate
        ga, gg = self._ga, self._gg
        def gma_attention(inps, w_qk, heads=1):
            torch.einsum = einsum
            try:
                return ga(inps, w_qk, heads)
            finally:
                torch.einsum = _einsum
        def gma_aggregate(attn, fmap, w_v, gamma):
            torch.einsum = einsum
            try:
                return gg(h16(attn), fmap, w_v, gamma)
            finally:
                torch.einsum = _einsum
        orc.gma_attention, orc.gma_aggregate = gma_attention, gma_aggregate
        return self

    def __exit__(self, *a):
        orc.F.conv2d, orc.gelu, orc.temporal_block = _conv2d, _gelu, _tblock
        orc.gma_attention, orc.gma_aggregate = self._ga, self._gg


def hard_case(seed):
    ps, fs, a, b = (21, 24, 22, 23) if seed == 21 else (seed, 100 + seed, 200 + seed, 300 + seed)
    P = syn.make_params(ps, 4)
    frames = torch.stack([(syn.randn(fs, f"frame{t}", (1, 3, 128, 192)).sigmoid() * 255.0) for t in range(4)], dim=1)
    imgs = 2 * (frames / 255.0) - 1.0
    return P, two.twins_csc_forward(imgs, syn.make_twins_params(a)), two.twins_csc_forward(imgs[:, :-1], syn.make_twins_params(b)), 4


def run(tag, P, fm, cn, iters):
    ref, _ = orc.hotpath_forward(fm, cn, P, iters)
    with autocast_emulated():
        ac, _ = orc.hotpath_forward(fm, cn, P, iters)
    assert orc.F.conv2d is _conv2d
    mag = float(torch.stack([o.norm(dim=1).mean() for o in ref]).mean())
    e = max(orc.epe(a, b) for a, b in zip(ac, ref))
    print(json.dumps({"case": tag, "mean_flow_px": round(mag, 2), "epe_reference_autocast_vs_reference_fp32": e,
                      "relative_to_flow": e / mag}), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    for seed in [int(a) for a in args] or [21, 11, 12, 13, 31, 32]:
        run(f"hard seed {seed} (128x192, 4 iterations)", *hard_case(seed))
    if "--headline" in sys.argv:
        for sd in (0, 1):
            P = syn.make_params(sd, 4)
            fm, cn = syn.make_features(1000 + 17 * sd if sd else 1000, 1, 4, 55, 128)
            run(f"headline seed {sd} (55x128 grid, 15 iterations)", P, fm, cn, 15)
