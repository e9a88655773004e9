"""CPU oracle for the Twins_CSC encoder  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, with explicit tensor arithmetic on the host, what the reference's encoder computes:

* the reference's OWN code: core/encoders/twins_csc.py:14-85 -- ``PatchEmbed`` over the T frames of a clip
  concatenated along the height (``'(b t) c h w -> b (t h w) c'``, token grid ``(T*h, w)``, :26-34), the stage loop
  (patch embed -> block 0 -> positional conv -> block 1, :68-76), two stages, output ``[B, T, 256, H/8, W/8]``;
* third-party arithmetic the reference imports from ``timm`` (``timm.create_model('twins_svt_large')``,
  twins_csc.py:40; timm is unpinned in install.sh and absent from this image): Twins-SVT-large stages 1-2 --
  embed dims (128, 256), heads (4, 8), mlp ratio 4, depths (2, 2), window 7, sub-sampling ratios (8, 4), LayerNorm
  eps 1e-6 in the blocks -- restated from timm's published ``timm/models/twins.py``:
  ``LocallyGroupedAttn`` (window attention; the token grid is zero-padded to a multiple of the window AFTER norm1, the
  padded tokens take part in the softmax with k = v = the qkv bias, their outputs are dropped), ``GlobalSubSampleAttn``
  (keys/values from a strided conv + LayerNorm(eps 1e-5) of the tokens), ``Block`` (pre-norm attention and MLP, both
  residual), ``PosConv`` (depthwise 3x3 conv + bias, residual).

Pinning: ``tests/golden/make_golden.py`` executes the reference's ``Twins_CSC.forward`` (its PatchEmbed, its loop) over a
``timm`` stand-in written as ordinary nn.Modules from the same published definitions; ``tests/test_oracle_golden.py``
checks this file against those vectors.  The timm boundary itself is **parity unpinned** (no reference test or golden
vector pins timm's arithmetic; only the checkpoint key names / shapes do) -- as for the temporal block in
streamflow_oracle.py.

Parameters: flat ``dict[str, Tensor]`` keyed like the reference state dict below the encoder, e.g.
``svt.blocks.0.1.attn.sr.weight`` (prefix ``fnet.`` / ``cnet.`` stripped by the caller).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

EMBED_DIMS = (128, 256)
NUM_HEADS = (4, 8)
SR_RATIOS = (8, 4)
PATCH = (4, 2)
WINDOW = 7
MLP_RATIO = 4
BLOCK_LN_EPS = 1e-6          # timm Twins: norm_layer = partial(nn.LayerNorm, eps=1e-6)
LN_EPS = 1e-5                # nn.LayerNorm default: PatchEmbed.norm (twins_csc.py:23) and GlobalSubSampleAttn.norm


def _ln(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _linear(x: Tensor, p: Params, name: str) -> Tensor:
    y = x @ p[name + ".weight"].t()
    return y + p[name + ".bias"] if name + ".bias" in p else y


def patch_embed(x: Tensor, p: Params, prefix: str, k: int) -> Tuple[Tensor, Tuple[int, int]]:
    """twins_csc.py:26-34.  x [B,T,C,H,W] -> tokens [B, T*h*w, E] (frame-major, then rows, then columns), grid (T*h, w)."""
    B, T, C, H, W = x.shape
    y = F.conv2d(x.reshape(B * T, C, H, W), p[prefix + ".proj.weight"], p[prefix + ".proj.bias"], stride=k)
    E, h, w = y.shape[1:]
    tok = y.reshape(B, T, E, h * w).permute(0, 1, 3, 2).reshape(B, T * h * w, E)
    return _ln(tok, p[prefix + ".norm.weight"], p[prefix + ".norm.bias"], LN_EPS), (T * h, w)


def _attend(q: Tensor, k: Tensor, v: Tensor, scale: float) -> Tensor:
    """[..., Nq, hd], [..., Nk, hd] -> softmax(scale q k^T) v."""
    return torch.softmax((q * scale) @ k.transpose(-2, -1), dim=-1) @ v


def locally_grouped_attn(x: Tensor, size: Tuple[int, int], p: Params, prefix: str, heads: int, ws: int = WINDOW) -> Tensor:
    """timm LocallyGroupedAttn.forward: attention inside non-overlapping ws x ws windows of the (H, W) token grid."""
    B, N, C = x.shape
    H, W = size
    hd = C // heads
    g = x.reshape(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    g = F.pad(g, (0, 0, 0, pad_r, 0, pad_b))                       # zero tokens (the input here is already norm1(x))
    Hp, Wp = H + pad_b, W + pad_r
    nh, nw = Hp // ws, Wp // ws
    win = g.reshape(B, nh, ws, nw, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, nh * nw, ws * ws, C)
    qkv = _linear(win, p, prefix + ".qkv").reshape(B, nh * nw, ws * ws, 3, heads, hd).permute(3, 0, 1, 4, 2, 5)
    o = _attend(qkv[0], qkv[1], qkv[2], hd ** -0.5)                # [B, windows, heads, ws*ws, hd]
    o = o.permute(0, 1, 3, 2, 4).reshape(B, nh, nw, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    o = o[:, :H, :W, :].reshape(B, N, C)
    return _linear(o, p, prefix + ".proj")


def global_subsample_attn(x: Tensor, size: Tuple[int, int], p: Params, prefix: str, heads: int, sr: int) -> Tensor:
    """timm GlobalSubSampleAttn.forward: every token attends to the sr x sr strided-conv summary of the grid."""
    B, N, C = x.shape
    H, W = size
    hd = C // heads
    q = _linear(x, p, prefix + ".q").reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    g = x.permute(0, 2, 1).reshape(B, C, H, W)
    s = F.conv2d(g, p[prefix + ".sr.weight"], p[prefix + ".sr.bias"], stride=sr).reshape(B, C, -1).permute(0, 2, 1)
    s = _ln(s, p[prefix + ".norm.weight"], p[prefix + ".norm.bias"], LN_EPS)
    kv = _linear(s, p, prefix + ".kv").reshape(B, -1, 2, heads, hd).permute(2, 0, 3, 1, 4)
    o = _attend(q, kv[0], kv[1], hd ** -0.5).permute(0, 2, 1, 3).reshape(B, N, C)
    return _linear(o, p, prefix + ".proj")


def block(x: Tensor, size: Tuple[int, int], p: Params, prefix: str, heads: int, sr: int, local: bool) -> Tensor:
    """timm twins Block.forward: x + attn(norm1(x)); x + mlp(norm2(x)), exact-erf GELU."""
    y = _ln(x, p[prefix + ".norm1.weight"], p[prefix + ".norm1.bias"], BLOCK_LN_EPS)
    a = (locally_grouped_attn(y, size, p, prefix + ".attn", heads) if local
         else global_subsample_attn(y, size, p, prefix + ".attn", heads, sr))
    x = x + a
    y = _ln(x, p[prefix + ".norm2.weight"], p[prefix + ".norm2.bias"], BLOCK_LN_EPS)
    return x + _linear(F.gelu(_linear(y, p, prefix + ".mlp.fc1")), p, prefix + ".mlp.fc2")


def pos_conv(x: Tensor, size: Tuple[int, int], p: Params, prefix: str) -> Tensor:
    """timm PosConv.forward (stride 1): depthwise 3x3 conv + bias over the token grid, plus the input."""
    B, N, C = x.shape
    g = x.permute(0, 2, 1).reshape(B, C, *size)
    y = F.conv2d(g, p[prefix + ".proj.0.weight"], p[prefix + ".proj.0.bias"], padding=1, groups=C) + g
    return y.reshape(B, C, N).permute(0, 2, 1)


def twins_csc_forward(images: Tensor, p: Params) -> Tensor:
    """Twins_CSC.forward (twins_csc.py:59-85).  images [B,T,3,H,W] (already normalised by the caller) ->
    [B, T, 256, H/8, W/8].  H and W must be multiples of 8."""
    x = images
    B, T, _, H, W = x.shape
    for i in range(2):
        x, size = patch_embed(x, p, f"svt.patch_embeds.{i}", PATCH[i])
        for j in range(2):
            x = block(x, size, p, f"svt.blocks.{i}.{j}", NUM_HEADS[i], SR_RATIOS[i], local=(j == 0))
            if j == 0:
                x = pos_conv(x, size, p, f"svt.pos_block.{i}")
        H, W = H // PATCH[i], W // PATCH[i]
        x = x.reshape(B, T, H, W, -1).permute(0, 1, 4, 2, 3).contiguous()          # 'b (t h w) c -> b t c h w'
    return x
