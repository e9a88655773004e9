"""Deterministic synthetic weights and inputs for the hot path.

Used by the golden-vector generator, the parity tests, ``bench.py`` and ``smoke()`` so that every
party can rebuild *identical* tensors from a seed without shipping multi-MB weight files.
Values come from numpy's PCG64 streams (stable across platforms), keyed by (seed, crc32(name)).

Key names and shapes are the reference ``state_dict`` contract (SURVEY.md section 8b;
reference core/update.py:12-29,313-326,739-762, core/gma.py:48,82-84, core/models/streamflow.py:47-49).
Zero-initialised reference parameters (``aggregator.gamma``, ``transformer_block.*``:
gma.py:84, update.py:453-457,505) are given non-zero values on purpose, otherwise the GMA and
temporal paths would be invisible to parity checks.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

CORR_LEVELS = 4
CORR_RADIUS = 4
COR_PLANES = CORR_LEVELS * (2 * CORR_RADIUS + 1) ** 2   # 324
HDIM = 128                                              # decoder_dim // 2
K_CONV = (1, 15)
GRU_CONV = (1, 7)


def skblock_shapes(prefix: str, c_in: int, c_out: int, k_conv: Sequence[int]) -> List[Tuple[str, Tuple[int, ...]]]:
    mid = int(1.5 * c_in)
    out = []
    for i, k in enumerate(k_conv):
        out += [(f"{prefix}.conv_list.{i}.weight", (c_in, 1, k, k)), (f"{prefix}.conv_list.{i}.bias", (c_in,))]
    out += [
        (f"{prefix}.ffn1.0.weight", (mid, c_in, 1, 1)), (f"{prefix}.ffn1.0.bias", (mid,)),
        (f"{prefix}.ffn1.2.weight", (c_in, mid, 1, 1)), (f"{prefix}.ffn1.2.bias", (c_in,)),
        (f"{prefix}.pw.weight", (c_in, c_in, 1, 1)), (f"{prefix}.pw.bias", (c_in,)),
        (f"{prefix}.ffn2.0.weight", (mid, c_in, 1, 1)), (f"{prefix}.ffn2.0.bias", (mid,)),
        (f"{prefix}.ffn2.2.weight", (c_out, mid, 1, 1)), (f"{prefix}.ffn2.2.bias", (c_out,)),
    ]
    return out


def hotpath_param_shapes(T: int = 4) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) for every hot-path parameter of the canonical configuration with T frames."""
    P = T - 1
    u = "update_block"
    e = u + ".encoder"
    tb = u + ".transformer_block.transformer_block"
    s: List[Tuple[str, Tuple[int, ...]]] = [("att.to_qk.weight", (2 * HDIM, HDIM, 1, 1))]
    s += skblock_shapes(e + ".convc1", COR_PLANES, 256, K_CONV)
    s += skblock_shapes(e + ".convc2", 256, 192, K_CONV)
    s += [(e + ".convf1.weight", (128, 2, 1, 1)), (e + ".convf1.bias", (128,))]
    s += skblock_shapes(e + ".convf2", 128, 64, K_CONV)
    s += skblock_shapes(e + ".conv", 64 + 192, HDIM - 2, K_CONV)
    s += [(u + ".aggregator.to_v.weight", (HDIM, HDIM, 1, 1)), (u + ".aggregator.gamma", (1,))]
    s += skblock_shapes(u + ".gru", HDIM * 5, HDIM, GRU_CONV)
    s += [(u + ".mask.0.weight", (2 * HDIM, HDIM, 3, 3)), (u + ".mask.0.bias", (2 * HDIM,)),
          (u + ".mask.2.weight", (576, 2 * HDIM, 1, 1)), (u + ".mask.2.bias", (576,))]
    s += [(tb + ".norm1.weight", (HDIM,)), (tb + ".norm1.bias", (HDIM,)),
          (tb + ".norm2.weight", (HDIM,)), (tb + ".norm2.bias", (HDIM,)),
          (tb + ".attn.qkv.weight", (3 * HDIM, HDIM)),
          (tb + ".attn.proj.weight", (HDIM, HDIM)), (tb + ".attn.proj.bias", (HDIM,)),
          (tb + ".mlp.fc1.weight", (2 * HDIM, HDIM)), (tb + ".mlp.fc1.bias", (2 * HDIM,)),
          (tb + ".mlp.fc2.weight", (HDIM, 2 * HDIM)), (tb + ".mlp.fc2.bias", (HDIM,))]
    s += skblock_shapes(u + ".flow_head", HDIM * P, 2 * P, K_CONV)
    return s


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def randn(seed: int, name: str, shape: Sequence[int], scale: float = 1.0) -> torch.Tensor:
    a = _rng(seed, name).standard_normal(tuple(shape), dtype=np.float32)
    return torch.from_numpy(a * np.float32(scale))


def make_params(seed: int = 0, T: int = 4, gain: float = 0.7, flow_gain: float = 2.0) -> Dict[str, torch.Tensor]:
    """fp32 CPU tensors for every hot-path key.  weights ~ N(0, gain^2/fan_in), biases ~ N(0, 0.05^2),
    LayerNorm weights 1+0.1N, gamma 0.5.  ``flow_head.ffn2.2`` is scaled by ``flow_gain`` so the
    per-iteration flow update is of the order of a pixel."""
    out: Dict[str, torch.Tensor] = {}
    for key, shape in hotpath_param_shapes(T):
        if key.endswith("gamma"):
            t = torch.full(shape, 0.5, dtype=torch.float32)
        elif ".norm" in key and key.endswith("weight"):
            t = 1.0 + randn(seed, key, shape, 0.1)
        elif key.endswith("bias"):
            t = randn(seed, key, shape, 0.05)
        else:
            fan_in = int(np.prod(shape[1:]))
            t = randn(seed, key, shape, gain / np.sqrt(fan_in))
            if key.endswith("flow_head.ffn2.2.weight"):
                t = t * flow_gain
        out[key] = t
    return out


def make_features(seed: int, B: int, T: int, h: int, w: int, D: int = 256) -> Tuple[torch.Tensor, torch.Tensor]:
    """(fmaps [B,T,D,h,w], cnets [B,T-1,2*HDIM,h,w]) ~ N(0,1), fp32 CPU.

    The hot path starts at encoder outputs (reference streamflow.py:106-108), so synthetic inputs
    are generated directly at 1/8 resolution.  To give the correlation volume
    realistic structure (a peak near the true displacement instead of white noise), frame t+1's
    features are frame t's shifted by a small integer displacement plus noise."""
    base = randn(seed, "fmap.base", (B, D, h, w))
    frames = [base]
    for t in range(1, T):
        shift = (1 + t % 2, -1 - t % 3)                     # (dy, dx) in grid cells
        nxt = torch.roll(frames[-1], shifts=shift, dims=(2, 3))
        nxt = 0.8 * nxt + 0.6 * randn(seed, f"fmap.noise{t}", (B, D, h, w))
        frames.append(nxt)
    fmaps = torch.stack(frames, dim=1).contiguous()
    cnets = randn(seed, "cnet", (B, T - 1, 2 * HDIM, h, w))
    return fmaps, cnets


# ---- Twins_CSC encoder (reference core/encoders/twins_csc.py:37-57: first two stages of timm's twins_svt_large) ----------
TWINS_DIMS = (128, 256)
TWINS_HEADS = (4, 8)
TWINS_SR = (8, 4)
TWINS_PATCH = (4, 2)


def twins_param_shapes() -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of the encoder's state dict below `fnet.` / `cnet.` (what remains after twins_csc.py:52-57 deletes
    stages 3-4 and the head; the 1024-wide final `svt.norm` of the timm model survives and is part of the checkpoint)."""
    s: List[Tuple[str, Tuple[int, ...]]] = []
    cin = 3
    for i, (E, k, sr) in enumerate(zip(TWINS_DIMS, TWINS_PATCH, TWINS_SR)):
        pe = f"svt.patch_embeds.{i}"
        s += [(pe + ".proj.weight", (E, cin, k, k)), (pe + ".proj.bias", (E,)), (pe + ".norm.weight", (E,)), (pe + ".norm.bias", (E,))]
        for j in range(2):
            b = f"svt.blocks.{i}.{j}"
            s += [(b + ".norm1.weight", (E,)), (b + ".norm1.bias", (E,))]
            if j == 0:                      # LocallyGroupedAttn
                s += [(b + ".attn.qkv.weight", (3 * E, E)), (b + ".attn.qkv.bias", (3 * E,))]
            else:                           # GlobalSubSampleAttn
                s += [(b + ".attn.q.weight", (E, E)), (b + ".attn.q.bias", (E,)),
                      (b + ".attn.kv.weight", (2 * E, E)), (b + ".attn.kv.bias", (2 * E,)),
                      (b + ".attn.sr.weight", (E, E, sr, sr)), (b + ".attn.sr.bias", (E,)),
                      (b + ".attn.norm.weight", (E,)), (b + ".attn.norm.bias", (E,))]
            s += [(b + ".attn.proj.weight", (E, E)), (b + ".attn.proj.bias", (E,)),
                  (b + ".norm2.weight", (E,)), (b + ".norm2.bias", (E,)),
                  (b + ".mlp.fc1.weight", (4 * E, E)), (b + ".mlp.fc1.bias", (4 * E,)),
                  (b + ".mlp.fc2.weight", (E, 4 * E)), (b + ".mlp.fc2.bias", (E,))]
        s += [(f"svt.pos_block.{i}.proj.0.weight", (E, 1, 3, 3)), (f"svt.pos_block.{i}.proj.0.bias", (E,))]
        cin = E
    s += [("svt.norm.weight", (1024,)), ("svt.norm.bias", (1024,))]
    return s


def make_twins_params(seed: int = 0, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """fp32 CPU tensors for every Twins_CSC key: weights ~ N(0, gain^2 / fan_in), biases ~ N(0, 0.1^2) (non-zero qkv bias
    matters: zero-padded window tokens act through it), LayerNorm weights 1 + 0.1 N."""
    out: Dict[str, torch.Tensor] = {}
    for key, shape in twins_param_shapes():
        if ".norm" in key and key.endswith("weight"):
            t = 1.0 + randn(seed, key, shape, 0.1)
        elif key.endswith("bias"):
            t = randn(seed, key, shape, 0.1)
        else:
            t = randn(seed, key, shape, gain / np.sqrt(int(np.prod(shape[1:]))))
        out[key] = t
    return out
