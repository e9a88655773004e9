"""Stand-in encoders.

The reference's encoder (Twins_CSC = first two stages of timm's twins_svt_large, core/encoders/twins_csc.py)
is OUT OF SCOPE for this build (its arithmetic lives in timm, which is not in the
image).  The hot path starts at encoder OUTPUTS, so the model classes accept any module mapping
[B,T,3,H,W] -> [B,T,256,H/8,W/8].  Two stand-ins are provided for plumbing and tests.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class InjectEncoder(nn.Module):
    """Returns the tensor stored in `.features` ([B,T,256,h,w]), sliced to the number of frames it is called
    with.  Used by parity tests to feed identical features to the reference and to this build."""

    def __init__(self, args=None, norm_fn=None):
        super().__init__()
        self.features = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.features is None:
            raise RuntimeError("InjectEncoder.features is not set")
        return self.features[:, : x.shape[1]]


class PatchEncoder(nn.Module):
    """Deterministic 8x8 patch embedding (+ a temporal mix) in plain PyTorch -- NOT Twins_CSC, plumbing only."""

    def __init__(self, args=None, norm_fn=None, out_dim: int = 256):
        super().__init__()
        g = torch.Generator().manual_seed(1234)
        self.register_buffer("proj", torch.randn(out_dim, 3, 8, 8, generator=g) / (3 * 64) ** 0.5, persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, C, H, W = x.shape
        f = F.conv2d(x.reshape(B * T, C, H, W).float(), self.proj, stride=8)
        return f.view(B, T, -1, H // 8, W // 8)


ENCODERS = {"InjectEncoder": InjectEncoder, "PatchEncoder": PatchEncoder}
