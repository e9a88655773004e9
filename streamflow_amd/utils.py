"""Grid / sampler helpers with the reference's signatures (core/utils/utils.py:7-31,65-85)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops


def coords_grid(batch: int, ht: int, wd: int, device=None) -> torch.Tensor:
    """[batch,2,ht,wd] float32, channel 0 = x, channel 1 = y (reference utils.py:82-85).
    The reference builds it on the host and `.to(device)`s it; here it is written by a kernel on `device`
    (default: the current GPU)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return ops.coords_grid(batch, ht, wd, device)


def bilinear_sampler(img: torch.Tensor, coords: torch.Tensor, mode: str = "bilinear", mask: bool = False):
    """Pixel-coordinate bilinear sampling, zero padding (reference utils.py:65-79).
    img [M,C,H,W], coords [M,Ho,Wo,2] (x,y)."""
    if mode != "bilinear":
        raise RuntimeError("only mode='bilinear' is supported")
    res = ops.bilinear_sampler(img.contiguous().float(), coords.contiguous().float(), want_mask=mask)
    return res


def forward_interpolate(flow: torch.Tensor) -> torch.Tensor:
    """Reference `forward_interpolate(flow)` (core/utils/utils.py:34-62): flow [2, H, W] -> [2, H, W] float32, the
    warm start for the next clip.  Runs on the GPU and returns a device tensor (the reference goes through numpy /
    scipy on the host and returns a CPU tensor that the caller moves back with `.cuda()`).  A leading batch
    dimension [B, 2, H, W] is accepted as an extension."""
    f = flow.detach().float().contiguous()
    if f.dim() == 3:
        return ops.forward_interpolate(f[None])[0]
    return ops.forward_interpolate(f)


class InputPadder:
    """Replicate-pads frames up to the next multiple of `factor` (8: three stride-2 encoder stages) and crops results back.
    Same public surface as the reference's class (utils.py:7-31: `pad`, `pad_list`, `unpad`, the `_pad` = [left, right, top, bottom]
    attribute, `ht` / `wd`): 'sintel' mode centres the frame in both directions, every other mode (KITTI) centres it horizontally and
    puts all vertical padding at the bottom.  Host-side plumbing around the hot path."""

    def __init__(self, dims, mode: str = "sintel", factor: int = 8):
        self.ht, self.wd = int(dims[-2]), int(dims[-1])
        extra_h, extra_w = (-self.ht) % 8, (-self.wd) % 8           # (the reference hard-codes 8 whatever `factor` says; so do we)
        left, top = extra_w // 2, (extra_h // 2 if mode == "sintel" else 0)
        self._pad = [left, extra_w - left, top, extra_h - top]

    def _apply(self, x):
        return F.pad(x, self._pad, mode="replicate")

    def pad(self, *inputs):
        return [self._apply(x) for x in inputs]

    def pad_list(self, inputs):
        return [self._apply(x) for x in inputs]

    def unpad(self, x):
        left, right, top, bottom = self._pad
        rows, cols = x.shape[-2:]
        return x[..., top:rows - bottom, left:cols - right]
