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
    """Pads images so H and W are divisible by 8 (reference utils.py:7-31; host-side plumbing)."""

    def __init__(self, dims, mode: str = "sintel", factor: int = 8):
        self.ht, self.wd = dims[-2:]
        pad_ht = (((self.ht // 8) + 1) * 8 - self.ht) % 8
        pad_wd = (((self.wd // 8) + 1) * 8 - self.wd) % 8
        if mode == "sintel":
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        return [F.pad(x, self._pad, mode="replicate") for x in inputs]

    def pad_list(self, inputs):
        return [F.pad(x, self._pad, mode="replicate") for x in inputs]

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return x[..., c[0]:c[1], c[2]:c[3]]
