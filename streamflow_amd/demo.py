"""Clip grouping of the reference's demo pipeline (demo.py:502-534), without its video I/O (decord / cv2).

The demo slides a window of T frames with stride T-1 over the video; the last window re-uses the final T frames
and drops the flow fields that an earlier window already produced (``flags == -1``).  Every consecutive frame pair
(j, j+1) therefore gets exactly one flow field, in order.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch

from .utils import InputPadder


def group_clips(n_frames: int, T: int = 4) -> List[Tuple[int, List[bool]]]:
    """[(first frame of the window, keep[k] for each of its T-1 pairs)] -- the demo's window/flag schedule."""
    if n_frames < T:
        raise ValueError(f"need at least T={T} frames, got {n_frames}")
    out = []
    i = 0
    while True:
        if i + T <= n_frames:
            start, flags = i, list(range(i, i + T))
        else:
            start = n_frames - T
            flags = [-1 if j < i else j for j in range(start, n_frames)]
        out.append((start, [flags[k] != -1 for k in range(T - 1)]))
        if i + T >= n_frames:
            break
        i += T - 1
    return out


@torch.no_grad()
def predict_frames(model: Callable, frames: Sequence[torch.Tensor], T: int = 4, device=None, mode: str = "sintel"
                   ) -> List[torch.Tensor]:
    """frames: list of [3,H,W] tensors already normalised to [-1,1] (demo.py:510).  Returns len(frames)-1 flow fields
    [2,H,W] on the CPU, as `read_video_and_group_predict` does.  `model(images[1,T,3,H',W'])` -> list of T-1 flows."""
    padder = InputPadder(frames[0].shape, mode=mode)
    padded = padder.pad_list([f[None] for f in frames])
    flows: List[torch.Tensor] = []
    for start, keep in group_clips(len(frames), T):
        imgs = torch.stack([padded[j][0] for j in range(start, start + T)], dim=0)[None]
        if device is not None:
            imgs = imgs.to(device)
        out = model(imgs)
        flows += [padder.unpad(out[k][0]).cpu() for k in range(T - 1) if keep[k]]
    return flows
