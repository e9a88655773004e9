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
    """[(first frame of the window, keep[k] for each of its T-1 pairs)] -- the demo's window / flag schedule, in closed form:
    `full = (n - 1) // (T - 1)` whole windows start at 0, T - 1, 2 (T - 1), ... and keep every pair; if pairs are left over
    (`(n - 1) % (T - 1) != 0`) ONE more window is aligned to the end of the video (start n - T) and keeps only the pairs that start
    at or behind frame `full * (T - 1)`, the first one no earlier window has produced."""
    if T < 2 or n_frames < T:
        raise ValueError(f"need at least T={T} >= 2 frames, got {n_frames}")
    step = T - 1
    full = (n_frames - 1) // step
    out = [(s * step, [True] * step) for s in range(full)]
    done = full * step                                   # pairs 0 .. done - 1 are covered
    if done < n_frames - 1:
        start = n_frames - T
        out.append((start, [start + k >= done for k in range(step)]))
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


def predict_clips_warm_start(model: Callable, clips: Sequence[Sequence[torch.Tensor]], iters: int = 15
                             ) -> List[List[torch.Tensor]]:
    """Warm-started evaluation of consecutive clips of ONE scene, the loop of the reference's
    `create_sintel_submission_mf_warmup` (evaluate_mf.py:286-304): every clip starts from the previous clip's
    low-resolution flows pushed forward along themselves (`forward_interpolate`, utils.py:34-62).

    clips: sequence of clips, each a list of T image tensors [1,3,H,W] in 0..255 (already padded to /8), as
    `SKFlow_MF8.forward` takes them.  Returns the list of per-clip flow lists.  Unlike the reference nothing leaves
    the GPU between clips: the reference round-trips every low-resolution flow through numpy / scipy on the host."""
    from .utils import forward_interpolate
    flow_prev = None
    out: List[List[torch.Tensor]] = []
    for images in clips:
        if flow_prev is None:
            # first clip of a scene: a zero initial flow is what `flow_init=None` means (coords1 = coords0 + 0,
            # streamflow.py:112-115) and makes the model return the low-resolution flows as well
            b, _, H, W = images[0].shape
            flow_prev = [torch.zeros(b, 2, H // 8, W // 8, device=images[0].device) for _ in range(len(images) - 1)]
        flows, lowres = model(list(images), iters=iters, flow_init=flow_prev, test_mode=True)
        flow_prev = [forward_interpolate(l[0])[None] for l in lowres]
        out.append(flows)
    return out
