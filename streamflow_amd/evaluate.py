"""Dataset scoring loops with the reference's signatures and result dictionaries.

* ``validate_sintel_mf(model, iters, root, nframes)`` -- reference evaluate_mf.py:468-503 over a Sintel-layout tree
  ``root/training/{clean,final}/<scene>/frame_XXXX.png`` + ``root/training/flow/<scene>/frame_XXXX.flo``: every scene is cut
  into clips of ``nframes`` frames that overlap by one frame; the tail of a scene is covered by ONE clip aligned to the scene's
  end whose already-scored pairs carry frame id -1 and are skipped (core/mf_datasets.py:1125-1149); pad -> model -> unpad;
  per-pixel EPE over all scored pairs; returns ``{'clean': epe, 'final': epe}`` (the 1 / 3 / 5 px rates are printed like the
  reference prints them and returned by ``sintel_report``).
* ``validate_kitti_mf(model, iters, multi_root, nframes)`` -- evaluate_mf.py:106-142 over the multi-frame KITTI-2015 layout
  ``multi_root/training/image_2/000NNN_FF.png`` + ``flow_occ/000NNN_10.png``: the clip is frames 12 - nframes .. 11, only the
  LAST pair (frames 10 -> 11) has ground truth (core/mf_datasets.py:946-952); 'kitti' padding; EPE and the F1-all outlier rate
  (epe > 3 px and epe / |gt| > 5 %) over pixels with valid >= 0.5; returns ``{'kitti_epe': .., 'kitti_f1': ..}``.

``model`` is anything with the reference's test-mode call ``model(images: list of [1,3,H,W] in 0..255, iters=.., test_mode=True)
-> list of nframes - 1 flows [1,2,H,W]`` (streamflow_amd.SKFlow_MF8, or the CPU oracle wrapped the same way in the tests).
Files are read with this package's own codecs (flow_io.py: PNG, .flo, KITTI 16-bit PNG).  Host-side plumbing: nothing here
launches a kernel itself.
"""
from __future__ import annotations

import glob
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import flow_io
from .utils import InputPadder


def sintel_clip_schedule(n_images: int, nframes: int) -> List[Tuple[int, List[int]]]:
    """[(first frame, frame ids)] of one scene (core/mf_datasets.py:1125-1149).  Clips start every nframes - 1 frames while a
    whole clip fits; if frames are left over, ONE more clip is aligned to the end of the scene and the frames it shares
    with the clips before it get id -1 (the pairs that start there are already scored).  Closed form: k = (n - 1) // (T - 1)
    full clips start at 0, T - 1, ..; a tail clip exists iff (n - 1) % (T - 1) != 0 and starts at n - T."""
    T = int(nframes)
    if n_images < T or T < 2:
        raise ValueError(f"a scene needs at least nframes = {T} >= 2 images, got {n_images}")
    full = (n_images - 1) // (T - 1)
    out = [(s * (T - 1), list(range(s * (T - 1), s * (T - 1) + T))) for s in range(full)]
    covered = full * (T - 1)                          # first frame whose outgoing pair is not scored yet
    if covered < n_images - 1:
        first = n_images - T
        out.append((first, [-1 if j < covered else j for j in range(first, n_images)]))
    return out


def _image(path: str) -> torch.Tensor:
    img = flow_io.read_png(path)
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, axis=2)
    return torch.from_numpy(np.ascontiguousarray(img[:, :, :3]).astype(np.uint8)).permute(2, 0, 1).float()


def _scenes(image_root: str) -> List[str]:
    return sorted(d for d in os.listdir(image_root) if os.path.isdir(os.path.join(image_root, d)))


def _device_of(model) -> torch.device:
    try:
        return next(model.parameters()).device
    except (AttributeError, StopIteration, TypeError):
        return torch.device("cpu")


@torch.no_grad()
def sintel_report(model: Callable, iters: int = 6, root: str = "/data/Sintel", nframes: int = 3,
                  dstypes: Sequence[str] = ("clean", "final"), device: Optional[torch.device] = None) -> Dict[str, Dict[str, float]]:
    """Per render pass: {'epe', '1px', '3px', '5px', 'pairs'} over every scored pair of every scene."""
    dev = device or _device_of(model)
    report = {}
    for dstype in dstypes:
        image_root = os.path.join(root, "training", dstype)
        flow_root = os.path.join(root, "training", "flow")
        epe_list = []
        for scene in _scenes(image_root):
            imgs = sorted(glob.glob(os.path.join(image_root, scene, "*.png")))
            flos = sorted(glob.glob(os.path.join(flow_root, scene, "*.flo")))
            if len(flos) != len(imgs) - 1:
                raise RuntimeError(f"{scene}: {len(imgs)} frames need {len(imgs) - 1} .flo files, found {len(flos)}")
            for first, ids in sintel_clip_schedule(len(imgs), nframes):
                images = [_image(p)[None].to(dev) for p in imgs[first:first + nframes]]
                padder = InputPadder(images[0].shape)
                flows = model(padder.pad_list(images), iters=iters, test_mode=True)
                flows = [padder.unpad(f[0]).float().cpu() for f in flows]
                for i in range(nframes - 1):
                    if ids[i] == -1:
                        continue
                    gt = torch.from_numpy(flow_io.read_flo(flos[first + i])).permute(2, 0, 1).float()
                    epe_list.append(torch.sum((flows[i] - gt) ** 2, dim=0).sqrt().view(-1).numpy())
        epe_all = np.concatenate(epe_list)
        m = flow_io.sintel_metrics(epe_all)
        m["pairs"] = len(epe_list)
        print("Validation (%s) EPE: %f, 1px: %f, 3px: %f, 5px: %f" % (dstype, m["epe"], m["1px"], m["3px"], m["5px"]))
        report[dstype] = m
    return report


@torch.no_grad()
def validate_sintel_mf(model: Callable, iters: int = 6, root: str = "/data/Sintel", tqdm_miniters: int = 1, nframes: int = 3,
                       device: Optional[torch.device] = None) -> Dict[str, float]:
    """The reference's return value: {'clean': mean EPE, 'final': mean EPE}  (evaluate_mf.py:468-503)."""
    return {k: v["epe"] for k, v in sintel_report(model, iters, root, nframes, device=device).items()}


@torch.no_grad()
def validate_kitti_mf(model: Callable, iters: int = 6, multi_root: Optional[str] = None, nframes: int = 3,
                      device: Optional[torch.device] = None) -> Dict[str, float]:
    """{'kitti_epe', 'kitti_f1'} over the sequences present under multi_root/training (the reference walks 000000 .. 000199)."""
    if multi_root is None:
        raise ValueError("validate_kitti_mf: multi_root (the multi-frame KITTI-2015 tree) is required")
    dev = device or _device_of(model)
    image_root = os.path.join(multi_root, "training", "image_2")
    flow_root = os.path.join(multi_root, "training", "flow_occ")
    seqs = sorted(os.path.basename(p)[:6] for p in glob.glob(os.path.join(flow_root, "??????_10.png")))
    if not seqs:
        raise RuntimeError(f"no ground truth under {flow_root}")
    out_list, epe_list = [], []
    for seq in seqs:
        images = [_image(os.path.join(image_root, "%s_%02d.png" % (seq, i)))[None].to(dev) for i in range(12 - nframes, 12)]
        gt_np, valid_np = flow_io.read_flow_kitti(os.path.join(flow_root, seq + "_10.png"))
        gt = torch.from_numpy(gt_np).permute(2, 0, 1).float()
        valid = torch.from_numpy(valid_np)
        padder = InputPadder(images[0].shape, mode="kitti")
        flows = model(padder.pad_list(images), iters=iters, test_mode=True)
        flow = padder.unpad(flows[nframes - 2][0]).float().cpu()          # only the last pair (frames 10 -> 11) has ground truth
        epe = torch.sum((flow - gt) ** 2, dim=0).sqrt().view(-1)
        mag = torch.sum(gt ** 2, dim=0).sqrt().view(-1)
        val = valid.view(-1) >= 0.5
        out = ((epe > 3.0) & ((epe / mag) > 0.05)).float()
        epe_list.append(epe[val].mean().item())
        out_list.append(out[val].numpy())
    epe = float(np.mean(np.array(epe_list)))
    f1 = float(100 * np.mean(np.concatenate(out_list)))
    print("Validation KITTI: %f, %f" % (epe, f1))
    return {"kitti_epe": epe, "kitti_f1": f1}
