"""Flow-field file formats and the evaluation metrics of the reference harness (host side, numpy).

* Middlebury ``.flo``: float32 magic 202021.25 (the bytes ``PIEH``), int32 width, int32 height, then height*width
  interleaved (u, v) float32 pairs, row-major (reference core/utils/frame_utils.py:13-29, 85-114).
* KITTI flow PNG arithmetic: uint16 channels (u, v, valid), value = 64*flow + 2^15
  (frame_utils.py:117-122, 137-141); only the codec arithmetic is here -- reading/writing 16-bit PNGs needs cv2,
  which this image does not have.
* Metrics: end-point error map, Sintel-style EPE / 1px / 3px / 5px (evaluate_mf.py:484-497) and the KITTI
  F1-all outlier rate: epe > 3 px and epe/|gt| > 5 % over valid pixels (evaluate_mf.py:124-133).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

FLO_MAGIC = np.float32(202021.25)


def write_flo(path: str, flow: np.ndarray) -> None:
    """flow [H, W, 2] (u, v) -> Middlebury .flo"""
    flow = np.asarray(flow)
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise ValueError(f"expected [H, W, 2], got {flow.shape}")
    h, w = flow.shape[:2]
    with open(path, "wb") as f:
        np.array([FLO_MAGIC], np.float32).tofile(f)
        np.array([w, h], np.int32).tofile(f)
        flow.astype(np.float32).tofile(f)


def read_flo(path: str) -> np.ndarray:
    """Middlebury .flo -> [H, W, 2] float32"""
    with open(path, "rb") as f:
        magic = np.fromfile(f, np.float32, count=1)
        if magic.size != 1 or magic[0] != FLO_MAGIC:
            raise IOError(f"{path}: bad .flo magic")
        w, h = (int(v) for v in np.fromfile(f, np.int32, count=2))
        data = np.fromfile(f, np.float32, count=2 * w * h)
        if data.size != 2 * w * h:
            raise IOError(f"{path}: truncated .flo file")
    return data.reshape(h, w, 2)


def kitti_encode(flow: np.ndarray) -> np.ndarray:
    """[H, W, 2] flow -> uint16 [H, W, 3] = (64*u + 2^15, 64*v + 2^15, 1)  (frame_utils.py:137-141)"""
    uv = 64.0 * np.asarray(flow, np.float64) + 2 ** 15
    valid = np.ones(uv.shape[:2] + (1,))
    return np.concatenate([uv, valid], axis=-1).astype(np.uint16)


def kitti_decode(png: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """uint16 [H, W, 3] (u, v, valid) -> (flow [H, W, 2] float32, valid [H, W])  (frame_utils.py:117-122)"""
    a = np.asarray(png).astype(np.float32)
    return (a[:, :, :2] - 2 ** 15) / 64.0, a[:, :, 2]


def epe_map(flow: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """[2, H, W] or [H, W, 2] -> per-pixel end-point error."""
    flow, gt = np.asarray(flow, np.float64), np.asarray(gt, np.float64)
    axis = 0 if flow.shape[0] == 2 and flow.ndim == 3 and flow.shape[-1] != 2 else -1
    return np.sqrt(((flow - gt) ** 2).sum(axis=axis))


def sintel_metrics(epe_all: np.ndarray) -> Dict[str, float]:
    """epe over all evaluated pixels -> EPE, 1px, 3px, 5px (evaluate_mf.py:489-495)"""
    e = np.asarray(epe_all).reshape(-1)
    return {"epe": float(e.mean()), "1px": float((e < 1).mean()), "3px": float((e < 3).mean()),
            "5px": float((e < 5).mean())}


def kitti_f1(flow: np.ndarray, gt: np.ndarray, valid: np.ndarray) -> Dict[str, float]:
    """KITTI-2015 metrics over valid pixels: mean EPE and F1-all (% of outliers)  (evaluate_mf.py:124-141)"""
    epe = epe_map(flow, gt).reshape(-1)
    gt = np.asarray(gt, np.float64)
    axis = 0 if gt.shape[0] == 2 and gt.shape[-1] != 2 else -1
    mag = np.sqrt((gt ** 2).sum(axis=axis)).reshape(-1)
    val = np.asarray(valid).reshape(-1) >= 0.5
    out = (epe > 3.0) & ((epe / np.maximum(mag, 1e-30)) > 0.05)
    return {"epe": float(epe[val].mean()), "f1": float(100.0 * out[val].mean())}
