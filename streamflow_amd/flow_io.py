"""Flow-field file formats and the evaluation metrics of the reference harness (host side, numpy).

* Middlebury ``.flo``: float32 magic 202021.25 (the bytes ``PIEH``), int32 width, int32 height, then height*width
  interleaved (u, v) float32 pairs, row-major (reference core/utils/frame_utils.py:13-29, 85-114).
* KITTI flow PNG: 16-bit RGB PNG with channels (R, G, B) = (u, v, valid), value = 64*flow + 2^15
  (frame_utils.py:117-122, 137-141: cv2 reads/writes BGR and the reference flips the channel axis, so the FILE holds
  u in R).  cv2 is not in this image: a small PNG codec (zlib + the five PNG row filters, 8/16-bit gray / RGB / RGBA,
  non-interlaced) is included, which is all the KITTI devkit files need.
* PFM (FlyingThings ground truth, frame_utils.py:48-83): 'PF' / 'Pf' header, rows stored bottom-up, sign of the scale
  = endianness.
* ``.flo5`` (Spring: an HDF5 container with one gzip dataset, frame_utils.py:31-47,130-135): ``flo5.py``, a codec for
  the subset of the HDF5 file format h5py emits for such a file (re-exported here as read_flo5 / write_flo5; h5py itself
  is not in this image, so that codec is checked against the format specification only).
* Metrics: end-point error map, Sintel-style EPE / 1px / 3px / 5px (evaluate_mf.py:484-497) and the KITTI
  F1-all outlier rate: epe > 3 px and epe/|gt| > 5 % over valid pixels (evaluate_mf.py:124-133).
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, Tuple

import numpy as np

from .flo5 import read_flo5, write_flo5  # noqa: F401  (Spring's .flo5)

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
    """[H, W, 2] flow -> uint16 [H, W, 3] = (64*u + 2^15, 64*v + 2^15, 1)  (frame_utils.py:137-141).  Computed in the DTYPE OF
    `flow` like the reference (a float32 flow gives float32 codes before the truncation to uint16: pinned by the golden)."""
    uv = 64.0 * np.asarray(flow) + 2 ** 15
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


# ---- PNG (16-bit capable), for the KITTI flow format ---------------------------------------------------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"
_PNG_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def write_png(path: str, img: np.ndarray) -> None:
    """uint8 / uint16 array [H, W] or [H, W, C] (C = 1, 2, 3, 4) -> non-interlaced PNG (filter 0 rows, zlib level 6)."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.dtype not in (np.uint8, np.uint16) or a.ndim != 3 or a.shape[2] not in (1, 2, 3, 4):
        raise ValueError(f"write_png: need uint8/uint16 [H, W(, C<=4)], got {a.dtype} {a.shape}")
    h, w, c = a.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    depth = 8 * a.dtype.itemsize
    rows = a.astype(">u2" if depth == 16 else np.uint8).reshape(h, -1).view(np.uint8).reshape(h, -1)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_png(path: str) -> np.ndarray:
    """Non-interlaced 8/16-bit gray / gray+alpha / RGB / RGBA PNG -> uint8 / uint16 [H, W, C] (C squeezed when 1)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _PNG_SIG:
        raise IOError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if len(body) != n or pos + 12 + n > len(data):
            raise IOError(f"{path}: truncated PNG chunk {tag!r}")
        if struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] != (zlib.crc32(tag + body) & 0xFFFFFFFF):
            raise IOError(f"{path}: CRC mismatch in PNG chunk {tag!r}")
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    if hdr is None:
        raise IOError(f"{path}: no IHDR chunk")
    w, h, depth, ctype, _, _, interlace = hdr
    if depth not in (8, 16) or ctype not in _PNG_CHANNELS or interlace:
        raise IOError(f"{path}: unsupported PNG (bit depth {depth}, colour type {ctype}, interlace {interlace})")
    c = _PNG_CHANNELS[ctype]
    bpp = c * depth // 8                                   # bytes per pixel = filter distance
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    if raw.size != h * (stride + 1):
        raise IOError(f"{path}: truncated image data")
    raw = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.uint8)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:]
        if ft == 0:
            cur = line
        elif ft == 2:                                      # Up: byte-wise sum with the row above (uint8 wraps mod 256)
            cur = line + prev
        elif ft == 1:                                      # Sub: a running sum per byte lane of the pixel
            cur = np.cumsum(line.reshape(w, bpp), axis=0, dtype=np.uint8).reshape(stride)
        elif ft in (3, 4):                                 # Average / Paeth: sequential in x; plain ints on bytes objects
            lb, pb = line.tobytes(), prev.tobytes()        # (a numpy scalar per byte was ~20x slower: libpng writes
            cb = bytearray(stride)                         # adaptive filters, so this IS the common KITTI path)
            if ft == 3:
                for x in range(stride):
                    a = cb[x - bpp] if x >= bpp else 0
                    cb[x] = (lb[x] + ((a + pb[x]) >> 1)) & 255
            else:
                for x in range(stride):
                    if x >= bpp:
                        a, c_ = cb[x - bpp], pb[x - bpp]
                    else:
                        a = c_ = 0
                    b_ = pb[x]
                    p = a + b_ - c_
                    pa, pb_, pc = abs(p - a), abs(p - b_), abs(p - c_)
                    pred = a if (pa <= pb_ and pa <= pc) else (b_ if pb_ <= pc else c_)
                    cb[x] = (lb[x] + pred) & 255
            cur = np.frombuffer(bytes(cb), np.uint8)
        else:
            raise IOError(f"{path}: bad filter type {ft}")
        out[y] = cur
        prev = out[y]
    img = out.view(">u2").astype(np.uint16) if depth == 16 else out
    img = img.reshape(h, w, c)
    return img[:, :, 0] if c == 1 else img


def write_flow_kitti(path: str, flow: np.ndarray) -> None:
    """reference writeFlowKITTI (frame_utils.py:137-141): [H, W, 2] -> 16-bit PNG, (R, G, B) = (u, v, 1) codes."""
    write_png(path, kitti_encode(flow))


def read_flow_kitti(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """reference readFlowKITTI (frame_utils.py:117-122): 16-bit RGB PNG -> (flow [H, W, 2] float32, valid [H, W])."""
    img = read_png(path)
    if img.ndim != 3 or img.shape[2] < 3 or img.dtype != np.uint16:
        raise IOError(f"{path}: KITTI flow needs a 16-bit RGB PNG, got {img.dtype} {img.shape}")
    return kitti_decode(img[:, :, :3])


# ---- PFM -----------------------------------------------------------------------------------------------------------------
def read_pfm(path: str) -> np.ndarray:
    """reference readPFM (frame_utils.py:48-83): float32 [H, W] ('Pf') or [H, W, 3] ('PF'), rows flipped to top-down."""
    with open(path, "rb") as f:
        header = f.readline().rstrip()
        if header not in (b"PF", b"Pf"):
            raise IOError(f"{path}: not a PFM file")
        dims = f.readline().split()
        if len(dims) != 2:
            raise IOError(f"{path}: malformed PFM header")
        w, h = int(dims[0]), int(dims[1])
        scale = float(f.readline().rstrip())
        data = np.fromfile(f, "<f4" if scale < 0 else ">f4")
    c = 3 if header == b"PF" else 1
    if data.size != w * h * c:
        raise IOError(f"{path}: truncated PFM file")
    img = data.reshape((h, w, 3) if c == 3 else (h, w)).astype(np.float32)
    return np.flipud(img).copy()


def write_pfm(path: str, img: np.ndarray) -> None:
    """float32 [H, W] or [H, W, 3] -> little-endian PFM (scale -1), rows bottom-up."""
    a = np.asarray(img, np.float32)
    if a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
        raise ValueError(f"write_pfm: need [H, W] or [H, W, 3], got {a.shape}")
    with open(path, "wb") as f:
        f.write(b"PF\n" if a.ndim == 3 else b"Pf\n")
        f.write(f"{a.shape[1]} {a.shape[0]}\n-1.0\n".encode())
        np.flipud(a).astype("<f4").tofile(f)
