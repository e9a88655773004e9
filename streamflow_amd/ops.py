"""Tensor-level wrappers over the C ABI (streamflow_amd/_lib.py).

Internal data layout: every activation is a set of *channel-major planes* ``[n_img][rows][P]``
(P = h*w contiguous, i.e. NCHW) living inside a flat fp32 device buffer; a :class:`Planes`
object is a (pointer, image stride, rows) view of it, so concatenations (``torch.cat`` in the
reference) are just kernels writing into row slices of a wider buffer.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, replace
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import (EPI_AXPY, EPI_GELU, EPI_NONE, EPI_RELU, EPI_RES, EPI_RES_GELU, EPI_RES_GELU_DW1,  # noqa: F401
                   LAYOUT_K_MAJOR, LAYOUT_K_MINOR, LAYOUT_SPLIT_F16, LAYOUT_F16_K_MINOR, PRECISION_F16, PRECISION_F16X2, PRECISION_F16X3,
                   PRECISION_FP32,
                   SfGemm)

# Arithmetic mode of every GEMM-shaped op (sf_gemm, corr build):
#   PRECISION_FP32  exact fp32 on v_mfma_f32_32x32x2_f32
#   PRECISION_F16X3 split precision (x = hi + lo in fp16, 3 MFMAs per product, fp32 accumulate; ~2^-22 relative)
#   PRECISION_F16X2 weights split hi + lo, activations rounded once to fp16 (2 MFMAs per product; 2^-11 relative on
#                   the activations: ~1e-4 px EPE on the flows)
#   PRECISION_F16   weights AND activations rounded once to fp16 (1 MFMA per product, fp32 accumulation): the reference's
#                   deployed arithmetic class (fp16 autocast), opt-in
PRECISION = PRECISION_F16X3
_PRECISION_NAMES = {"fp32": PRECISION_FP32, "f16x3": PRECISION_F16X3, "f16x2": PRECISION_F16X2, "f16": PRECISION_F16}


def set_precision(mode) -> int:
    """mode: 'fp32' | 'f16x3' | 'f16x2' (or the integer constants).  Returns the previous mode."""
    global PRECISION
    prev = PRECISION
    PRECISION = _PRECISION_NAMES[mode] if isinstance(mode, str) else int(mode)
    if PRECISION not in (PRECISION_FP32, PRECISION_F16X3, PRECISION_F16X2, PRECISION_F16):
        PRECISION = prev
        raise RuntimeError(f"unknown precision {mode}")
    return prev


def precision_name() -> str:
    return {v: k for k, v in _PRECISION_NAMES.items()}[PRECISION]


@dataclass(frozen=True)
class Ctx:
    """Everything a launch depends on besides its operands -- passed explicitly (`cx=`) by the engine and the encoder, so
    that two engines with different presets can be driven from two threads (there is no process-wide launch
    state; the module-level PRECISION below is only the DEFAULT of calls that do not pass a context, like a default dtype).
    precision: PRECISION_*.  split_ws: scratch that lets sf_gemm split K for small grids (None: never).
    shadows / shadow_fused: use and maintain Planes.shadow; producers write the k-octet copy in their own epilogue.
    flash_stats: the fused GMA kernel uses the softmax statistics stored once per clip.
    hidden_f16 / hidden_koct / pw_fold / x2_f16: hand-over formats inside an SK block (engine.run_skblock)."""
    precision: int
    split_ws: Optional[torch.Tensor] = None
    shadows: bool = True
    shadow_fused: bool = True
    flash_stats: bool = True
    hidden_f16: bool = True
    hidden_koct: bool = True
    split_handover: bool = False     # f16x3: GEMM-to-GEMM tensors stored already split (hi, lo k-octet images) for the consumer's DMA (measured slower: off)
    pw_fold: bool = True
    x2_f16: bool = True
    ffn_pairs: bool = True      # an SK block's ffn1 / ffn2 as ONE launch where sf_ffn_pair has the shape (csrc/ffn_pair.hip)
    head_pairs: bool = False    # ... also the flow head's (grouped view, fp32 residual)
    sk_tail: bool = True        # an SK block's pw -> ffn2.0 -> ffn2.2 as ONE launch where sf_sk_tail has the shape (csrc/sk_tail.hip)
    sk_tail_all: bool = False   # ... also where it does not pay (the flow head's shape, small launches)

    def no_split(self) -> "Ctx":
        """The same context without the split-K scratch (ONE buffer: only one stream may use it at a time)."""
        return self if self.split_ws is None else replace(self, split_ws=None)


def _cx(cx: Optional["Ctx"]) -> "Ctx":
    return cx if cx is not None else Ctx(PRECISION)


class Profiler:
    """Per-launch HIP-event timing on the stream the kernels are launched on (torch's current stream).
    Used by bench.py's instrumented pass: every wrapped op records (name, algorithmic flops, algorithmic
    bytes, start event, end event)."""

    def __init__(self):
        self.records = []

    def launch(self, name, flops, nbytes, fn, products=1.0):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records.append((name, float(flops), float(nbytes), s, e, float(flops) * float(products)))

    def summary(self):
        """per name: launches, ms, algorithmic flops and bytes, and mfma_flops = flops x MFMA products per algorithmic
        product (3 / 2 / 1 in f16x3 / f16x2 / f16 or a single-product layer; fp32 MFMA counts as 1)."""
        torch.cuda.synchronize()
        out = {}
        for name, fl, by, s, e, mf in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "mfma_flops": 0.0})
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += by
            d["mfma_flops"] += mf
        return out


# SF_DEBUG_RANGE=1: before every contraction that rounds / splits an fp32 operand to fp16, check max |x| < 65504 (the fp16
# range: above it `hi` saturates silently) -- a host-side, synchronising debug aid; raises RuntimeError
DEBUG_RANGE = os.environ.get("SF_DEBUG_RANGE", "0") == "1"
F16_MAX = 65504.0


def _check_range(X: "Planes", what: str) -> None:
    if X.f16 or X.group:
        return
    m = float(X.tensor().abs().max())
    if not (m < F16_MAX):
        raise RuntimeError(f"SF_DEBUG_RANGE: {what}: max |activation| = {m:.4g} does not fit fp16 (65504): the split-precision "
                           "modes would saturate it silently; run this layer with precision='fp32'")


PROFILER: Optional[Profiler] = None
PROFILE_SHAPES = False      # tag GEMM launches with their shape in the profiler (bench.py --gemm-shapes)


def _launch(name: str, flops: float, nbytes: float, fn, products: float = 1.0) -> None:
    if PROFILER is None:
        fn()
    else:
        PROFILER.launch(name, flops, nbytes, fn, products)


def _products(prec: int) -> float:
    return {PRECISION_FP32: 1.0, PRECISION_F16X3: 3.0, PRECISION_F16X2: 2.0, PRECISION_F16: 1.0}[prec]


def on_tensor_device(fn):
    """Run `fn` with the device of its first tensor / Planes / torch.device argument current.  The C ABI enqueues on
    the stream it is given and `_lib.stream()` is the CURRENT device's stream, so a launch on tensors of another GPU
    would otherwise mix a device-0 stream with device-1 pointers."""
    import functools

    def _device(a):
        if isinstance(a, torch.Tensor):
            return a.device if a.is_cuda else None
        if isinstance(a, Planes):
            return a.base.device
        if isinstance(a, BlockedVolume):
            return a.buf.device
        if isinstance(a, torch.device):
            return a if a.type == "cuda" else None
        if isinstance(a, (list, tuple)) and a:
            return _device(a[0])
        return None

    @functools.wraps(fn)
    def wrapped(*args, **kw):
        for a in list(args) + list(kw.values()):
            d = _device(a)
            if d is not None:
                with torch.cuda.device(d):
                    return fn(*args, **kw)
        return fn(*args, **kw)

    return wrapped


def _dev_check(t: torch.Tensor) -> None:
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("streamflow_amd ops need contiguous float32 tensors on the GPU "
                           f"(got device={t.device}, dtype={t.dtype}, contiguous={t.is_contiguous()}); "
                           "there is no CPU fallback")


@dataclass(frozen=True)
class Planes:
    """View of channel-major planes [n_img][rows][P] inside ``base`` (a flat fp32 device tensor)."""
    base: torch.Tensor
    off: int            # float offset of (img 0, row 0)
    img_stride: int     # floats between consecutive images
    n_img: int
    rows: int
    P: int
    group: int = 0          # rows per group for '(B T) C -> B (T C)' views (0 = plain)
    group_stride: int = 0
    f16: bool = False       # the planes hold IEEE fp16 values (img_stride then counts halves; `off` stays in floats of
                            # the underlying fp32 allocation): GEMM-to-GEMM hand-over in the f16x2 mode (SfGemm.c_f16)
    koct: bool = False      # (with f16) k-octet planes [ceil(rows/8)][P][8] instead of rows [rows][P]: the consumer GEMM's
                            # LDS image, moved there by DMA (SF_LAYOUT_F16_KOCT / c_f16 = 2)
    split: bool = False     # (with f16 and koct, f16x3 mode) the values ALREADY split for the next GEMM, x = hi + lo: two k-octet images
                            # per image, lo behind hi (SF_LAYOUT_SPLIT_KOCT / c_f16 = 4) -- 4 bytes per element like fp32 planes
    shadow: Optional["Planes"] = None   # fp16 k-octet COPY of these fp32 planes, kept current by every producer (f16x2 mode):
                                        # a GEMM on the DMA-fed tile reads it as its B operand, everything else (residuals,
                                        # element-wise kernels) keeps reading the fp32 planes

    @staticmethod
    def of(t: torch.Tensor) -> "Planes":
        """Wrap a contiguous [n, C, h, w] or [n, C, P] tensor."""
        _dev_check(t)
        n, c = t.shape[0], t.shape[1]
        p = t.numel() // (n * c)
        return Planes(t, 0, c * p, n, c, p)

    @property
    def ptr(self) -> int:
        return self.base.data_ptr() + 4 * self.off

    def slice(self, r0: int, r1: int, unshadowed: bool = False) -> "Planes":
        """Rows r0 .. r1-1.  A slice of shadowed planes that does not start on an octet cannot carry the k-octet copy along:
        the caller must say so (`unshadowed=True`) and keep that copy current itself -- a producer writing through such a
        slice would otherwise leave the parent's copy stale without any error."""
        assert 0 <= r0 < r1 <= self.rows and self.group == 0
        if self.f16:                                # k-octet planes: whole octets only
            assert self.koct and r0 % 8 == 0 and self.shadow is None and not self.split
            return replace(self, off=self.off + (r0 // 8) * self.P * 4, rows=r1 - r0)
        if self.shadow is not None and r0 % 8 != 0 and not unshadowed:
            raise RuntimeError(f"Planes.slice({r0}, {r1}): shadowed planes can only be sliced at octet boundaries "
                               "(pass unshadowed=True and maintain the k-octet copy explicitly)")
        sh = self.shadow.slice(r0, r1) if (self.shadow is not None and r0 % 8 == 0) else None
        return replace(self, off=self.off + r0 * self.P, rows=r1 - r0, shadow=sh)

    def tensor(self) -> torch.Tensor:
        """Materialise as a [n_img, rows, P] torch view (only for contiguous-row, ungrouped views)."""
        assert self.group == 0
        if self.f16 and self.koct and self.split:      # hi + lo as float32 (tests only)
            oc = (self.rows + 7) // 8
            hi = replace(self, split=False)
            lo = replace(self, split=False, off=self.off + oc * self.P * 4)
            return hi.tensor().float() + lo.tensor().float()
        if self.f16 and self.koct:                     # logical [n_img, rows, P] copy out of the octet planes (tests only)
            h = self.base.view(torch.float16)
            oc = (self.rows + 7) // 8
            t = torch.as_strided(h, (self.n_img, oc, self.P, 8), (self.img_stride, self.P * 8, 8, 1),
                                 2 * (self.base.storage_offset() + self.off))
            return t.permute(0, 1, 3, 2).reshape(self.n_img, oc * 8, self.P)[:, : self.rows]
        if self.f16:
            h = self.base.view(torch.float16)
            return torch.as_strided(h, (self.n_img, self.rows, self.P), (self.img_stride, self.P, 1),
                                    2 * (self.base.storage_offset() + self.off))
        return torch.as_strided(self.base, (self.n_img, self.rows, self.P), (self.img_stride, self.P, 1),
                                self.base.storage_offset() + self.off)


class Workspace:
    """Bump allocator over one flat device buffer; `take` hands out contiguous [n_img][rows][P] planes."""

    def __init__(self, floats: int, device):
        self.buf = torch.empty(int(floats), dtype=torch.float32, device=device)
        self.top = 0

    def take(self, n_img: int, rows: int, P: int) -> Planes:
        size = n_img * rows * P
        start = (self.top + 63) // 64 * 64          # 256-byte aligned
        if start + size > self.buf.numel():
            raise RuntimeError("workspace exhausted")
        self.top = start + size
        return Planes(self.buf, start, rows * P, n_img, rows, P)

    def take_flat(self, floats: int) -> torch.Tensor:
        start = (self.top + 63) // 64 * 64
        if start + floats > self.buf.numel():
            raise RuntimeError("workspace exhausted")
        self.top = start + floats
        return self.buf[start:start + floats]


class PackedLinear:
    """A 1x1-conv / Linear weight [Cout, Cin(,1,1)] repacked K-major ([Cin][Cout padded to 4]) for sf_gemm."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], device, conv3x3: bool = False):
        # single: in the f16x2 mode this layer's weights enter the products as their round-to-nearest fp16 `hi` image alone
        # (ONE MFMA per product instead of two, no `lo` plane read): set per layer by HotPathEngine(single_layers=...)
        self.single = False
        w = weight.detach().to(device=device, dtype=torch.float32)
        if conv3x3:                                   # [Cout, Cin, 3, 3] -> k = (ky*3+kx)*Cin + c
            cout = w.shape[0]
            w2 = w.permute(2, 3, 1, 0).reshape(-1, cout)
        else:
            w2 = w.reshape(w.shape[0], -1).t()
        self.K, self.M = w2.shape
        self.lda = (self.M + 127) // 128 * 128            # zero padded to [K up to 32][M up to 128] (a_padded)
        wt = torch.zeros((self.K + 31) // 32 * 32, self.lda, dtype=torch.float32, device=device)
        wt[: self.K, : self.M] = w2
        self.wt = wt.contiguous()
        self.bias = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
        self.conv3x3 = conv3x3
        # split-precision image: w = hi + lo (fp16 each) in k-octet planes [K padded to 32 / 8][M padded to 128][8], zero
        # padded (SF_LAYOUT_SPLIT_F16): 16 bytes = one MFMA operand octet of one output row.
        # The tensor is first scaled by a power of two so that max|w| lands in [1, 2): fp16 keeps 11 bits only for
        # |x| >= 6.1e-5, so the lo part of an unscaled small weight (|w| < 0.12) would fall into fp16 subnormals
        # (absolute spacing 6e-8) and a layer whose weights are all ~1e-3 would keep ~15 bits instead of ~21.  The
        # scale is exact (power of two) and is undone in the epilogue: alpha' = alpha / s, bias' = bias * s.
        # (K padded to 128: the tiled kernels read K rounded up to 32, the activation-stationary one whole weight stages)
        mp, kp = (self.M + 127) // 128 * 128, (self.K + 127) // 128 * 128
        self.k_pad = 128
        wmax = float(w2.abs().max()) if w2.numel() else 0.0
        if not torch.isfinite(torch.tensor(wmax)):
            raise RuntimeError("PackedLinear: non-finite weight")
        self.split_scale = 2.0 ** min(14, max(-14, -math.floor(math.log2(wmax)))) if wmax > 0 else 1.0
        wm = torch.zeros(mp, kp, dtype=torch.float32, device=device)
        wm[: self.M, : self.K] = w2.t() * self.split_scale
        hi = wm.to(torch.float16)
        lo = (wm - hi.float()).to(torch.float16)
        self.hi = hi.view(mp, kp // 8, 8).permute(1, 0, 2).contiguous()
        self.lo = lo.view(mp, kp // 8, 8).permute(1, 0, 2).contiguous()
        self.lda_h = mp
        self.bias_split = None if self.bias is None else (self.bias * self.split_scale).contiguous()
        # achieved split accuracy relative to the largest weight (reported, used by tests)
        self.split_error = float(((hi.float() + lo.float()) - wm).abs().max() / max(wmax * self.split_scale, 1e-30))


class PackedPair:
    """The two 1x1 convolutions of an SK block's FFN (update.py:14-16: conv -> GELU -> conv) as ONE weight stream for
    sf_ffn_pair (csrc/ffn_pair.hip): 1-KB MFMA fragments in consumption order, built per (products of layer 1, products of
    layer 2) on first use.  `first` / `second` are the layers' PackedLinear objects (their `single` flags, scales and biases
    are the source of truth: the pair computes exactly what the two sf_gemm launches compute)."""

    S = 16                                               # fragments per stage (csrc/ffn_pair.hip)

    def __init__(self, first: "PackedLinear", second: "PackedLinear"):
        assert first.M == second.K and not first.conv3x3 and not second.conv3x3
        self.first, self.second = first, second
        self.K1, self.H, self.M2 = first.K, first.M, second.M
        self._streams = {}

    def products(self, cx: "Ctx"):
        one = cx.precision == PRECISION_F16
        return (1 if (one or self.first.single) else 2), (1 if (one or self.second.single) else 2)

    @staticmethod
    def _split(A: "PackedLinear", rows: int, cols: int):
        """(hi, lo) fp16 images of split_scale * W, zero-padded to [rows, cols] (the same rounding as PackedLinear.hi / .lo)."""
        w = torch.zeros(rows, cols, dtype=torch.float32, device=A.wt.device)
        w[: A.M, : A.K] = A.wt[: A.K, : A.M].t() * A.split_scale
        hi = w.to(torch.float16)
        return hi, (w - hi.float()).to(torch.float16)

    def stream(self, pm1: int, pm2: int) -> torch.Tensor:
        key = (pm1, pm2)
        if key in self._streams:
            return self._streams[key]
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("PackedPair.stream: weight stream requested for the first time inside a graph capture; build it "
                               "before (HotPathEngine does at construction)")
        nk1, nt2, hp = (self.K1 + 31) // 32, (self.M2 + 15) // 16, (self.H + 31) // 32
        fpad = int(_lib.load().sf_ffn_pair_frags(self.K1, self.M2, pm1, pm2))
        h1, l1 = self._split(self.first, hp * 32, nk1 * 32)
        h2, l2 = self._split(self.second, nt2 * 16, hp * 32)
        dev = h1.device
        # layer 1: fragment (m, u, s, plane) = rows 32 m + 16 u .. + 15, k 32 s .. + 31; lane (row, kq) holds k = 8 kq .. + 7
        planes1 = [l1, h1] if pm1 == 2 else [h1]
        f1 = torch.stack([w.view(hp, 2, 16, nk1, 4, 8) for w in planes1], dim=0)          # [plane, m, u, row, s, kq, i]
        f1 = f1.permute(1, 2, 4, 0, 5, 3, 6).reshape(hp, 2 * nk1 * pm1, 64 * 8)          # [m][(u, s, plane)][(kq, row), i]
        # layer 2: fragment (m, t, plane) = rows 16 t .. + 15, hidden rows 32 m .. + 31 in the order the layer-1 accumulators
        # hold them: k = 8 kq + i  <->  4 kq + i (i < 4) | 16 + 4 kq + i - 4 (i >= 4)
        kq = torch.arange(4, device=dev).view(4, 1)
        i = torch.arange(8, device=dev).view(1, 8)
        perm = torch.where(i < 4, 4 * kq + i, 16 + 4 * kq + i - 4).reshape(-1)             # [32] hidden row of (kq, i)
        planes2 = [l2, h2] if pm2 == 2 else [h2]
        f2 = torch.stack([w.view(nt2, 16, hp, 32)[..., perm].reshape(nt2, 16, hp, 4, 8) for w in planes2], dim=0)
        f2 = f2.permute(3, 1, 0, 4, 2, 5).reshape(hp, nt2 * pm2, 64 * 8)                  # [m][(t, plane)][(kq, row), i]
        pad = torch.zeros(hp, fpad - f1.shape[1] - f2.shape[1], 64 * 8, dtype=torch.float16, device=dev)
        st = torch.cat([f1, f2, pad], dim=1).contiguous().view(-1)
        assert st.numel() * 2 == hp * fpad * 1024
        self._streams[key] = st
        return st


class PackedTail:
    """The back half of an SK block -- pw (residual folded: W + I), ffn2.0, ffn2.2 (update.py:35-36, :14-16) -- as ONE weight stream for
    sf_sk_tail (csrc/sk_tail.hip): 1-KB fragments of 32 rows x 16 k in consumption order, units padded to 16-fragment stages (layout:
    include/streamflow_hip.h).  The PackedLinear objects stay the source of truth (rounding, power-of-two scales, biases, `single`)."""

    S = 16

    def __init__(self, pw_res: "PackedLinear", f0: "PackedLinear", f2: "PackedLinear"):
        assert pw_res.M == pw_res.K == f0.K and f0.M == f2.K and not (pw_res.conv3x3 or f0.conv3x3 or f2.conv3x3)
        self.layers = (pw_res, f0, f2)
        self.C, self.H, self.M2 = pw_res.K, f0.M, f2.M
        self._streams = {}

    def products(self, cx: "Ctx") -> Optional[int]:
        """MFMA products per weight: 1 or 2 -- the kernel takes ONE value for the three layers; a mixed set keeps the three launches."""
        if cx.precision == PRECISION_F16 or all(l.single for l in self.layers):
            return 1
        return 2 if not any(l.single for l in self.layers) else None

    def built(self, pm: Optional[int]) -> bool:
        return pm is not None and int(_lib.load().sf_sk_tail_frags(self.C, self.H, self.M2, pm)) > 0

    def stream(self, pm: int) -> torch.Tensor:
        if pm in self._streams:
            return self._streams[pm]
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("PackedTail.stream: weight stream requested for the first time inside a graph capture; build it before "
                               "(HotPathEngine does at construction)")
        nc, nh, nm = (self.C + 31) // 32, self.H // 32, (self.M2 + 31) // 32
        ks = 2 * nc
        stage, group, one = C.c_int(0), C.c_int(0), C.c_int(0)
        total = int(_lib.load().sf_sk_tail_layout(self.C, self.H, self.M2, pm, C.byref(stage), C.byref(group), C.byref(one)))
        assert total > 0 and nh % group.value == 0
        w1, w2, w3 = (PackedPair._split(l, r, c) for l, (r, c) in zip(self.layers, ((nc * 32, nc * 32), (nh * 32, nc * 32), (nm * 32, nh * 32))))
        dev = w1[0].device
        khalf = torch.arange(2, device=dev).view(2, 1)
        i = torch.arange(8, device=dev).view(1, 8)
        natural = (8 * khalf + i).reshape(-1)                                          # column of (khalf, i) inside a natural k-step
        acc_order = ((i & 3) + 8 * (i >> 2) + 4 * khalf).reshape(-1)                   # ... inside a k-step made of accumulator registers
        out = []
        zero = torch.zeros(512, dtype=torch.float16, device=dev)

        def emit(w, r0, cols):                                                         # lo before hi; lane (khalf, row m): 8 halves
            for plane in ((w[1], w[0]) if pm == 2 else (w[0],)):
                out.append(plane[r0:r0 + 32][:, cols].reshape(32, 2, 8).permute(1, 0, 2).reshape(512))

        def pad():
            while (len(out) % stage.value) != 0:
                out.append(zero)

        for t in range(nc):                                                            # phase 1: pw row tiles over the natural k-steps
            for k in range(ks):
                emit(w1, 32 * t, 16 * k + natural)
            if not one.value:
                pad()
        pad()
        for th in range(nh):                                                           # phase 2: per 32 hidden rows, `group` tiles per unit
            for k in range(ks):                                                        # ffn2.0 over x4 (k-step = (tile k / 2, half k % 2))
                emit(w2, 32 * th, 32 * (k // 2) + 16 * (k % 2) + acc_order)
            for s_ in range(2):                                                        # ffn2.2's two k-steps from this hidden tile
                for m in range(nm):
                    emit(w3, 32 * m, 32 * th + 16 * s_ + acc_order)
            if (th + 1) % group.value == 0:
                pad()
        st = torch.cat(out).contiguous()
        assert st.numel() == int(_lib.load().sf_sk_tail_frags(self.C, self.H, self.M2, pm)) * 512, (st.numel() // 512, self.C, self.H, self.M2, pm)
        self._streams[pm] = st
        return st


def sk_tail_ok(tail: Optional[PackedTail], X: Planes, Y: Planes, cx: Optional["Ctx"] = None) -> bool:
    """Does sf_sk_tail run this block's back half?  fp16-activation arithmetic, x3 as fp16 ROWS, one product count for the three
    layers, a built shape, k-octet and / or fp32 outputs."""
    cx = _cx(cx)
    if tail is None or not cx.sk_tail or cx.precision not in (PRECISION_F16X2, PRECISION_F16) or not (cx.hidden_f16 and cx.hidden_koct):
        return False
    if not (X.f16 and not X.koct and X.group == 0 and Y.group == 0 and X.rows == tail.C and Y.rows == tail.M2 and X.P % 4 == 0):
        return False
    if Y.f16 and not Y.koct:
        return False
    return tail.built(tail.products(cx))


@on_tensor_device
def sk_tail(tail: PackedTail, X: Planes, Y: Planes, gelu_out: bool = False, cx: Optional["Ctx"] = None) -> None:
    """Y = ffn2(gelu(x3 + pw(x3))) [gelu'ed with gelu_out] (update.py:35-36): X = x3 as fp16 rows; Y fp16 k-octet planes, or fp32 planes
    (+ their k-octet copy Y.shadow)."""
    cx = _cx(cx)
    assert sk_tail_ok(tail, X, Y, cx) and Y.n_img == X.n_img and Y.P == X.P
    pm = tail.products(cx)
    st = tail.stream(pm)
    A1, A2, A3 = tail.layers
    g = _lib.SfSkTail()
    g.X, g.strideX, g.ldx = X.ptr, X.img_stride, X.P
    g.wstream, g.wstream_bytes = st.data_ptr(), st.numel() * 2
    g.bias1 = None if A1.bias_split is None else A1.bias_split.data_ptr()
    g.bias2 = None if A2.bias_split is None else A2.bias_split.data_ptr()
    g.bias3 = None if A3.bias_split is None else A3.bias_split.data_ptr()
    g.alpha1, g.alpha2, g.alpha3 = 1.0 / A1.split_scale, 1.0 / A2.split_scale, 1.0 / A3.split_scale
    g.N, g.batch, g.C, g.H, g.M2, g.pm, g.gelu_out = X.P, X.n_img, tail.C, tail.H, tail.M2, pm, int(bool(gelu_out))
    out_bytes = 2.0
    if Y.f16:
        g.Y16, g.strideY16, g.ldy16 = Y.ptr, Y.img_stride, Y.P
    else:
        g.Y, g.strideY, g.ldy = Y.ptr, Y.img_stride, Y.P
        out_bytes = 4.0
        if Y.shadow is not None and cx.shadows:
            sh = Y.shadow
            g.Y16, g.strideY16, g.ldy16 = sh.ptr, sh.img_stride, sh.P
            g.y16_partial = 1 if tail.M2 % 8 else 0
            out_bytes = 6.0
    n, P = X.n_img, X.P
    macs = tail.C * tail.C + tail.C * tail.H + tail.H * tail.M2
    name = "sk_tail" if not PROFILE_SHAPES else f"sk_tail C{tail.C} H{tail.H} M{tail.M2} b{n}"
    _launch(name, 2.0 * n * P * macs, n * P * (2.0 * tail.C + out_bytes * tail.M2) + 2.0 * macs * pm,
            lambda: _lib.check(_lib.load().sf_sk_tail(C.byref(g), _lib.stream()), "sf_sk_tail"), products=float(pm))
    if g.Y and not g.Y16:
        refresh_shadow(Y, cx)


class PackedTemporal:
    """The four contractions of the temporal transformer block (update.py:459-484 -> timm Block: qkv, proj, fc1, fc2) as ONE weight
    stream for sf_temporal_block (csrc/temporal.hip): 1-KB MFMA fragments in consumption order (layout: include/streamflow_hip.h).
    The PackedLinear objects stay the source of truth (rounding, power-of-two scales, biases, `single` flags)."""

    def __init__(self, qkv: "PackedLinear", proj: "PackedLinear", fc1: "PackedLinear", fc2: "PackedLinear"):
        self.layers = (qkv, proj, fc1, fc2)
        self.C, self.H = qkv.K, fc1.M
        self._streams = {}

    def built(self) -> bool:
        qkv, proj, fc1, fc2 = self.layers
        return (self.C == 128 and self.H == 256 and qkv.M == 384 and qkv.bias is None and (proj.M, proj.K) == (128, 128) and
                fc1.K == 128 and (fc2.M, fc2.K) == (128, 256) and not any(l.conv3x3 for l in self.layers))

    def products(self, cx: "Ctx") -> Optional[int]:
        """MFMA products per element: 1 (fp16 weights) or 2 (hi + lo) -- the kernel takes ONE value for the block; a layer set that
        mixes the two keeps the unfused launches (None)."""
        if cx.precision == PRECISION_F16 or all(l.single for l in self.layers):
            return 1
        return 2 if not any(l.single for l in self.layers) else None

    def stream(self, pm: int) -> torch.Tensor:
        if pm in self._streams:
            return self._streams[pm]
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("PackedTemporal.stream: weight stream requested for the first time inside a graph capture; build it "
                               "before (HotPathEngine does at construction)")
        qkv, proj, fc1, fc2 = (PackedPair._split(l, l.M, l.K) for l in self.layers)       # (hi, lo) of scale * W, [M, K] fp16
        dev = qkv[0].device
        kq = torch.arange(4, device=dev).view(4, 1)
        i = torch.arange(8, device=dev).view(1, 8)
        perm = torch.where(i < 4, 4 * kq + i, 16 + 4 * kq + i - 4).reshape(-1)             # input row of column (kq, i) of a k-step
        out = []

        def emit(w, m, s, permuted):
            for plane in ((w[1], w[0]) if pm == 2 else (w[0],)):                           # lo before hi
                blk = plane[16 * m:16 * m + 16, 32 * s:32 * s + 32]
                if permuted:
                    blk = blk[:, perm]
                out.append(blk.reshape(16, 4, 8).permute(1, 0, 2).reshape(512))           # lane (kq, row): 8 halves

        for m in range(16):                                                                # q and k rows
            for s in range(4):
                emit(qkv, m, s, False)
        for p_ in range(4):                                                                # v two tiles at a time, then proj's k-step
            for tile in (16 + 2 * p_, 16 + 2 * p_ + 1):
                for s in range(4):
                    emit(qkv, tile, s, False)
            for m in range(8):
                emit(proj, m, p_, True)
        for h in range(8):                                                                 # fc1 two tiles at a time, then fc2's k-step
            for tile in (2 * h, 2 * h + 1):
                for s in range(4):
                    emit(fc1, tile, s, True)
            for m in range(8):
                emit(fc2, m, h, True)
        st = torch.cat(out).contiguous()
        assert st.numel() == int(_lib.load().sf_temporal_block_frags(pm)) * 512
        self._streams[pm] = st
        return st


class PackedMask:
    """mask.2 (update.py:758: Conv1x1 256 -> 576) as the weight stream of sf_mask_upsample (csrc/mask_upsample.hip): row tile 0 .. 35,
    k-step 0 .. 7, `lo` before `hi`."""

    def __init__(self, layer: "PackedLinear"):
        self.layer = layer
        self._streams = {}

    def built(self) -> bool:
        return (self.layer.M, self.layer.K) == (576, 256) and not self.layer.conv3x3

    def products(self, cx: "Ctx") -> int:
        return 1 if (cx.precision == PRECISION_F16 or self.layer.single) else 2

    def stream(self, pm: int) -> torch.Tensor:
        if pm in self._streams:
            return self._streams[pm]
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("PackedMask.stream: weight stream requested for the first time inside a graph capture")
        hi, lo = PackedPair._split(self.layer, 576, 256)
        planes = [lo, hi] if pm == 2 else [hi]
        f = torch.stack([w.view(36, 16, 8, 4, 8) for w in planes], dim=0)                   # [plane, tile, row, s, kq, i]
        st = f.permute(1, 3, 0, 4, 2, 5).reshape(-1).contiguous()                           # [tile][s][plane][(kq, row), i]
        assert st.numel() == int(_lib.load().sf_mask_upsample_frags(pm)) * 512
        self._streams[pm] = st
        return st


def mask_upsample_ok(pack: Optional[PackedMask], M256: Planes, cx: Optional["Ctx"] = None) -> bool:
    cx = _cx(cx)
    return (pack is not None and pack.built() and cx.precision in (PRECISION_F16X2, PRECISION_F16) and cx.shadows and
            M256.shadow is not None and M256.rows == 256 and M256.group == 0)


@on_tensor_device
def mask_upsample(pack: PackedMask, M256: Planes, flow: torch.Tensor, out: torch.Tensor, h: int, w: int,
                  cx: Optional["Ctx"] = None) -> None:
    """out [n, 2, 8h, 8w] = convex upsampling of flow [n, 2, h, w] with mask = 0.25 * mask.2(M256) (streamflow.py:82-93), the mask
    never written: M256 = relu(mask.0(net)) as planes with a k-octet copy (the operand)."""
    cx = _cx(cx)
    assert mask_upsample_ok(pack, M256, cx) and M256.P == h * w and flow.is_contiguous() and out.is_contiguous()
    sh, A = M256.shadow, pack.layer
    pm = pack.products(cx)
    st = pack.stream(pm)
    g = _lib.SfMaskUpsample()
    g.X16, g.strideX, g.ldx = sh.ptr, sh.img_stride, sh.P
    g.wstream, g.wstream_bytes = st.data_ptr(), st.numel() * 2
    g.bias = None if A.bias_split is None else A.bias_split.data_ptr()
    g.flow, g.out = flow.data_ptr(), out.data_ptr()
    g.n_img, g.h, g.w, g.K, g.M, g.pm = M256.n_img, h, w, 256, 576, pm
    g.alpha = 0.25 / A.split_scale
    n, P = M256.n_img, h * w
    _launch("mask_upsample", 2.0 * 576 * 256 * n * P, (2.0 * 256 + 4.0 * 2 + 4.0 * 128) * n * P,
            lambda: _lib.check(_lib.load().sf_mask_upsample(C.byref(g), _lib.stream()), "sf_mask_upsample"), products=float(pm))


def temporal_block_ok(pack: Optional[PackedTemporal], X: Planes, TT: int, cx: Optional["Ctx"] = None) -> bool:
    """Does sf_temporal_block run this block?  fp16-activation arithmetic with the k-octet copy of the tokens at hand, the built
    shape (C = 128, hidden 256, <= 3 tokens per pixel), one product count for all four layers."""
    cx = _cx(cx)
    if pack is None or cx.precision not in (PRECISION_F16X2, PRECISION_F16) or not (cx.shadows and cx.hidden_f16 and cx.hidden_koct):
        return False
    return (pack.built() and pack.products(cx) is not None and 1 <= TT <= 3 and X.shadow is not None and X.rows == 128 and
            X.group == 0 and X.n_img % TT == 0)


@on_tensor_device
def temporal_block(pack: PackedTemporal, X: Planes, Y: Planes, TT: int, ln1, ln2, eps: float = 1e-5, cx: Optional["Ctx"] = None) -> None:
    """Y = Block(X) over the TT tokens of every pixel (update.py:481-484): X fp32 planes with a k-octet copy (the operand), image =
    clip * TT + frame; Y fp32 planes (+ their k-octet copy Y.shadow).  ln1 / ln2 = (weight, bias) of the two LayerNorms."""
    cx = _cx(cx)
    assert temporal_block_ok(pack, X, TT, cx)
    sh = X.shadow
    assert sh.f16 and sh.koct and not Y.f16 and Y.rows == 128 and Y.n_img == X.n_img and Y.P == X.P and Y.group == 0
    pm = pack.products(cx)
    st = pack.stream(pm)
    qkv, proj, fc1, fc2 = pack.layers
    g = _lib.SfTemporalBlock()
    g.X16, g.strideX, g.ldx = sh.ptr, sh.img_stride, sh.P
    g.wstream, g.wstream_bytes = st.data_ptr(), st.numel() * 2
    g.ln1_w, g.ln1_b, g.ln2_w, g.ln2_b = ln1[0].data_ptr(), ln1[1].data_ptr(), ln2[0].data_ptr(), ln2[1].data_ptr()
    g.bias_proj = None if proj.bias_split is None else proj.bias_split.data_ptr()
    g.bias_fc1 = None if fc1.bias_split is None else fc1.bias_split.data_ptr()
    g.bias_fc2 = None if fc2.bias_split is None else fc2.bias_split.data_ptr()
    g.Y, g.strideY, g.ldy = Y.ptr, Y.img_stride, Y.P
    nbytes = 2.0 + 4.0
    if Y.shadow is not None and cx.shadows:
        g.Y16, g.strideY16, g.ldy16 = Y.shadow.ptr, Y.shadow.img_stride, Y.shadow.P
        nbytes += 2.0
    g.N, g.B, g.TT, g.C, g.H, g.pm = X.P, X.n_img // TT, TT, 128, 256, pm
    g.alpha_qkv, g.alpha_proj, g.alpha_fc1, g.alpha_fc2 = (1.0 / l.split_scale for l in pack.layers)
    g.ss_proj, g.ss_fc2, g.eps, g.scale = proj.split_scale, fc2.split_scale, eps, 128 ** -0.5
    n, P = X.n_img, X.P
    flops = 2.0 * n * P * (384 * 128 + 128 * 128 + 256 * 128 + 128 * 256) + 4.0 * n * P * TT * 128
    _launch("temporal_block", flops, nbytes * n * 128 * P,
            lambda: _lib.check(_lib.load().sf_temporal_block(C.byref(g), _lib.stream()), "sf_temporal_block"), products=float(pm))


# (K1, M2) of the shapes csrc/ffn_pair.hip is built for, by mode (0: an ffn2 pair, 1: an ffn1 pair)
PAIR_SHAPES = {1: {(128, 128), (256, 256), (324, 324), (384, 384)},
               0: {(128, 64), (256, 192), (256, 126), (324, 256), (384, 6), (256, 4), (128, 2)}}


def ffn_pair_ok(pair: Optional[PackedPair], X: Planes, mode: int, cx: Optional["Ctx"] = None) -> bool:
    """Does sf_ffn_pair run this FFN?  fp16-activation arithmetic with k-octet hand-over, a k-octet operand (the planes themselves
    or their copy; grouped '(B T) C -> B (T C)' views in groups of a multiple of 32 rows), one of the built shapes."""
    cx = _cx(cx)
    if pair is None or cx.precision not in (PRECISION_F16X2, PRECISION_F16) or not (cx.shadows and cx.hidden_f16 and cx.hidden_koct):
        return False
    src = X if (X.f16 and X.koct) else X.shadow
    if src is None or X.P % 4 or (src.group and (src.group % 32 or pair.K1 % src.group)):
        return False
    pm = pair.products(cx)
    return (pair.K1, pair.M2) in PAIR_SHAPES[mode] and pm in ((1, 1), (2, 1), (2, 2))


@on_tensor_device
def ffn_pair(pair: PackedPair, X: Planes, Y: Planes, mode: int, dw_w: Optional[torch.Tensor] = None,
             dw_b: Optional[torch.Tensor] = None, gelu_out: bool = False, cx: Optional["Ctx"] = None) -> None:
    """mode 0: Y = ffn(X) [gelu'ed with gelu_out]: Y fp16 k-octet planes, or fp32 planes (+ their k-octet copy Y.shadow).
    mode 1: Y (fp16 ROWS) = gelu(x1 + dw_w * x1 + dw_b), x1 = gelu(X + ffn(X))  (update.py:31-32).  X: k-octet planes or planes
    with a k-octet copy."""
    cx = _cx(cx)
    assert ffn_pair_ok(pair, X, mode, cx)
    src = X if (X.f16 and X.koct) else X.shadow
    pm1, pm2 = pair.products(cx)
    st = pair.stream(pm1, pm2)
    g = _lib.SfFfnPair()
    g.X, g.strideX, g.ldx = src.ptr, src.img_stride, src.P
    g.x_group, g.x_group_stride = src.group, src.group_stride
    g.wstream, g.wstream_bytes = st.data_ptr(), st.numel() * 2
    A1, A2 = pair.first, pair.second
    g.bias1 = None if A1.bias_split is None else A1.bias_split.data_ptr()
    g.bias2 = None if A2.bias_split is None else A2.bias_split.data_ptr()
    g.alpha1, g.alpha2 = 1.0 / A1.split_scale, 1.0 / A2.split_scale
    g.N, g.batch, g.K1, g.H, g.M2 = X.P, X.n_img, pair.K1, pair.H, pair.M2
    g.pm1, g.pm2, g.mode, g.gelu_out = pm1, pm2, int(mode), int(bool(gelu_out))
    assert Y.n_img == X.n_img and Y.P == X.P and Y.rows == pair.M2 and Y.group == 0
    out_bytes = 2.0
    if mode == 1:
        assert Y.f16 and not Y.koct and dw_w is not None and dw_b is not None
        g.dw_w, g.dw_b = dw_w.data_ptr(), dw_b.data_ptr()
        if not X.f16:                                 # fp32 planes with a k-octet copy: the residual stays the fp32 value
            assert X.group == src.group
            g.R32, g.strideR32, g.ldr32, g.r32_group_stride = X.ptr, X.img_stride, X.P, X.group_stride
        g.C16, g.strideC16, g.ldc16 = Y.ptr, Y.img_stride, Y.P
    elif Y.f16:
        assert Y.koct
        g.C16, g.strideC16, g.ldc16 = Y.ptr, Y.img_stride, Y.P
    else:
        g.C, g.strideC, g.ldc = Y.ptr, Y.img_stride, Y.P
        out_bytes = 4.0
        if Y.shadow is not None and cx.shadows:
            sh = Y.shadow
            g.C16, g.strideC16, g.ldc16 = sh.ptr, sh.img_stride, sh.P
            g.c16_partial = 1 if pair.M2 % 8 else 0
            out_bytes = 6.0
    n, P = X.n_img, X.P
    flops = 2.0 * n * P * (pair.K1 * pair.H + pair.H * pair.M2)
    mf = 2.0 * n * P * (pair.K1 * pair.H * pm1 + pair.H * pair.M2 * pm2)
    nbytes = n * P * (2.0 * pair.K1 * (2 if mode == 1 else 1) + out_bytes * pair.M2) + 2.0 * (pair.K1 * pair.H * pm1 + pair.H * pair.M2 * pm2)
    name = "ffn_pair" if not PROFILE_SHAPES else f"ffn_pair K{pair.K1} H{pair.H} M{pair.M2} b{n} m{mode}"
    _launch(name, flops, nbytes, lambda: _lib.check(_lib.load().sf_ffn_pair(C.byref(g), _lib.stream()), "sf_ffn_pair"),
            products=mf / flops)
    if g.C and not g.C16:
        refresh_shadow(Y, cx)


def new_shadow(X: Planes, device) -> Planes:
    """Allocate the k-octet copy of fp32 planes X (zeroed: rows past X.rows in the last octet must stay finite)."""
    assert not X.f16 and X.group == 0
    noct = (X.rows + 7) // 8
    base = torch.zeros(X.n_img * noct * X.P * 4, dtype=torch.float32, device=device)       # 8 halves = 4 floats
    return Planes(base, 0, noct * X.P * 8, X.n_img, X.rows, X.P, f16=True, koct=True)


@on_tensor_device
def pack_koct(X: Planes, Y: Planes) -> None:
    """fp32 planes -> their k-octet fp16 copy (sf_pack_koct)."""
    assert Y.f16 and Y.koct and not X.f16 and X.group == 0 and (X.rows, X.P, X.n_img) == (Y.rows, Y.P, Y.n_img)
    _launch("pack_koct", 0, 6.0 * X.n_img * X.rows * X.P,
            lambda: _lib.check(_lib.load().sf_pack_koct(X.ptr, X.img_stride, X.n_img, X.rows, X.P, Y.ptr, Y.img_stride,
                                                        _lib.stream()), "sf_pack_koct"))


def clock_probe(out: torch.Tensor, spin_us: int) -> None:
    """Enqueue sf_clock_probe on the CURRENT stream: out (int64[2], device) <- (shader cycles, 100 MHz ticks) over spin_us."""
    assert out.is_cuda and out.dtype == torch.int64 and out.numel() >= 2 and out.is_contiguous()
    _lib.check(_lib.load().sf_clock_probe(out.data_ptr(), int(spin_us), _lib.stream()), "sf_clock_probe")


def refresh_shadow(X: Planes, cx: Optional[Ctx] = None) -> None:
    """After a producer without a fused k-octet output wrote X: bring X.shadow up to date."""
    if X.shadow is not None and _cx(cx).shadows:
        pack_koct(replace(X, shadow=None), X.shadow)


def uses_dma_tile(M: int) -> bool:
    """Does sf_gemm run an M-row problem on the 128-row, DMA-fed tile (gemm_split.hip pick_tile)?  Of the tiled kernels only
    that one takes a k-octet fp16 B operand."""
    return (M + 127) // 128 * 128 * 4 <= M * 5


def takes_koct(M: int, K: int) -> bool:
    """Can sf_gemm read the B operand of an M x K layer as fp16 k-octet planes?  The activation-stationary kernel does for
    64 < K <= 640 at every M (gemm_bstat.hip), the tiled family on its 128-row tile."""
    return 64 < K <= 640 or uses_dma_tile(M)


def gemm(A: PackedLinear, X: Planes, Y: Planes, epilogue: int = EPI_NONE, R: Optional[Planes] = None,
         dw_w: Optional[torch.Tensor] = None, dw_b: Optional[torch.Tensor] = None, alpha: float = 1.0,
         hw: Optional[Sequence[int]] = None, algo: int = 0, cx: Optional[Ctx] = None) -> None:
    """Y[img] = epilogue(alpha * (W @ X[img] + bias)) for every image (batched over grid.z).
    algo: _lib.ALGO_AUTO / ALGO_TILED / ALGO_BSTAT (SfGemm.algo: force one kernel family; tests and A/B timing).
    cx: launch context (arithmetic mode, split-K scratch, hand-over switches); None = Ctx(ops.PRECISION)."""
    cx = _cx(cx)
    PRECISION, SHADOWS, SPLIT_WS = cx.precision, cx.shadows, cx.split_ws
    assert X.rows * (9 if A.conv3x3 else 1) == A.K, (X.rows, A.K)
    assert Y.rows == A.M and X.n_img == Y.n_img and X.P == Y.P, (Y.rows, A.M)
    if (X.shadow is not None and SHADOWS and PRECISION in (PRECISION_F16X2, PRECISION_F16) and not A.conv3x3 and
            takes_koct(A.M, A.K)):
        X = X.shadow                                  # the fp16 k-octet copy: both operands by LDS-DMA
    if DEBUG_RANGE and PRECISION != PRECISION_FP32:
        _check_range(X, f"sf_gemm M{A.M} K{A.K}")
    g = SfGemm()
    g.A, g.B, g.C = A.wt.data_ptr(), X.ptr, Y.ptr
    g.bias = None if A.bias is None else A.bias.data_ptr()
    g.M, g.N, g.K, g.batch = A.M, X.P, A.K, X.n_img
    g.lda, g.ldb, g.ldc = A.lda, X.P, Y.P
    g.strideA, g.strideB, g.strideC = 0, X.img_stride, Y.img_stride
    g.a_layout, g.b_layout = LAYOUT_K_MAJOR, LAYOUT_K_MAJOR
    g.a_padded = 1
    prec = PRECISION
    if A.single and prec == PRECISION_F16X2:
        prec = PRECISION_F16                          # per-layer single-product weights (same operand formats)
    if X.split or Y.split:                            # f16x3: the GEMM-to-GEMM tensor already split (hi, lo k-octet images)
        if prec != PRECISION_F16X3 or (X.f16 and not X.split) or (Y.f16 and not Y.split):
            raise RuntimeError("split k-octet planes are a hand-over format of the f16x3 mode only")
        if X.split:
            assert X.koct and X.group == 0
            g.b_layout = _lib.LAYOUT_SPLIT_KOCT
        if Y.split:
            assert Y.koct
            g.c_f16 = 4
    elif X.f16 or Y.f16:
        if prec not in (PRECISION_F16X2, PRECISION_F16):
            raise RuntimeError("fp16 activation planes are a hand-over format of the f16x2 / f16 modes only")
        if X.f16:
            g.b_layout = _lib.LAYOUT_F16_KOCT if X.koct else _lib.LAYOUT_F16_K_MAJOR
        if Y.f16:
            g.c_f16 = 2 if Y.koct else 1
    fused_shadow = False
    if (Y.shadow is not None and SHADOWS and not Y.f16 and prec in (PRECISION_F16X2, PRECISION_F16) and Y.P % 4 == 0
            and cx.shadow_fused):
        g.c_f16, g.C16, g.strideC16 = 3, Y.shadow.ptr, Y.shadow.img_stride       # fp32 planes + their k-octet copy
        fused_shadow = True
    if prec != PRECISION_FP32:
        g.a_layout = LAYOUT_SPLIT_F16
        g.A_hi, g.A_lo, g.lda_h, g.a_k_pad = A.hi.data_ptr(), A.lo.data_ptr(), A.lda_h, A.k_pad
        # the split image holds split_scale * W: C = alpha/s * (s W X + s b)
        alpha = alpha / A.split_scale
        g.bias = None if A.bias_split is None else A.bias_split.data_ptr()
    g.b_group, g.b_group_stride = X.group, X.group_stride
    if R is not None:
        assert R.rows == A.M and R.n_img == Y.n_img
        g.R, g.ldr, g.strideR = R.ptr, R.P, R.img_stride
        g.r_group, g.r_group_stride = R.group, R.group_stride
        if R.f16:                                     # the block input exists only as fp16 k-octets (correlation features)
            assert R.koct and R.group == 0
            g.r_f16 = 2
    if dw_w is not None:
        g.dw_w, g.dw_b = dw_w.data_ptr(), dw_b.data_ptr()
    if A.conv3x3:
        g.conv3x3, g.h, g.w = 1, int(hw[0]), int(hw[1])
    g.alpha, g.epilogue, g.precision, g.algo = float(alpha), int(epilogue), prec, int(algo)
    if SPLIT_WS is not None and prec != PRECISION_FP32:
        g.split_ws, g.split_ws_floats = SPLIT_WS.data_ptr(), SPLIT_WS.numel()
    name = "gemm" if not PROFILE_SHAPES else f"gemm M{g.M} K{g.K} b{g.batch} e{g.epilogue}"
    # algorithmic bytes: activations in (each input row once) + result out (+ residual in) per image, weights once
    nbytes = (g.batch * g.N * ((2.0 if (X.f16 and not X.split) else 4.0) * X.rows + (2.0 if (Y.f16 and not Y.split) else 6.0 if fused_shadow else 4.0) * g.M +
                               ((2.0 if R.f16 else 4.0) * g.M if R is not None else 0.0)) + 4.0 * g.M * g.K)
    _launch(name, 2.0 * g.M * g.N * g.K * g.batch, nbytes,
            lambda: _lib.check(_lib.load().sf_gemm(C.byref(g), _lib.stream()), "sf_gemm"), products=_products(prec))
    if not fused_shadow:
        refresh_shadow(Y, cx)


def gemm_split_ws_floats(M: int, N: int, K: int, batch: int, cx: Optional[Ctx] = None) -> int:
    """floats of scratch with which sf_gemm would split K on its own for this shape (0: it would not); see Ctx.split_ws."""
    if _cx(cx).precision == PRECISION_FP32:
        return 0
    return int(_lib.load().sf_gemm_split_ws_floats(M, N, K, batch))


def gemm_raw(cx: Optional[Ctx] = None, **kw) -> None:
    """Fully explicit sf_gemm call (used for the attention logits and attn @ v contractions)."""
    PRECISION = _cx(cx).precision
    g = SfGemm()
    g.alpha = 1.0
    for k, v in kw.items():
        setattr(g, k, v)
    g.precision = PRECISION
    name = "gemm_attn" if g.a_layout == LAYOUT_K_MINOR else "gemm"
    eb = 2.0 if g.b_layout == LAYOUT_F16_K_MINOR else 4.0
    nbytes = g.batch * (4.0 * g.M * g.K + eb * g.K * g.N + 4.0 * g.M * g.N * (2 if g.R else 1) *
                        (g.k_splits if g.k_splits > 1 else 1))
    _launch(name, 2.0 * g.M * g.N * g.K * g.batch, nbytes,
            lambda: _lib.check(_lib.load().sf_gemm(C.byref(g), _lib.stream()), "sf_gemm"), products=_products(PRECISION))


@on_tensor_device
def splitk_combine(partial: torch.Tensor, split_stride: int, k_splits: int, part_img_stride: int, R: Planes,
                   gamma: torch.Tensor, out: Planes) -> None:
    assert R.rows == out.rows and R.n_img == out.n_img
    _launch("splitk_combine", 0, 4.0 * (k_splits + 2) * R.n_img * R.rows * R.P,
            lambda: _lib.check(_lib.load().sf_splitk_combine(
                partial.data_ptr(), split_stride, k_splits, part_img_stride, R.ptr, R.img_stride, gamma.data_ptr(),
                out.ptr, out.img_stride, R.n_img, R.rows * R.P, _lib.stream()), "sf_splitk_combine"))


@on_tensor_device
def dwconv_res_gelu(X: Planes, wgt: torch.Tensor, bias: torch.Tensor, Y: Planes, h: int, w: int, k: int,
                    single: bool = False, cx: Optional[Ctx] = None) -> None:
    """Y = gelu(X + dwconv_KxK(X) + bias).  Y may be fp16 row planes (Planes.f16, not koct): the hand-over to a GEMM.
    X may be fp16 row planes too (f16x2 / f16 arithmetic, fp16 Y): sf_dwconv_res_gelu_f16in.
    single (f16x2 mode only): the weights enter the products as ONE fp16 value (a single-product layer)."""
    PRECISION = _cx(cx).precision
    assert X.rows == Y.rows == wgt.shape[0] and X.P == h * w and not X.koct and not Y.koct
    if DEBUG_RANGE and PRECISION != PRECISION_FP32:
        _check_range(X, f"sf_dwconv_res_gelu C{X.rows} k{k}")
    prec = PRECISION_F16 if (single and PRECISION == PRECISION_F16X2) else PRECISION
    if X.f16:
        assert Y.f16, "fp16 input needs fp16 output"
        call = lambda: _lib.check(_lib.load().sf_dwconv_res_gelu_f16in(X.ptr, X.img_stride, wgt.data_ptr(), bias.data_ptr(),
                                                                       Y.ptr, Y.img_stride, X.n_img, X.rows, h, w, k, prec,
                                                                       _lib.stream()), "sf_dwconv_res_gelu_f16in")
    else:
        call = lambda: _lib.check(_lib.load().sf_dwconv_res_gelu(X.ptr, X.img_stride, wgt.data_ptr(), bias.data_ptr(),
                                                                 Y.ptr, Y.img_stride, int(Y.f16), X.n_img, X.rows, h, w, k,
                                                                 prec, _lib.stream()), "sf_dwconv_res_gelu")
    _launch("dwconv%d" % k, 2.0 * k * k * X.n_img * X.rows * h * w,
            ((2.0 if X.f16 else 4.0) + (2.0 if Y.f16 else 4.0)) * X.n_img * X.rows * h * w, call,
            # matrix-core work: banded Toeplitz GEMMs (32 / 15 of the algorithmic flops) for K = 15 in the split modes, 2 or 3
            # products; the fp32 and 7 x 7 stencils run on the VALU
            products=((32.0 / 15.0) * _products(prec)) if (k == 15 and PRECISION != PRECISION_FP32) else 0.0)


@on_tensor_device
def layernorm_cm(X: Planes, gamma: torch.Tensor, beta: torch.Tensor, Y: Planes, eps: float = 1e-5) -> None:
    """Y fp32 planes, or fp16 k-octet planes (Planes.koct: the hand-over to a GEMM on the DMA-fed tile)."""
    assert not X.f16 and (not Y.f16 or Y.koct)
    ko = Y.f16
    _launch("layernorm", 0, (6.0 if ko else 8.0) * X.n_img * X.rows * X.P,
            lambda: _lib.check(_lib.load().sf_layernorm_cm(X.ptr, X.img_stride, gamma.data_ptr(), beta.data_ptr(),
                                                           None if ko else Y.ptr, 0 if ko else Y.img_stride,
                                                           Y.ptr if ko else None, Y.img_stride if ko else 0,
                                                           X.n_img, X.rows, X.P, eps, _lib.stream()), "sf_layernorm_cm"))


@on_tensor_device
def temporal_attn(QKV: Planes, OUT: Planes, B: int, TT: int, C_: int) -> None:
    assert QKV.img_stride == 3 * C_ * QKV.P and OUT.img_stride == C_ * OUT.P and QKV.n_img == B * TT
    assert (not OUT.f16 or OUT.koct) and not QKV.koct
    ko = OUT.f16                                          # fp16 k-octet planes: the hand-over to the proj GEMM
    fn = _lib.load().sf_temporal_attn_f16in if QKV.f16 else _lib.load().sf_temporal_attn     # (fp16 rows: the qkv GEMM's c_f16 = 1)
    _launch("temporal_attn", 0, ((6.0 if QKV.f16 else 12.0) + (2.0 if ko else 4.0)) * QKV.n_img * C_ * QKV.P,
            lambda: _lib.check(fn(QKV.ptr, None if ko else OUT.ptr, OUT.ptr if ko else None,
                                  B, TT, C_, QKV.P, _lib.stream()), "sf_temporal_attn"))


@on_tensor_device
def softmax_rows(x: torch.Tensor, rows: int, cols: int, out16: Optional[torch.Tensor] = None) -> None:
    """In place, or (out16: fp16 [rows][cols]) written as half precision with x left as scratch."""
    assert out16 is None or (out16.dtype == torch.float16 and out16.numel() >= rows * cols)
    _launch("softmax_rows", 0, (8.0 if out16 is None else 6.0) * rows * cols,
            lambda: _lib.check(_lib.load().sf_softmax_rows(x.data_ptr(), rows, cols,
                                                           out16.data_ptr() if out16 is not None else None,
                                                           _lib.stream()), "sf_softmax_rows"))


@on_tensor_device
def window_attn(QKV: Planes, qkv_bias: torch.Tensor, OUT: Planes, heads: int, H: int, W: int, ws: int = 7,
                cx: Optional[Ctx] = None, exact: bool = False) -> None:
    """timm LocallyGroupedAttn core on token planes (encoder; see include/streamflow_hip.h).  PRECISION_FP32: the exact
    VALU kernel; every other class: the matrix-core kernel (3 products per contraction for F16X3, 1 for the fp16 classes)."""
    PRECISION = _cx(cx).precision
    C = OUT.rows
    assert QKV.rows == 3 * C and QKV.P == H * W == OUT.P and qkv_bias.numel() == 3 * C and (not OUT.f16 or OUT.koct)
    assert not QKV.f16 or (QKV.koct and PRECISION in (PRECISION_F16X2, PRECISION_F16))
    flops, nbytes = 4.0 * QKV.n_img * H * W * ws * ws * C, (2.0 if QKV.f16 else 4.0) * QKV.n_img * 3 * C * H * W + 4.0 * QKV.n_img * C * H * W
    if PRECISION == PRECISION_FP32 or exact:
        assert not OUT.f16 and not QKV.f16
        _launch("window_attn", flops, nbytes,
                lambda: _lib.check(_lib.load().sf_window_attn(QKV.ptr, QKV.img_stride, qkv_bias.data_ptr(), OUT.ptr, OUT.img_stride,
                                                              QKV.n_img, C, heads, H, W, ws, _lib.stream()), "sf_window_attn"))
        return
    prec = PRECISION
    ko = OUT.f16                                          # fp16 k-octet planes: the hand-over to the proj GEMM
    _launch("window_attn_mfma", flops, nbytes - (2.0 * QKV.n_img * C * H * W if ko else 0.0),
            lambda: _lib.check(_lib.load().sf_window_attn_mfma(QKV.ptr, QKV.img_stride, int(QKV.f16), qkv_bias.data_ptr(),
                                                               None if ko else OUT.ptr, 0 if ko else OUT.img_stride,
                                                               OUT.ptr if ko else None, OUT.img_stride if ko else 0,
                                                               QKV.n_img, C, heads, H, W, ws, prec, _lib.stream()),
                               "sf_window_attn_mfma"),
            products=3 if prec == PRECISION_F16X3 else 1)


@on_tensor_device
def subsample_attn(Q: Planes, KV: Planes, OUT: Planes, heads: int, ws: Optional[torch.Tensor] = None,
                   cx: Optional[Ctx] = None, exact: bool = False) -> None:
    """timm GlobalSubSampleAttn core: OUT = softmax(q k^T / sqrt(32)) v, keys/values = the M sub-sampled tokens.
    PRECISION_FP32: the exact VALU kernel; every other class: the matrix-core kernel (3 products per contraction for
    F16X3, 1 for the fp16 classes) over `ws` (uint8, >= subsample_attn_ws_bytes; allocated here when not given)."""
    PRECISION = _cx(cx).precision
    C = Q.rows
    assert KV.rows == 2 * C and OUT.rows == C and Q.P == OUT.P and Q.n_img == KV.n_img == OUT.n_img and (not OUT.f16 or OUT.koct)
    nbytes = 4.0 * Q.n_img * C * (2 * Q.P + 2 * KV.P)
    if PRECISION == PRECISION_FP32 or exact:
        assert not OUT.f16
        _launch("subsample_attn", 4.0 * Q.n_img * Q.P * KV.P * C, nbytes,
                lambda: _lib.check(_lib.load().sf_subsample_attn(Q.ptr, Q.img_stride, KV.ptr, KV.img_stride, OUT.ptr, OUT.img_stride,
                                                                 Q.n_img, C, heads, Q.P, KV.P, _lib.stream()), "sf_subsample_attn"))
        return
    need = subsample_attn_ws_bytes(Q.n_img, heads, KV.P)
    if ws is None:
        ws = torch.empty(need, dtype=torch.uint8, device=Q.base.device)
    assert ws.dtype == torch.uint8 and ws.numel() >= need and ws.device == Q.base.device
    prec = PRECISION
    ko = OUT.f16
    _launch("subsample_attn_mfma", 4.0 * Q.n_img * Q.P * KV.P * C, nbytes - (2.0 * Q.n_img * C * Q.P if ko else 0.0),
            lambda: _lib.check(_lib.load().sf_subsample_attn_mfma(Q.ptr, Q.img_stride, KV.ptr, KV.img_stride,
                                                                  None if ko else OUT.ptr, 0 if ko else OUT.img_stride,
                                                                  OUT.ptr if ko else None, OUT.img_stride if ko else 0,
                                                                  Q.n_img, C, heads, Q.P, KV.P, ws.data_ptr(), ws.numel(), prec,
                                                                  _lib.stream()), "sf_subsample_attn_mfma"),
            products=3 if prec == PRECISION_F16X3 else 1)


def subsample_attn_ws_bytes(n_img: int, heads: int, M: int) -> int:
    return int(_lib.load().sf_subsample_attn_ws_bytes(n_img, heads, M))


@on_tensor_device
def dwconv3x3_res(X: Planes, wgt: torch.Tensor, bias: torch.Tensor, Y: Planes, H: int, W: int) -> None:
    """timm PosConv: Y = X + depthwise3x3(X) + b on [n_img][C][H][W]."""
    assert X.rows == Y.rows == wgt.shape[0] and X.P == H * W
    _launch("dwconv3", 18.0 * X.n_img * X.rows * H * W, 8.0 * X.n_img * X.rows * H * W,
            lambda: _lib.check(_lib.load().sf_dwconv3x3_res(X.ptr, X.img_stride, wgt.data_ptr(), bias.data_ptr(), Y.ptr,
                                                            Y.img_stride, X.n_img, X.rows, H, W, _lib.stream()), "sf_dwconv3x3_res"))


def gma_flash_ws_bytes(n_img: int, P: int) -> int:
    return int(_lib.load().sf_gma_flash_ws_bytes(n_img, P))


def gma_flash_img_bytes(P: int) -> int:
    """Bytes of ONE image's section of the fused GMA workspace (the images lie back to back; the key-split partial buffers of a
    small launch follow the last image and are not part of it)."""
    big = 4096                                                        # (enough query tiles that no key split is planned)
    return (gma_flash_ws_bytes(big, P) - gma_flash_ws_bytes(big - 1, P))


def gma_no_key_split(n_img: int, P: int) -> bool:
    """The fused / stored GMA launch over n_img images runs WITHOUT its key-split form (whose partial buffers live behind the last
    image of the workspace): the condition for launching a sub-range of images out of a larger workspace."""
    return gma_flash_ws_bytes(n_img, P) == n_img * gma_flash_img_bytes(P)


@on_tensor_device
def gma_flash_pack_qk(QK: Planes, ws: torch.Tensor, scale: float, stats_qk_products: int = 0, cx: Optional[Ctx] = None) -> None:
    """QK [n_img][256][P] (to_qk output) -> packed fp16 operand images in ws (once per clip); stats_qk_products = 1 / 2 / 3
    also stores every query's softmax statistics for gma_flash_aggregate(..., use_stats=True) with the same product count."""
    assert QK.rows == 256 and ws.dtype == torch.uint8
    sq = int(stats_qk_products) if _cx(cx).flash_stats else 0
    _launch("flash_pack_qk", 2.0 * QK.n_img * QK.P * QK.P * 128 * (1 if sq else 0), 4.0 * QK.n_img * 256 * QK.P * 2,
            lambda: _lib.check(_lib.load().sf_gma_flash_pack_qk(QK.ptr, QK.img_stride, ws.data_ptr(), ws.numel(), QK.n_img,
                                                                QK.P, float(scale), sq, _lib.stream()), "sf_gma_flash_pack_qk"),
            products=(sq + 1) / 2.0 if sq else 1.0)


def gma_flash_project_ok(A: "PackedLinear", X: Planes, cx: Optional[Ctx] = None) -> bool:
    """Can to_v + the v pack run as ONE launch (sf_gma_flash_project_v)?  fp16-activation arithmetic, a 128 x 128 weight, and the
    k-octet copy of the motion features at hand."""
    cx = _cx(cx)
    return (cx.precision in (PRECISION_F16X2, PRECISION_F16) and cx.shadows and A.M == 128 and A.K == 128 and A.bias is None and
            X.shadow is not None and X.rows == 128)


@on_tensor_device
def gma_flash_project_v(ws: torch.Tensor, A: "PackedLinear", X: Planes, cx: Optional[Ctx] = None) -> None:
    """v = to_v(X) (gma.py:93) written straight into the packed v planes of the fused kernel's workspace; X.shadow (the k-octet
    fp16 copy) is the operand.  The following gma_flash_aggregate call passes V=None."""
    cx = _cx(cx)
    sh = X.shadow
    assert gma_flash_project_ok(A, X, cx) and sh.f16 and sh.koct
    products = 1 if (cx.precision == PRECISION_F16 or A.single) else 2
    n, P = X.n_img, X.P
    _launch("gma_project_v", 2.0 * 128 * 128 * n * P, 4.0 * n * 128 * P,
            lambda: _lib.check(_lib.load().sf_gma_flash_project_v(
                ws.data_ptr(), ws.numel(), sh.ptr, sh.img_stride, sh.P, A.hi.data_ptr(), A.lo.data_ptr(), A.lda_h,
                1.0 / A.split_scale, products, n, P, _lib.stream()), "sf_gma_flash_project_v"), products=float(products))


@on_tensor_device
def gma_flash_aggregate(ws: torch.Tensor, V: Optional[Planes], MF: Planes, gamma: torch.Tensor, OUT: Planes, qk_products: int = 3,
                        use_stats: bool = False, cx: Optional[Ctx] = None) -> None:
    """OUT = MF + gamma * softmax(scale q k^T) V, fused (no N x N tensor); q, k come packed in ws.  use_stats: the softmax
    statistics stored by gma_flash_pack_qk(..., stats_qk_products=qk_products) are used instead of an online softmax.
    V=None: the v planes of ws were written by gma_flash_project_v."""
    cx = _cx(cx)
    use_stats = bool(use_stats) and cx.flash_stats
    if V is None:
        V = Planes(MF.base, 0, 0, MF.n_img, 128, MF.P, f16=True)             # placeholder: pointer 0 is passed below
    assert V.rows == MF.rows == OUT.rows == 128 and V.n_img == MF.n_img == OUT.n_img and not V.koct
    n, P = V.n_img, V.P
    fn = _lib.load().sf_gma_flash_aggregate_f16v if V.f16 else _lib.load().sf_gma_flash_aggregate     # (v as fp16 rows)
    # algorithmic: the two contractions; bytes: v, mf in, out (q/k/v tiles are re-read from L2 by every query tile)
    sh = OUT.shadow if (OUT.shadow is not None and cx.shadows and cx.shadow_fused) else None
    _launch("gma_flash", 4.0 * n * P * P * 128,
            (2.0 if V.f16 else 4.0) * n * 128 * P + 4.0 * n * 128 * P * 2 + 2.0 * n * 128 * P * (2 if sh is None else 3),
            lambda: _lib.check(fn(
                ws.data_ptr(), ws.numel(), (V.ptr if V.img_stride else None), V.img_stride, MF.ptr, MF.img_stride, gamma.data_ptr(), OUT.ptr,
                OUT.img_stride, None if sh is None else sh.ptr, 0 if sh is None else sh.img_stride, n, P,
                int(qk_products), int(use_stats), _lib.stream()), "sf_gma_flash_aggregate"), products=(qk_products + 1) / 2.0)
    if sh is None:
        refresh_shadow(OUT, cx)


def gma_stored_p_bytes(n_img: int, P: int) -> int:
    return int(_lib.load().sf_gma_stored_p_bytes(n_img, P))


@on_tensor_device
def gma_flash_store_p(ws: torch.Tensor, pbuf: torch.Tensor, n_img: int, P: int, qk_products: int, cx: Optional[Ctx] = None) -> None:
    """Once per clip, after gma_flash_pack_qk(stats_qk_products=qk_products): the softmax weights of every (query, key) as fp16 in
    the fragment order of the second contraction -> pbuf (gma.py:53-65: `attn`, kept for the refinement loop)."""
    assert ws.dtype == torch.uint8 and pbuf.dtype == torch.uint8
    _launch("gma_store_p", 2.0 * n_img * P * P * 128, 2.0 * n_img * P * P,
            lambda: _lib.check(_lib.load().sf_gma_flash_store_p(ws.data_ptr(), ws.numel(), pbuf.data_ptr(), pbuf.numel(), n_img, P,
                                                                int(qk_products), _lib.stream()), "sf_gma_flash_store_p"),
            products=(qk_products + 1) / 2.0)


@on_tensor_device
def gma_stored_aggregate(ws: torch.Tensor, pbuf: torch.Tensor, V: Optional[Planes], MF: Planes, gamma: torch.Tensor, OUT: Planes,
                         cx: Optional[Ctx] = None) -> None:
    """OUT = MF + gamma * attn V with the stored weights (gma.py:99-102), bit-identical to gma_flash_aggregate(use_stats=True) on the
    same workspace.  V=None: the v planes of ws were written by gma_flash_project_v."""
    cx = _cx(cx)
    if V is None:
        V = Planes(MF.base, 0, 0, MF.n_img, 128, MF.P, f16=True)             # placeholder: pointer 0 is passed below
    assert V.rows == MF.rows == OUT.rows == 128 and V.n_img == MF.n_img == OUT.n_img and not V.koct
    n, P = V.n_img, V.P
    sh = OUT.shadow if (OUT.shadow is not None and cx.shadows and cx.shadow_fused) else None
    # algorithmic: ONE contraction; bytes: the stored weights (once), v, mf in, out
    _launch("gma_stored", 2.0 * n * P * P * 128,
            2.0 * n * P * P + (2.0 if V.f16 else 4.0) * n * 128 * P + 4.0 * n * 128 * P * 2 + (2.0 * n * 128 * P if sh is not None else 0.0),
            lambda: _lib.check(_lib.load().sf_gma_stored_aggregate(
                ws.data_ptr(), ws.numel(), pbuf.data_ptr(), pbuf.numel(), (V.ptr if V.img_stride else None), int(V.f16), V.img_stride,
                MF.ptr, MF.img_stride, gamma.data_ptr(), OUT.ptr, OUT.img_stride, None if sh is None else sh.ptr,
                0 if sh is None else sh.img_stride, n, P, _lib.stream()), "sf_gma_stored_aggregate"), products=1.0)
    if sh is None:
        refresh_shadow(OUT, cx)


@on_tensor_device
def coords_grid(batch: int, ht: int, wd: int, device) -> torch.Tensor:
    out = torch.empty(batch, 2, ht, wd, dtype=torch.float32, device=device)
    _lib.ptr(out)
    _lib.check(_lib.load().sf_coords_grid(out.data_ptr(), batch, ht, wd, _lib.stream()), "sf_coords_grid")
    return out


@on_tensor_device
def context_split(cnets: torch.Tensor, nets: Planes, inps: Planes, hdim: int) -> None:
    """cnets [n_img, 2*hdim, P]-like contiguous tensor."""
    _dev_check(cnets)
    _lib.check(_lib.load().sf_context_split(cnets.data_ptr(), nets.ptr, nets.img_stride, inps.ptr, inps.img_stride,
                                            nets.n_img, hdim, nets.P, _lib.stream()), "sf_context_split")


@on_tensor_device
def flow_update(coords1: Planes, delta: Optional[Planes], flow_a: Optional[Planes], flow_b: Optional[Planes],
                n_img: int, h: int, w: int, koct: Optional[Planes] = None, koct_row: int = 0, cx: Optional[Ctx] = None) -> None:
    """koct / koct_row: k-octet planes (Planes.shadow of the tensor flow_b is a slice of) and the row the x component
    goes to: keeps that copy's flow rows current."""
    assert coords1.img_stride == 2 * h * w and (delta is None or delta.img_stride == 2 * h * w)
    if not _cx(cx).shadows:
        koct = None
    assert koct is None or (koct.f16 and koct.koct and koct_row + 2 <= (koct.rows + 7) // 8 * 8)
    _launch("flow_update", 0, 0, lambda: _lib.check(_lib.load().sf_flow_update(
        coords1.ptr, None if delta is None else delta.ptr,
        None if flow_a is None else flow_a.ptr, 0 if flow_a is None else flow_a.img_stride,
        None if flow_b is None else flow_b.ptr, 0 if flow_b is None else flow_b.img_stride,
        None if koct is None else koct.ptr, 0 if koct is None else koct.img_stride, int(koct_row),
        n_img, h, w, _lib.stream()), "sf_flow_update"))


@on_tensor_device
def upsample_flow(flow: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """flow [n,2,h,w], mask [n,576,h,w] -> [n,2,8h,8w] (reference streamflow.py:82-93)."""
    _dev_check(flow)
    _dev_check(mask)
    n, _, h, w = flow.shape
    out = torch.empty(n, 2, 8 * h, 8 * w, dtype=torch.float32, device=flow.device)
    _lib.check(_lib.load().sf_upsample_flow(flow.data_ptr(), mask.data_ptr(), out.data_ptr(), n, h, w, _lib.stream()),
               "sf_upsample_flow")
    return out


@on_tensor_device
def forward_interpolate(flow: torch.Tensor) -> torch.Tensor:
    """flow [n,2,h,w] -> [n,2,h,w] (reference utils.py:34-62, on the device)."""
    _dev_check(flow)
    n, _, h, w = flow.shape
    out = torch.empty_like(flow)
    _lib.check(_lib.load().sf_forward_interpolate(flow.data_ptr(), out.data_ptr(), n, h, w, _lib.stream()),
               "sf_forward_interpolate")
    return out


def pair_strides(arr: Optional[Sequence[int]]):
    if arr is None:
        return None
    return (C.c_int64 * 4)(*[int(a) for a in arr])


def corr_build_ws_bytes(B: int, pairs: int, D: int, h: int, w: int) -> int:
    return int(_lib.load().sf_corr_build_ws_bytes(B, pairs, D, h, w))


def corr_pitch(h: int, w: int) -> Optional[tuple]:
    """Row pitch (cells) of the fp32 volume maps of an h x w feature grid, or None where the reference's dense [N, h_l, w_l] layout
    already starts every level-0 row on a cache line (w % 32 == 0: Sintel 128, Spring 240 is NOT): every level's rows are rounded
    up to 32 cells = 128 bytes (sf_corr_build_pyramid_pitched; DESIGN.md section 12.7: KITTI's 156-cell rows)."""
    if w % 32 == 0:
        return None
    return tuple(((w >> l) + 31) // 32 * 32 for l in range(4))


def _pitch_arg(pitch):
    return None if pitch is None else (C.c_int32 * 4)(*[int(p) for p in pitch])


@on_tensor_device
def corr_build(f1_ptr: int, f2_ptr: int, clip_stride: int, pair_stride: int, lvls: Sequence[torch.Tensor],
               lvl_pair_stride: Optional[Sequence[int]], B: int, pairs: int, D: int, h: int, w: int,
               ws: Optional[torch.Tensor] = None, cx: Optional[Ctx] = None, pitch: Optional[Sequence[int]] = None) -> None:
    """ws: scratch of corr_build_ws_bytes() bytes for the split-precision / fp16 builds (allocated here when omitted).
    The dtype of `lvls` selects the volume format: float32 (arithmetic = the package precision, fp32 or f16x3) or
    float16 (SF_PRECISION_F16: single f16 products, fp16 cells).  pitch: row pitch of the maps in cells per level (corr_pitch),
    None = dense."""
    N = h * w
    vol16 = lvls[0].dtype == torch.float16
    assert all(t.dtype == lvls[0].dtype for t in lvls) and lvls[0].dtype in (torch.float16, torch.float32)
    prec = PRECISION_F16 if vol16 else min(_cx(cx).precision, PRECISION_F16X3)
    need = corr_build_ws_bytes(B, pairs, D, h, w) if prec != PRECISION_FP32 else 0
    if need and (ws is None or ws.numel() * ws.element_size() < need):
        ws = torch.empty(need, dtype=torch.uint8, device=lvls[0].device)
    cells = sum((h >> l) * (w >> l) for l in range(4))
    # algorithmic bytes per (clip, pair): both feature maps read once + every pyramid cell written once
    nbytes = B * pairs * (2.0 * N * D * 4 + (2.0 if vol16 else 4.0) * N * cells)
    _launch("corr_build", 2.0 * N * N * D * B * pairs, nbytes, lambda: _lib.check(
        _lib.load().sf_corr_build_pyramid_pitched(
            f1_ptr, f2_ptr, clip_stride, pair_stride, lvls[0].data_ptr(), lvls[1].data_ptr(), lvls[2].data_ptr(),
            lvls[3].data_ptr(), pair_strides(lvl_pair_stride), _pitch_arg(pitch), B, pairs, D, h, w, 4, prec,
            ws.data_ptr() if need else None, need, _lib.stream()),
        "sf_corr_build_pyramid"), products=_products(prec))


@on_tensor_device
def corr_lookup(lvls: Sequence[torch.Tensor], lvl_pair_stride: Optional[Sequence[int]], coords: Planes,
                out: Planes, B: int, pairs: int, h: int, w: int, cx: Optional[Ctx] = None,
                pitch: Optional[Sequence[int]] = None) -> None:
    assert out.rows == 324 and out.n_img == B * pairs and coords.img_stride == 2 * h * w
    N = h * w
    vol16 = lvls[0].dtype == torch.float16
    # algorithmic bytes per image: 10x10 footprint x 4 levels read + coords + 324 output channels
    # the k-octet copy of the output (Planes.shadow) comes out of the same kernel when the volumes are fp16
    cx = _cx(cx)
    sh = out.shadow if (out.shadow is not None and cx.shadows and vol16 and cx.shadow_fused) else None
    # (footprints + coords + the 324 fp32 output channels; the k-octet copy is extra traffic)
    nbytes = B * pairs * (N * 4 * 100 * (2.0 if vol16 else 4.0) + N * 2 * 4.0 + N * 324 * 4.0)
    _launch("corr_lookup", 0, nbytes, lambda: _lib.check(_lib.load().sf_corr_lookup_pitched(
        lvls[0].data_ptr(), lvls[1].data_ptr(), lvls[2].data_ptr(), lvls[3].data_ptr(),
        pair_strides(lvl_pair_stride), _pitch_arg(pitch), coords.ptr, out.ptr, out.img_stride,
        None if sh is None else sh.ptr, 0 if sh is None else sh.img_stride, B, pairs, h, w, 4, 4,
        PRECISION_F16 if vol16 else PRECISION_FP32, _lib.stream()), "sf_corr_lookup"))
    if sh is None:
        refresh_shadow(out, cx)


@dataclass(frozen=True)
class BlockedVolume:
    """Correlation pyramids of `n_img` images in the blocked layout (one buffer): fp16 cells in 8 x 8-cell blocks
    (csrc/corr_blocked.hip) or, `f32`, fp32 cells in 4-row x 8-column blocks (csrc/corr_blocked32.hip)."""
    buf: torch.Tensor          # uint8
    img_stride: int            # bytes
    n_img: int
    h: int
    w: int
    rec: int                   # bytes per source pixel
    off: tuple                 # byte offset of each level inside a record
    nby: tuple
    nbx: tuple
    src_rows: int              # records per image (h*w rounded up to 128)
    f32: bool = False          # fp32 cells, blocks of 4 rows x 8 columns

    def levels(self):
        """The four levels as [n_img, h*w, hl, wl] tensors of the cell type (copies; tests and API parity only)."""
        N = self.h * self.w
        bh = 4 if self.f32 else 8                                         # block rows (8 columns either way)
        out = []
        for l in range(4):
            hl, wl, nby, nbx = self.h >> l, self.w >> l, self.nby[l], self.nbx[l]
            blk = torch.as_strided(self.buf[self.off[l]:], (self.n_img, N, nby * nbx * 128), (self.img_stride, self.rec, 1))
            blk = blk.contiguous().view(torch.float32 if self.f32 else torch.float16)
            blk = blk.view(self.n_img, N, nby, nbx, 8, bh)                # [by][bx][tx % 8][ty % bh]
            out.append(blk.permute(0, 1, 2, 5, 3, 4).reshape(self.n_img, N, nby * bh, nbx * 8)[:, :, :hl, :wl].contiguous())
        return out


def blocked_geometry(h: int, w: int, f32: bool = False):
    rec, src = C.c_int64(), C.c_int64()
    off, nby, nbx = (C.c_int64 * 4)(), (C.c_int32 * 4)(), (C.c_int32 * 4)()
    name = "sf_corr_blocked32_geometry" if f32 else "sf_corr_blocked_geometry"
    _lib.check(getattr(_lib.load(), name)(h, w, C.byref(rec), off, nby, nbx, C.byref(src)), name)
    return int(rec.value), tuple(int(v) for v in off), tuple(int(v) for v in nby), tuple(int(v) for v in nbx), int(src.value)


def new_blocked_volume(n_img: int, h: int, w: int, device, f32: bool = False) -> BlockedVolume:
    rec, off, nby, nbx, src = blocked_geometry(h, w, f32)
    stride = src * rec                                      # a multiple of 128 (rec is)
    buf = torch.empty(n_img * stride + 128, dtype=torch.uint8, device=device)
    pad = (-buf.data_ptr()) % 128
    return BlockedVolume(buf[pad: pad + n_img * stride], stride, n_img, h, w, rec, off, nby, nbx, src, f32)


def corr_build_blocked_ws_bytes(n_img: int, D: int, h: int, w: int, f32: bool = False) -> int:
    if f32:
        return int(_lib.load().sf_corr_build_blocked32_ws_bytes(n_img, D, h, w))
    return int(_lib.load().sf_corr_build_blocked_ws_bytes(n_img, D, h, w))


@on_tensor_device
def corr_build_blocked(f1_ptr: int, f2_ptr: int, clip_stride: int, pair_stride: int, vol: BlockedVolume, B: int,
                       pairs: int, D: int, ws: Optional[torch.Tensor] = None) -> None:
    """a1 + a2 into a BlockedVolume: fp16 cells (single f16 MFMA products, fp32 accumulation) or, for an f32 volume, fp32
    cells (split fp16 operands, three products, fp32 accumulation)."""
    h, w = vol.h, vol.w
    N = h * w
    assert vol.n_img == B * pairs
    need = corr_build_blocked_ws_bytes(B * pairs, D, h, w, vol.f32)
    if ws is None or ws.numel() * ws.element_size() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=vol.buf.device)
    cells = sum((h >> l) * (w >> l) for l in range(4))
    e = 4.0 if vol.f32 else 2.0
    nbytes = B * pairs * (2.0 * N * D * 4 + e * N * cells)
    name = "sf_corr_build_blocked32" if vol.f32 else "sf_corr_build_blocked"
    _launch("corr_build", 2.0 * N * N * D * B * pairs, nbytes, lambda: _lib.check(
        getattr(_lib.load(), name)(f1_ptr, f2_ptr, clip_stride, pair_stride, vol.buf.data_ptr(), vol.img_stride,
                                   B, pairs, D, h, w, ws.data_ptr(), need, _lib.stream()), name))


@on_tensor_device
def corr_lookup_blocked(vol: BlockedVolume, coords: Planes, out: Optional[Planes], out_koct: Optional[Planes], B: int,
                        pairs: int) -> None:
    """a3 from a BlockedVolume: `out_koct` (fp16 k-octet planes of the 324 features) and / or `out` (fp32 planes)."""
    h, w = vol.h, vol.w
    N = h * w
    assert coords.img_stride == 2 * N and vol.n_img == B * pairs
    assert out is None or (out.rows == 324 and out.n_img == B * pairs and not out.f16)
    assert out_koct is None or (out_koct.rows == 324 and out_koct.n_img == B * pairs and out_koct.koct)
    if vol.f32:                                           # fp32 cells: fp32 planes out (the fp32-class operand format)
        assert out is not None and out_koct is None
        nbytes = B * pairs * (N * 4 * 100 * 4.0 + N * 2 * 4.0 + N * 324 * 4.0)  # SURVEY.md section 8d, e = 4
        _launch("corr_lookup", 0, nbytes, lambda: _lib.check(_lib.load().sf_corr_lookup_blocked32(
            vol.buf.data_ptr(), vol.img_stride, coords.ptr, out.ptr, out.img_stride, B, pairs, h, w, _lib.stream()),
            "sf_corr_lookup_blocked32"))
        return
    nbytes = B * pairs * (N * 4 * 100 * 2.0 + N * 2 * 4.0 + N * 324 * 4.0)      # SURVEY.md section 8d, e = 2
    _launch("corr_lookup", 0, nbytes, lambda: _lib.check(_lib.load().sf_corr_lookup_blocked(
        vol.buf.data_ptr(), vol.img_stride, coords.ptr, None if out is None else out.ptr,
        0 if out is None else out.img_stride, None if out_koct is None else out_koct.ptr,
        0 if out_koct is None else out_koct.img_stride, B, pairs, h, w, _lib.stream()), "sf_corr_lookup_blocked"))


@on_tensor_device
def bilinear_sampler(img: torch.Tensor, coords: torch.Tensor, want_mask: bool = False):
    _dev_check(img)
    _dev_check(coords)
    M, Cc, Hi, Wi = img.shape
    _, Ho, Wo, _ = coords.shape
    out = torch.empty(M, Cc, Ho, Wo, dtype=torch.float32, device=img.device)
    mask = torch.empty(M, Ho, Wo, 1, dtype=torch.float32, device=img.device) if want_mask else None
    _lib.check(_lib.load().sf_bilinear_sampler(img.data_ptr(), coords.data_ptr(), out.data_ptr(),
                                               None if mask is None else mask.data_ptr(), M, Cc, Hi, Wi, Ho, Wo,
                                               _lib.stream()), "sf_bilinear_sampler")
    return (out, mask) if want_mask else out
