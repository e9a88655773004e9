"""CorrBlock with the reference's interface (core/corr.py:6-54), backed by the HIP kernels."""
from __future__ import annotations

import math

import torch

from . import ops
from .ops import Planes


class CorrBlock:
    """All-pairs correlation pyramid + windowed lookup for ONE frame pair.

    ``CorrBlock(fmap1, fmap2, num_levels=4, radius=4)`` builds ``corr_pyramid`` (list of
    ``[B*h*w, 1, h>>l, w>>l]`` tensors, reference corr.py:13-21) in a single fused kernel;
    ``blk(coords)`` returns ``[B, 4*81, h, w]`` float32 contiguous (corr.py:23-44).
    ``dtype=torch.float16`` (an extension; the reference always keeps fp32 volumes) stores the pyramid as fp16
    cells built with single f16 MFMA products -- the bf16/fp16 volume configurations of BASELINE.json.
    ``layout="blocked"`` keeps them in cache-line blocks -- 8 x 8 fp16 cells (csrc/corr_blocked.hip) or 4 rows x 8 columns of
    fp32 cells (csrc/corr_blocked32.hip) -- what the fused engine uses; ``corr_pyramid`` is then a row-major COPY made on first access.
    """

    @ops.on_tensor_device
    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4,
                 dtype: torch.dtype = torch.float32, layout: str = "rows"):
        if num_levels != 4 or radius != 4:
            raise RuntimeError("CorrBlock: the HIP path is built for num_levels=4, radius=4 "
                               "(the only values the StreamFlow model uses, streamflow.py:38-39)")
        f1 = fmap1.contiguous().float()
        f2 = fmap2.contiguous().float()
        ops._dev_check(f1)
        ops._dev_check(f2)
        B, D, h, w = f1.shape
        self.num_levels, self.radius = num_levels, radius
        self.shape = (B, h, w)
        N = h * w
        if dtype not in (torch.float32, torch.float16):
            raise RuntimeError(f"CorrBlock: volume dtype must be float32 or float16, got {dtype}")
        if layout not in ("rows", "blocked"):
            raise RuntimeError("CorrBlock: layout must be 'rows' or 'blocked'")
        self._keep = (f1, f2)
        self.vol = None
        if layout == "blocked":
            self.vol = ops.new_blocked_volume(B, h, w, f1.device, f32=dtype == torch.float32)
            ops.corr_build_blocked(f1.data_ptr(), f2.data_ptr(), D * N, 0, self.vol, B, 1, D)
            self._pyr = self._store = self.pitch = None
            return
        # fp32 maps whose rows would straddle cache lines are kept with a row pitch of a multiple of 32 cells (ops.corr_pitch:
        # KITTI's 156-cell rows); `corr_pyramid` is then the strided [..., :w_l] view of the pitched tensors: the reference's
        # shape and values (corr.py:13-21), no copy
        self.pitch = ops.corr_pitch(h, w) if dtype == torch.float32 else None
        pw = self.pitch or tuple(w >> l for l in range(4))
        self._store = [torch.empty(B * N, 1, h >> l, pw[l], dtype=dtype, device=f1.device) for l in range(4)]
        ops.corr_build(f1.data_ptr(), f2.data_ptr(), D * N, 0, self._store, None, B, 1, D, h, w, pitch=self.pitch)
        self._pyr = [t[..., : w >> l] for l, t in enumerate(self._store)]

    @property
    def corr_pyramid(self):
        if self._pyr is None:                                  # blocked volumes: row-major copy, reference shapes
            B, h, w = self.shape
            self._pyr = [t.reshape(B * h * w, 1, h >> l, w >> l) for l, t in enumerate(self.vol.levels())]
        return self._pyr

    @ops.on_tensor_device
    def __call__(self, coords: torch.Tensor) -> torch.Tensor:
        B, h, w = self.shape
        c = coords.contiguous().float()
        ops._dev_check(c)
        assert tuple(c.shape) == (B, 2, h, w), (c.shape, self.shape)
        out = torch.empty(B, 4 * 81, h, w, dtype=torch.float32, device=c.device)
        if self.vol is not None:
            ops.corr_lookup_blocked(self.vol, Planes.of(c), Planes.of(out), None, B, 1)
        else:
            ops.corr_lookup(self._store, None, Planes.of(c), Planes.of(out), B, 1, h, w, pitch=self.pitch)
        return out

    @staticmethod
    @ops.on_tensor_device
    def corr(fmap1: torch.Tensor, fmap2: torch.Tensor) -> torch.Tensor:
        """[B,h,w,1,h,w] = f1^T f2 / sqrt(D) (corr.py:46-54)."""
        f1 = fmap1.contiguous().float()
        f2 = fmap2.contiguous().float()
        ops._dev_check(f1)
        ops._dev_check(f2)
        B, D, h, w = f1.shape
        N = h * w
        out = torch.empty(B, N, N, dtype=torch.float32, device=f1.device)
        ops.gemm_raw(A=f1.data_ptr(), B=f2.data_ptr(), C=out.data_ptr(), M=N, N=N, K=D, batch=B, lda=N, ldb=N, ldc=N,
                     strideA=D * N, strideB=D * N, strideC=N * N, a_layout=ops.LAYOUT_K_MAJOR,
                     b_layout=ops.LAYOUT_K_MAJOR, alpha=1.0 / math.sqrt(D), epilogue=ops.EPI_NONE)
        return out.view(B, h, w, 1, h, w)
