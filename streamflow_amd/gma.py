"""GMA Attention / Aggregate with the reference's interface (core/gma.py:34-104)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .ops import PackedLinear, Planes


class _Packed:
    """Caches kernel-layout copies of the module's parameters; rebuilt when a parameter changes."""

    def _packed(self, build):
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in self.parameters())
        if getattr(self, "_pack_key", None) != key:
            self._pack = build()
            self._pack_key = key
        return self._pack


class Attention(nn.Module, _Packed):
    """attn = softmax(scale * q k^T) over all pixel pairs, [b, heads, N, N] (gma.py:53-65)."""

    def __init__(self, *, args=None, dim, max_pos_size=100, heads=4, dim_head=128):
        super().__init__()
        self.args = args
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.to_qk = nn.Conv2d(dim, heads * dim_head * 2, 1, bias=False)

    @ops.on_tensor_device
    def forward(self, fmap: torch.Tensor) -> torch.Tensor:
        x = fmap.contiguous().float()
        ops._dev_check(x)
        b, c, h, w = x.shape
        P, H, dh = h * w, self.heads, self.dim_head
        W = self._packed(lambda: PackedLinear(self.to_qk.weight, None, x.device))
        qk = torch.empty(b, 2 * H * dh, P, dtype=torch.float32, device=x.device)
        ops.gemm(W, Planes.of(x), Planes.of(qk))
        attn = torch.empty(b, H, P, P, dtype=torch.float32, device=x.device)
        for hd in range(H):          # q rows [hd*dh, (hd+1)*dh), k rows H*dh + the same
            ops.gemm_raw(A=qk.data_ptr() + 4 * hd * dh * P, B=qk.data_ptr() + 4 * (H + hd) * dh * P,
                         C=attn.data_ptr() + 4 * hd * P * P, M=P, N=P, K=dh, batch=b, lda=P, ldb=P, ldc=P,
                         strideA=2 * H * dh * P, strideB=2 * H * dh * P, strideC=H * P * P,
                         a_layout=ops.LAYOUT_K_MAJOR, b_layout=ops.LAYOUT_K_MAJOR, alpha=float(self.scale),
                         epilogue=ops.EPI_NONE)
        ops.softmax_rows(attn, b * H * P, P)
        return attn


class Aggregate(nn.Module, _Packed):
    """out = fmap + gamma * (attn @ to_v(fmap)) (gma.py:91-104).  The reference adds a `project` conv only
    when heads*dim_head != dim; the StreamFlow configuration never does (dim == inner dim)."""

    def __init__(self, args=None, dim=128, heads=4, dim_head=128):
        super().__init__()
        self.args = args
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        inner = heads * dim_head
        self.to_v = nn.Conv2d(dim, inner, 1, bias=False)
        self.gamma = nn.Parameter(torch.zeros(1))
        if dim != inner:
            raise RuntimeError("Aggregate: heads*dim_head must equal dim (no `project` conv on the HIP path)")
        self.project = None

    @ops.on_tensor_device
    def forward(self, attn: torch.Tensor, fmap: torch.Tensor) -> torch.Tensor:
        x = fmap.contiguous().float()
        a = attn.contiguous().float()
        ops._dev_check(x)
        ops._dev_check(a)
        b, c, h, w = x.shape
        P, H, dh = h * w, self.heads, self.dim_head
        W = self._packed(lambda: PackedLinear(self.to_v.weight, None, x.device))
        v = torch.empty(b, H * dh, P, dtype=torch.float32, device=x.device)
        ops.gemm(W, Planes.of(x), Planes.of(v))
        out = torch.empty_like(x)
        g = self.gamma.detach().float().contiguous()
        for hd in range(H):
            off = 4 * hd * dh * P
            ops.gemm_raw(A=v.data_ptr() + off, B=a.data_ptr() + 4 * hd * P * P, C=out.data_ptr() + off,
                         R=x.data_ptr() + off, gamma=g.data_ptr(), M=dh, N=P, K=P, batch=b, lda=P, ldb=P, ldc=P,
                         ldr=P, strideA=H * dh * P, strideB=H * P * P, strideC=c * P, strideR=c * P,
                         a_layout=ops.LAYOUT_K_MINOR, b_layout=ops.LAYOUT_K_MINOR, alpha=1.0, epilogue=ops.EPI_AXPY)
        return out
