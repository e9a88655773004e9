"""Update-block modules with the reference's class names, constructor signatures, forward signatures
and state-dict keys (core/update.py:12-36, 313-339, 453-513, 739-782), executing on the HIP kernels.

The nn.Conv2d / nn.Linear / nn.LayerNorm children exist only to hold parameters under the
reference's key names (so published checkpoints load with strict=True); their own forward methods
are never used.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .engine import SKBlockWeights, run_skblock, _scratch
from .gma import Aggregate, _Packed
from .ops import (EPI_GELU, EPI_NONE, EPI_RELU, EPI_RES, PackedLinear, Planes)


def _planes_like(n: int, rows: int, P: int, device) -> Planes:
    return Planes.of(torch.empty(n, rows, P, dtype=torch.float32, device=device))


class PCBlock4_Deep_nopool_res(nn.Module, _Packed):
    """SKBlock: x=gelu(x+ffn1(x)); x=gelu(x+dw_k(x)) for k in k_conv; x=gelu(x+pw(x)); ffn2(x)
    (reference update.py:12-36; `SKBlock` in demo.py:163-187).  k_conv must be [1, K], K in {7, 15}."""

    def __init__(self, C_in: int, C_out: int, k_conv):
        super().__init__()
        self.conv_list = nn.ModuleList(
            [nn.Conv2d(C_in, C_in, k, stride=1, padding=k // 2, groups=C_in) for k in k_conv])
        mid = int(1.5 * C_in)
        self.ffn1 = nn.Sequential(nn.Conv2d(C_in, mid, 1), nn.GELU(), nn.Conv2d(mid, C_in, 1))
        self.pw = nn.Conv2d(C_in, C_in, 1)
        self.ffn2 = nn.Sequential(nn.Conv2d(C_in, mid, 1), nn.GELU(), nn.Conv2d(mid, C_out, 1))
        self.C_in, self.C_out, self.k_conv = C_in, C_out, list(k_conv)
        if len(self.k_conv) != 2 or self.k_conv[0] != 1 or self.k_conv[1] not in (7, 15):
            raise RuntimeError(f"k_conv={k_conv}: the HIP path is built for [1,15] and [1,7] "
                               "(the values every StreamFlow script uses)")

    def weights(self, device) -> SKBlockWeights:
        return self._packed(lambda: SKBlockWeights({"." + k: v for k, v in self.state_dict().items()}, "", device))

    def run(self, X: Planes, Y: Planes, h: int, w: int, final_gelu: bool = False) -> None:
        W = self.weights(X.base.device)
        dev = X.base.device
        r8 = lambda r: (r + 7) // 8 * 8               # (whole k-octets: the split hand-over of the f16x3 mode writes them)
        hid = _planes_like(X.n_img, r8(W.c_mid), X.P, dev)
        xa = _planes_like(X.n_img, r8(W.c_in), X.P, dev)
        xb = _planes_like(X.n_img, r8(W.c_in), X.P, dev)
        run_skblock(W, X, Y, hid, xa, xb, h, w, final_gelu)

    @ops.on_tensor_device
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.contiguous().float()
        ops._dev_check(x)
        n, c, h, w = x.shape
        y = torch.empty(n, self.C_out, h, w, dtype=torch.float32, device=x.device)
        self.run(Planes.of(x), Planes.of(y), h, w)
        return y


SKBlock = PCBlock4_Deep_nopool_res


class SKMotionEncoder6_Deep_nopool_res(nn.Module, _Packed):
    """reference update.py:313-339."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        out_dim = args.decoder_dim // 2
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.convc1 = PCBlock4_Deep_nopool_res(cor_planes, 256, args.k_conv)
        self.convc2 = PCBlock4_Deep_nopool_res(256, 192, args.k_conv)
        self.convf1 = nn.Conv2d(2, 128, 1, 1, 0)
        self.convf2 = PCBlock4_Deep_nopool_res(128, 64, args.k_conv)
        self.conv = PCBlock4_Deep_nopool_res(64 + 192, out_dim - 2, args.k_conv)
        self.out_dim = out_dim

    def run(self, flow: Planes, corr: Planes, out: Planes, h: int, w: int) -> None:
        """out (out_dim rows) = cat(conv(cat(convc2(gelu(convc1(corr))), convf2(convf1(flow)))), flow)."""
        dev = flow.base.device
        n, P = flow.n_img, flow.P
        cor256 = _planes_like(n, 256, P, dev)
        cat256 = _planes_like(n, 256, P, dev)
        f128 = _planes_like(n, 128, P, dev)
        self.convc1.run(corr, cor256, h, w, final_gelu=True)
        self.convc2.run(cor256, cat256.slice(0, 192), h, w)
        Wf = self._packed(lambda: PackedLinear(self.convf1.weight, self.convf1.bias, dev))
        ops.gemm(Wf, flow, f128, EPI_NONE)
        self.convf2.run(f128, cat256.slice(192, 256), h, w)
        self.conv.run(cat256, out.slice(0, self.out_dim - 2), h, w)
        out.slice(self.out_dim - 2, self.out_dim).tensor().copy_(flow.tensor())

    @ops.on_tensor_device
    def forward(self, flow: torch.Tensor, corr: torch.Tensor, attention=None) -> torch.Tensor:
        flow = flow.contiguous().float()
        corr = corr.contiguous().float()
        ops._dev_check(flow)
        ops._dev_check(corr)
        n, _, h, w = flow.shape
        out = torch.empty(n, self.out_dim, h, w, dtype=torch.float32, device=flow.device)
        self.run(Planes.of(flow), Planes.of(corr), Planes.of(out), h, w)
        return out


def zero_module(module: nn.Module) -> nn.Module:
    """reference update.py:453-457."""
    for p in module.parameters():
        p.detach().zero_()
    return module


class _TimmStyleAttention(nn.Module):
    """Parameter holder with timm's key names: qkv (no bias), proj."""

    def __init__(self, dim: int):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)


class _TimmStyleMlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class TransformerBlock(nn.Module, _Packed):
    """Pre-LN 1-head attention over the token axis + pre-LN MLP, both residual (reference update.py:459-484;
    Attention/Mlp arithmetic is timm's: qkv split [q|k|v], softmax(q k^T / sqrt(dim)) v, proj; fc2(GELU(fc1)))."""

    def __init__(self, dim: int, num_heads: int = 1, mlp_ratio: int = 2, drop_rate: float = 0.0):
        super().__init__()
        if num_heads != 1:
            raise RuntimeError("TransformerBlock: only num_heads=1 (the reference default) is built")
        self.dim = dim
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.attn = _TimmStyleAttention(dim)
        self.mlp = _TimmStyleMlp(dim, int(dim * mlp_ratio))

    def run(self, X: Planes, Y: Planes, B: int, TT: int) -> None:
        """X, Y: [B*TT][dim][P] planes; token (b, t, p) is column p of image b*TT+t."""
        dev = X.base.device
        n, P, C = X.n_img, X.P, self.dim

        def build():
            f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
            return dict(qkv=PackedLinear(self.attn.qkv.weight, None, dev),
                        proj=PackedLinear(self.attn.proj.weight, self.attn.proj.bias, dev),
                        fc1=PackedLinear(self.mlp.fc1.weight, self.mlp.fc1.bias, dev),
                        fc2=PackedLinear(self.mlp.fc2.weight, self.mlp.fc2.bias, dev),
                        n1w=f(self.norm1.weight), n1b=f(self.norm1.bias), n2w=f(self.norm2.weight),
                        n2b=f(self.norm2.bias))
        W = self._packed(build)
        ln = _planes_like(n, C, P, dev)
        qkv = _planes_like(n, 3 * C, P, dev)
        att = _planes_like(n, C, P, dev)
        tx = _planes_like(n, C, P, dev)
        hid = _planes_like(n, W["fc1"].M, P, dev)
        ops.layernorm_cm(X, W["n1w"], W["n1b"], ln, self.norm1.eps)
        ops.gemm(W["qkv"], ln, qkv, EPI_NONE)
        ops.temporal_attn(qkv, att, B, TT, C)
        ops.gemm(W["proj"], att, tx, EPI_RES, R=X)
        ops.layernorm_cm(tx, W["n2w"], W["n2b"], ln, self.norm2.eps)
        ops.gemm(W["fc1"], ln, hid, EPI_GELU)
        ops.gemm(W["fc2"], hid, Y, EPI_RES, R=tx)

    @ops.on_tensor_device
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [S, T, C] tokens (reference layout) -> [S, T, C]."""
        S, TT, C = x.shape
        xp = x.float().permute(1, 2, 0).contiguous()            # [TT, C, S]: one 'clip' with P = S
        ops._dev_check(xp)
        y = torch.empty_like(xp)
        self.run(Planes.of(xp), Planes.of(y), 1, TT)
        return y.permute(2, 0, 1).contiguous()


class TemporalLayer2(nn.Module):
    """reference update.py:502-513: zero-initialised TransformerBlock; input '(B H W) T C', output '(B T) C H W'."""

    def __init__(self, dim: int):
        super().__init__()
        self.transformer_block = zero_module(TransformerBlock(dim))

    @ops.on_tensor_device
    def forward(self, x: torch.Tensor, HW):
        H, W = HW[0], HW[1]
        S, TT, C = x.shape
        B = S // (H * W)
        y = self.transformer_block(x)                            # [(B H W), T, C]
        return y.view(B, H, W, TT, C).permute(0, 3, 4, 1, 2).reshape(B * TT, C, H, W).contiguous()


class SKUpdateBlock_TAM_v3(nn.Module, _Packed):
    """reference update.py:739-782.  forward(nets, inps, corrs, flows, attentions, T) ->
    (nets [BT,128,h,w], masks [B,T,576,h,w], delta_flows [B,T,2,h,w])."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.encoder = SKMotionEncoder6_Deep_nopool_res(args)
        ratio = 16 if getattr(args, "Encoder", "") == "UMT" else 8
        if not getattr(args, "use_gma", True):
            raise RuntimeError("SKUpdateBlock_TAM_v3: use_gma=False is not built (every StreamFlow script sets --use_gma)")
        self.gma = args.use_gma
        embed_dim = args.decoder_dim // 2
        self.embed_dim = embed_dim
        self.aggregator = Aggregate(args=args, dim=embed_dim, dim_head=embed_dim, heads=args.num_heads)
        self.gru = PCBlock4_Deep_nopool_res(embed_dim * 5, embed_dim, k_conv=args.PCUpdater_conv)
        self.mask = nn.Sequential(nn.Conv2d(embed_dim, embed_dim * 2, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(embed_dim * 2, ratio * ratio * 9, 1, padding=0))
        self.transformer_block = TemporalLayer2(dim=embed_dim)
        self.flow_head = PCBlock4_Deep_nopool_res(embed_dim * (args.T - 1), 2 * (args.T - 1), args.k_conv)

    @ops.on_tensor_device
    def forward(self, nets, inps, corrs, flows, attentions, T: Optional[int] = None):
        nets = nets.contiguous().float()
        ops._dev_check(nets)
        BT, C, H, W = nets.shape
        if T is None:
            T = self.args.T - 1
        B = BT // T
        P, dev = H * W, nets.device
        concat = torch.empty(BT, 5 * C, P, dtype=torch.float32, device=dev)
        cp = Planes.of(concat)
        cp.slice(0, C).tensor().copy_(nets.view(BT, C, P))
        cp.slice(C, 2 * C).tensor().copy_(inps.float().reshape(BT, C, P))
        mf = cp.slice(2 * C, 3 * C)
        self.encoder.run(Planes.of(flows.contiguous().float()), Planes.of(corrs.contiguous().float()), mf, H, W)
        mfg = self.aggregator(attentions, mf.tensor().contiguous().view(BT, C, H, W))
        cp.slice(3 * C, 4 * C).tensor().copy_(mfg.view(BT, C, P))
        self.transformer_block.transformer_block.run(mf, cp.slice(4 * C, 5 * C), B, T)
        nets_out = torch.empty(BT, C, H, W, dtype=torch.float32, device=dev)
        self.gru.run(cp, Planes.of(nets_out), H, W)
        delta = torch.empty(B, 2 * T, H, W, dtype=torch.float32, device=dev)
        self.flow_head.run(Planes.of(nets_out.view(B, T * C, H, W)), Planes.of(delta), H, W)
        Wm = self._packed(lambda: (PackedLinear(self.mask[0].weight, self.mask[0].bias, dev, conv3x3=True),
                                   PackedLinear(self.mask[2].weight, self.mask[2].bias, dev)))
        m256 = _planes_like(BT, Wm[0].M, P, dev)
        masks = torch.empty(BT, Wm[1].M, H, W, dtype=torch.float32, device=dev)
        ops.gemm(Wm[0], Planes.of(nets_out), m256, EPI_RELU, hw=(H, W))
        ops.gemm(Wm[1], m256, Planes.of(masks), EPI_NONE, alpha=0.25)
        return nets_out, masks.view(B, T, -1, H, W), delta.view(B, T, 2, H, W)
