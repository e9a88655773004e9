"""CPU oracle for the StreamFlow hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a from-scratch CPU restatement (PyTorch fp32 tensors on the host, explicit
arithmetic) of the reference algorithm for the multi-frame optical-flow inner loop.  It is the
*checker* the HIP path is compared against.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; nothing under ``streamflow_amd/`` does.

Pinning: the reference ships no tests/golden vectors, so the oracle is pinned
against outputs of the reference's own modules executed in the build container
(``tests/golden/make_golden.py`` imports ``/root/reference/core/{corr,gma,update}.py`` and
``core/models/streamflow.py``; vectors are committed under ``tests/golden/``).
``tests/test_oracle_golden.py`` checks every function below against those vectors.
One piece stays "parity unpinned": the temporal transformer block's arithmetic lives in the
third-party ``timm`` package (unpinned in the reference's install.sh, absent from this image); its
published ``Attention``/``Mlp`` definition is restated here and in the golden generator's stub.

All ``file:line`` citations are relative to /root/reference.
Parameters are passed as a flat ``dict[str, Tensor]`` keyed exactly like the reference
``state_dict``, e.g. ``update_block.encoder.convc1.ffn1.0.weight``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# a5: coords_grid                                   core/utils/utils.py:82-85
# ----------------------------------------------------------------------------------------------
def coords_grid(batch: int, ht: int, wd: int) -> Tensor:
    """[B,2,ht,wd] float32; channel 0 = x (column index), channel 1 = y (row index)."""
    xs = torch.arange(wd, dtype=torch.float32).view(1, wd).expand(ht, wd)
    ys = torch.arange(ht, dtype=torch.float32).view(ht, 1).expand(ht, wd)
    return torch.stack([xs, ys], dim=0)[None].expand(batch, -1, -1, -1).contiguous()


# ----------------------------------------------------------------------------------------------
# a4: bilinear_sampler                              core/utils/utils.py:65-79
# ----------------------------------------------------------------------------------------------
def bilinear_sampler(img: Tensor, coords: Tensor, mask: bool = False):
    """img [M,C,Hi,Wi]; coords [M,Ho,Wo,2] pixel coordinates (x,y) -> [M,C,Ho,Wo]
    (and, with mask=True, the [M,Ho,Wo,1] float indicator of utils.py:75-77: normalised coordinates strictly
    inside (-1, 1)).

    The reference normalises pixel coords to [-1,1] (utils.py:69-70) and hands them to
    ``F.grid_sample(align_corners=True)`` (bilinear, zero padding), which maps them back with
    ``((g+1)/2)*(size-1)``.  The same fp32 round trip is reproduced here so that the floor()
    decisions agree with the reference as closely as fp32 allows; the 4-tap gather is explicit.
    """
    M, C, Hi, Wi = img.shape
    x = coords[..., 0]
    y = coords[..., 1]
    gx = 2 * x / (Wi - 1) - 1
    gy = 2 * y / (Hi - 1) - 1
    ix = ((gx + 1) / 2) * (Wi - 1)
    iy = ((gy + 1) / 2) * (Hi - 1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    fx = ix - x0
    fy = iy - y0
    x0 = x0.long()
    y0 = y0.long()
    flat = img.reshape(M, C, Hi * Wi)

    def tap(yy: Tensor, xx: Tensor) -> Tensor:
        ok = (xx >= 0) & (xx < Wi) & (yy >= 0) & (yy < Hi)
        idx = (yy.clamp(0, Hi - 1) * Wi + xx.clamp(0, Wi - 1)).reshape(M, 1, -1).expand(M, C, -1)
        v = torch.gather(flat, 2, idx).reshape(M, C, *xx.shape[1:])
        return v * ok.unsqueeze(1).to(v.dtype)

    w00 = ((1 - fx) * (1 - fy)).unsqueeze(1)
    w01 = (fx * (1 - fy)).unsqueeze(1)
    w10 = ((1 - fx) * fy).unsqueeze(1)
    w11 = (fx * fy).unsqueeze(1)
    out = (tap(y0, x0) * w00 + tap(y0, x0 + 1) * w01
           + tap(y0 + 1, x0) * w10 + tap(y0 + 1, x0 + 1) * w11)
    if mask:
        inside = (gx > -1) & (gy > -1) & (gx < 1) & (gy < 1)
        return out, inside.unsqueeze(-1).float()
    return out


# ----------------------------------------------------------------------------------------------
# a1/a2: all-pairs correlation + pyramid            core/corr.py:7-21, 46-54
# ----------------------------------------------------------------------------------------------
def corr_volume(f1: Tensor, f2: Tensor) -> Tensor:
    """f1,f2 [B,D,h,w] -> [B, h*w, h*w] = f1^T f2 / sqrt(D) (corr.py:46-54)."""
    B, D, h, w = f1.shape
    a = f1.reshape(B, D, h * w).transpose(1, 2)
    b = f2.reshape(B, D, h * w)
    return torch.matmul(a, b) / math.sqrt(D)


def avg_pool_2x2(x: Tensor) -> Tensor:
    """2x2/stride-2 mean over the last two dims with floor on odd sizes (F.avg_pool2d, corr.py:20)."""
    H, W = x.shape[-2:]
    H2, W2 = H // 2, W // 2
    x = x[..., : 2 * H2, : 2 * W2]
    x = x.reshape(*x.shape[:-2], H2, 2, W2, 2)
    return (x[..., 0, :, 0] + x[..., 0, :, 1] + x[..., 1, :, 0] + x[..., 1, :, 1]) * 0.25


def corr_pyramid(f1: Tensor, f2: Tensor, num_levels: int = 4) -> List[Tensor]:
    """List of [B*h*w, 1, h/2^l, w/2^l]; pooling is over the TARGET dims only (corr.py:13-21)."""
    B, D, h, w = f1.shape
    vol = corr_volume(f1, f2).reshape(B * h * w, 1, h, w)
    pyr = [vol]
    for _ in range(num_levels - 1):
        vol = avg_pool_2x2(vol)
        pyr.append(vol)
    return pyr


# ----------------------------------------------------------------------------------------------
# a3: pyramid lookup                                core/corr.py:23-44
# ----------------------------------------------------------------------------------------------
def corr_lookup(pyramid: Sequence[Tensor], coords: Tensor, radius: int = 4) -> Tensor:
    """coords [B,2,h,w] (ch0=x, ch1=y) -> [B, L*(2r+1)^2, h, w] float32.

    Output channel l*(2r+1)^2 + a*(2r+1) + b samples level l at (x/2^l + a - r, y/2^l + b - r):
    the FIRST window axis moves x, because the reference adds a (dy,dx)-ordered meshgrid to
    (x,y)-ordered coordinates (corr.py:31-37).
    """
    B, _, h, w = coords.shape
    r = radius
    n = 2 * r + 1
    c = coords.permute(0, 2, 3, 1).reshape(B * h * w, 1, 1, 2)
    d = torch.linspace(-r, r, n)
    # delta[a, b] = (d[a], d[b]) added to (x, y)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, n, n, 2)
    outs = []
    for lvl, vol in enumerate(pyramid):
        pts = c / 2 ** lvl + delta
        s = bilinear_sampler(vol, pts)          # [B*h*w, 1, n, n]
        outs.append(s.reshape(B, h, w, n * n))
    out = torch.cat(outs, dim=-1)
    return out.permute(0, 3, 1, 2).contiguous().float()


# ----------------------------------------------------------------------------------------------
# a6/a7: GMA attention + aggregate                  core/gma.py:53-65, 91-104
# ----------------------------------------------------------------------------------------------
def gma_attention(inps: Tensor, w_qk: Tensor, heads: int = 1) -> Tensor:
    """inps [BT,C,h,w]; w_qk [2*heads*dh, C, 1, 1] -> attn [BT, heads, N, N] (softmax over last)."""
    BT, C, h, w = inps.shape
    qk = F.conv2d(inps, w_qk)
    q, k = qk.chunk(2, dim=1)
    dh = q.shape[1] // heads
    scale = dh ** -0.5
    q = (scale * q).reshape(BT, heads, dh, h * w)
    k = k.reshape(BT, heads, dh, h * w)
    sim = torch.einsum("bhdi,bhdj->bhij", q, k)
    return sim.softmax(dim=-1)


def gma_aggregate(attn: Tensor, fmap: Tensor, w_v: Tensor, gamma: Tensor) -> Tensor:
    """out = fmap + gamma * (attn @ to_v(fmap)); heads*dim_head == dim so no projection (gma.py:86-104)."""
    BT, C, h, w = fmap.shape
    heads = attn.shape[1]
    v = F.conv2d(fmap, w_v).reshape(BT, heads, -1, h * w)            # [BT, heads, d, N]
    out = torch.einsum("bhij,bhdj->bhdi", attn, v).reshape(BT, -1, h, w)
    return fmap + gamma * out


# ----------------------------------------------------------------------------------------------
# a8: SKBlock (PCBlock4_Deep_nopool_res)            core/update.py:12-36
# ----------------------------------------------------------------------------------------------
def gelu(x: Tensor) -> Tensor:
    """Exact erf GELU (F.gelu default)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def skblock(x: Tensor, p: Params, prefix: str, k_conv: Sequence[int]) -> Tensor:
    g = lambda name: p[prefix + "." + name]
    C = x.shape[1]
    y = F.conv2d(x, g("ffn1.0.weight"), g("ffn1.0.bias"))
    y = F.conv2d(gelu(y), g("ffn1.2.weight"), g("ffn1.2.bias"))
    x = gelu(x + y)
    for i, k in enumerate(k_conv):
        y = F.conv2d(x, g(f"conv_list.{i}.weight"), g(f"conv_list.{i}.bias"), padding=k // 2, groups=C)
        x = gelu(x + y)
    x = gelu(x + F.conv2d(x, g("pw.weight"), g("pw.bias")))
    y = F.conv2d(x, g("ffn2.0.weight"), g("ffn2.0.bias"))
    return F.conv2d(gelu(y), g("ffn2.2.weight"), g("ffn2.2.bias"))


# ----------------------------------------------------------------------------------------------
# a9: motion encoder                                core/update.py:313-339
# ----------------------------------------------------------------------------------------------
def motion_encoder(flow: Tensor, corr: Tensor, p: Params, prefix: str, k_conv: Sequence[int]) -> Tensor:
    cor = gelu(skblock(corr, p, prefix + ".convc1", k_conv))
    cor = skblock(cor, p, prefix + ".convc2", k_conv)
    flo = F.conv2d(flow, p[prefix + ".convf1.weight"], p[prefix + ".convf1.bias"])
    flo = skblock(flo, p, prefix + ".convf2", k_conv)
    out = skblock(torch.cat([cor, flo], dim=1), p, prefix + ".conv", k_conv)
    return torch.cat([out, flow], dim=1)


# ----------------------------------------------------------------------------------------------
# a10: per-pixel temporal transformer block         core/update.py:459-484, 502-513 (+ timm)
# ----------------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def temporal_block(tokens: Tensor, p: Params, prefix: str) -> Tensor:
    """tokens [S, T, C] -> [S, T, C].  Pre-LN 1-head attention over the T axis + pre-LN MLP, both
    residual.  Attention/Mlp semantics are timm's published ones (parity unpinned, see header):
    qkv = Linear(C,3C,no bias) split as [q|k|v]; softmax(q k^T * C^-0.5) v; proj = Linear(C,C);
    Mlp = fc2(GELU(fc1(x)))."""
    g = lambda name: p[prefix + "." + name]
    S, T, C = tokens.shape
    h = layer_norm(tokens, g("norm1.weight"), g("norm1.bias"))
    qkv = h @ g("attn.qkv.weight").t()
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    a = torch.softmax((q * C ** -0.5) @ k.transpose(1, 2), dim=-1)
    o = (a @ v) @ g("attn.proj.weight").t() + g("attn.proj.bias")
    x = tokens + o
    h = layer_norm(x, g("norm2.weight"), g("norm2.bias"))
    h = gelu(h @ g("mlp.fc1.weight").t() + g("mlp.fc1.bias"))
    h = h @ g("mlp.fc2.weight").t() + g("mlp.fc2.bias")
    return x + h


# ----------------------------------------------------------------------------------------------
# a11: SKUpdateBlock_TAM_v3.forward                 core/update.py:764-782
# ----------------------------------------------------------------------------------------------
def update_block(nets: Tensor, inps: Tensor, corrs: Tensor, flows: Tensor, attn: Tensor, T: int,
                 p: Params, prefix: str = "update_block",
                 k_conv: Sequence[int] = (1, 15), gru_conv: Sequence[int] = (1, 7)
                 ) -> Tuple[Tensor, Tensor, Tensor]:
    """nets,inps [B*T,128,h,w]; corrs [B*T,324,h,w]; flows [B*T,2,h,w]; attn [B*T,1,N,N]; T = frames-1.
    Returns nets [B*T,128,h,w], masks [B,T,576,h,w], delta_flows [B,T,2,h,w]."""
    BT, C, h, w = nets.shape
    B = BT // T
    mf = motion_encoder(flows, corrs, p, prefix + ".encoder", k_conv)
    mf_global = gma_aggregate(attn, mf, p[prefix + ".aggregator.to_v.weight"], p[prefix + ".aggregator.gamma"])
    tok = mf.reshape(B, T, C, h * w).permute(0, 3, 1, 2).reshape(B * h * w, T, C)
    tok = temporal_block(tok, p, prefix + ".transformer_block.transformer_block")
    mf_temporal = tok.reshape(B, h * w, T, C).permute(0, 2, 3, 1).reshape(BT, C, h, w)
    x = torch.cat([nets, inps, mf, mf_global, mf_temporal], dim=1)
    nets = skblock(x, p, prefix + ".gru", gru_conv)
    dflow = skblock(nets.reshape(B, T * C, h, w), p, prefix + ".flow_head", k_conv)
    m = F.conv2d(nets, p[prefix + ".mask.0.weight"], p[prefix + ".mask.0.bias"], padding=1)
    m = F.conv2d(torch.relu(m), p[prefix + ".mask.2.weight"], p[prefix + ".mask.2.bias"])
    masks = (0.25 * m).reshape(B, T, -1, h, w)
    return nets, masks, dflow.reshape(B, T, 2, h, w)


# ----------------------------------------------------------------------------------------------
# K12: convex upsampling                            core/models/streamflow.py:82-93
# ----------------------------------------------------------------------------------------------
def upsample_flow(flow: Tensor, mask: Tensor, ratio: int = 8) -> Tensor:
    """flow [N,2,h,w], mask [N,9*ratio^2,h,w] -> [N,2,ratio*h,ratio*w]."""
    N, _, h, w = flow.shape
    m = torch.softmax(mask.reshape(N, 1, 9, ratio, ratio, h, w), dim=2)
    fp = F.pad(ratio * flow, (1, 1, 1, 1))
    nb = torch.stack([fp[:, :, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], dim=2)
    up = (m * nb.reshape(N, 2, 9, 1, 1, h, w)).sum(dim=2)          # [N,2,r,r,h,w]
    return up.permute(0, 1, 4, 2, 5, 3).reshape(N, 2, ratio * h, ratio * w)


# ----------------------------------------------------------------------------------------------
# a12: the refinement loop                          core/models/streamflow.py:110-147
# ----------------------------------------------------------------------------------------------
def hotpath_forward(fmaps: Tensor, cnets: Tensor, p: Params, iters: int,
                    flow_init: Optional[Sequence[Tensor]] = None, num_heads: int = 1,
                    all_iters: bool = False):
    """fmaps [B,T,D,h,w] fp32 (encoder features), cnets [B,T-1,256,h,w] (context features).

    Returns (flows_up, flows_lowres): lists of T-1 tensors [B,2,8h,8w] / [B,2,h,w] after the
    last iteration (``test_mode`` semantics, streamflow.py:142-147).  With ``all_iters`` the first
    element is a list (per pair) of lists (per iteration) instead (training-mode return, :149).
    """
    B, T, D, h, w = fmaps.shape
    P = T - 1
    pyramids = [corr_pyramid(fmaps[:, i], fmaps[:, i + 1]) for i in range(P)]
    coords0 = [coords_grid(B, h, w) for _ in range(P)]
    coords1 = [coords_grid(B, h, w) for _ in range(P)]
    if flow_init is not None:
        coords1 = [coords1[i] + flow_init[i] for i in range(len(flow_init))]
    hdim = cnets.shape[2] // 2
    nets = torch.tanh(cnets[:, :, :hdim]).reshape(B * P, hdim, h, w)
    inps = torch.relu(cnets[:, :, hdim:]).reshape(B * P, hdim, h, w)
    attn = gma_attention(inps, p["att.to_qk.weight"], heads=num_heads)
    preds: List[List[Tensor]] = [[] for _ in range(P)]
    masks = None
    for _ in range(iters):
        corrs = torch.stack([corr_lookup(pyramids[i], coords1[i]) for i in range(P)], dim=1)
        corrs = corrs.reshape(B * P, -1, h, w)
        flows = torch.stack([coords1[i] - coords0[i] for i in range(P)], dim=1).reshape(B * P, 2, h, w)
        nets, masks, dflow = update_block(nets, inps, corrs, flows, attn, P, p)
        coords1 = [coords1[i] + dflow[:, i] for i in range(P)]
        if all_iters:
            for i in range(P):
                preds[i].append(upsample_flow(coords1[i] - coords0[i], masks[:, i]))
    lowres = [coords1[i] - coords0[i] for i in range(P)]
    if all_iters:
        return preds, lowres
    ups = [upsample_flow(lowres[i], masks[:, i]) for i in range(P)]
    return ups, lowres


def epe(a: Tensor, b: Tensor) -> float:
    """Mean end-point error between two [.,2,H,W] flow fields (evaluate_mf.py:146-147)."""
    return torch.sqrt(((a - b) ** 2).sum(dim=-3)).mean().item()


def forward_interpolate(flow: torch.Tensor) -> torch.Tensor:
    """core/utils/utils.py:34-62 (warm start of the next clip, evaluate_mf.py:305): every source pixel is pushed
    along its flow (float64 position = integer grid + float32 flow, numpy promotion), points landing outside
    (0, w) x (0, h) are dropped, and each grid pixel takes the flow of the NEAREST remaining point
    (scipy.interpolate.griddata(method='nearest') = exact nearest neighbour in the Euclidean metric).
    Restated as a brute-force argmin over squared float64 distances; ties (measure zero) go to the lowest
    source index.  With no valid point the result is zero (scipy raises there)."""
    f = flow.detach().cpu().double()
    ht, wd = f.shape[-2:]
    y0, x0 = torch.meshgrid(torch.arange(ht, dtype=torch.float64), torch.arange(wd, dtype=torch.float64), indexing="ij")
    x1, y1 = (x0 + f[0]).reshape(-1), (y0 + f[1]).reshape(-1)
    dx, dy = f[0].reshape(-1), f[1].reshape(-1)
    valid = (x1 > 0) & (x1 < wd) & (y1 > 0) & (y1 < ht)
    if not bool(valid.any()):
        return torch.zeros(2, ht, wd)
    x1, y1, dx, dy = x1[valid], y1[valid], dx[valid], dy[valid]
    gx, gy = x0.reshape(-1, 1), y0.reshape(-1, 1)
    d2 = (x1[None, :] - gx) ** 2 + (y1[None, :] - gy) ** 2          # [targets, points]
    idx = d2.argmin(dim=1)
    return torch.stack([dx[idx].reshape(ht, wd), dy[idx].reshape(ht, wd)]).float()
