"""Fused, pre-planned execution of the StreamFlow hot path on one MI355X.

Covers reference core/models/streamflow.py:110-147 (the refinement loop) with everything it
calls: corr.py (volume + pyramid + lookup), gma.py (Attention/Aggregate), update.py
(SKUpdateBlock_TAM_v3 and its SKBlocks, motion encoder, temporal transformer block, mask head) and
the convex upsampling.  Weights are repacked once (K-major, padded) at construction; all
activation buffers for a given (clips, frames, h, w) are carved once out of a single workspace, so a
forward is a fixed sequence of kernel launches with constant pointers -- capturable as one HIP
graph (``use_graph=True``) and replayed per clip.

Image index convention everywhere: img = clip * (T-1) + pair   (the reference's '(B T)' packing).
"""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass, fields, replace
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib, ops
from .ops import (EPI_AXPY, EPI_GELU, EPI_NONE, EPI_RELU, EPI_RES, EPI_RES_GELU, EPI_RES_GELU_DW1, LAYOUT_F16_K_MINOR,
                  LAYOUT_K_MAJOR, LAYOUT_K_MINOR, PackedLinear, Planes, Workspace)

HDIM = 128
COR_PLANES = 324


@dataclass(frozen=True)
class EngineOptions:
    """Scheduling and hand-over switches of HotPathEngine (the A/B knobs that used to be environment variables).  Results are
    the same arithmetic under every setting unless noted; every field is part of the engine's graph key.
    `HotPathEngine(options=EngineOptions(split_solo=0))`, or from the shell for an experiment: SF_ENGINE_OPTS="split_solo=0,pw_fold=0"."""
    parallel_branches: bool = True   # independent chains of an iteration on a second stream
    split_solo: int = 2              # 0 / 2 / 4: sections without a concurrent branch run as half-batch chains on own streams
    split_uneven: bool = False       # ... also when the clip count does not divide (a single clip, an odd batch: two unequal ranges of images).
                                     # Off (round 6): one chain is faster there -- 1 clip 229.8 -> 237.2 ff/s, 3 clips 341.7 -> 348.9, one Spring
                                     # clip 50.0 -> 50.8; even splits keep their chains (2 / 6 / 8 clips: +2-3 %; tools/clip_split_sweep.sh)
    auto_split_k: bool = True        # let sf_gemm split K for small grids (changes summation order at toy shapes only)
    attn_chunk_rows: int = 0         # > 0 forces the chunked recompute of the attention matrix
    attn_k_splits: int = 3           # split-K of attn @ v (materialised matrix), <= 4
    corr_blocked: bool = True        # fp16 volumes in the blocked layout (k-octet-only hand-over of the correlation features)
    corr_blocked32: bool = True      # fp32 volumes in cache-line blocks of 4 rows x 8 columns (csrc/corr_blocked32.hip) instead of row-major maps
    shadows: bool = True             # fp16 k-octet copies of the SK blocks' inputs
    shadow_fused: bool = True        # ... written by their producers' epilogues instead of a pack pass
    hidden_f16: bool = True          # GEMM-to-GEMM tensors as fp16 in the f16x2 / f16 modes
    hidden_koct: bool = True         # ... as k-octet planes where the consumer takes them
    split_handover: bool = False     # f16x3 (fp32_class): GEMM-to-GEMM tensors stored already split, hi + lo k-octet images (SF_LAYOUT_SPLIT_KOCT):
                                     # bit-identical; the consumers run 7-10 % faster on two DMA-fed operands, the producers' k-octet epilogue
                                     # costs more than that (fp32_class 182.6 -> 178.2 ff/s; tools/gemm_split_koct_bench.py): off
    pw_fold: bool = True             # pw residual folded into the weights (x3 handed over in fp16)
    koct_io: bool = True             # motion-encoder tensors between SK blocks as fp16 k-octets only (no fp32 planes)
    x2_f16: bool = True              # single-reader tensors as fp16 ROWS: x2 (ffn1.2 -> depthwise), qkv (-> temporal attention), v (-> GMA pack)
    flash_stats: bool = True         # fused GMA: softmax statistics computed once per clip
    ffn_pairs: bool = True           # the SK blocks' ffn1 / ffn2 pairs as one launch each where the shape is built (sf_ffn_pair)
    project_v: bool = True           # fused GMA, fp16 activations: to_v + the v pack as one launch (sf_gma_flash_project_v)
    temporal_block: bool = True      # the temporal transformer block as ONE launch (sf_temporal_block) instead of seven
    sk_tail: bool = True             # an SK block's back half (pw -> GELU -> ffn2.0 -> GELU -> ffn2.2) as ONE launch where that pays (sf_sk_tail)
    sk_tail_all: bool = False        # ... wherever the shape is built (the flow head, small launches): tests, A/B
    head_pairs: bool = False         # the flow head's FFN pairs on its grouped view (sf_ffn_pair x_group / R32).  Off: +0.4 % on the step
                                     # (382 vs 380 ff/s) against EPE samples of 3.6 / 1.9 / 2.6e-4 px instead of 2.9 / 1.7 / 2.6e-4
    mask_upsample: bool = True       # mask head's second layer + convex upsampling as ONE launch (sf_mask_upsample): the mask is never written
    clock_probe_us: int = 0          # measurement aid: > 0 forks sf_clock_probe for that long beside every forward (results in
                                     # engine.clock_counts: shader cycles, 100 MHz ticks); a graph branch like any other
    setup_overlap: bool = True       # the context chain of the setup (split, to_qk, GMA pack / statistics) beside the volume build
    gma_per_chain: bool = True       # the fused GMA kernel as one launch per half-batch chain, each chain going straight on into its GRU
                                     # (the partial last round of one chain's GMA grid is filled by the other chain's kernels: 1320
                                     # workgroups on 768 slots = 1.72 rounds; 62.0 -> 61.6 ms per step, bit-identical)
    hybrid_store_pct: int = 50       # gma_mode 'hybrid': share of the images whose softmax weights are stored (the rest is recomputed beside them)
    stored_auto_px: int = 16384      # gma_mode 'flash': grids of at least this many pixels -- and split-precision logits (flash_qk_products >= 2)
                                     # at any grid -- keep the softmax weights for the loop instead (the 'stored' form, bit-identical): at
                                     # 1080p the stream is 19 % faster than the one-product recompute (1164 vs 1438 us per iteration, 49.1 vs
                                     # 47.0 ff/s for a Spring clip), 2.5x faster than the three-product one; at Sintel / KITTI grids with
                                     # one product it is not.  0: never
    stored_auto_max_gb: int = 16     # ... when the weight buffer stays under this size
    stored_p_max_gb: int = 64        # gma_mode 'stored': largest weight buffer (n_img * Ppad^2 * 2 bytes) kept; beyond it the fused recompute runs
    max_plans: int = 4               # buffer sets (and graphs) kept, least recently used evicted

    @staticmethod
    def from_env(base: Optional["EngineOptions"] = None) -> "EngineOptions":
        opt = base or EngineOptions()
        spec = os.environ.get("SF_ENGINE_OPTS", "")
        if not spec:
            return opt
        types = {f.name: f.type for f in fields(EngineOptions)}
        kw = {}
        for item in spec.split(","):
            k, _, v = item.partition("=")
            k = k.strip()
            if k not in types:
                raise RuntimeError(f"SF_ENGINE_OPTS: unknown option {k!r} (have {sorted(types)})")
            kw[k] = (v.strip() not in ("0", "false", "False", "")) if types[k] in (bool, "bool") else int(v)
        return replace(opt, **kw)


class SKBlockWeights:
    """Packed parameters of one PCBlock4_Deep_nopool_res (reference update.py:12-29)."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, device):
        g = lambda n: sd[prefix + "." + n]
        self.ffn1_0 = PackedLinear(g("ffn1.0.weight"), g("ffn1.0.bias"), device)
        self.ffn1_2 = PackedLinear(g("ffn1.2.weight"), g("ffn1.2.bias"), device)
        self.pw = PackedLinear(g("pw.weight"), g("pw.bias"), device)
        # x4 = gelu(x3 + pw(x3)) = gelu((W + I) x3 + b): with the residual folded into the weights x3 has ONE reader and
        # can be handed over in fp16 like the FFN hiddens (run_skblock, f16x2 / f16 modes only)
        wpw = g("pw.weight")
        eye = torch.eye(wpw.shape[0], dtype=wpw.dtype, device=wpw.device).reshape(wpw.shape[0], wpw.shape[0], *wpw.shape[2:])
        self.pw_res = PackedLinear(wpw + eye, g("pw.bias"), device)
        self.ffn2_0 = PackedLinear(g("ffn2.0.weight"), g("ffn2.0.bias"), device)
        self.ffn2_2 = PackedLinear(g("ffn2.2.weight"), g("ffn2.2.bias"), device)
        # the two FFNs as single launches (update.py:14-16: nn.Sequential(conv, GELU, conv)); weight streams are built on first use
        self.pair1 = ops.PackedPair(self.ffn1_0, self.ffn1_2)
        self.pair2 = ops.PackedPair(self.ffn2_0, self.ffn2_2)
        # ... and the whole back half, pw (residual folded) -> ffn2.0 -> ffn2.2, as one launch (update.py:35-36; csrc/sk_tail.hip)
        self.tail = ops.PackedTail(self.pw_res, self.ffn2_0, self.ffn2_2)
        self.c_in, self.c_mid, self.c_out = self.ffn1_0.K, self.ffn1_0.M, self.ffn2_2.M
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        w0 = g("conv_list.0.weight")
        if w0.shape[-1] != 1:
            raise RuntimeError("SKBlock: first depthwise kernel must be 1x1 (k_conv=[1,K]), got %s" % (tuple(w0.shape),))
        self.dw1_w = f32(w0.reshape(-1))
        self.dw1_b = f32(g("conv_list.0.bias"))
        wk = g("conv_list.1.weight")
        self.k = int(wk.shape[-1])
        self.dwk_w = f32(wk.reshape(wk.shape[0], -1))
        self.dwk_b = f32(g("conv_list.1.bias"))
        self.dw_single = False                # the K x K depthwise weights as one fp16 value (layer name '<block>.dw')


def _scratch(buf: Planes, n_img: int, rows: int, f16: bool = False, koct: bool = False, split: bool = False) -> Planes:
    """Reinterpret a scratch allocation as contiguous [n_img][rows][P] planes (fp32, or fp16 values in the same memory;
    koct: fp16 k-octet planes [ceil(rows/8)][P][8], the DMA-able image of the consuming GEMM; split: two such images per
    image, hi then lo -- the f16x3 hand-over, as many bytes as the fp32 planes)."""
    rows_alloc = (rows + 7) // 8 * 8 if koct else rows
    assert not split or (f16 and koct)
    assert n_img * rows_alloc * (1 if (not f16 or split) else 0.5) <= buf.n_img * buf.rows, "scratch too small"
    return Planes(buf.base, buf.off, rows_alloc * buf.P * (2 if split else 1), n_img, rows, buf.P, f16=f16, koct=f16 and koct,
                  split=split)


def _handover(cx: ops.Ctx, buf: Planes, n_img: int, rows: int, P: int, consumer_rows: int, allow_koct: bool = True) -> Planes:
    # (rows = K of the consuming layer, consumer_rows = its M)
    """Scratch view for a tensor that is written by one GEMM and read only as the B operand of the next: fp32 planes in
    the exact / f16x3 modes; in f16x2 fp16 values -- as k-octet planes when the consumer runs on the DMA-fed 128-row tile
    (both of its operands then go HBM/L2 -> LDS without touching registers), as fp16 rows otherwise."""
    if cx.precision == ops.PRECISION_F16X3 and cx.split_handover and ops.uses_dma_tile(consumer_rows) and P % 4 == 0:
        # f16x3: the (hi, lo) pair the consumer would compute, stored by the producer: same bytes, same values, both operands by DMA
        return _scratch(buf, n_img, rows, f16=True, koct=True, split=True)
    f16 = hidden_f16_ok(cx, P)
    koct = f16 and allow_koct and ops.takes_koct(consumer_rows, rows) and cx.hidden_koct
    return _scratch(buf, n_img, rows, f16=f16, koct=koct)


def _sub(X: Planes, i0: int, cnt: int) -> Planes:
    """Images i0 .. i0 + cnt - 1 of X (and of its k-octet copy)."""
    assert 0 <= i0 and i0 + cnt <= X.n_img
    step = X.img_stride // (2 if X.f16 else 1)                     # `off` counts floats of the underlying allocation
    assert not X.f16 or X.img_stride % 2 == 0
    sh = _sub(X.shadow, i0, cnt) if X.shadow is not None else None
    return replace(X, off=X.off + i0 * step, n_img=cnt, shadow=sh)


def _part(buf: Planes, i0: int, cnt: int) -> Planes:
    """The piece of a scratch allocation that belongs to images i0 .. i0 + cnt - 1 (concurrent chains over disjoint images)."""
    return replace(buf, off=buf.off + i0 * buf.img_stride, n_img=cnt)


def hidden_f16_ok(cx: ops.Ctx, P: int) -> bool:
    """FFN hidden activations are handed from GEMM to GEMM as fp16 in the f16x2 mode (bit-identical: that mode rounds a
    B operand to fp16 on load anyway) when the plane geometry allows 8-byte stores / dword loads."""
    return cx.precision in (ops.PRECISION_F16X2, ops.PRECISION_F16) and P % 4 == 0 and cx.hidden_f16


def run_skblock(W: SKBlockWeights, X: Planes, Y: Planes, hid: Planes, xa: Planes, xb: Planes, h: int, w: int,
                final_gelu: bool = False, cx: Optional[ops.Ctx] = None) -> None:
    """One PCBlock4_Deep_nopool_res forward, op order of reference update.py:30-36:
    x1 = gelu(x + ffn1(x)); x2 = gelu(x1 + dw1x1(x1)); x3 = gelu(x2 + dwKxK(x2)); x4 = gelu(x3 + pw(x3));
    y = ffn2(x4).  hid/xa/xb are scratch allocations (capacity >= n_img*c_mid / n_img*c_in rows)."""
    cx = ops._cx(cx)
    C = W.c_in
    assert X.rows == C and Y.rows == W.c_out and X.n_img == Y.n_img
    a, b = _scratch(xa, X.n_img, C), _scratch(xb, X.n_img, C)
    fold = hidden_f16_ok(cx, X.P) and cx.pw_fold
    if fold and cx.x2_f16:
        # x2 has ONE reader, the depthwise layer, whose two-product arithmetic multiplies fp16(x2) anyway: handed over as fp16
        # rows (half the bytes out of this GEMM and into the depthwise kernel, which then stages its strips by DMA)
        a = _scratch(xa, X.n_img, C, f16=True)
    # x1 = gelu(x + ffn1(x)); x2 = gelu(x1 + dw1x1(x1))  (both fused in the epilogue)
    # (the flow head's pairs pay with enough pixels to fill the chip at 64 per workgroup: 8 clips 63.3 vs 63.8 ms per step, one
    # clip 220 vs 226 ff/s the other way -- its GEMM launches split M for small grids)
    head_pairs = cx.head_pairs and X.group and X.shadow is not None and X.n_img * X.P >= 4 * 7040
    if cx.ffn_pairs and a.f16 and ((X.f16 and X.koct) or head_pairs) and ops.ffn_pair_ok(W.pair1, X, 1, cx):
        # the block input exists as k-octets only (operand AND residual) -- or it is the flow head's grouped view of the hidden
        # state, whose k-octet copy is the operand and (round 5) the residual: the whole ffn1 + epilogue is one launch, the 1.5 C
        # hidden tensor stays in registers (csrc/ffn_pair.hip)
        ops.ffn_pair(W.pair1, X, a, 1, dw_w=W.dw1_w, dw_b=W.dw1_b, cx=cx)
    else:
        hidden = _handover(cx, hid, X.n_img, W.c_mid, X.P, consumer_rows=C)         # ffn1.0 -> ffn1.2
        ops.gemm(W.ffn1_0, X, hidden, EPI_GELU, cx=cx)
        ops.gemm(W.ffn1_2, hidden, a, EPI_RES_GELU_DW1, R=X, dw_w=W.dw1_w, dw_b=W.dw1_b, cx=cx)
    # x4 is read by ffn2.0 only: the same GEMM-to-GEMM hand-over as the hidden activations (x2 in `xa` is dead by now)
    if fold:
        # x3 in fp16 rows straight out of the depthwise kernel, residual folded into the pw weights: the pw GEMM reads half
        # the bytes, has a residual-free epilogue and may therefore write k-octets
        b16 = _scratch(xb, X.n_img, C, f16=True)
        ops.dwconv_res_gelu(a, W.dwk_w, W.dwk_b, b16, h, w, W.k, single=W.dw_single, cx=cx)   # x3 = gelu(x2 + dwKxK(x2))
        # the back half as ONE launch: x4 and the 1.5 C hidden never leave the registers (csrc/sk_tail.hip; round 6).  Measured
        # (profiles/r06_sk_tail_ab.txt, 24 images): 195 / 178 / 46 us against 218 / 219 / 58 us for the three launches at C = 256 / 256 /
        # 128 -- used there; the flow head's shape (C = 384: one wave per SIMD for the registers) is 133 against 122 us, and a launch
        # too small to fill the chip (a single clip) is faster as three launches whose grids split M: both keep the three launches
        if (ops.sk_tail_ok(W.tail, b16, Y, cx) and W.c_in <= 256 and X.n_img * X.P >= 4 * 7040) or \
                (cx.sk_tail_all and ops.sk_tail_ok(W.tail, b16, Y, cx)):
            ops.sk_tail(W.tail, b16, Y, gelu_out=final_gelu, cx=cx)
            return
        a4 = _handover(cx, xa, X.n_img, C, X.P, consumer_rows=W.c_mid)
        ops.gemm(W.pw_res, b16, a4, EPI_GELU, cx=cx)                            # x4 = gelu((pw + I) x3)
    else:
        ops.dwconv_res_gelu(a, W.dwk_w, W.dwk_b, b, h, w, W.k, single=W.dw_single, cx=cx)   # x3 = gelu(x2 + dwKxK(x2))
        # (fp16 ROWS, not k-octets: the k-octet epilogue fetches its residual with 8 dword loads per octet and made the
        # pw GEMMs 15-20 % slower -- more than their consumers gained)
        a4 = _handover(cx, xa, X.n_img, C, X.P, consumer_rows=W.c_mid, allow_koct=False)
        ops.gemm(W.pw, b, a4, EPI_RES_GELU, R=b, cx=cx)                         # x4 = gelu(x3 + pw(x3))
    # (ffn2 pairs with hi + lo weights measured no faster than their two launches at C >= 256 -- 144 vs ~135 us at 256 -> 384 -> 192:
    # twice the MFMAs on 16 x 16 x 32 tiles, whose fragment reads bind -- so those stay on the 32 x 32 x 16 kernels)
    pair2_pays = W.c_in <= 128 or W.pair2.products(cx) == (1, 1) or (W.c_out <= 16 and head_pairs)   # (the flow head: layer 2 is one 16-row tile)
    if X.n_img * X.P < 4 * 7040 and not X.group:            # (a small launch is latency-bound: one launch instead of two, +0.7 % on a single clip)
        pair2_pays = True
    if (cx.ffn_pairs and pair2_pays and a4.f16 and a4.koct and ops.ffn_pair_ok(W.pair2, a4, 0, cx) and (not Y.f16 or Y.koct)):
        ops.ffn_pair(W.pair2, a4, Y, 0, gelu_out=final_gelu, cx=cx)                 # y = ffn2(x4): one launch
        return
    hidden = _handover(cx, hid, X.n_img, W.c_mid, X.P, consumer_rows=W.c_out)       # ffn2.0 -> ffn2.2
    ops.gemm(W.ffn2_0, a4, hidden, EPI_GELU, cx=cx)
    ops.gemm(W.ffn2_2, hidden, Y, EPI_GELU if final_gelu else EPI_NONE, cx=cx)


class HotPathWeights:
    """All hot-path parameters, repacked for the kernels.  `sd` uses the reference state-dict keys
    (SURVEY.md section 8b); a leading 'module.' (DataParallel) is stripped."""

    def __init__(self, sd: Dict[str, torch.Tensor], device, T: Optional[int] = None):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        u = "update_block"
        e = u + ".encoder"
        tb = u + ".transformer_block.transformer_block"
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        self.device = device
        self.to_qk = PackedLinear(sd["att.to_qk.weight"], None, device)
        if self.to_qk.M != 2 * HDIM:
            raise RuntimeError("only num_heads=1, dim_head=128 is supported (canonical StreamFlow configuration)")
        self.convc1 = SKBlockWeights(sd, e + ".convc1", device)
        self.convc2 = SKBlockWeights(sd, e + ".convc2", device)
        self.convf1 = PackedLinear(sd[e + ".convf1.weight"], sd[e + ".convf1.bias"], device)
        self.convf2 = SKBlockWeights(sd, e + ".convf2", device)
        self.conv = SKBlockWeights(sd, e + ".conv", device)
        self.to_v = PackedLinear(sd[u + ".aggregator.to_v.weight"], None, device)
        self.gamma = f32(sd[u + ".aggregator.gamma"])
        self.gru = SKBlockWeights(sd, u + ".gru", device)
        self.flow_head = SKBlockWeights(sd, u + ".flow_head", device)
        self.pairs = self.flow_head.c_in // HDIM
        if T is not None and self.pairs != T - 1:
            raise RuntimeError(f"flow_head expects T-1={self.pairs} pairs, model was asked for T={T}")
        self.mask0 = PackedLinear(sd[u + ".mask.0.weight"], sd[u + ".mask.0.bias"], device, conv3x3=True)
        self.mask2 = PackedLinear(sd[u + ".mask.2.weight"], sd[u + ".mask.2.bias"], device)
        self.ln1_w, self.ln1_b = f32(sd[tb + ".norm1.weight"]), f32(sd[tb + ".norm1.bias"])
        self.ln2_w, self.ln2_b = f32(sd[tb + ".norm2.weight"]), f32(sd[tb + ".norm2.bias"])
        self.qkv = PackedLinear(sd[tb + ".attn.qkv.weight"], sd.get(tb + ".attn.qkv.bias"), device)
        self.proj = PackedLinear(sd[tb + ".attn.proj.weight"], sd[tb + ".attn.proj.bias"], device)
        self.fc1 = PackedLinear(sd[tb + ".mlp.fc1.weight"], sd[tb + ".mlp.fc1.bias"], device)
        self.fc2 = PackedLinear(sd[tb + ".mlp.fc2.weight"], sd[tb + ".mlp.fc2.bias"], device)
        self.temporal = ops.PackedTemporal(self.qkv, self.proj, self.fc1, self.fc2)
        self.mask_pack = ops.PackedMask(self.mask2)

    SK_BLOCKS = ("convc1", "convc2", "convf2", "conv", "gru", "flow_head")
    SK_LAYERS = ("ffn1_0", "ffn1_2", "pw", "ffn2_0", "ffn2_2")
    PLAIN_LAYERS = ("to_qk", "convf1", "to_v", "qkv", "proj", "fc1", "fc2", "mask0", "mask2")

    def layers(self) -> Dict[str, PackedLinear]:
        """Every contraction's packed weights by name ('gru.ffn1_0', 'qkv', ...; 'X.pw' covers both forms of the pw layer)."""
        out = {n: getattr(self, n) for n in self.PLAIN_LAYERS}
        for b in self.SK_BLOCKS:
            for l in self.SK_LAYERS:
                out[f"{b}.{l}"] = getattr(getattr(self, b), l)
        return out

    def set_single(self, names) -> None:
        """Mark the named layers (or 'all') as single-product in the f16x2 mode (ops.PackedLinear.single)."""
        lay = self.layers()
        dws = [b + ".dw" for b in self.SK_BLOCKS]
        names = (list(lay) + dws) if names == "all" else list(names or [])
        unknown = [n for n in names if n not in lay and n not in dws]
        if unknown:
            raise RuntimeError(f"unknown layer name(s) {unknown}; have {sorted(lay) + dws}")
        for b in self.SK_BLOCKS:
            getattr(self, b).dw_single = (b + ".dw") in names
        for n, pl in lay.items():
            pl.single = n in names
            if n.endswith(".pw"):
                getattr(self, n.split(".")[0]).pw_res.single = pl.single


ATTN_ROW_LIMIT = 1 << 28          # floats per image the GEMM epilogue can address with 32-bit offsets (1 GiB)


class _Plan:
    """Buffers for one (clips, pairs, h, w) shape."""

    def __init__(self, W: HotPathWeights, Bc: int, h: int, w: int, D: int, device, attn_chunk_rows: int = 0,
                 attn_f16: bool = False, corr_f16: bool = False, flash: Optional[bool] = None, shadows: bool = False,
                 corr_blocked: bool = False, koct_io: bool = False, stored_p: bool = False, stored_p_max_bytes: int = 0,
                 stored_frac: float = 1.0, corr_blocked32: bool = False):
        Pn = W.pairs
        n, P = Bc * Pn, h * w
        self.Bc, self.Pn, self.h, self.w, self.n, self.P, self.D = Bc, Pn, h, w, n, P, D
        if (h >> 3) < 1 or (w >> 3) < 1:
            raise RuntimeError(f"feature grid {h}x{w} too small for a 4-level pyramid (need >= 8x8)")
        if (h >> 3) < 2 or (w >> 3) < 2:
            # the reference divides by (W-1) of the coarsest level (utils.py:69-70): size-1 levels give NaN there
            raise RuntimeError(f"feature grid {h}x{w}: coarsest pyramid level would be 1 pixel wide; the reference "
                               "is undefined there (needs images >= 128 px per side)")
        dims = [(h >> l, w >> l) for l in range(4)]
        # fp32 maps: rows padded to whole cache lines where the dense layout would straddle them (ops.corr_pitch)
        self.corr_pitch = None if corr_f16 else ops.corr_pitch(h, w)
        self.lvl_pair_stride = [Bc * P * hl * (self.corr_pitch[l] if self.corr_pitch else wl) for l, (hl, wl) in enumerate(dims)]
        # blocked fp16 volumes (csrc/corr_blocked.hip): one buffer, 8 x 8-cell blocks = cache lines; the lookup then hands
        # the correlation features over as fp16 k-octets ONLY (operand and residual of convc1's first block)
        self.corr_blocked = bool(corr_blocked and corr_f16 and shadows)
        # blocked fp32 volumes (csrc/corr_blocked32.hip): same idea with 4-row x 8-column blocks; the features leave as fp32 planes
        self.corr_blocked32 = bool(corr_blocked32 and not corr_f16)
        if self.corr_blocked32:
            self.vol = ops.new_blocked_volume(n, h, w, device, f32=True)
            self.lvls = None
            self.corr_pitch = None
            self.corr_ws = torch.empty(max(ops.corr_build_blocked_ws_bytes(n, D, h, w, True), 16), dtype=torch.uint8, device=device)
        elif self.corr_blocked:
            self.vol = ops.new_blocked_volume(n, h, w, device)
            self.lvls = None
            self.corr_ws = torch.empty(max(ops.corr_build_blocked_ws_bytes(n, D, h, w), 16), dtype=torch.uint8, device=device)
        else:
            self.vol = None
            # correlation pyramids of all pairs: fp32 cells, or fp16 cells (corr_dtype='f16': half the bytes of the
            # HBM-bound build and lookups; BASELINE.json configs 2 and 5)
            self.lvls = [torch.empty(Pn * s, dtype=torch.float16 if corr_f16 else torch.float32, device=device)
                         for s in self.lvl_pair_stride]
            # scratch of the split-precision volume build: (hi, lo) fp16 planes of every f1 / f2 image
            self.corr_ws = torch.empty(max(ops.corr_build_ws_bytes(Bc, Pn, D, h, w), 16), dtype=torch.uint8, device=device)
        # GMA attention: materialise the N x N matrix once (reference core/gma.py) when an image's matrix fits the
        # kernels' 32-bit offsets; otherwise (high resolution, e.g. 1080p: N = 32400 -> 4.2 GB per image) recompute
        # softmax(q k^T) v in row chunks every iteration, which is what the reference's demo does through
        # flash-attn (demo.py:235-258): same mathematics, no N^2 tensor kept.
        if attn_chunk_rows <= 0 and P * P < ATTN_ROW_LIMIT:
            self.attn_rows = P
        else:
            lim = max(1, (ATTN_ROW_LIMIT - 1) // P)
            self.attn_rows = min(P, lim if attn_chunk_rows <= 0 else min(attn_chunk_rows, lim))
        # fused path (sf_gma_flash_*): softmax(q k^T) v recomputed inside one kernel per iteration, logits never leave
        # the CU.  flash=None: used whenever the matrix would have to be chunked (split precisions only).
        self.flash = bool(flash) if flash is not None else (attn_f16 and self.attn_rows < P and attn_chunk_rows <= 0)
        self.flash_ws = None
        self.pbuf = None
        self.n_store = 0
        if self.flash:
            self.flash_ws = torch.empty(ops.gma_flash_ws_bytes(n, P), dtype=torch.uint8, device=device)
            self.attn_rows = 1                     # no attention matrix at all: placeholders only
            # 'stored': the softmax weights kept for the loop in the fragment order of the second contraction (sf_gma_flash_store_p)
            # ('hybrid': for the first n_store images only -- the stream of stored weights is HBM-bound, the recompute kernel is
            # bound by the chip's power budget: run side by side on disjoint images they use different resources, section 12)
            self.n_store = n if stored_frac >= 1.0 else max(0, min(n, int(round(n * stored_frac))))
            if self.n_store < n and not (self.n_store > 0 and ops.gma_no_key_split(self.n_store, P) and ops.gma_no_key_split(n - self.n_store, P)):
                self.n_store = 0                   # (a launch small enough for the key-split form keeps the recompute kernel alone)
            if stored_p and self.n_store > 0 and ops.gma_stored_p_bytes(self.n_store, P) <= stored_p_max_bytes:
                self.pbuf = torch.empty(ops.gma_stored_p_bytes(self.n_store, P), dtype=torch.uint8, device=device)
            else:
                self.n_store = 0
        self.attn = torch.empty(n, self.attn_rows, P, dtype=torch.float32, device=device)
        # split-precision modes keep the materialised matrix in fp16 (half the bytes of the HBM-bound attn @ v that
        # every iteration repeats; measured effect on the final flow: 4e-6 px mean EPE); self.attn is then only the
        # logits scratch of the one-time softmax
        self.attn16 = (torch.empty(n, self.attn_rows, P, dtype=torch.float16, device=device)
                       if attn_f16 and P % 2 == 0 else None)       # whole matrix, or one row chunk (high resolution)
        spec = [("qk", 2 * HDIM), ("corr", COR_PLANES), ("flow", 2), ("hid", 960), ("xa", 640), ("xb", 640),
                ("cor256", 256), ("cat256", 256), ("f128", 128),
                ("concat", 640),                       # [nets | inps | mf | mf_global | mf_temporal]
                ("v128", 128), ("ln128", 128), ("qkv", 384), ("att128", 128), ("tx128", 128), ("h256", 256),
                ("delta", 2),                          # == [Bc][2*Pn][P]
                ("m256", 256), ("mask", 576), ("coords1", 2),
                ("hid2", 192), ("xa2", 128), ("xb2", 128),       # scratch of the flow branch (runs concurrently)
                ("part", 128 * 4),                               # split-K slabs of attn @ v: [4][n][128][P]
                ("splitws", 1024)]                                # scratch for sf_gemm's automatic split-K
        if self.corr_blocked:
            spec = [(nm, r) for nm, r in spec if nm != "corr"]          # no fp32 planes of the correlation features at all
        # tensors of the motion encoder that are ONLY input / output of SK blocks (cor = convc1's output, cat(cor, flo), convf1's
        # output) exist as fp16 k-octet planes alone in the fp16 hand-over modes: their producers write 2 instead of 6 bytes per
        # element, the next block takes operand AND residual from the k-octets (SfGemm.r_f16 = 2).  EPE-neutral
        self.koct_io = bool(koct_io and shadows)
        koct_only = {"cor256": 256, "cat256": 256, "f128": 128} if self.koct_io else {}
        spec = [(nm, r) for nm, r in spec if nm not in koct_only]
        ws = Workspace(sum(n * r * P + 64 for _, r in spec), device)
        self.ws = ws
        for name, r in spec:
            setattr(self, name, ws.take(n, r, P))
        if self.corr_blocked:
            self.corr = ops.new_shadow(Planes(ws.buf, 0, COR_PLANES * P, n, COR_PLANES, P), device)   # k-octet planes only
        for name, r in koct_only.items():
            setattr(self, name, ops.new_shadow(Planes(ws.buf, 0, r * P, n, r, P), device))
        # f16x2 mode: the SK blocks' INPUT tensors keep an fp16 k-octet copy next to the fp32 planes (ops.Planes.shadow):
        # the first GEMM of a block reads the copy by LDS-DMA, its residual epilogue and every other kernel the planes
        self.shadows = bool(shadows)
        if self.shadows:
            for name in (() if self.koct_io else ("cor256", "cat256")) + ("concat", "m256") + (() if self.corr_blocked else ("corr",)):
                setattr(self, name, replace(getattr(self, name), shadow=ops.new_shadow(getattr(self, name), device)))
        self.nets = self.concat.slice(0, 128)
        self.inps = self.concat.slice(128, 256)
        self.mf = self.concat.slice(256, 384)
        self.mfg = self.concat.slice(384, 512)
        self.mft = self.concat.slice(512, 640)
        # flow-head view of nets: '(B T) C H W -> B (T C) H W' without a copy
        self.nets_grouped = Planes(self.concat.base, self.concat.off, Pn * self.concat.img_stride, Bc, HDIM * Pn, P,
                                   group=HDIM, group_stride=self.concat.img_stride)
        if self.shadows:
            sh = self.concat.shadow
            self.nets_grouped = replace(self.nets_grouped, shadow=Planes(
                sh.base, sh.off, Pn * sh.img_stride, Bc, HDIM * Pn, P, f16=True, koct=True, group=HDIM,
                group_stride=sh.img_stride))
        self.delta_fh = Planes(self.delta.base, self.delta.off, 2 * Pn * P, Bc, 2 * Pn, P)
        self.up = torch.empty(n, 2, 8 * h, 8 * w, dtype=torch.float32, device=device)
        # static input staging (so a captured graph sees constant pointers)
        self.fmaps_in = None
        self.cnets_in = None
        self.graph = None
        self.graph_key = None
        self.graph_calls = 0


def _level_maps(self, l: int) -> torch.Tensor:
    """Level l of the row-major volumes as [pairs, clips * N, h_l, w_l] -- per pair the reference's corr_pyramid[l] (corr.py:13-21)
    without its singleton dimension -- a strided VIEW of the (possibly row-pitched, _Plan.corr_pitch) storage; a copy for blocked volumes."""
    hl, wl = self.h >> l, self.w >> l
    if self.lvls is None:                      # blocked volumes (image = clip * pairs + pair): a row-major COPY, tests only
        t = self.vol.levels()[l].view(self.Bc, self.Pn, self.P, hl, wl)
        return t.permute(1, 0, 2, 3, 4).reshape(self.Pn, self.Bc * self.P, hl, wl)
    pit = self.corr_pitch[l] if self.corr_pitch else wl
    return self.lvls[l].view(self.Pn, self.Bc * self.P, hl, pit)[..., :wl]


_Plan.level_maps = _level_maps


class HotPathEngine:
    """`forward(fmaps, cnets, iters)` == reference loop in test_mode (streamflow.py:110-147)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda:0", T: Optional[int] = None,
                 use_graph: bool = False, precision: Optional[str] = None, corr_dtype: str = "f32",
                 gma_mode: Optional[str] = None, flash_qk_products: Optional[int] = None,
                 single_layers: Optional[Sequence[str]] = None, options: Optional[EngineOptions] = None):
        """precision: 'f16x3' (split fp16, fp32-class accuracy, default), 'fp32' (exact fp32 MFMA), 'f16x2'
        (weights split, activations rounded to fp16: ~1e-4 px EPE, faster) or 'f16' (weights and activations fp16);
        None = the package-wide setting (streamflow_amd.ops.PRECISION).
        corr_dtype: 'f32' keeps the correlation pyramids in fp32 as the reference does (corr.py:13, arithmetic =
        `precision`); 'f16' stores them as fp16 and builds them with single f16 MFMA products (SF_PRECISION_F16).
        gma_mode: 'matrix' = attention matrix materialised once per clip (gma.py), 'flash' = fused recompute kernel every
        iteration (demo.py:235-258), 'stored' = the fused kernel's own softmax weights stored once per clip as fp16 in fragment
        order and streamed every iteration (bit-identical to 'flash'; n_img * Ppad^2 * 2 bytes, options.stored_p_max_gb),
        'auto' (default) = flash exactly when the matrix cannot be kept (high resolution).
        flash_qk_products: MFMA products per logit of the fused kernel (3 = split precision, 1 = fp16 q and k).
        options: scheduling / hand-over switches (EngineOptions; SF_ENGINE_OPTS overrides single fields for experiments).
        See streamflow_amd.presets for the named configurations."""
        _lib.load()
        self.options = EngineOptions.from_env(options)
        self.precision = ops.PRECISION if precision is None else ops._PRECISION_NAMES[precision]
        if corr_dtype not in ("f32", "f16"):
            raise RuntimeError(f"corr_dtype must be 'f32' or 'f16', got {corr_dtype!r}")
        self.corr_f16 = corr_dtype == "f16"
        # fp16 volumes in the blocked layout (+ k-octet-only hand-over of the correlation features) wherever the f16x2 / f16
        # hand-over formats are active; options.corr_blocked = False keeps the row-major fp16 volumes
        self.corr_blocked = self.corr_f16 and self.options.corr_blocked
        self.corr_blocked32 = (not self.corr_f16) and self.options.corr_blocked32
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HotPathEngine needs an MI355X device (cuda:N); there is no CPU fallback")
        # independent chains of one iteration (flow branch || corr branch of the motion encoder; temporal block ||
        # global aggregation) are enqueued on a second stream, so the tails of one chain's small kernels are
        # filled by the other; inside a captured graph these become parallel branches.
        self.parallel_branches = self.options.parallel_branches
        self.auto_split_k = self.options.auto_split_k
        self.attn_chunk_rows = int(self.options.attn_chunk_rows)            # > 0 forces the recompute path
        self.attn_k_splits = min(4, int(self.options.attn_k_splits))        # 1 = no split-K (<= 4)
        # GMA aggregation: 'matrix' = attention matrix materialised once (gma.py), chunked recompute when it cannot be
        # kept; 'flash' = fused recompute kernel every iteration (demo.py:235-258); 'auto' = flash exactly when the
        # matrix would have to be chunked (high resolution)
        self.gma_mode = gma_mode or "auto"
        if self.gma_mode not in ("auto", "matrix", "flash", "stored", "hybrid"):
            raise RuntimeError(f"gma_mode must be auto, matrix, flash, stored or hybrid, got {self.gma_mode!r}")
        # MFMA products per logit of the fused kernel: 3 = split precision (fp32-class), 2 / 1 = k / q and k in fp16
        self.flash_qk_products = (int(flash_qk_products or 0)
                                  or {ops.PRECISION_F16X2: 2, ops.PRECISION_F16: 1}.get(self.precision, 3))
        if self.flash_qk_products not in (1, 2, 3):
            raise RuntimeError(f"flash_qk_products must be 1, 2 or 3, got {self.flash_qk_products}")
        self._side = torch.cuda.Stream(device=self.device)
        if int(self.options.clock_probe_us) > 0:
            self._probe_stream = torch.cuda.Stream(device=self.device)
            self.clock_counts = torch.zeros(2, dtype=torch.int64, device=self.device)
        self._chain_streams = [torch.cuda.Stream(device=self.device) for _ in range(3)]
        # Where the main stream has no concurrent branch (corr encoder, motion-encoder tail, GRU + flow head) a batch of an
        # even number of clips is cut in two halves that run as two chains on two streams: a dependent kernel boundary costs
        # ~8 us on this part and each chain hides the other's.  options.split_solo = 0 / 2 / 4 chains.
        self.split_solo = int(self.options.split_solo)
        self.W = HotPathWeights(state_dict, self.device, T)
        # layers whose weights are used as ONE fp16 value in the f16x2 mode (see presets.py: chosen by measured EPE)
        self.single_layers = tuple(single_layers or ())
        self.W.set_single("all" if self.single_layers == ("all",) else self.single_layers)
        # the FFN pairs' weight streams are built NOW (device-synchronous), not lazily inside a forward: a forward enqueues on several
        # streams (and may be a graph capture) -- a stream built on one of them would be read by another before it exists
        if (self.options.ffn_pairs or self.options.sk_tail) and self.precision in (ops.PRECISION_F16X2, ops.PRECISION_F16):
            cxp = ops.Ctx(precision=self.precision)
            with torch.cuda.device(self.device):
                for b in HotPathWeights.SK_BLOCKS:
                    blk = getattr(self.W, b)
                    for pair in (blk.pair1, blk.pair2) if self.options.ffn_pairs else ():
                        if (pair.K1, pair.M2) in ops.PAIR_SHAPES[0] | ops.PAIR_SHAPES[1]:
                            pair.stream(*pair.products(cxp))
                    if self.options.sk_tail and blk.tail.built(blk.tail.products(cxp)):
                        blk.tail.stream(blk.tail.products(cxp))
                torch.cuda.synchronize(self.device)
        if self.options.mask_upsample and self.precision in (ops.PRECISION_F16X2, ops.PRECISION_F16) and self.W.mask_pack.built():
            with torch.cuda.device(self.device):
                self.W.mask_pack.stream(self.W.mask_pack.products(ops.Ctx(precision=self.precision)))
                torch.cuda.synchronize(self.device)
        if self.options.temporal_block and self.precision in (ops.PRECISION_F16X2, ops.PRECISION_F16) and self.W.temporal.built():
            pm = self.W.temporal.products(ops.Ctx(precision=self.precision))
            if pm is not None:
                with torch.cuda.device(self.device):
                    self.W.temporal.stream(pm)
                    torch.cuda.synchronize(self.device)
        self.use_graph = use_graph
        self._plans: Dict[Tuple[int, int, int, int], _Plan] = {}
        self.max_plans = int(self.options.max_plans)

    # ---------------------------------------------------------------------------------------------
    def plan(self, Bc: int, h: int, w: int, D: int) -> _Plan:
        """Buffers (and captured graph) for one input shape; at most `max_plans` shapes are kept, least recently used
        evicted first (a dataset with varying image sizes would otherwise pin several GB per distinct shape)."""
        key = (Bc, h, w, D)
        pl = self._plans.pop(key, None)
        if pl is None:
            while len(self._plans) >= max(1, self.max_plans):
                self._plans.pop(next(iter(self._plans)))            # dicts keep insertion order: first = oldest
            split = self.precision != ops.PRECISION_FP32
            if self.gma_mode in ("flash", "stored", "hybrid") and not split:
                raise RuntimeError(f"gma_mode={self.gma_mode!r} needs a split precision (f16x3 / f16x2); the exact fp32 mode keeps "
                                   "the materialised / chunked attention path")
            flash = {"auto": None, "matrix": False, "flash": True, "stored": True, "hybrid": True}[self.gma_mode]
            # high-resolution grids: the fused path keeps its softmax weights for the loop (options.stored_auto_px; same results)
            n_img = Bc * self.W.pairs
            # ... and so do split-precision logits at every grid (two / three MFMA products per logit make the recompute 2.5x the stream:
            # 1483 vs 608 us per iteration at the Sintel grid, fp32_class 168 -> 182 ff/s)
            auto_stored = (self.gma_mode == "flash" and int(self.options.stored_auto_px) > 0 and
                           (h * w >= int(self.options.stored_auto_px) or self.flash_qk_products >= 2)
                           and ops.gma_stored_p_bytes(n_img, h * w) <= (int(self.options.stored_auto_max_gb) << 30))
            pl = _Plan(self.W, Bc, h, w, D, self.device, self.attn_chunk_rows, attn_f16=split, corr_f16=self.corr_f16,
                       flash=flash, shadows=(self.options.shadows and (h * w) % 4 == 0 and
                                             self.precision in (ops.PRECISION_F16X2, ops.PRECISION_F16) and
                                             self.options.hidden_f16),
                       corr_blocked=self.corr_blocked, corr_blocked32=self.corr_blocked32,
                       koct_io=self.options.koct_io and self.options.hidden_koct,   # (k-octet-only blocks need k-octet producers)
                       stored_p=((self.gma_mode in ("stored", "hybrid") or auto_stored) and self.options.flash_stats),
                       stored_frac=(1.0 if (self.gma_mode == "stored" or auto_stored) else float(self.options.hybrid_store_pct) / 100.0),
                       stored_p_max_bytes=int(self.options.stored_p_max_gb) << 30)
            pl.auto_stored = bool(auto_stored and pl.pbuf is not None)
        self._plans[key] = pl                                        # (re)insert as most recent
        return pl

    # ---------------------------------------------------------------------------------------------
    def _ctx(self, pl: _Plan) -> ops.Ctx:
        """The launch context of one forward over plan `pl`: passed to every op (no process-wide launch state)."""
        o = self.options
        return ops.Ctx(precision=self.precision,
                       split_ws=pl.splitws.tensor().view(-1) if self.auto_split_k else None,
                       shadows=o.shadows, shadow_fused=o.shadow_fused, flash_stats=o.flash_stats,
                       hidden_f16=o.hidden_f16, hidden_koct=o.hidden_koct, split_handover=o.split_handover, pw_fold=o.pw_fold, x2_f16=o.x2_f16,
                       ffn_pairs=o.ffn_pairs, head_pairs=o.head_pairs, sk_tail=o.sk_tail, sk_tail_all=o.sk_tail_all)

    @staticmethod
    def _flash_ws_sub(pl: _Plan, i0: int, cnt: int) -> torch.Tensor:
        """The section of the fused GMA workspace that belongs to images i0 .. i0 + cnt - 1 (launched as a workspace of its own: legal
        when that launch has no key split, ops.gma_no_key_split)."""
        if i0 == 0 and cnt == pl.n:
            return pl.flash_ws                     # (the whole batch: with the key-split partial buffers of a small launch)
        ib = ops.gma_flash_img_bytes(pl.P)
        return pl.flash_ws[i0 * ib:(i0 + cnt) * ib]

    def _attention_rows(self, cx: ops.Ctx, pl: _Plan, i0: int, rows: int) -> None:
        """attn[:, :rows, :] = softmax(scale * q[:, i0:i0+rows]^T k)   (gma.py:53-65) for every image."""
        P, n = pl.P, pl.n
        ops.gemm_raw(cx, A=pl.qk.ptr + 4 * i0, B=pl.qk.ptr + 4 * HDIM * P, C=pl.attn.data_ptr(), M=rows, N=P, K=HDIM,
                     batch=n, lda=P, ldb=P, ldc=P, strideA=pl.qk.img_stride, strideB=pl.qk.img_stride,
                     strideC=pl.attn_rows * P, a_layout=LAYOUT_K_MAJOR, b_layout=LAYOUT_K_MAJOR,
                     alpha=float(HDIM) ** -0.5, epilogue=EPI_NONE)
        if rows == pl.attn_rows:
            ops.softmax_rows(pl.attn, n * rows, P, out16=pl.attn16)
        else:                                     # last, shorter chunk: rows of each image are not contiguous
            for img in range(n):
                ops.softmax_rows(pl.attn[img], rows, P, out16=None if pl.attn16 is None else pl.attn16[img])

    # ---------------------------------------------------------------------------------------------
    def _setup(self, cx: ops.Ctx, pl: _Plan, fmaps: torch.Tensor, cnets: torch.Tensor) -> None:
        """Everything before the iteration loop: volumes, context split, GMA attention matrix."""
        W = self.W
        Bc, Pn, h, w, P, n, D = pl.Bc, pl.Pn, pl.h, pl.w, pl.P, pl.n, pl.D
        T = Pn + 1
        # The context chain (split -> to_qk -> GMA pack + softmax statistics / attention matrix) reads only `cnets` and the volume
        # build only `fmaps`: two independent chains, the context chain on the side stream (options.setup_overlap)
        main = torch.cuda.current_stream()
        side = self._side if (self.parallel_branches and self.options.setup_overlap) else main
        cs = cx.no_split() if side is not main else cx
        if side is not main:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            # streamflow.py:119-122: nets = tanh(.), inps = relu(.)
            ops.context_split(cnets, pl.nets, pl.inps, HDIM)
            ops.refresh_shadow(pl.nets, cs)
            ops.refresh_shadow(pl.inps, cs)
            # a6: attn = softmax(scale * q k^T) over the context features (gma.py:53-65), computed once
            ops.gemm(W.to_qk, pl.inps, pl.qk, EPI_NONE, cx=cs)
            if pl.flash:
                # q, k are constant over the loop: packed once, and the softmax statistics of every query with them
                ops.gma_flash_pack_qk(pl.qk, pl.flash_ws, float(HDIM) ** -0.5, stats_qk_products=self.flash_qk_products, cx=cs)
                if pl.pbuf is not None:             # ... and the softmax weights themselves (gma.py:53-65: `attn`), kept for the loop
                    ops.gma_flash_store_p(self._flash_ws_sub(pl, 0, pl.n_store), pl.pbuf, pl.n_store, P, self.flash_qk_products, cx=cs)
            elif pl.attn_rows == P:
                self._attention_rows(cs, pl, 0, P)
        # a1+a2: all pairs, one launch.  pair t = (frame t, frame t+1)
        if pl.corr_blocked or pl.corr_blocked32:
            ops.corr_build_blocked(fmaps.data_ptr(), fmaps.data_ptr() + 4 * D * P, T * D * P, D * P, pl.vol, Bc, Pn, D,
                                   ws=pl.corr_ws)
        else:
            ops.corr_build(fmaps.data_ptr(), fmaps.data_ptr() + 4 * D * P, T * D * P, D * P, pl.lvls, pl.lvl_pair_stride,
                           Bc, Pn, D, h, w, ws=pl.corr_ws, cx=cx, pitch=pl.corr_pitch)
        if side is not main:
            main.wait_stream(side)

    def _iteration(self, cx: ops.Ctx, pl: _Plan, with_mask: bool, fuse_mask: bool = False) -> None:
        W = self.W
        Bc, Pn, h, w, P, n = pl.Bc, pl.Pn, pl.h, pl.w, pl.P, pl.n
        sk = lambda Wt, X, Y, fg=False: run_skblock(Wt, X, Y, pl.hid, pl.xa, pl.xb, h, w, fg, cx=cx)
        main = torch.cuda.current_stream()
        side = self._side if self.parallel_branches else main
        # the automatic split-K scratch is ONE buffer: only the main stream may use it while two streams run concurrently
        cs = cx.no_split() if side is not main else cx

        def fork():
            if side is not main:
                side.wait_stream(main)

        @contextlib.contextmanager
        def on_side():
            with torch.cuda.stream(side):
                yield

        def join():
            if side is not main:
                main.wait_stream(side)

        # a9: motion encoder (update.py:329-339).  flow branch (convf1 -> convf2) on the side stream ...
        fork()
        with on_side():
            ops.gemm(W.convf1, pl.flow, pl.f128, EPI_NONE, cx=cs)
            run_skblock(W.convf2, pl.f128, pl.cat256.slice(192, 256), pl.hid2, pl.xa2, pl.xb2, h, w, cx=cs)
        # ... while the main stream does a3 (correlation lookup for all pairs, streamflow.py:132) and the corr branch
        if pl.corr_blocked:
            ops.corr_lookup_blocked(pl.vol, pl.coords1, None, pl.corr, Bc, Pn)
        elif pl.corr_blocked32:
            ops.corr_lookup_blocked(pl.vol, pl.coords1, replace(pl.corr, shadow=None), None, Bc, Pn)
            ops.refresh_shadow(pl.corr, cx)
        else:
            ops.corr_lookup(pl.lvls, pl.lvl_pair_stride, pl.coords1, pl.corr, Bc, Pn, h, w, cx=cx, pitch=pl.corr_pitch)
        # image ranges of the chains: whole clips when the clip count divides, else (a single clip, an odd batch) two
        # ranges of images -- every block but the flow head works image by image
        nch = self.split_solo if (self.split_solo in (2, 4) and side is not main) else 1
        if nch > 1 and Bc % nch == 0:
            parts, by_clip = [(c * (n // nch), n // nch) for c in range(nch)], True
        elif nch > 1 and n >= 2 and self.options.split_uneven:
            parts, by_clip = [(0, (n + 1) // 2), ((n + 1) // 2, n // 2)], False
        else:
            parts, by_clip = [(0, n)], True
        split = len(parts) > 1

        # One scratch buffer: no automatic split-K while the chains run (context `cc`).  auto_splits() only fires for grids of
        # fewer than 96 workgroups with K >= 256 (gemm_split.hip) -- never at the Sintel / KITTI / Spring shapes, whose
        # smallest GEMM has 165 workgroups per clip -- so the two-chain schedule changes results (summation order) only at toy
        # shapes; the bitwise test of the two schedules therefore runs with auto_split_k = False.
        cc = cx.no_split()

        def two_chains(fn):
            """fn(image0, count): chain 0 on the main stream, the others on their own streams."""
            for c in range(1, len(parts)):
                sc = self._chain_streams[c - 1]
                sc.wait_stream(main)
                with torch.cuda.stream(sc):
                    fn(*parts[c])
            fn(*parts[0])
            for c in range(1, len(parts)):
                main.wait_stream(self._chain_streams[c - 1])

        if split:
            def corr_chain(i0, cnt):
                hid, xa, xb = _part(pl.hid, i0, cnt), _part(pl.xa, i0, cnt), _part(pl.xb, i0, cnt)
                run_skblock(W.convc1, _sub(pl.corr, i0, cnt), _sub(pl.cor256, i0, cnt), hid, xa, xb, h, w, True, cx=cc)
                run_skblock(W.convc2, _sub(pl.cor256, i0, cnt), _sub(pl.cat256.slice(0, 192), i0, cnt), hid, xa, xb, h, w, cx=cc)
            two_chains(corr_chain)
        else:
            sk(W.convc1, pl.corr, pl.cor256, True)                 # cor = gelu(convc1(corr))
            sk(W.convc2, pl.cor256, pl.cat256.slice(0, 192))
        join()
        if split:                                                   # mf = cat(out, flow); flow rows kept by flow_update
            two_chains(lambda i0, cnt: run_skblock(W.conv, _sub(pl.cat256, i0, cnt), _sub(pl.mf.slice(0, HDIM - 2), i0, cnt),
                                                   _part(pl.hid, i0, cnt), _part(pl.xa, i0, cnt), _part(pl.xb, i0, cnt), h, w,
                                                   cx=cc))
        else:
            sk(W.conv, pl.cat256, pl.mf.slice(0, HDIM - 2))
        # a10: temporal transformer block over the T-1 tokens of each pixel (update.py:481-484,770), side stream
        fork()
        with on_side():
            if self.options.temporal_block and ops.temporal_block_ok(W.temporal, pl.mf, Pn, cs):
                # the whole block in one launch from the k-octet copy of the motion features (csrc/temporal.hip)
                ops.temporal_block(W.temporal, pl.mf, pl.mft, Pn, (W.ln1_w, W.ln1_b), (W.ln2_w, W.ln2_b), cx=cs)
            else:
                # LayerNorm / attention outputs have ONE reader, a GEMM: in the f16x2 mode they leave as its k-octet operand
                ko = lambda buf, M: (lambda t: t if (t.koct and not t.split) else buf)(_handover(cs, buf, pl.n, HDIM, P, consumer_rows=M))
                ln, att = ko(pl.ln128, W.qkv.M), ko(pl.att128, W.proj.M)
                ops.layernorm_cm(pl.mf, W.ln1_w, W.ln1_b, ln)
                # qkv has ONE reader, the attention core: fp16 rows where the fp16 hand-over is active (EPE-neutral)
                qkv = _scratch(pl.qkv, pl.n, 3 * HDIM, f16=True) if (hidden_f16_ok(cs, P) and cs.x2_f16) else pl.qkv
                ops.gemm(W.qkv, ln, qkv, EPI_NONE, cx=cs)
                ops.temporal_attn(qkv, att, Bc, Pn, HDIM)
                ops.gemm(W.proj, att, pl.tx128, EPI_RES, R=pl.mf, cx=cs)
                ln = ko(pl.ln128, W.fc1.M)
                ops.layernorm_cm(pl.tx128, W.ln2_w, W.ln2_b, ln)
                h256 = _handover(cs, pl.h256, pl.n, 256, P, consumer_rows=HDIM)        # fc1 -> fc2 only
                ops.gemm(W.fc1, ln, h256, EPI_GELU, cx=cs)
                ops.gemm(W.fc2, h256, pl.mft, EPI_RES, R=pl.tx128, cx=cs)
        # a7: global aggregation  mfg = mf + gamma * attn @ to_v(mf)   (gma.py:91-104), main stream
        # v has ONE reader, the pack of the fused kernel (which rounds it to fp16): fp16 rows where the fp16 hand-over is active
        v128 = _scratch(pl.v128, pl.n, HDIM, f16=True) if (pl.flash and hidden_f16_ok(cx, P) and cx.x2_f16) else pl.v128
        # fused recompute path with fp16 activations: to_v and the v pack are ONE launch from mf's k-octet copy (options.project_v)
        project = pl.flash and self.options.project_v and ops.gma_flash_project_ok(W.to_v, pl.mf, cx)
        # options.gma_per_chain: one fused launch per half-batch chain (each a workspace sub-range without the key-split form)
        gma_in_chains = bool(self.options.gma_per_chain and pl.flash and pl.pbuf is None and project and split and
                             all(ops.gma_no_key_split(cnt, P) for _, cnt in parts))
        if project:
            ops.gma_flash_project_v(pl.flash_ws, W.to_v, pl.mf, cx=cx)
            v128 = None
        else:
            ops.gemm(W.to_v, pl.mf, v128, EPI_NONE, cx=cx)
        ks = self.attn_k_splits if self.precision != ops.PRECISION_FP32 else 1
        attn_ptr, attn_lay = ((pl.attn16.data_ptr(), LAYOUT_F16_K_MINOR) if pl.attn16 is not None
                              else (pl.attn.data_ptr(), LAYOUT_K_MINOR))
        if pl.flash and pl.pbuf is not None and pl.n_store == n:
            # the stored weights streamed past v (gma.py:99-102): half the matrix-core work of the recompute, HBM-bound
            ops.gma_stored_aggregate(pl.flash_ws, pl.pbuf, v128, pl.mf, W.gamma, pl.mfg, cx=cx)
        elif pl.flash and pl.pbuf is not None:
            # hybrid: images [0, n_store) through the stored weights on a chain stream (HBM-bound), the rest through the recompute
            # kernel on the main stream (power-bound): two launches that want different resources, side by side
            ns = pl.n_store
            assert project and v128 is None, "hybrid GMA needs the fused v projection (fp16-activation presets)"
            sc = self._chain_streams[0] if side is not main else main
            if sc is not main:
                sc.wait_stream(main)
            with torch.cuda.stream(sc):
                ops.gma_stored_aggregate(self._flash_ws_sub(pl, 0, ns), pl.pbuf, None, _sub(pl.mf, 0, ns), W.gamma, _sub(pl.mfg, 0, ns), cx=cs)
            ops.gma_flash_aggregate(self._flash_ws_sub(pl, ns, n - ns), None, _sub(pl.mf, ns, n - ns), W.gamma, _sub(pl.mfg, ns, n - ns),
                                    self.flash_qk_products, use_stats=True, cx=cx)
            if sc is not main:
                main.wait_stream(sc)
        elif pl.flash and gma_in_chains:
            pass                                                    # launched per chain below, in front of each chain's GRU
        elif pl.flash:
            # fused recompute (K6' of SURVEY.md): one kernel, online softmax, logits never written
            ops.gma_flash_aggregate(pl.flash_ws, v128, pl.mf, W.gamma, pl.mfg, self.flash_qk_products, use_stats=True, cx=cx)
        elif pl.attn_rows < P:
            # high-resolution path: recompute the attention rows chunk by chunk (K6' of SURVEY.md)
            for i0 in range(0, P, pl.attn_rows):
                rows = min(pl.attn_rows, P - i0)
                self._attention_rows(cx, pl, i0, rows)
                ops.gemm_raw(cx, A=pl.v128.ptr, B=attn_ptr, C=pl.mfg.ptr + 4 * i0, R=pl.mf.ptr + 4 * i0,
                             gamma=W.gamma.data_ptr(), M=HDIM, N=rows, K=P, batch=n, lda=P, ldb=P, ldc=P, ldr=P,
                             strideA=pl.v128.img_stride, strideB=pl.attn_rows * P, strideC=pl.mfg.img_stride,
                             strideR=pl.mf.img_stride, a_layout=LAYOUT_K_MINOR, b_layout=attn_lay, alpha=1.0,
                             epilogue=EPI_AXPY)
        elif ks > 1 and P % 4 == 0:
            # attn @ v streams the N x N matrix (HBM-bound) but has only N/128 * images workgroups: split K so that
            # enough bytes are in flight; partial products go to slabs, combined with the AXPY of gma.py:102
            ops.gemm_raw(cx, A=pl.v128.ptr, B=attn_ptr, C=pl.part.ptr, M=HDIM, N=P, K=P, batch=n, lda=P, ldb=P,
                         ldc=P, strideA=pl.v128.img_stride, strideB=P * P, strideC=HDIM * P,       # slabs: [split][img][128][P]
                         a_layout=LAYOUT_K_MINOR, b_layout=attn_lay, alpha=1.0, epilogue=EPI_NONE,
                         k_splits=ks, split_stride=n * HDIM * P)
            ops.splitk_combine(pl.part.base[pl.part.off:], n * HDIM * P, ks, HDIM * P, pl.mf, W.gamma, pl.mfg)
        else:
            ops.gemm_raw(cx, A=pl.v128.ptr, B=attn_ptr, C=pl.mfg.ptr, R=pl.mf.ptr, gamma=W.gamma.data_ptr(),
                         M=HDIM, N=P, K=P, batch=n, lda=P, ldb=P, ldc=P, ldr=P, strideA=pl.v128.img_stride,
                         strideB=P * P, strideC=pl.mfg.img_stride, strideR=pl.mf.img_stride,
                         a_layout=LAYOUT_K_MINOR, b_layout=attn_lay, alpha=1.0, epilogue=EPI_AXPY)
        if not pl.flash:
            ops.refresh_shadow(pl.mfg, cx)                          # (the flash kernel writes the k-octet copy itself)
        if not gma_in_chains:
            join()
        # "gru": SKBlock(640 -> 128) over cat[nets, inps, mf, mfg, mft]; new nets overwrite the nets slice
        if split:
            def gru_chain(i0, cnt):
                hid, xa, xb = _part(pl.hid, i0, cnt), _part(pl.xa, i0, cnt), _part(pl.xb, i0, cnt)
                if gma_in_chains:
                    ops.gma_flash_aggregate(self._flash_ws_sub(pl, i0, cnt), None, _sub(pl.mf, i0, cnt), W.gamma, _sub(pl.mfg, i0, cnt),
                                            self.flash_qk_products, use_stats=True, cx=cc)
                    if side is not main:                            # the temporal block's output (side stream) is a GRU input
                        torch.cuda.current_stream().wait_stream(side)
                run_skblock(W.gru, _sub(pl.concat, i0, cnt), _sub(pl.nets, i0, cnt), hid, xa, xb, h, w, cx=cc)
                if by_clip:     # flow head sees all T-1 hidden states of a clip jointly (update.py:774): clips i0 / Pn ..
                    run_skblock(W.flow_head, _sub(pl.nets_grouped, i0 // Pn, cnt // Pn),
                                _sub(pl.delta_fh, i0 // Pn, cnt // Pn), hid, xa, xb, h, w, cx=cc)
            two_chains(gru_chain)
            if not by_clip:
                sk(W.flow_head, pl.nets_grouped, pl.delta_fh)
        else:
            sk(W.gru, pl.concat, pl.nets)
            # flow head sees all T-1 hidden states of a clip jointly (update.py:774)
            sk(W.flow_head, pl.nets_grouped, pl.delta_fh)
        if with_mask:                                               # update.py:756-759,777
            ops.gemm(W.mask0, pl.nets, pl.m256, EPI_RELU, hw=(h, w), cx=cx)
            if not fuse_mask:                                       # (fused: mask.2 runs inside sf_mask_upsample, after the loop)
                ops.gemm(W.mask2, pl.m256, pl.mask, EPI_NONE, alpha=0.25, cx=cx)
        # streamflow.py:138 + :133 for the next iteration
        ops.flow_update(pl.coords1, pl.delta, pl.flow, pl.mf.slice(HDIM - 2, HDIM, unshadowed=True), n, h, w,
                        koct=pl.mf.shadow, koct_row=HDIM - 2, cx=cx)   # (+ the flow rows of mf's k-octet copy)

    def _run(self, pl: _Plan, fmaps: torch.Tensor, cnets: torch.Tensor, iters: int, all_masks: bool) -> None:
        cx = self._ctx(pl)
        probe = int(self.options.clock_probe_us)
        if probe > 0:                                   # the clock under THIS forward: a branch of the launch sequence (and of its graph)
            main = torch.cuda.current_stream()
            self._probe_stream.wait_stream(main)
            with torch.cuda.stream(self._probe_stream):
                ops.clock_probe(self.clock_counts, probe)
        self._setup(cx, pl, fmaps, cnets)
        # only the last iteration's mask is used (test mode): its second layer + the convex upsampling are one launch then
        fuse = (self.options.mask_upsample and not all_masks and iters > 0 and ops.mask_upsample_ok(self.W.mask_pack, pl.m256, cx))
        for it in range(iters):
            self._iteration(cx, pl, with_mask=all_masks or it == iters - 1, fuse_mask=fuse)
        if probe > 0:
            torch.cuda.current_stream().wait_stream(self._probe_stream)
        flow_t = pl.flow.tensor().view(pl.n, 2, pl.h, pl.w)
        if fuse:
            ops.mask_upsample(self.W.mask_pack, pl.m256, flow_t, pl.up, pl.h, pl.w, cx=cx)
            return
        mask_t = pl.mask.tensor().view(pl.n, 576, pl.h, pl.w)
        _lib.check(_lib.load().sf_upsample_flow(flow_t.data_ptr(), mask_t.data_ptr(), pl.up.data_ptr(), pl.n, pl.h,
                                                pl.w, _lib.stream()), "sf_upsample_flow")

    # ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, fmaps: torch.Tensor, cnets: torch.Tensor, iters: int = 15,
                flow_init: Optional[Sequence[torch.Tensor]] = None, all_masks: bool = False
                ) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        """fmaps [B,T,D,h,w], cnets [B,T-1,256,h,w] fp32 on the GPU.
        Returns (flows_up, flows_lowres): T-1 tensors [B,2,8h,8w] and [B,2,h,w] (views into engine buffers,
        valid until the next forward of the same shape)."""
        with torch.cuda.device(self.device):      # streams, launches and graph capture all belong to self.device
            pl = self._begin(fmaps, cnets, flow_init)
            Bc, T, h, w = pl.Bc, pl.Pn + 1, pl.h, pl.w
            if self.use_graph:
                self._forward_graph(pl, fmaps, cnets, iters, all_masks)
            else:
                self._run(pl, fmaps, cnets, iters, all_masks)
            up = pl.up.view(Bc, T - 1, 2, 8 * h, 8 * w)
            low = pl.flow.tensor().view(Bc, T - 1, 2, h, w)
            return [up[:, i] for i in range(T - 1)], [low[:, i] for i in range(T - 1)]

    def _begin(self, fmaps: torch.Tensor, cnets: torch.Tensor, flow_init) -> _Plan:
        """Shared entry checks of both forward flavours + loop-state initialisation (streamflow.py:111-115)."""
        ops._dev_check(fmaps)
        ops._dev_check(cnets)
        if fmaps.device != self.device or cnets.device != self.device:
            raise RuntimeError(f"inputs live on {fmaps.device} / {cnets.device}, engine was built for {self.device}")
        Bc, T, D, h, w = fmaps.shape
        if T - 1 != self.W.pairs or tuple(cnets.shape) != (Bc, T - 1, 2 * HDIM, h, w):
            raise RuntimeError(f"shape mismatch: fmaps {tuple(fmaps.shape)}, cnets {tuple(cnets.shape)}, "
                               f"weights built for T={self.W.pairs + 1}")
        pl = self.plan(Bc, h, w, D)
        n = pl.n
        coords1 = pl.coords1.tensor().view(n, 2, h, w)
        _lib.check(_lib.load().sf_coords_grid(coords1.data_ptr(), n, h, w, _lib.stream()), "sf_coords_grid")
        if flow_init is not None:                          # streamflow.py:114-115
            if len(flow_init) != T - 1:
                raise RuntimeError(f"flow_init needs {T - 1} tensors, got {len(flow_init)}")
            for i, f in enumerate(flow_init):
                coords1.view(Bc, T - 1, 2, h, w)[:, i] += f.to(coords1)
        ops.flow_update(pl.coords1, None, pl.flow, pl.mf.slice(HDIM - 2, HDIM, unshadowed=True), n, h, w,
                        koct=pl.mf.shadow, koct_row=HDIM - 2, cx=self._ctx(pl))
        return pl

    @torch.no_grad()
    def forward_all_iterations(self, fmaps: torch.Tensor, cnets: torch.Tensor, iters: int = 12,
                               flow_init: Optional[Sequence[torch.Tensor]] = None) -> List[List[torch.Tensor]]:
        """Training-mode return of the reference (streamflow.py:139-149): for every pair, the list of upsampled
        predictions after each iteration (mask head + convex upsampling run every iteration; no graph)."""
        with torch.cuda.device(self.device):
            pl = self._begin(fmaps, cnets, flow_init)
            Bc, T, h, w, n = pl.Bc, pl.Pn + 1, pl.h, pl.w, pl.n
            preds: List[List[torch.Tensor]] = [[] for _ in range(T - 1)]
            cx = self._ctx(pl)
            self._setup(cx, pl, fmaps, cnets)
            for _ in range(iters):
                self._iteration(cx, pl, with_mask=True)
                up = ops.upsample_flow(pl.flow.tensor().view(n, 2, h, w), pl.mask.tensor().view(n, 576, h, w))
                up = up.view(Bc, T - 1, 2, 8 * h, 8 * w)
                for i in range(T - 1):
                    preds[i].append(up[:, i])
            return preds

    def _forward_graph(self, pl: _Plan, fmaps, cnets, iters, all_masks) -> None:
        key = (iters, all_masks)
        if pl.fmaps_in is None:
            pl.fmaps_in = torch.empty_like(fmaps)
            pl.cnets_in = torch.empty_like(cnets)
        pl.fmaps_in.copy_(fmaps)
        pl.cnets_in.copy_(cnets)
        # (the schedule reads these MUTABLE attributes -- bench.py / tests assign eng.parallel_branches etc. after construction:
        # the values in force at capture time are part of the key)
        key = key + (self.precision, self.options, self.gma_mode, self.flash_qk_products, self.corr_f16, self.single_layers,
                     self.parallel_branches, self.split_solo, self.auto_split_k, self.attn_chunk_rows, self.attn_k_splits)
        if pl.graph is None or pl.graph_key != key:
            # warm-up outside capture, then capture the whole clip as one graph.  The loop state (coords1, flow)
            # is re-initialised by the caller before every replay, so the graph itself is stateless.
            state = (pl.coords1.tensor().clone(), pl.flow.tensor().clone(), pl.mf.tensor().clone())
            s = torch.cuda.Stream(device=self.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._run(pl, pl.fmaps_in, pl.cnets_in, 1, all_masks)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            calls0 = _lib.CALLS
            with torch.cuda.graph(g, stream=s):
                self._run(pl, pl.fmaps_in, pl.cnets_in, iters, all_masks)
            pl.graph, pl.graph_key = g, key
            pl.graph_calls = _lib.CALLS - calls0           # C-ABI launch calls captured (>= kernel nodes / fused setup calls): a diagnostic
            pl.coords1.tensor().copy_(state[0])
            pl.flow.tensor().copy_(state[1])
            pl.mf.tensor().copy_(state[2])
            ops.refresh_shadow(pl.mf.slice(HDIM - 8, HDIM), self._ctx(pl))       # ... and the flow rows of mf's k-octet copy
        pl.graph.replay()
