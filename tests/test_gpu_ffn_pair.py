"""sf_ffn_pair (csrc/ffn_pair.hip): an SK block's FFN -- conv1x1 -> GELU -> conv1x1 (+ the block's epilogue), core/update.py:14-16,
30-36 -- as ONE launch, through the C ABI (-m gpu): against float64 on the same fp16-rounded operands and against the two-launch
sf_gemm path it replaces, for every built shape, both modes, the three product combinations, ragged pixel counts."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

FFN1 = [(128, 192), (256, 384), (324, 486), (384, 576)]          # (C, 1.5 C): mode 1, M2 = C
FFN2 = [(128, 192, 64), (256, 384, 192), (256, 384, 126), (324, 486, 256), (384, 576, 6), (256, 384, 4), (128, 192, 2)]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _koct(x, dev, ops):
    from streamflow_amd.ops import Planes
    n, K, P = x.shape
    Ka = (K + 7) // 8 * 8
    Y = Planes(torch.zeros(n * Ka * P // 2 + 8, device=dev), 0, Ka * P, n, K, P, f16=True, koct=True)
    ops.pack_koct(Planes.of(x.to(dev).contiguous()), Y)
    return Y


def _weff(A, single):
    hi = A.hi.float().permute(1, 0, 2).reshape(A.lda_h, -1)[: A.M, : A.K].double().cpu()
    lo = A.lo.float().permute(1, 0, 2).reshape(A.lda_h, -1)[: A.M, : A.K].double().cpu()
    return (hi if single else hi + lo) / A.split_scale


def _layers(K1, H, M2, seed, dev, pm):
    from streamflow_amd.ops import PackedLinear, PackedPair
    g = torch.Generator().manual_seed(seed)
    W1, b1 = torch.randn(H, K1, generator=g) / K1 ** 0.5, torch.randn(H, generator=g) * 0.2
    W2, b2 = torch.randn(M2, H, generator=g) / H ** 0.5 * 1.7, torch.randn(M2, generator=g) * 0.2
    A1, A2 = PackedLinear(W1.view(H, K1, 1, 1), b1, dev), PackedLinear(W2.view(M2, H, 1, 1), b2, dev)
    A1.single, A2.single = pm[0] == 1, pm[1] == 1
    return A1, A2, PackedPair(A1, A2), b1, b2, g


@pytest.mark.parametrize("pm", [(1, 1), (2, 1), (2, 2)])
@pytest.mark.parametrize("P", [64, 1000, 7040])
@pytest.mark.parametrize("shape", FFN2)
def test_ffn2_pair_vs_float64_and_two_launches(dev, shape, P, pm):
    from dataclasses import replace
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    K1, H, M2 = shape
    n = 3 if P < 7040 else 2
    A1, A2, pair, b1, b2, g = _layers(K1, H, M2, K1 * 7 + M2 + P, dev, pm)
    x = torch.randn(n, K1, P, generator=g)
    X = _koct(x, dev, ops)
    cx = ops.Ctx(precision=ops.PRECISION_F16X2)
    gelu_out = M2 == 256
    assert ops.ffn_pair_ok(pair, X, 0, cx)
    # float64 on the values the kernels see: fp16 input, fp16-rounded hidden
    x16 = x.half().double()
    hid = F.gelu(torch.einsum("hk,nkp->nhp", _weff(A1, pm[0] == 1), x16) + b1.double()[None, :, None]).half().double()
    ref = torch.einsum("mh,nhp->nmp", _weff(A2, pm[1] == 1), hid) + b2.double()[None, :, None]
    if gelu_out:
        ref = F.gelu(ref)
    scale = max(1.0, ref.abs().max().item())
    Mo = (M2 + 7) // 8 * 8
    if M2 % 8 == 0:
        # k-octet-only output (NaN-filled: every cell must be written)
        Y = Planes(torch.full((n * Mo * P // 2 + 8,), float("nan"), device=dev), 0, Mo * P, n, M2, P, f16=True, koct=True)
        ops.ffn_pair(pair, X, Y, 0, gelu_out=gelu_out, cx=cx)
        got = Y.tensor().double().cpu()
        err = (got - ref).abs()
        # one fp16 rounding of the result + the hidden's rounding carried through layer 2 (~2^-11 relative of O(1) sums)
        assert bool((err <= 2.0 ** -10 * ref.abs() + 2e-3 * scale).all()), (shape, P, pm, err.max().item())
    # fp32 planes + k-octet copy (rows >= M2 of the last octet must be left alone)
    y32 = torch.full((n, M2, P), float("nan"), device=dev)
    Y = Planes.of(y32)
    sh = ops.new_shadow(Y, dev)
    sh.base.view(torch.float16).fill_(7.0)
    ops.ffn_pair(pair, X, replace(Y, shadow=sh), 0, gelu_out=gelu_out, cx=cx)
    torch.cuda.synchronize()
    err32 = (y32.double().cpu() - ref).abs().max().item()
    print(f"ffn2 pair {shape} P={P} pm={pm}: max abs err vs float64 = {err32:.2e} (scale {scale:.1f})")
    assert err32 < 2e-3 * scale, (shape, P, pm, err32)
    assert torch.equal(sh.tensor().float(), y32.half().float())
    if M2 % 8:
        oc = sh.base.view(torch.float16)[: n * Mo * P].view(n, Mo // 8, P, 8)[:, -1, :, M2 % 8:]
        assert bool((oc == 7.0).all())
    # the two-launch path: same arithmetic, the hidden handed over as fp16 k-octets
    hid_p = Planes(torch.zeros(n * ((H + 7) // 8 * 8) * P // 2 + 8, device=dev), 0, (H + 7) // 8 * 8 * P, n, H, P, f16=True, koct=True)
    ops.gemm(A1, X, hid_p, ops.EPI_GELU, cx=cx)
    y2 = torch.full((n, M2, P), float("nan"), device=dev)
    ops.gemm(A2, hid_p, Planes.of(y2), ops.EPI_GELU if gelu_out else ops.EPI_NONE, cx=cx)
    torch.cuda.synchronize()
    d = (y2 - y32).abs().max().item()
    print(f"   vs two launches: {d:.2e}")
    assert d < 2.5e-3 * scale, (shape, P, pm, d)          # (a hidden value on an fp16 rounding boundary may round the other way)


@pytest.mark.parametrize("pm", [(1, 1), (2, 2)])
@pytest.mark.parametrize("P", [64, 1000, 7040])
@pytest.mark.parametrize("shape", FFN1)
def test_ffn1_pair_vs_float64_and_two_launches(dev, shape, P, pm):
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    C, H = shape
    n = 3 if P < 7040 else 2
    A1, A2, pair, b1, b2, g = _layers(C, H, C, C * 5 + P, dev, pm)
    x = torch.randn(n, C, P, generator=g)
    dw_w, dw_b = torch.randn(C, generator=g) * 0.5, torch.randn(C, generator=g) * 0.1
    X = _koct(x, dev, ops)
    cx = ops.Ctx(precision=ops.PRECISION_F16X2)
    assert ops.ffn_pair_ok(pair, X, 1, cx)
    x16 = x.half().double()
    hid = F.gelu(torch.einsum("hk,nkp->nhp", _weff(A1, pm[0] == 1), x16) + b1.double()[None, :, None]).half().double()
    y = torch.einsum("mh,nhp->nmp", _weff(A2, pm[1] == 1), hid) + b2.double()[None, :, None]
    x1 = F.gelu(x16 + y)
    ref = F.gelu(x1 + (dw_w.double()[None, :, None] * x1 + dw_b.double()[None, :, None]))
    out = torch.full((n * C * P // 2 + 8,), float("nan"), device=dev)
    Y = Planes(out, 0, C * P, n, C, P, f16=True)
    ops.ffn_pair(pair, X, Y, 1, dw_w=dw_w.to(dev), dw_b=dw_b.to(dev), cx=cx)
    torch.cuda.synchronize()
    got = Y.tensor().double().cpu()
    err = (got - ref).abs()
    print(f"ffn1 pair {shape} P={P} pm={pm}: max abs err vs float64 = {err.max().item():.2e}")
    assert bool(torch.isfinite(got).all())
    assert bool((err <= 2.0 ** -10 * ref.abs() + 3e-3).all()), (shape, P, pm, err.max().item())
    # the two-launch path (residual from the k-octet operand, fp16 rows out)
    hid_p = Planes(torch.zeros(n * ((H + 7) // 8 * 8) * P // 2 + 8, device=dev), 0, (H + 7) // 8 * 8 * P, n, H, P, f16=True, koct=True)
    ops.gemm(A1, X, hid_p, ops.EPI_GELU, cx=cx)
    out2 = torch.full((n * C * P // 2 + 8,), float("nan"), device=dev)
    Y2 = Planes(out2, 0, C * P, n, C, P, f16=True)
    ops.gemm(A2, hid_p, Y2, ops.EPI_RES_GELU_DW1, R=X, dw_w=dw_w.to(dev), dw_b=dw_b.to(dev), cx=cx)
    torch.cuda.synchronize()
    d = (Y2.tensor().float() - Y.tensor().float()).abs().max().item()
    print(f"   vs two launches: {d:.2e}")
    # (same arithmetic, another summation order: one fp16 ulp of the stored result)
    assert d <= 2.0 ** -9 * max(1.0, ref.abs().max().item()), (shape, P, pm, d)


def test_ffn_pair_refuses_unbuilt_shapes(dev):
    from streamflow_amd import ops
    from streamflow_amd.ops import PackedLinear, PackedPair
    A1 = PackedLinear(torch.randn(960, 640, 1, 1), None, dev)
    A2 = PackedLinear(torch.randn(128, 960, 1, 1), None, dev)
    x = _koct(torch.randn(1, 640, 64), dev, ops)
    assert not ops.ffn_pair_ok(PackedPair(A1, A2), x, 0, ops.Ctx(precision=ops.PRECISION_F16X2))
    assert not ops.ffn_pair_ok(None, x, 0, ops.Ctx(precision=ops.PRECISION_F16X2))


@pytest.mark.parametrize("mode,shape", [(1, (256, 384, 256)), (0, (256, 384, 192)), (1, (324, 486, 324)), (0, (324, 486, 256))])
def test_ffn_pair_is_deterministic(dev, mode, shape):
    """Twenty launches on the same operands, bit-identical results (the weight ring's waits and barriers: a stage read before its
    DMA pieces landed, or overwritten while another wave still reads it, shows up as run-to-run differences)."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    K1, H, M2 = shape
    n, P = 24, 7040
    A1, A2, pair, b1, b2, g = _layers(K1, H, M2, 99, dev, (2, 2))
    X = _koct(torch.randn(n, K1, P, generator=g), dev, ops)
    cx = ops.Ctx(precision=ops.PRECISION_F16X2)
    dw_w, dw_b = torch.randn(M2, generator=g).to(dev) * 0.5, torch.randn(M2, generator=g).to(dev) * 0.1
    outs = []
    side = torch.cuda.Stream(device=dev)
    for rep in range(20):
        if mode == 1:
            Y = Planes(torch.zeros(n * M2 * P // 2 + 8, device=dev), 0, M2 * P, n, M2, P, f16=True)
        else:
            Y = Planes(torch.zeros(n * M2 * P // 2 + 8, device=dev), 0, M2 * P, n, M2, P, f16=True, koct=True)
        if rep % 2:                                  # (a competing launch on another stream: uneven load)
            with torch.cuda.stream(side):
                torch.randn(1 << 22, device=dev).sum()
        ops.ffn_pair(pair, X, Y, mode, dw_w=dw_w if mode else None, dw_b=dw_b if mode else None, cx=cx)
        outs.append(Y.base.clone())
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int32), outs[0].view(torch.int32))


@pytest.mark.parametrize("qkp", [1, 3])
def test_gma_flash_is_deterministic(dev, qkp):
    """The fused GMA kernel's single V stage is refilled right behind a barrier (csrc/attn.hip): twenty launches, bit-identical."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    n, P = 24, 7040
    g = torch.Generator().manual_seed(5)
    qk = torch.randn(n, 256, P, generator=g).to(dev)
    v = torch.randn(n, 128, P, generator=g).to(dev)
    mf = torch.randn(n, 128, P, generator=g).to(dev)
    gamma = torch.tensor([0.5], device=dev)
    cx = ops.Ctx(precision=ops.PRECISION_F16X2)
    ws = torch.empty(ops.gma_flash_ws_bytes(n, P), dtype=torch.uint8, device=dev)
    ops.gma_flash_pack_qk(Planes.of(qk), ws, 128 ** -0.5, stats_qk_products=qkp, cx=cx)
    outs = []
    side = torch.cuda.Stream(device=dev)
    for rep in range(20):
        out = torch.zeros(n, 128, P, device=dev)
        if rep % 2:
            with torch.cuda.stream(side):
                torch.randn(1 << 22, device=dev).sum()
        ops.gma_flash_aggregate(ws, Planes.of(v), Planes.of(mf), gamma, Planes.of(out), qkp, use_stats=True, cx=cx)
        outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("pm", [(1, 1), (2, 2)])
@pytest.mark.parametrize("frames", [1, 2, 3])
def test_flow_head_pairs_on_the_grouped_view(dev, frames, pm):
    """The flow head's input is the '(B T) C -> B (T C)' view of the hidden state (update.py:775: 128-row slices of T - 1 consecutive
    images): sf_ffn_pair addresses it in place (x_group = 128).  Both pairs of the block against the same launches on a gathered,
    plain copy of that view (bit-identical: same fragments, same order)."""
    from dataclasses import replace
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    B, P, Cg = 3, 1000, 128
    C, H, M2 = Cg * frames, Cg * frames * 3 // 2, 2 * frames
    g = torch.Generator().manual_seed(frames * 10 + pm[0])
    # a wider parent buffer (like the engine's 640-row concat): the hidden state in rows 0 .. 127 of every image
    rows = 640
    parent = torch.randn(B * frames, rows, P, generator=g)
    par16 = _koct(parent, dev, ops)                                            # k-octet copy of the whole parent
    img16 = par16.img_stride
    Xg = Planes(par16.base, par16.off, frames * img16, B, C, P, f16=True, koct=True, group=Cg, group_stride=img16)
    plain = parent[:, :Cg].reshape(B, frames * Cg, P).contiguous()
    Xp = _koct(plain, dev, ops)
    cx = ops.Ctx(precision=ops.PRECISION_F16X2)
    # ffn1 pair (mode 1)
    A1, A2, pair, b1, b2, g2 = _layers(C, H, C, 3 + frames, dev, pm)
    dw_w, dw_b = (torch.randn(C, generator=g2) * 0.5).to(dev), (torch.randn(C, generator=g2) * 0.1).to(dev)
    assert ops.ffn_pair_ok(pair, Xg, 1, cx) and ops.ffn_pair_ok(pair, Xp, 1, cx)
    outs = []
    for X in (Xg, Xp):
        Y = Planes(torch.full((B * C * P // 2 + 8,), float("nan"), device=dev), 0, C * P, B, C, P, f16=True)
        ops.ffn_pair(pair, X, Y, 1, dw_w=dw_w, dw_b=dw_b, cx=cx)
        outs.append(Y.tensor().clone())
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])
    # the engine's form: fp32 planes (grouped the same way) with the k-octet copy as Planes.shadow -- the operand is the copy, the
    # RESIDUAL the fp32 value (SfFfnPair.R32): float64 with exactly that split
    par32 = parent.to(dev).contiguous()
    X32 = Planes(par32.view(-1), 0, frames * rows * P, B, C, P, group=Cg, group_stride=rows * P, shadow=Xg)
    assert ops.ffn_pair_ok(pair, X32, 1, cx)
    Y = Planes(torch.full((B * C * P // 2 + 8,), float("nan"), device=dev), 0, C * P, B, C, P, f16=True)
    ops.ffn_pair(pair, X32, Y, 1, dw_w=dw_w, dw_b=dw_b, cx=cx)
    torch.cuda.synchronize()
    x16, x32 = plain.half().double(), plain.double()
    hid = F.gelu(torch.einsum("hk,nkp->nhp", _weff(A1, pm[0] == 1), x16) + b1.double()[None, :, None]).half().double()
    yv = torch.einsum("mh,nhp->nmp", _weff(A2, pm[1] == 1), hid) + b2.double()[None, :, None]
    x1 = F.gelu(x32 + yv)
    ref1 = F.gelu(x1 + (dw_w.double().cpu()[None, :, None] * x1 + dw_b.double().cpu()[None, :, None]))
    got = Y.tensor().double().cpu()
    assert bool(torch.isfinite(got).all())
    err = (got - ref1).abs()
    assert bool((err <= 2.0 ** -10 * ref1.abs() + 3e-3).all()), (frames, pm, err.max().item())
    # (and it differs from the fp16-residual result where the rounding of x matters: the option is live)
    assert not torch.equal(Y.tensor(), outs[0])
    # ffn2 pair (mode 0, 2 (T - 1) output rows, fp32 planes)
    A1, A2, pair, b1, b2, g2 = _layers(C, H, M2, 5 + frames, dev, pm)
    assert ops.ffn_pair_ok(pair, Xg, 0, cx)
    outs = []
    for X in (Xg, Xp):
        y = torch.full((B, M2, P), float("nan"), device=dev)
        ops.ffn_pair(pair, X, Planes.of(y), 0, cx=cx)
        outs.append(y.clone())
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])
    x16 = plain.half().double()
    hid = F.gelu(torch.einsum("hk,nkp->nhp", _weff(A1, pm[0] == 1), x16) + b1.double()[None, :, None]).half().double()
    ref = torch.einsum("mh,nhp->nmp", _weff(A2, pm[1] == 1), hid) + b2.double()[None, :, None]
    assert (outs[0].double().cpu() - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())
