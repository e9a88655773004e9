"""sf_temporal_block (csrc/temporal.hip): the temporal transformer block of the update step -- core/update.py:459-484,502-513 (timm
Block: pre-LN 1-head attention over the T - 1 frames of a pixel + pre-LN MLP, both residual) -- as ONE launch, through the C ABI
(-m gpu): against float64 on the same fp16-rounded tokens, against the CPU oracle's restatement, and against the seven launches it
replaces; 1 .. 3 tokens per pixel, one and two products, ragged pixel counts; every output cell written; run-to-run identical."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
C, H = 128, 256


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _weff(A, single):
    hi = A.hi.float().permute(1, 0, 2).reshape(A.lda_h, -1)[: A.M, : A.K].double().cpu()
    lo = A.lo.float().permute(1, 0, 2).reshape(A.lda_h, -1)[: A.M, : A.K].double().cpu()
    return (hi if single else hi + lo) / A.split_scale


def _block(seed, dev, pm):
    from streamflow_amd.ops import PackedLinear, PackedTemporal
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    p = {"qkv": r(3 * C, C) / C ** 0.5 * 1.5, "proj": r(C, C) / C ** 0.5, "proj_b": r(C) * 0.2, "fc1": r(H, C) / C ** 0.5, "fc1_b": r(H) * 0.2,
         "fc2": r(C, H) / H ** 0.5, "fc2_b": r(C) * 0.2, "ln1_w": 1 + 0.3 * r(C), "ln1_b": 0.2 * r(C), "ln2_w": 1 + 0.3 * r(C), "ln2_b": 0.2 * r(C)}
    L = [PackedLinear(p["qkv"].view(3 * C, C, 1, 1), None, dev), PackedLinear(p["proj"].view(C, C, 1, 1), p["proj_b"], dev),
         PackedLinear(p["fc1"].view(H, C, 1, 1), p["fc1_b"], dev), PackedLinear(p["fc2"].view(C, H, 1, 1), p["fc2_b"], dev)]
    for l in L:
        l.single = pm == 1
    return p, L, PackedTemporal(*L), g


def _ref64(x16, p, L, pm, TT):
    """float64 over the values the kernel sees: tokens x16 [B, TT, C, P] (fp16-rounded), effective (hi [+ lo]) weights; the
    intermediate fp16 roundings of the kernel (LN output, q / k, attention output, LN2 output, GELU output) are NOT modelled: the
    tolerance below is theirs."""
    Wq, Wp, W1, W2 = (_weff(l, pm == 1) for l in L)
    d = lambda k: p[k].double()
    x = x16.double().permute(0, 3, 1, 2)                       # [B, P, TT, C]
    h = F.layer_norm(x, (C,), d("ln1_w"), d("ln1_b"), 1e-5)
    qkv = h @ Wq.t()
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    a = torch.softmax((q * C ** -0.5) @ k.transpose(-1, -2), dim=-1)
    x = x + (a @ v) @ Wp.t() + d("proj_b")
    h = F.layer_norm(x, (C,), d("ln2_w"), d("ln2_b"), 1e-5)
    x = x + F.gelu(h @ W1.t() + d("fc1_b")) @ W2.t() + d("fc2_b")
    return x.permute(0, 2, 3, 1)                               # [B, TT, C, P]


def _planes_with_copy(x, dev, ops):
    """fp32 planes [n, C, P] + their k-octet fp16 copy (Planes.shadow), as the engine holds the motion features."""
    from dataclasses import replace
    from streamflow_amd.ops import Planes
    n, K, P = x.shape
    X = Planes.of(x.to(dev).contiguous())
    sh = Planes(torch.zeros(n * K * P // 2 + 8, device=dev), 0, K * P, n, K, P, f16=True, koct=True)
    ops.pack_koct(X, sh)
    return replace(X, shadow=sh)


def _run(dev, ops, pack, x, TT, p, nan_fill=True):
    from dataclasses import replace
    from streamflow_amd.ops import Planes
    n, _, P = x.shape
    X = _planes_with_copy(x, dev, ops)
    y32 = torch.full((n, C, P), float("nan") if nan_fill else 0.0, device=dev)
    ysh = Planes(torch.full((n * C * P // 2 + 8,), float("nan") if nan_fill else 0.0, device=dev), 0, C * P, n, C, P, f16=True, koct=True)
    Y = replace(Planes.of(y32), shadow=ysh)
    cx = ops.Ctx(precision=ops.PRECISION_F16X2)
    assert ops.temporal_block_ok(pack, X, TT, cx)
    ln1 = (p["ln1_w"].to(dev).contiguous(), p["ln1_b"].to(dev).contiguous())
    ln2 = (p["ln2_w"].to(dev).contiguous(), p["ln2_b"].to(dev).contiguous())
    ops.temporal_block(pack, X, Y, TT, ln1, ln2, cx=cx)
    torch.cuda.synchronize()
    return y32, ysh, X, (ln1, ln2), cx


@pytest.mark.parametrize("pm", [1, 2])
@pytest.mark.parametrize("P", [37, 64, 1000, 7040])
@pytest.mark.parametrize("TT", [1, 2, 3])
def test_temporal_block_vs_float64(dev, TT, P, pm):
    from streamflow_amd import ops
    B = 2 if P >= 7040 else 3
    p, L, pack, g = _block(100 * TT + P + pm, dev, pm)
    x = torch.randn(B * TT, C, P, generator=g) * 1.5
    y32, ysh, X, _, _ = _run(dev, ops, pack, x, TT, p)
    ref = _ref64(x.half().view(B, TT, C, P), p, L, pm, TT).reshape(B * TT, C, P)
    got = y32.double().cpu()
    assert bool(torch.isfinite(got).all()), "a cell of the fp32 output was not written"
    scale = max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    # five fp16 hand-overs inside the block (2^-11 relative each, through O(1)-gain layers)
    assert err <= 4e-3 * scale, (TT, P, pm, err, scale)
    rms = ((got - ref) ** 2).mean().sqrt().item()
    assert rms <= 6e-4 * scale, (TT, P, pm, rms)
    # the k-octet copy is the fp16 rounding of the fp32 result, every cell written
    k16 = ysh.tensor().float().cpu()
    assert torch.equal(k16, y32.half().float().cpu())


@pytest.mark.parametrize("pm", [1, 2])
@pytest.mark.parametrize("TT", [1, 3])
def test_temporal_block_vs_the_seven_launches(dev, TT, pm):
    """The unfused form the engine ran before (sf_layernorm_cm, sf_gemm, sf_temporal_attn_f16in, sf_gemm, sf_layernorm_cm, sf_gemm,
    sf_gemm with the same hand-over formats) on the same operands: the two agree to the fp16 hand-over noise."""
    from dataclasses import replace
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    B, P = 2, 3000
    n = B * TT
    p, L, pack, g = _block(7 * TT + pm, dev, pm)
    x = torch.randn(n, C, P, generator=g) * 1.5
    y32, _, X, (ln1, ln2), cx = _run(dev, ops, pack, x, TT, p)
    qkvA, projA, fc1A, fc2A = L
    ko = lambda rows: Planes(torch.zeros(n * rows * P // 2 + 8, device=dev), 0, rows * P, n, rows, P, f16=True, koct=True)
    ln, att, h256 = ko(C), ko(C), ko(H)
    ops.layernorm_cm(replace(X, shadow=None), ln1[0], ln1[1], ln)
    qkv = Planes(torch.zeros(n * 3 * C * P // 2 + 8, device=dev), 0, 3 * C * P, n, 3 * C, P, f16=True)
    ops.gemm(qkvA, ln, qkv, ops.EPI_NONE, cx=cx)
    ops.temporal_attn(qkv, att, B, TT, C)
    tx = Planes.of(torch.empty(n, C, P, device=dev))
    ops.gemm(projA, att, tx, ops.EPI_RES, R=replace(X, shadow=None), cx=cx)
    ops.layernorm_cm(tx, ln2[0], ln2[1], ln)
    ops.gemm(fc1A, ln, h256, ops.EPI_GELU, cx=cx)
    out = Planes.of(torch.empty(n, C, P, device=dev))
    ops.gemm(fc2A, h256, out, ops.EPI_RES, R=tx, cx=cx)
    torch.cuda.synchronize()
    a, b = y32.double().cpu(), out.tensor().double().cpu()
    scale = max(1.0, b.abs().max().item())
    # (the fused kernel takes LayerNorm 1 and the residual from the fp16 copy of the tokens; the unfused path from the fp32 planes)
    assert (a - b).abs().max().item() <= 5e-3 * scale
    assert ((a - b) ** 2).mean().sqrt().item() <= 8e-4 * scale


def test_temporal_block_vs_oracle_restatement(dev):
    """oracle.temporal_block (the restatement pinned against the reference's TemporalLayer2 golden) on the same parameters."""
    from streamflow_amd import ops
    from oracle import streamflow_oracle as orc
    TT, B, P, pm = 3, 2, 640, 2
    p, L, pack, g = _block(5, dev, pm)
    x = torch.randn(B * TT, C, P, generator=g)
    y32, _, _, _, _ = _run(dev, ops, pack, x, TT, p)
    pre = "tb"
    params = {pre + ".norm1.weight": p["ln1_w"], pre + ".norm1.bias": p["ln1_b"], pre + ".norm2.weight": p["ln2_w"], pre + ".norm2.bias": p["ln2_b"],
              pre + ".attn.qkv.weight": p["qkv"], pre + ".attn.proj.weight": p["proj"], pre + ".attn.proj.bias": p["proj_b"],
              pre + ".mlp.fc1.weight": p["fc1"], pre + ".mlp.fc1.bias": p["fc1_b"], pre + ".mlp.fc2.weight": p["fc2"], pre + ".mlp.fc2.bias": p["fc2_b"]}
    tok = x.view(B, TT, C, P).permute(0, 3, 1, 2).reshape(B * P, TT, C)
    ref = orc.temporal_block(tok, params, pre).reshape(B, P, TT, C).permute(0, 2, 3, 1).reshape(B * TT, C, P)
    err = (y32.cpu() - ref).abs().max().item()
    assert err <= 5e-3 * max(1.0, ref.abs().max().item()), err


def test_temporal_block_is_deterministic(dev):
    """Twenty launches beside a competing stream: bit-identical (the weight ring is refilled by DMA behind a barrier)."""
    from streamflow_amd import ops
    TT, B, P, pm = 3, 8, 7040, 2
    p, L, pack, g = _block(9, dev, pm)
    x = torch.randn(B * TT, C, P, generator=g)
    first = None
    side = torch.cuda.Stream(device=dev)
    junk = torch.randn(4096, 4096, device=dev)
    for i in range(20):
        with torch.cuda.stream(side):
            for _ in range(3):
                junk = torch.tanh(junk) * 1.0001
        y32, ysh, _, _, _ = _run(dev, ops, pack, x, TT, p, nan_fill=(i % 2 == 0))
        cur = (y32.clone(), ysh.tensor().clone())
        if first is None:
            first = cur
        else:
            assert torch.equal(first[0], cur[0]) and torch.equal(first[1], cur[1]), i
    torch.cuda.synchronize()


def test_temporal_block_rejects_what_it_was_not_built_for(dev):
    from streamflow_amd import ops, _lib
    import ctypes as Ct
    p, L, pack, g = _block(1, dev, 2)
    L[2].single = True                                   # mixed product counts: the caller keeps the seven launches
    x = _planes_with_copy(torch.randn(4, C, 64, generator=g), dev, ops)
    cx = ops.Ctx(precision=ops.PRECISION_F16X2)
    assert pack.products(cx) is None and not ops.temporal_block_ok(pack, x, 2, cx)
    assert not ops.temporal_block_ok(pack, x, 4, cx)     # (not even with one product count: 4 tokens per pixel)
    g_ = _lib.SfTemporalBlock()
    assert _lib.load().sf_temporal_block(Ct.byref(g_), None) != 0
    assert _lib.load().sf_temporal_block_frags(3) == 0 and _lib.load().sf_temporal_block_frags(2) == 512


def test_engine_with_and_without_the_fused_temporal_block_vs_oracle(dev):
    """the one-launch temporal block takes its residual and LayerNorm input from the fp16 k-octet copy of the motion
    features instead of the fp32 planes -- isolated accuracy evidence at the engine level: the same clip through the engine with
    EngineOptions.temporal_block on and off (15 iterations, config2_fp16 so that nothing else differs), each against the CPU
    oracle; the fused arm must stay within the class bound AND within 1.5x of the unfused arm (+ a floor for the noise between
    two equivalent fp16 launch sequences)."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import EngineOptions, HotPathEngine
    T, B, h, w, iters = 4, 1, 24, 32, 15
    params = syn.make_params(9, T)
    fmaps, cnets = syn.make_features(79, B, T, h, w)
    ups_o, _ = orc.hotpath_forward(fmaps, cnets, params, iters)
    kw = presets.engine_kwargs("config2_fp16")
    epe = {}
    for on in (True, False):
        eng = HotPathEngine(params, device=dev, T=T, options=EngineOptions(temporal_block=on), **kw)
        ups, _ = eng.forward(fmaps.to(dev), cnets.to(dev), iters=iters)
        epe[on] = max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))
    print(f"temporal block fused / unfused: EPE vs oracle {epe[True]:.3e} / {epe[False]:.3e} px (15 iterations, 24 x 32 grid)")
    assert epe[True] <= 1e-3 and epe[False] <= 1e-3
    assert epe[True] <= 1.5 * epe[False] + 5e-5, epe
