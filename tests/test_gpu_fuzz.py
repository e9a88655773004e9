"""Randomised shape sweep of the fused GEMM, the implicit 3x3 convolution and the correlation build / lookup through
the C ABI against float64 torch references (-m gpu).  Seeds are fixed; shapes are drawn to hit the kernels' tails:
M, N, K that are not multiples of the 128 / 128 / 32 tiles, single rows, ragged widths, batches."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _gelu(x):
    return F.gelu(x)          # exact (erf) form


def _ref_epilogue(epi, v, R, dw_w, dw_b, ops):
    if epi == ops.EPI_GELU:
        return _gelu(v)
    if epi == ops.EPI_RELU:
        return torch.relu(v)
    if epi == ops.EPI_RES:
        return R + v
    if epi == ops.EPI_RES_GELU:
        return _gelu(R + v)
    if epi == ops.EPI_RES_GELU_DW1:
        t = _gelu(R + v)
        return _gelu(t + (dw_w[None, :, None] * t + dw_b[None, :, None]))
    return v


@pytest.mark.parametrize("seed", range(16))
@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
def test_gemm_random_shapes(dev, seed, prec):
    from streamflow_amd import ops
    from streamflow_amd.ops import PackedLinear, Planes
    rng = np.random.default_rng(1000 + seed)
    M = int(rng.choice([1, 6, 31, 64, 126, 128, 129, 200, 324, 385]))
    K = int(rng.choice([2, 7, 32, 33, 100, 128, 192, 324, 800]))
    P = int(rng.choice([1, 17, 127, 128, 129, 300, 1000]))
    n = int(rng.integers(1, 4))
    epi = int(rng.choice([ops.EPI_NONE, ops.EPI_GELU, ops.EPI_RELU, ops.EPI_RES, ops.EPI_RES_GELU, ops.EPI_RES_GELU_DW1]))
    alpha = float(rng.choice([1.0, 0.25]))
    g = torch.Generator().manual_seed(seed)
    Wt = torch.randn(M, K, generator=g) / max(K, 1) ** 0.5
    bias = torch.randn(M, generator=g) * 0.1 if rng.random() < 0.8 else None
    X = torch.randn(n, K, P, generator=g)
    R = torch.randn(n, M, P, generator=g)
    dw_w, dw_b = torch.randn(M, generator=g) * 0.5, torch.randn(M, generator=g) * 0.1
    prev = ops.set_precision(prec)
    try:
        A = PackedLinear(Wt, bias, dev)
        Y = torch.full((n, M, P), float("nan"), device=dev)
        ops.gemm(A, Planes.of(X.to(dev)), Planes.of(Y), epi, R=Planes.of(R.to(dev)), dw_w=dw_w.to(dev),
                 dw_b=dw_b.to(dev), alpha=alpha)
        torch.cuda.synchronize()
    finally:
        ops.set_precision(prev)
    v = alpha * (torch.einsum("mk,zkp->zmp", Wt.double(), X.double()) + (bias.double()[None, :, None] if bias is not None else 0.0))
    ref = _ref_epilogue(epi, v, R.double(), dw_w.double(), dw_b.double(), ops)
    err = (Y.double().cpu() - ref).abs().max().item()
    tol = 3e-5 if prec == "f16x3" else 2e-5
    assert err < tol, (M, K, P, n, epi, prec, err)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
def test_conv3x3_random_shapes(dev, seed, prec):
    """Mask-head style implicit 3x3 convolution (update.py:756-759) on ragged images."""
    from streamflow_amd import ops
    from streamflow_amd.ops import PackedLinear, Planes
    rng = np.random.default_rng(2000 + seed)
    cin = int(rng.choice([32, 64, 128]))
    cout = int(rng.choice([5, 64, 130, 256]))
    h, w = int(rng.integers(3, 20)), int(rng.integers(3, 40))
    n = int(rng.integers(1, 3))
    g = torch.Generator().manual_seed(seed)
    Wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    X = torch.randn(n, cin, h, w, generator=g)
    prev = ops.set_precision(prec)
    try:
        A = PackedLinear(Wt, bias, dev, conv3x3=True)
        Y = torch.full((n, cout, h * w), float("nan"), device=dev)
        ops.gemm(A, Planes.of(X.to(dev)), Planes.of(Y), ops.EPI_RELU, hw=(h, w))
        torch.cuda.synchronize()
    finally:
        ops.set_precision(prev)
    ref = torch.relu(F.conv2d(X.double(), Wt.double(), bias.double(), padding=1)).reshape(n, cout, h * w)
    err = (Y.double().cpu() - ref).abs().max().item()
    assert err < 3e-5, (cin, cout, h, w, n, prec, err)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
def test_corr_random_shapes(dev, seed, prec):
    """CorrBlock (core/corr.py) on ragged grids against the CPU oracle: pyramid levels and a lookup."""
    from oracle import streamflow_oracle as orc
    import streamflow_amd as sfa
    from streamflow_amd import ops
    rng = np.random.default_rng(3000 + seed)
    B = int(rng.integers(1, 3))
    D = int(rng.choice([16, 40, 256]))
    h, w = int(rng.integers(16, 40)), int(rng.integers(16, 50))
    g = torch.Generator().manual_seed(seed)
    f1, f2 = torch.randn(B, D, h, w, generator=g), torch.randn(B, D, h, w, generator=g)
    coords = orc.coords_grid(B, h, w) + torch.randn(B, 2, h, w, generator=g) * 4.0
    prev = ops.set_precision(prec)
    try:
        blk = sfa.CorrBlock(f1.to(dev), f2.to(dev), num_levels=4, radius=4)
        out = blk(coords.to(dev)).cpu()
        lv = [t.cpu() for t in blk.corr_pyramid]
    finally:
        ops.set_precision(prev)
    pyr = orc.corr_pyramid(f1, f2, 4)
    for a, b_ in zip(lv, pyr):
        assert a.shape == b_.shape and (a - b_).abs().max().item() < 3e-5
    assert (out - orc.corr_lookup(pyr, coords, 4)).abs().max().item() < 5e-5


@pytest.mark.parametrize("seed", range(6))
def test_corr_fp16_volume_random_shapes(dev, seed):
    """fp16 correlation volumes (SF_PRECISION_F16: single f16 products, fp16 cells; BASELINE configs 2/5) on ragged
    grids.  Build: against the oracle pyramid of the fp16-ROUNDED features (what the kernel multiplies), tolerance =
    one fp16 rounding of the stored cell (2^-11 relative) + fp32 accumulation noise.  Lookup: against the oracle
    lookup run on the GPU's own stored cells, so only the fp32 tap blending differs."""
    from oracle import streamflow_oracle as orc
    import streamflow_amd as sfa
    rng = np.random.default_rng(4000 + seed)
    B = int(rng.integers(1, 3))
    D = int(rng.choice([16, 40, 256]))
    h, w = int(rng.integers(16, 40)), int(rng.integers(16, 50))
    g = torch.Generator().manual_seed(100 + seed)
    f1, f2 = torch.randn(B, D, h, w, generator=g), torch.randn(B, D, h, w, generator=g)
    coords = orc.coords_grid(B, h, w) + torch.randn(B, 2, h, w, generator=g) * 4.0
    coords[:, :, 0, 0] = torch.tensor([-6.0, 2.0])
    coords[:, :, 1, 1] = torch.tensor([float(w + 9), float(h + 9)])
    coords[:, :, 2, 2] = torch.tensor([3.0, 4.0])
    blk = sfa.CorrBlock(f1.to(dev), f2.to(dev), num_levels=4, radius=4, dtype=torch.float16)
    out = blk(coords.to(dev)).cpu()
    lv = [t.cpu() for t in blk.corr_pyramid]
    assert all(t.dtype == torch.float16 for t in lv)
    pyr = orc.corr_pyramid(f1.half().float(), f2.half().float(), 4)
    for l, (a, b_) in enumerate(zip(lv, pyr)):
        assert a.shape == b_.shape
        err = (a.float() - b_).abs()
        tol = 2.0 ** -11 * b_.abs() + 3e-5
        assert (err <= tol).all(), (l, (err - tol).max().item())
    ref = orc.corr_lookup([t.float() for t in lv], coords, 4)
    assert out.dtype == torch.float32 and (out - ref).abs().max().item() < 5e-5
    # and the fp16 path stays within fp16-storage distance of the exact fp32 volume
    exact = orc.corr_pyramid(f1, f2, 4)[0]
    assert (lv[0].float() - exact).abs().max().item() < 2e-3 * max(1.0, exact.abs().max().item())


@pytest.mark.parametrize("wscale", [1e-3, 1.0, 1e3])
def test_split_gemm_weight_magnitudes(dev, wscale):
    """the fp16 (hi, lo) split loses bits when the lo part falls into fp16 subnormals (small weights) and
    saturates above 65504.  Weights are therefore scaled by a per-tensor power of two before splitting (undone exactly in
    the epilogue): the split GEMM must keep its ~2^-20 relative accuracy for layers of tiny and of huge weights."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes, PackedLinear
    g = torch.Generator().manual_seed(11)
    M, K, P, n = 200, 160, 300, 2
    Wt = torch.randn(M, K, generator=g) * wscale / K ** 0.5
    bias = torch.randn(M, generator=g) * wscale * 0.1
    X = torch.randn(n, K, P, generator=g)
    A = PackedLinear(Wt.view(M, K, 1, 1), bias, dev)
    Y = torch.full((n, M, P), float("nan"), device=dev)
    prev = ops.set_precision("f16x3")
    try:
        ops.gemm(A, Planes.of(X.to(dev)), Planes.of(Y), ops.EPI_NONE, alpha=0.5)
        torch.cuda.synchronize()
    finally:
        ops.set_precision(prev)
    ref = 0.5 * (torch.einsum("mk,nkp->nmp", Wt.double(), X.double()) + bias.double()[None, :, None])
    err = (Y.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"weights x {wscale:g}: split scale {A.split_scale:g}, split error {A.split_error:.1e}, max rel err {err:.2e}")
    assert err < 3e-6, (wscale, err)


@pytest.mark.parametrize("seed", range(8))
def test_fp16_hidden_handover_is_bit_identical(dev, seed):
    """In the f16x2 mode a B operand is rounded to fp16 when it is staged, so a producer that already stores the rounded
    value (SfGemm.c_f16 -> SF_LAYOUT_F16_K_MAJOR in the consumer) must give BIT-IDENTICAL results to the fp32 hand-over,
    for ragged M / K (partial k-tiles, partial row tiles) and every tile configuration."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes, PackedLinear
    rng = np.random.default_rng(7000 + seed)
    C = int(rng.choice([128, 324, 640]))
    H = int(rng.choice([192, 486, 960]))
    Cout = int(rng.choice([6, 64, 126, 256, 640]))
    P = int(rng.choice([96, 1000, 7040]))                      # multiples of 4
    n = 2
    g = torch.Generator().manual_seed(seed)
    W1 = PackedLinear(torch.randn(H, C, 1, 1, generator=g) / C ** 0.5, torch.randn(H, generator=g) * 0.1, dev)
    W2 = PackedLinear(torch.randn(Cout, H, 1, 1, generator=g) / H ** 0.5, torch.randn(Cout, generator=g) * 0.1, dev)
    X = Planes.of(torch.randn(n, C, P, generator=g).to(dev))
    prev = ops.set_precision("f16x2")
    try:
        hid32 = Planes.of(torch.empty(n, H, P, device=dev))
        y32 = torch.full((n, Cout, P), float("nan"), device=dev)
        ops.gemm(W1, X, hid32, ops.EPI_GELU)
        ops.gemm(W2, hid32, Planes.of(y32), ops.EPI_NONE)
        store = torch.empty(n, H, P, device=dev)               # fp32 allocation reused as fp16 planes
        hid16 = Planes(store.view(-1), 0, H * P, n, H, P, f16=True)
        y16 = torch.full((n, Cout, P), float("nan"), device=dev)
        ops.gemm(W1, X, hid16, ops.EPI_GELU)
        ops.gemm(W2, hid16, Planes.of(y16), ops.EPI_NONE)
        # k-octet planes: the producer writes the consumer's LDS image, the consumer DMAs it (128-row tile only)
        yko = None
        if ops.uses_dma_tile(Cout):
            Ha = (H + 7) // 8 * 8
            store2 = torch.zeros(n, Ha, P, device=dev)
            hidko = Planes(store2.view(-1), 0, Ha * P, n, H, P, f16=True, koct=True)
            yko = torch.full((n, Cout, P), float("nan"), device=dev)
            ops.gemm(W1, X, hidko, ops.EPI_GELU)
            ops.gemm(W2, hidko, Planes.of(yko), ops.EPI_NONE)
        torch.cuda.synchronize()
    finally:
        ops.set_precision(prev)
    assert torch.equal(hid16.tensor().float(), hid32.tensor().half().float())
    assert torch.equal(y16, y32), (C, H, Cout, P, (y16 - y32).abs().max().item())
    if yko is not None:
        # the k-octet epilogue evaluates GELU as a polynomial (<= 5.2e-5 absolute; sf_common.h gelu_poly2) before the same
        # fp16 rounding: the hidden tensor agrees to that plus one rounding, the consumer's result to the propagated bound
        h32 = hid32.tensor()
        assert bool(((hidko.tensor().float() - h32).abs() <= 6e-5 + 2.0 ** -10 * h32.abs()).all())
        assert (yko - y32).abs().max().item() <= 2e-3 * max(1.0, y32.abs().max().item()), (C, H, Cout, P)
        # with a GELU-free producer the hand-over is bit-identical again
        ops.set_precision("f16x2")
        try:
            ops.gemm(W1, X, hid32, ops.EPI_NONE)
            ops.gemm(W2, hid32, Planes.of(y32), ops.EPI_NONE)
            ops.gemm(W1, X, hidko, ops.EPI_NONE)
            ops.gemm(W2, hidko, Planes.of(yko), ops.EPI_NONE)
            torch.cuda.synchronize()
        finally:
            ops.set_precision(prev)
        # (round 4: the k-octet producer is the activation-stationary kernel, the fp32 one a tiled kernel: the same products in
        # the same k order, but the bias enters the accumulator first instead of last -- fp32 results one ulp apart, so their
        # fp16 roundings differ by one fp16 ulp on the rare value that sat on a rounding boundary)
        hk, hr = hidko.tensor().float(), hid32.tensor().half().float()
        # (MFMA accumulation truncates relative to |accumulator|: with the bias inside it, sums that cancel to ~0 carry an
        # absolute error of ~1e-6 instead of ~1e-7)
        assert bool(((hk - hr).abs() <= 2.0 ** -10 * hr.abs() + 2e-6).all()) and (hk != hr).float().mean().item() < 1e-2
        assert (yko - y32).abs().max().item() <= 2e-4 * max(1.0, y32.abs().max().item()), (C, H, Cout, P)


@pytest.mark.parametrize("seed", range(6))
def test_dual_output_koct_copy(dev, seed):
    """SfGemm.c_f16 = 3 (f16x2 mode): the fp32 result is unchanged and its k-octet fp16 copy (Planes.shadow) equals the
    rounded fp32 result, for every tile configuration, residual epilogues and a ragged M whose last octet is shared with
    another producer (rows >= M of that octet must survive).  A grouped k-octet view then feeds a consumer GEMM."""
    from dataclasses import replace
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes, PackedLinear
    rng = np.random.default_rng(9100 + seed)
    K = int(rng.choice([64, 192, 384]))
    M = int(rng.choice([6, 64, 126, 128, 256, 640]))
    P = int(rng.choice([96, 1000, 7040]))
    epi = [ops.EPI_NONE, ops.EPI_GELU, ops.EPI_RES][seed % 3]
    n = 3
    g = torch.Generator().manual_seed(seed)
    W = PackedLinear(torch.randn(M, K, 1, 1, generator=g) / K ** 0.5, torch.randn(M, generator=g) * 0.1, dev)
    X = Planes.of(torch.randn(n, K, P, generator=g).to(dev))
    R = Planes.of(torch.randn(n, M, P, generator=g).to(dev)) if epi == ops.EPI_RES else None
    Ma = (M + 7) // 8 * 8
    prev = ops.set_precision("f16x2")
    try:
        y_plain = torch.full((n, M, P), float("nan"), device=dev)
        ops.gemm(W, X, Planes.of(y_plain), epi, R=R)
        y = torch.full((n, M, P), float("nan"), device=dev)
        sh = ops.new_shadow(Planes.of(torch.empty(n, Ma, P, device=dev)), dev)
        sh.base.view(torch.float16).fill_(7.0)                                    # sentinel in the rows past M
        Y = replace(Planes.of(y), shadow=replace(sh, rows=M))
        ops.gemm(W, X, Y, epi, R=R)
        torch.cuda.synchronize()
        assert torch.equal(y, y_plain)
        full = replace(sh, rows=Ma).tensor().float()
        assert torch.equal(full[:, :M], y.half().float())
        assert bool((full[:, M:] == 7.0).all())
        # consumer: the copy as B (k-octets by DMA) == the fp32 planes as B (rounded on load)
        if M % 32 == 0:
            W2 = PackedLinear(torch.randn(256, M, 1, 1, generator=g) / M ** 0.5, None, dev)
            z0 = torch.empty(n, 256, P, device=dev)
            z1 = torch.empty(n, 256, P, device=dev)
            ops.gemm(W2, Planes.of(y), Planes.of(z0), ops.EPI_NONE)
            ops.gemm(W2, Y, Planes.of(z1), ops.EPI_NONE)
            # '(B T) C -> B (T C)' view of the copy: groups of M rows, one group per image
            W3 = PackedLinear(torch.randn(256, n * M, 1, 1, generator=g) / (n * M) ** 0.5, None, dev)
            yg = Planes(Y.base, 0, n * M * P, 1, n * M, P, group=M, group_stride=M * P)
            shg = Planes(sh.base, 0, n * sh.img_stride, 1, n * M, P, f16=True, koct=True, group=M, group_stride=sh.img_stride)
            z2 = torch.empty(1, 256, P, device=dev)
            z3 = torch.empty(1, 256, P, device=dev)
            ops.gemm(W3, yg, Planes.of(z2), ops.EPI_NONE)
            ops.gemm(W3, replace(yg, shadow=shg), Planes.of(z3), ops.EPI_NONE)
            torch.cuda.synchronize()
            assert torch.equal(z0, z1)
            assert torch.equal(z2, z3)
    finally:
        ops.set_precision(prev)


@pytest.mark.parametrize("hw", [(55, 128), (16, 24), (47, 156)])
def test_lookup_koct_copy(dev, hw):
    """sf_corr_lookup's optional second output (fp16 volumes): the 324 correlation channels as fp16 k-octet planes equal
    the rounded fp32 output, rows 324..327 of the last octet are zero, and the fp32 output itself is unchanged."""
    from dataclasses import replace
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    h, w = hw
    B, pairs, D, N = 2, 2, 64, h * w
    g = torch.Generator().manual_seed(h)
    f = torch.randn(B, pairs + 1, D, N, generator=g).to(dev) * 0.3
    dims = [(h >> l, w >> l) for l in range(4)]
    stride = [B * N * a * b for a, b in dims]
    lvls = [torch.empty(pairs * s, dtype=torch.float16, device=dev) for s in stride]
    wsb = torch.empty(max(ops.corr_build_ws_bytes(B, pairs, D, h, w), 16), dtype=torch.uint8, device=dev)
    ops.corr_build(f.data_ptr(), f.data_ptr() + 4 * D * N, (pairs + 1) * D * N, D * N, lvls, stride, B, pairs, D, h, w, ws=wsb)
    coords = torch.rand(B * pairs, 2, N, generator=g).to(dev) * torch.tensor([w * 1.2, h * 1.2], device=dev).view(1, 2, 1) - 3.0
    cp = Planes.of(coords)
    plain = torch.full((B * pairs, 324, N), float("nan"), device=dev)
    ops.corr_lookup(lvls, stride, cp, Planes.of(plain), B, pairs, h, w)
    out = torch.full((B * pairs, 324, N), float("nan"), device=dev)
    sh = ops.new_shadow(Planes.of(torch.empty(B * pairs, 328, N, device=dev)), dev)
    sh.base.view(torch.float16).fill_(5.0)
    ops.corr_lookup(lvls, stride, cp, replace(Planes.of(out), shadow=replace(sh, rows=324)), B, pairs, h, w)
    torch.cuda.synchronize()
    assert torch.equal(out, plain)
    full = sh.tensor().float()
    assert torch.equal(full[:, :324], out.half().float())
    assert bool((full[:, 324:] == 0).all())


def test_flow_update_koct_rows(dev):
    """sf_flow_update's k-octet output: rows 126 / 127 of a 128-row fp16 k-octet tensor receive the rounded flow, the
    other rows of that octet are untouched."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    n, h, w = 3, 9, 20
    P = h * w
    g = torch.Generator().manual_seed(3)
    coords = ops.coords_grid(n, h, w, dev).view(n, 2, P) + torch.randn(n, 2, P, generator=g).to(dev)
    delta = torch.randn(n, 2, P, generator=g).to(dev)
    mf = torch.zeros(n, 128, P, device=dev)
    sh = ops.new_shadow(Planes.of(mf), dev)
    sh.base.view(torch.float16).fill_(3.0)
    flow = torch.empty(n, 2, P, device=dev)
    cp = Planes.of(coords.clone())
    ops.flow_update(cp, Planes.of(delta), Planes.of(flow), Planes.of(mf).slice(126, 128), n, h, w, koct=sh, koct_row=126)   # (unshadowed planes: plain slice)
    torch.cuda.synchronize()
    got = sh.tensor().float()
    assert torch.equal(got[:, 126:128], flow.half().float())
    assert torch.equal(mf[:, 126:128], flow)
    assert bool((got[:, :126] == 3.0).all())


@pytest.mark.parametrize("P", [7040, 1000, 323])
def test_layernorm_and_temporal_attn_koct_outputs(dev, P):
    """sf_layernorm_cm / sf_temporal_attn with the fp16 k-octet output (the hand-over to the qkv / fc1 / proj GEMMs in the
    f16x2 mode): equal to the rounded fp32 output, fp32 output untouched when not requested."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    g = torch.Generator().manual_seed(P)
    n, C, TT = 6, 128, 3
    x = (torch.randn(n, C, P, generator=g) * 2 + 0.3).to(dev)
    gam, bet = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    y = torch.empty(n, C, P, device=dev)
    ops.layernorm_cm(Planes.of(x), gam, bet, Planes.of(y))
    sh = ops.new_shadow(Planes.of(y), dev)
    ops.layernorm_cm(Planes.of(x), gam, bet, sh)
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.double().permute(0, 2, 1), (C,), gam.double(), bet.double()).permute(0, 2, 1)
    assert (y.double() - ref).abs().max().item() < 2e-5
    # (the two instantiations may contract the final fma differently: equal up to one fp16 rounding of the fp32 result)
    close = lambda a, b: bool(((a - b).abs() <= 2.0 ** -11 * b.abs() * 1.01 + 1e-6).all())
    assert close(sh.tensor().float(), y)
    qkv = torch.randn(n, 3 * C, P, generator=g).to(dev)
    out = torch.empty(n, C, P, device=dev)
    ops.temporal_attn(Planes.of(qkv), Planes.of(out), n // TT, TT, C)
    sh2 = ops.new_shadow(Planes.of(out), dev)
    ops.temporal_attn(Planes.of(qkv), sh2, n // TT, TT, C)
    torch.cuda.synchronize()
    assert close(sh2.tensor().float(), out)
    # qkv as fp16 rows (sf_temporal_attn_f16in, the config-2 hand-over): float64 attention over the SAME fp16 values, and the
    # fp32-input kernel fed those values, for both output formats (NaN-filled first)
    q16 = qkv.half().contiguous()
    Q16 = Planes(q16.view(-1).view(torch.float32), 0, 3 * C * P, n, 3 * C, P, f16=True)
    out_h = torch.full((n, C, P), float("nan"), device=dev)
    ops.temporal_attn(Q16, Planes.of(out_h), n // TT, TT, C)
    sh3 = ops.new_shadow(Planes.of(out_h), dev)
    sh3.base.view(torch.float16).fill_(float("nan"))
    ops.temporal_attn(Q16, sh3, n // TT, TT, C)
    out_f = torch.empty(n, C, P, device=dev)
    ops.temporal_attn(Planes.of(q16.float()), Planes.of(out_f), n // TT, TT, C)
    torch.cuda.synchronize()
    qd = q16.double().view(n // TT, TT, 3, C, P)
    att = torch.softmax(torch.einsum("btcp,bucp->bptu", qd[:, :, 0], qd[:, :, 1]) / C ** 0.5, dim=-1)
    ref_a = torch.einsum("bptu,bucp->btcp", att, qd[:, :, 2]).reshape(n, C, P)
    assert (out_h.double() - ref_a).abs().max().item() < 2e-5
    assert (out_h - out_f).abs().max().item() < 2e-6
    assert close(sh3.tensor().float(), out_h)


@pytest.mark.parametrize("preset", ["config2_fp16", "fp32_class"])
def test_graph_capture_call_equals_replays_and_eager(dev, preset):
    """The call that captures the HIP graph (warm-up iteration outside capture, loop state restored, first replay) must
    return exactly what later replays and the eager engine return -- with a warm-start flow_init, so that the restored
    loop state (including the flow rows of the k-octet copy of the motion features) matters."""
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    B, T, h, w, iters = 2, 4, 24, 40, 4
    P = syn.make_params(11, T)
    fmaps, cnets = syn.make_features(11, B, T, h, w)
    g = torch.Generator().manual_seed(1)
    finit = [(torch.randn(B, 2, h, w, generator=g) * 2).to(dev) for _ in range(T - 1)]
    kw = presets.engine_kwargs(preset)
    eng_g = HotPathEngine(P, device=dev, T=T, use_graph=True, **kw)
    eng_e = HotPathEngine(P, device=dev, T=T, use_graph=False, **kw)
    fd, cd = fmaps.to(dev), cnets.to(dev)
    first = [u.clone() for u in eng_g.forward(fd, cd, iters=iters, flow_init=finit)[0]]
    second = [u.clone() for u in eng_g.forward(fd, cd, iters=iters, flow_init=finit)[0]]
    eager = eng_e.forward(fd, cd, iters=iters, flow_init=finit)[0]
    torch.cuda.synchronize()
    for a, b2, c in zip(first, second, eager):
        assert torch.equal(a, b2)
        assert torch.equal(a, c)


@pytest.mark.parametrize("case", [(2, 3, 16, 24, True), (1, 2, 17, 21, False), (3, 5, 16, 20, True), (1, 4, 17, 19, False),
                                  (2, 2, 24, 36, True)])
def test_config2_preset_shapes_vs_oracle(dev, case):
    """The bench preset (f16x2 + fp16 volumes + fused GMA + fp16 / k-octet operand hand-over) over clip lengths T = 2..5,
    batches, and grids whose pixel count is or is not a multiple of 4 (the fp16 hand-over formats need P % 4 == 0 and are
    switched off otherwise), graph and eager: every flow against the CPU oracle after 4 iterations."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    B, T, h, w, graph = case
    P = syn.make_params(20 + T, T)
    fmaps, cnets = syn.make_features(30 + h, B, T, h, w)
    ups_o, _ = orc.hotpath_forward(fmaps, cnets, P, 4)
    eng = HotPathEngine(P, device=dev, T=T, use_graph=graph, **presets.engine_kwargs("config2_fp16"))
    for _ in range(2):
        ups, _ = eng.forward(fmaps.to(dev), cnets.to(dev), iters=4)
    assert eng.plan(B, h, w, 256).shadows == ((h * w) % 4 == 0)
    e = max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))
    assert e <= 1e-3, (case, e)


def test_two_product_mode_gelu_accuracy(dev):
    """The f16x2 / f16 modes evaluate GELU as a polynomial (sf_common.h gelu_poly2) where the result is stored as fp16
    (k-octet epilogue).  Through an identity GEMM (K = M, W = I, inputs exactly representable in fp16, so the contraction
    is exact) the epilogue output IS gelu(x): the stored fp16 value must be within the polynomial's 6e-5 absolute plus one
    fp16 rounding of float64 over [-12, 12] AND far outside (|x| up to 2000: the polynomial saturates, its error must not
    grow with |x| -- ADVICE r2); fp32 results keep the erf-rational form (<= 2e-6) in every mode."""
    from streamflow_amd import ops
    from streamflow_amd.ops import PackedLinear, Planes
    M, P = 128, 4096
    x = (torch.linspace(-12, 12, M * P).view(1, P, M).permute(0, 2, 1)).contiguous().half().float()   # [1, M, P]
    x[0, :, :64] = torch.linspace(-2000, -12, M * 64).view(64, M).t().half().float()                   # far negative tail
    x[0, :, 64:128] = torch.linspace(12, 2000, M * 64).view(64, M).t().half().float()
    W = PackedLinear(torch.eye(M).view(M, M, 1, 1), None, dev)
    ref = torch.nn.functional.gelu(x.double())
    big = x >= 0.5
    for prec in ("f16x2", "f16x3"):
        prev = ops.set_precision(prec)
        try:
            y = torch.empty(1, M, P, device=dev)
            ops.gemm(W, Planes.of(x.to(dev)), Planes.of(y), ops.EPI_GELU)            # fp32 result: erf-rational GELU
            yk = None
            if prec == "f16x2":                                                       # k-octet result: polynomial GELU
                yk = ops.new_shadow(Planes.of(y), dev)
                ops.gemm(W, Planes.of(x.to(dev)), yk, ops.EPI_GELU)
            torch.cuda.synchronize()
        finally:
            ops.set_precision(prev)
        err = (y.double().cpu() - ref).abs()
        assert bool((err <= 2e-6 + 2e-7 * ref.abs()).all()) and (err[big] / ref[big]).max().item() < 2e-6, (prec, err.max().item())
        if yk is not None:
            got = yk.tensor().double().cpu()
            err = (got - ref).abs()
            # polynomial (<= 5.2e-5 absolute, 1.1e-5 relative) + the fp16 rounding of the stored value (2^-11 relative)
            assert bool((err <= 6e-5 + 2.0 ** -11 * ref.abs() * 1.01).all())
            assert (err[big] / ref[big]).max().item() < 2.0 ** -11 * 1.05


@pytest.mark.parametrize("preset", ["config2_fp16", "fp32_class"])
def test_two_chain_schedule_is_bitwise_the_one_chain_schedule(dev, preset):
    """The engine runs the sections without a concurrent branch as two half-batch chains on two streams (EngineOptions.split_solo,
    default 2).  Clips are independent and a kernel's arithmetic does not depend on the batch it is launched with, so the
    flows must be bit-identical to the single-chain schedule -- graph and eager, B = 2 and 4.  (The automatic split-K of
    small grids is switched off for the comparison: it is not available inside the chains -- one scratch buffer -- and
    changes the summation order of the GEMMs it applies to at this test's small shape.)"""
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import EngineOptions, HotPathEngine
    T, h, w, iters = 3, 24, 32, 3
    P = syn.make_params(7, T)
    kw = presets.engine_kwargs(preset)
    for B, graph in ((2, True), (4, False), (1, True), (3, False)):     # (odd batches: two chains over IMAGE ranges)
        fmaps, cnets = syn.make_features(40 + B, B, T, h, w)
        outs = {}
        for chains in ("0", "2", "4"):
            eng = HotPathEngine(P, device=dev, T=T, use_graph=graph, options=EngineOptions(auto_split_k=False, split_solo=int(chains), split_uneven=True), **kw)
            for _ in range(2):
                ups = eng.forward(fmaps.to(dev), cnets.to(dev), iters=iters)[0]
            outs[chains] = [u.clone() for u in ups]
        torch.cuda.synchronize()
        for chains in ("2", "4"):
            for a, b2 in zip(outs["0"], outs[chains]):
                assert torch.equal(a, b2), (preset, B, graph, chains)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("prec", ["fp32", "f16x3", "f16x2"])
def test_subsample_attn_random_shapes(dev, seed, prec):
    """Encoder f1: GlobalSubSampleAttn core (exact VALU kernel for fp32, the matrix-core kernel for the split / fp16 classes)
    against a float64 softmax(q k^T / sqrt(32)) v: ragged key counts (M % 32 != 0), query counts that end inside a wave's
    64-query strip, several heads and images, logits large enough that the running maximum matters."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    rng = np.random.default_rng(7000 + seed)
    heads = int(rng.choice([1, 4, 8]))
    N = int(rng.choice([1, 31, 64, 65, 200, 257, 1000]))
    M = int(rng.choice([1, 5, 32, 33, 64, 100, 330]))
    n = int(rng.integers(1, 4))
    C = heads * 32
    g = torch.Generator().manual_seed(seed)
    gain = float(rng.choice([1.0, 4.0]))
    q = torch.randn(n, C, N, generator=g) * gain
    kv = torch.randn(n, 2 * C, M, generator=g)
    kv[:, :C] *= gain
    out = torch.full((n, C, N), float("nan"), device=dev)
    prev = ops.set_precision(prec)
    try:
        ops.subsample_attn(Planes.of(q.to(dev)), Planes.of(kv.to(dev)), Planes.of(out), heads)
    finally:
        ops.set_precision(prev)
    qd = q.double().view(n, heads, 32, N)
    kd, vd = kv[:, :C].double().view(n, heads, 32, M), kv[:, C:].double().view(n, heads, 32, M)
    att = torch.softmax(torch.einsum("bhdn,bhdm->bhnm", qd, kd) * 32 ** -0.5, dim=-1)
    ref = torch.einsum("bhnm,bhdm->bhdn", att, vd).reshape(n, C, N)
    err = (out.cpu().double() - ref).abs().max().item()
    # fp16 classes: logits carry the fp16 rounding of q and k (2^-11 relative each, |q||k| up to ~ 16 * 32 * gain^2 / sqrt(32))
    tol = {"fp32": 2e-5, "f16x3": 2e-5, "f16x2": 4e-3 * gain * gain}[prec]
    print(f"subsample_attn {prec} heads={heads} N={N} M={M} n={n} gain={gain}: max abs err {err:.3e}")
    assert err <= tol, (err, tol)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("prec", ["fp32", "f16x3", "f16x2"])
def test_window_attn_random_shapes(dev, seed, prec):
    """Encoder f1: LocallyGroupedAttn core against float64 on grids that are not multiples of the window (padded tokens
    carry k = v = the qkv bias and take part in the softmax, as in timm's pad-after-norm), window sizes 2..7."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    rng = np.random.default_rng(8000 + seed)
    heads = int(rng.choice([4, 8]))
    ws = int(rng.choice([2, 3, 5, 7, 7, 7]))
    H, W = int(rng.integers(3, 30)), int(rng.integers(3, 40))
    n = int(rng.integers(1, 3))
    C = heads * 32
    g = torch.Generator().manual_seed(seed)
    gain = float(rng.choice([1.0, 3.0]))
    qkv = torch.randn(n, 3 * C, H * W, generator=g)
    qkv[:, : 2 * C] *= gain
    bias = torch.randn(3 * C, generator=g)
    out = torch.full((n, C, H * W), float("nan"), device=dev)
    prev = ops.set_precision(prec)
    try:
        ops.window_attn(Planes.of(qkv.to(dev)), bias.to(dev), Planes.of(out), heads, H, W, ws)
    finally:
        ops.set_precision(prev)
    # float64 reference: pad the grid with "bias tokens" to a multiple of ws, attend inside windows, crop
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    full = bias.double().view(1, 3 * C, 1, 1).expand(n, 3 * C, Hp, Wp).clone()
    full[:, :, :H, :W] = qkv.double().view(n, 3 * C, H, W)
    t = full.view(n, 3, heads, 32, Hp // ws, ws, Wp // ws, ws).permute(1, 0, 2, 4, 6, 5, 7, 3).reshape(3, n, heads, -1, ws * ws, 32)
    att = torch.softmax(t[0] @ t[1].transpose(-1, -2) * 32 ** -0.5, dim=-1) @ t[2]          # [n, heads, windows, ws*ws, 32]
    ref = att.view(n, heads, Hp // ws, Wp // ws, ws, ws, 32).permute(0, 1, 6, 2, 4, 3, 5).reshape(n, C, Hp, Wp)[:, :, :H, :W]
    err = (out.cpu().double().view(n, C, H, W) - ref).abs().max().item()
    tol = {"fp32": 2e-5, "f16x3": 2e-5, "f16x2": 4e-3 * gain * gain}[prec]
    print(f"window_attn {prec} heads={heads} ws={ws} H={H} W={W} n={n} gain={gain}: max abs err {err:.3e}")
    assert err <= tol, (err, tol)


def _koct_planes(ops, n, rows, P, dev):
    from streamflow_amd.ops import Planes
    return Planes(torch.zeros(n * rows * P // 2, dtype=torch.float32, device=dev), 0, rows * P, n, rows, P, f16=True, koct=True)


@pytest.mark.parametrize("seed", range(4))
def test_attention_cores_koct_handover(dev, seed):
    """The encoder's fp16 hand-over (f16x2 / f16 classes): the window core fed with k-octet q / k / v (the qkv GEMM's
    c_f16 = 2 output) and both cores writing k-octet outputs must agree with the same cores on fp32 planes of the SAME
    fp16-rounded inputs to within the output rounding."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    rng = np.random.default_rng(9000 + seed)
    heads = int(rng.choice([4, 8]))
    C = heads * 32
    H, W = int(rng.integers(5, 20)), 4 * int(rng.integers(2, 9))
    N, n = H * W, int(rng.integers(1, 3))
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(n, 3 * C, N, generator=g) * 1.5).half().float().to(dev)      # exactly representable: both paths see the same operands
    bias = torch.randn(3 * C, generator=g).half().float().to(dev)
    prev = ops.set_precision("f16x2")
    try:
        ref = torch.empty(n, C, N, device=dev)
        ops.window_attn(Planes.of(qkv), bias, Planes.of(ref), heads, H, W, 7)
        qk = _koct_planes(ops, n, 3 * C, N, dev)
        ops.pack_koct(Planes.of(qkv), qk)
        outk = _koct_planes(ops, n, C, N, dev)
        ops.window_attn(qk, bias, outk, heads, H, W, 7)
        got = outk.tensor().float()
        err = (got - ref).abs().max().item()
        print(f"window core, k-octet in/out vs fp32 planes: {err:.3e}")
        assert err <= 2e-3 + 1e-3 * ref.abs().max().item()        # q is rounded before / after its scale factor; fp16 output
        # sub-sample core: k-octet output only
        M = int(rng.choice([33, 64, 100]))
        q = torch.randn(n, C, N, generator=g).to(dev)
        kv = torch.randn(n, 2 * C, M, generator=g).to(dev)
        ref2 = torch.empty(n, C, N, device=dev)
        ops.subsample_attn(Planes.of(q), Planes.of(kv), Planes.of(ref2), heads)
        out2 = _koct_planes(ops, n, C, N, dev)
        ops.subsample_attn(Planes.of(q), Planes.of(kv), out2, heads)
        err2 = (out2.tensor().float() - ref2).abs().max().item()
        assert err2 <= 1e-3 * max(1.0, ref2.abs().max().item()), err2              # only the fp16 rounding of the result
    finally:
        ops.set_precision(prev)


@pytest.mark.parametrize("M,K", [(128, 128), (128, 192), (128, 960), (126, 384), (200, 136), (256, 136), (384, 256), (486, 324), (640, 960), (960, 640)])
@pytest.mark.parametrize("single", [False, True])
@pytest.mark.parametrize("epi", ["none", "gelu_koct", "res_gelu_dw1"])
def test_koct_gemm_dispatch_branches_vs_float64(dev, M, K, single, epi):
    """Every dispatch branch of the k-octet-fed GEMM (csrc/gemm_split.hip, lay == 13): the 128 x 128 DMA kernel (M <= 128 with
    a long K), the B-direct kernel as 1 x 8 waves (single-product layers, and 192 <= M < 512) and 2 x 4 waves (two-product,
    M >= 512), the short-K one-row-tile rule -- with split and single-product weights and the three epilogue families of the
    update block, ragged pixel counts and two images, against a float64 product of the SAME fp16-rounded operands."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes, PackedLinear
    g = torch.Generator().manual_seed(M * 7 + K + int(single))
    n, P = 2, 1012                                           # not a multiple of the 256-pixel tile; a multiple of 4
    Wt = torch.randn(M, K, generator=g) / K ** 0.5
    bias = torch.randn(M, generator=g) * 0.1
    x = torch.randn(n, K, P, generator=g).half().float()     # exactly representable activations
    R = torch.randn(n, M, P, generator=g)
    dw_w, dw_b = torch.randn(M, generator=g) * 0.5, torch.randn(M, generator=g) * 0.1
    prev = ops.set_precision("f16x2")
    try:
        A = PackedLinear(Wt.reshape(M, K, 1, 1), bias, dev)
        A.single = single
        Ka = (K + 7) // 8 * 8
        X = _koct_planes(ops, n, Ka, P, dev)
        X = Planes(X.base, 0, Ka * P, n, K, P, f16=True, koct=True)
        if not ops.uses_dma_tile(M):                         # M = 200 pads to 256 (28 % waste): the tiled family's 64-row tile has no
            from streamflow_amd import _lib                  # k-octet path (it says so); SF_ALGO_AUTO runs such shapes on the
            with pytest.raises(RuntimeError, match="SF_LAYOUT_F16_KOCT"):                      # activation-stationary kernel
                ops.gemm(A, X, Planes.of(torch.empty(n, M, P, device=dev)), ops.EPI_NONE, algo=_lib.ALGO_TILED)
        xs = torch.zeros(n, Ka, P)
        xs[:, :K] = x
        ops.pack_koct(Planes.of(xs.to(dev)), Planes(X.base, 0, Ka * P, n, Ka, P, f16=True, koct=True))
        if epi == "gelu_koct":
            Ma = (M + 7) // 8 * 8
            Yk = _koct_planes(ops, n, Ma, P, dev)
            Y = Planes(Yk.base, 0, Ma * P, n, M, P, f16=True, koct=True)
            ops.gemm(A, X, Y, ops.EPI_GELU)
            got = Y.tensor().float().cpu()
        elif epi == "res_gelu_dw1":
            Yt = torch.full((n, M, P), float("nan"), device=dev)
            ops.gemm(A, X, Planes.of(Yt), ops.EPI_RES_GELU_DW1, R=Planes.of(R.to(dev)), dw_w=dw_w.to(dev), dw_b=dw_b.to(dev))
            got = Yt.cpu()
        else:
            Yt = torch.full((n, M, P), float("nan"), device=dev)
            ops.gemm(A, X, Planes.of(Yt), ops.EPI_NONE)
            got = Yt.cpu()
        torch.cuda.synchronize()
        # the weights the kernel multiplies by: hi (+ lo) of the scaled split image
        hi = A.hi.float().permute(1, 0, 2).reshape(A.lda_h, -1)[:M, :K].cpu()
        lo = A.lo.float().permute(1, 0, 2).reshape(A.lda_h, -1)[:M, :K].cpu()
        Wk = (hi if single else hi + lo).double() / A.split_scale
    finally:
        ops.set_precision(prev)
    v = torch.einsum("mk,zkp->zmp", Wk, x.double()) + bias.double()[None, :, None]
    if epi == "gelu_koct":
        ref, tol = F.gelu(v), 2e-3 + 1e-3 * F.gelu(v).abs().max().item()       # fp16 output, polynomial GELU
    elif epi == "res_gelu_dw1":
        t = F.gelu(R.double() + v)
        ref, tol = F.gelu(t + (dw_w.double()[None, :, None] * t + dw_b.double()[None, :, None])), 5e-5
    else:
        ref, tol = v, 5e-5
    err = (got.double() - ref).abs().max().item()
    assert err <= tol, (M, K, single, epi, err)


@pytest.mark.parametrize("graph", [False, True])
def test_two_engines_two_threads_match_their_serial_runs(dev, graph):
    """No process-wide launch state: two engines with DIFFERENT presets (arithmetic class, volume format, GMA path, hand-over
    formats) driven concurrently from two host threads, each on its own stream, give bit for bit the flows each gives alone.
    Eager launches interleave call by call (every launch takes its precision / formats from the engine's own context);
    graph engines are captured one after the other and then replayed concurrently."""
    import threading
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    T, h, w, iters, B = 3, 24, 32, 3, 2
    P = syn.make_params(11, T)
    jobs = []
    for i, preset in enumerate(("config2_mixed", "fp32_class")):
        eng = HotPathEngine(P, device=dev, T=T, use_graph=graph, **presets.engine_kwargs(preset))
        fm, cn = syn.make_features(70 + i, B, T, h, w)
        fm, cn = fm.to(dev), cn.to(dev)
        for _ in range(2):                                        # serial reference (second run: graph replay)
            ref = [u.clone() for u in eng.forward(fm, cn, iters=iters)[0]]
        jobs.append((eng, fm, cn, ref))
    torch.cuda.synchronize()
    results, errors = [None, None], []
    gate = threading.Barrier(2)

    def run(i):
        try:
            eng, fm, cn, _ = jobs[i]
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                gate.wait(timeout=60)
                for _ in range(4):
                    ups = eng.forward(fm, cn, iters=iters)[0]
                results[i] = [u.clone() for u in ups]
            st.synchronize()
        except Exception as e:                                    # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(e)))

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    torch.cuda.synchronize()
    assert not errors, errors
    for i in range(2):
        assert results[i] is not None
        for a, b in zip(results[i], jobs[i][3]):
            assert torch.equal(a, b), ("engine", i, "graph", graph)


@pytest.mark.parametrize("opts", ["koct_io=0", "x2_f16=0", "koct_io=0,x2_f16=0,pw_fold=0", "hidden_koct=0", "shadows=0"])
def test_handover_switches_stay_within_the_class(dev, opts):
    """EngineOptions hand-over switches (A/B knobs): every combination computes the same network in the same arithmetic class --
    flows within the config-2 budget of the oracle AND close to the default engine's (the switches change where an activation
    is rounded to fp16, not what is computed)."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import EngineOptions, HotPathEngine
    T, h, w, iters, B = 3, 24, 32, 4, 2
    P = syn.make_params(5, T)
    fm, cn = syn.make_features(55, B, T, h, w)
    ups_o, _ = orc.hotpath_forward(fm, cn, P, iters)
    kw = presets.engine_kwargs("config2_mixed")
    spec = dict(item.split("=") for item in opts.split(","))
    alt = EngineOptions(**{k: (v != "0") for k, v in spec.items()})
    outs = []
    for options in (EngineOptions(), alt):
        eng = HotPathEngine(P, device=dev, T=T, use_graph=False, options=options, **kw)
        ups = eng.forward(fm.to(dev), cn.to(dev), iters=iters)[0]
        outs.append([u.cpu() for u in ups])
    e_def = max(orc.epe(u, o) for u, o in zip(outs[0], ups_o))
    e_alt = max(orc.epe(u, o) for u, o in zip(outs[1], ups_o))
    d = max(orc.epe(a, b) for a, b in zip(outs[0], outs[1]))
    assert e_def <= 1e-3 and e_alt <= 1e-3 and d <= 1e-3, (opts, e_def, e_alt, d)
