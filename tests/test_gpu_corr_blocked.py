"""Blocked fp16 correlation volumes (csrc/corr_blocked.hip; reference core/corr.py:7-54) through the C ABI:
build vs the oracle pyramid of the fp16-rounded features, lookup vs the oracle lookup on the stored cells, the k-octet
hand-over, robustness against garbage in the padding cells, and the k-octet residual of sf_gemm (SfGemm.r_f16)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _coords(orc, B, h, w, g, scale=4.0):
    c = orc.coords_grid(B, h, w) + torch.randn(B, 2, h, w, generator=g) * scale
    c[:, :, 0, 0] = torch.tensor([-6.0, 2.0])                        # window partly outside
    c[:, :, 1, 1] = torch.tensor([float(w + 9), float(h + 9)])       # fully outside
    c[:, :, 2, 2] = torch.tensor([3.0, 4.0])                         # exactly integer
    c[:, :, 3, 3] = torch.tensor([float(w - 1), float(h - 1)])       # last cell
    c[:, :, 4, 4] = torch.tensor([float("nan"), 1.0])                # swallowed: samples zero padding
    c[:, :, 5, 5] = torch.tensor([3.0, float(h) - 0.5])              # footprint crosses the padded last block row
    return c


# shapes: block-aligned, ragged in both dims, minimal (16 x 16: a 1-pixel coarsest level is NaN in the reference, utils.py:69-70), odd pooled sizes (KITTI-like 47 x 156 levels 23x78 / 11x39 / 5x19)
SHAPES = [(1, 32, 16, 24), (2, 16, 17, 19), (1, 40, 16, 16), (1, 256, 47, 156), (3, 16, 16, 40), (1, 64, 33, 65), (2, 256, 24, 41)]


@pytest.mark.parametrize("B,D,h,w", SHAPES)
def test_blocked_build_and_lookup_vs_oracle(dev, B, D, h, w):
    from oracle import streamflow_oracle as orc
    import streamflow_amd as sfa
    g = torch.Generator().manual_seed(B * 1000 + h * 10 + w)
    f1, f2 = torch.randn(B, D, h, w, generator=g), torch.randn(B, D, h, w, generator=g)
    coords = _coords(orc, B, h, w, g)
    blk = sfa.CorrBlock(f1.to(dev), f2.to(dev), num_levels=4, radius=4, dtype=torch.float16, layout="blocked")
    out = blk(coords.to(dev)).cpu()
    lv = [t.cpu() for t in blk.corr_pyramid]
    pyr = orc.corr_pyramid(f1.half().float(), f2.half().float(), 4)
    for l, (a, b_) in enumerate(zip(lv, pyr)):
        assert a.dtype == torch.float16 and a.shape == b_.shape, (l, a.shape, b_.shape)
        err = (a.float() - b_).abs()
        tol = 2.0 ** -11 * b_.abs() + 3e-5
        assert (err <= tol).all(), (l, (err - tol).max().item())
    cc = coords.clone()
    cc[torch.isnan(cc)] = -1.0e6
    ref = orc.corr_lookup([t.float() for t in lv], cc, 4)
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < 5e-5


@pytest.mark.parametrize("B,pairs,D,h,w", [(2, 3, 32, 17, 28), (1, 1, 16, 47, 156), (1, 3, 256, 55, 128)])
def test_blocked_koct_output_and_padding_garbage(dev, B, pairs, D, h, w):
    """The k-octet product is the fp16 rounding of the fp32 planes of the same launch, bit for bit; rows 324..327 are
    zero; and neither output changes when the volume buffer was full of NaN bit patterns before the build (padding cells
    of partial blocks are never read as data)."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    n, N = B * pairs, h * w
    g = torch.Generator().manual_seed(7 + h)
    fm = torch.randn(B, pairs + 1, D, h, w, generator=g).to(dev)
    coords = _coords(orc, n, h, w, g, 3.0).to(dev).contiguous()
    outs = []
    for fill in (0x00, 0xFF):
        vol = ops.new_blocked_volume(n, h, w, dev)
        vol.buf.fill_(fill)
        ops.corr_build_blocked(fm.data_ptr(), fm.data_ptr() + 4 * D * N, (pairs + 1) * D * N, D * N, vol, B, pairs, D)
        out = torch.full((n, 324, N), float("nan"), device=dev)
        ko = ops.new_shadow(Planes.of(out), dev)
        ko.base.fill_(float("nan"))
        ops.corr_lookup_blocked(vol, Planes.of(coords), Planes.of(out), ko, B, pairs)
        torch.cuda.synchronize()
        raw = ko.base.view(torch.float16).view(n, 41, N, 8)
        assert (raw[:, 40, :, 4:] == 0).all()
        outs.append((out.cpu(), ko.tensor().cpu(), [t.cpu() for t in vol.levels()]))
    a, b = outs
    assert torch.isfinite(a[0]).all() and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[1], a[0].half())
    for x, y in zip(a[2], b[2]):
        assert torch.equal(x, y)
    # per-pair addressing: image (clip b, pair t) is the volume of frames (t, t + 1) of clip b
    lv = a[2]
    fmc = fm.cpu()
    for (b_, t) in ((0, 0), (B - 1, pairs - 1)):
        pyr = orc.corr_pyramid(fmc[b_:b_ + 1, t].half().float(), fmc[b_:b_ + 1, t + 1].half().float(), 4)
        for l in range(4):
            got = lv[l][b_ * pairs + t].float().reshape(pyr[l].shape)
            assert ((got - pyr[l]).abs() <= 2.0 ** -11 * pyr[l].abs() + 3e-5).all(), (b_, t, l)


def test_blocked_matches_row_major_fp16_path(dev):
    """Same arithmetic, two layouts: the cells of the blocked build equal the row-major fp16 build to one fp16 ulp (the
    blocked build folds 1 / sqrt(D) = 2^-4 into the packed source features: exact except where a scaled feature becomes
    an fp16 subnormal)."""
    import streamflow_amd as sfa
    g = torch.Generator().manual_seed(3)
    f1, f2 = torch.randn(2, 256, 23, 37, generator=g).to(dev), torch.randn(2, 256, 23, 37, generator=g).to(dev)
    a = sfa.CorrBlock(f1, f2, dtype=torch.float16)
    b = sfa.CorrBlock(f1, f2, dtype=torch.float16, layout="blocked")
    for x, y in zip(a.corr_pyramid, b.corr_pyramid):
        assert ((x.float() - y.float()).abs() <= 2.0 ** -10 * x.float().abs() + 1e-6).all()
        assert (x != y).float().mean().item() < 0.01


@pytest.mark.parametrize("M,K,P,n", [(324, 486, 7040, 2), (200, 96, 300, 3), (128, 64, 36, 1)])
def test_gemm_koct_residual(dev, M, K, P, n):
    """SfGemm.r_f16 = 2: C = gelu(t + dw_w t + dw_b), t = gelu(R + W X + b), with R an fp16 k-octet image (here: the
    k-octet copy of X's block input, as in convc1's ffn1.2) -- against float64 on the same fp16 residual values."""
    import torch.nn.functional as F
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes, PackedLinear
    g = torch.Generator().manual_seed(M + K)
    Wt, bias = torch.randn(M, K, generator=g) / K ** 0.5, torch.randn(M, generator=g) * 0.1
    dw_w, dw_b = torch.randn(M, generator=g) * 0.3, torch.randn(M, generator=g) * 0.1
    X, R = torch.randn(n, K, P, generator=g), torch.randn(n, M, P, generator=g)
    prev = ops.set_precision("f16x2")
    try:
        A = PackedLinear(Wt.reshape(M, K, 1, 1), bias, dev)
        Rp = Planes.of(R.to(dev))
        Rk = ops.new_shadow(Rp, dev)
        ops.pack_koct(Rp, Rk)
        Y = torch.full((n, M, P), float("nan"), device=dev)
        ops.gemm(A, Planes.of(X.to(dev)), Planes.of(Y), ops.EPI_RES_GELU_DW1, R=Rk, dw_w=dw_w.to(dev), dw_b=dw_b.to(dev))
        Y2 = torch.full((n, M, P), float("nan"), device=dev)
        ops.gemm(A, Planes.of(X.to(dev)), Planes.of(Y2), ops.EPI_RES_GELU_DW1, R=Planes.of(R.half().float().to(dev)),
                 dw_w=dw_w.to(dev), dw_b=dw_b.to(dev))
        torch.cuda.synchronize()
    finally:
        ops.set_precision(prev)
    ge = lambda t: F.gelu(t)
    t = ge(R.half().double() + torch.einsum("mk,nkp->nmp", Wt.double(), X.half().double()) + bias.double()[None, :, None])
    ref = ge(t + dw_w.double()[None, :, None] * t + dw_b.double()[None, :, None])
    assert (Y.double().cpu() - ref).abs().max().item() < 2e-4
    # and within fp32 rounding of the fp32-residual epilogue fed the same (fp16-representable) residual values
    assert (Y - Y2).abs().max().item() < 2e-5


def test_integration_md_binding_examples_run(dev):
    """The ctypes stubs INTEGRATION.md section 2 shows a reference maintainer are executed as written (only the library
    path is made absolute) and their results checked against the oracle: row-major pyramid, blocked build, blocked lookup."""
    import os
    import re
    from oracle import streamflow_oracle as orc
    from streamflow_amd import synthetic as syn, _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = md[md.index("## 2. Binding the C ABI"):md.index("## 3. Build")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 3
    ns = {}
    for code in blocks:
        exec(code.replace('"streamflow_amd/libstreamflow_hip.so"', repr(_lib.LIB_PATH)), ns)
    B, T, D, h, w = 1, 3, 256, 16, 24
    fmaps = syn.randn(77, "fmaps", (B, T, D, h, w)).to(dev).contiguous()
    # row-major fp32 pyramid of pair 0 (split precision f16x3)
    lv = ns["corr_pyramid"](fmaps[:, 0].contiguous(), fmaps[:, 1].contiguous())
    ref = orc.corr_pyramid(fmaps[:, 0].cpu(), fmaps[:, 1].cpu())
    for a, b in zip(lv, ref):
        assert (a.cpu() - b).abs().max().item() < 2e-4
    # blocked fp16 volume + lookup into k-octets, both pairs
    vol, img_bytes = ns["corr_blocked"](fmaps, B, T, D, h, w)
    n_img, pairs = B * (T - 1), T - 1
    coords = orc.coords_grid(n_img, h, w) + syn.randn(78, "dc", (n_img, 2, h, w)) * 3.0
    koct = ns["lookup_blocked"](vol, img_bytes, coords.to(dev).contiguous(), n_img, pairs, h, w)
    torch.cuda.synchronize()
    got = koct.float().permute(0, 1, 3, 2).reshape(n_img, 328, h * w)[:, :324].cpu()
    for t in range(pairs):
        pyr = orc.corr_pyramid(fmaps[:, t].cpu(), fmaps[:, t + 1].cpu())
        want = orc.corr_lookup(pyr, coords[t:t + 1]).reshape(324, h * w)
        err = (got[t] - want).abs().max().item()
        assert err < 2e-2 * max(1.0, want.abs().max().item() / 8), (t, err)       # fp16 cells and fp16 hand-over
    # blocked fp32 volume + lookup into the reference's [n_img, 324, h, w] tensor
    vol32, img_bytes32 = ns["corr_blocked32"](fmaps, B, T, D, h, w)
    out32 = ns["lookup_blocked32"](vol32, img_bytes32, coords.to(dev).contiguous(), n_img, pairs, h, w)
    torch.cuda.synchronize()
    for t in range(pairs):
        pyr = orc.corr_pyramid(fmaps[:, t].cpu(), fmaps[:, t + 1].cpu())
        want = orc.corr_lookup(pyr, coords[t:t + 1])[0]
        assert (out32[t].cpu() - want).abs().max().item() < 5e-5, t


def test_blocked_build_and_lookup_are_deterministic_at_the_headline_shape(dev):
    """Twelve builds + lookups of the same 8-clip batch beside a competing stream: every volume byte and every looked-up feature
    bit-identical.  (Guards the kernels whose epilogues issue 16-byte buffer stores with a register offset: csrc/mask_upsample.hip's
    first version lost cells of its LAST workgroups that way, run to run.)"""
    import hashlib
    from oracle import streamflow_oracle as orc
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    B, pairs, D, h, w = 8, 3, 256, 55, 128
    n, N = B * pairs, h * w
    g = torch.Generator().manual_seed(3)
    fm = torch.randn(B, pairs + 1, D, h, w, generator=g).to(dev)
    coords = (orc.coords_grid(n, h, w) + torch.randn(n, 2, h, w, generator=g) * 6.0).to(dev).contiguous()
    side = torch.cuda.Stream(device=dev)
    junk = torch.randn(2048, 2048, device=dev)
    first = None
    vol = ops.new_blocked_volume(n, h, w, dev)
    for rep in range(12):
        if rep % 2:
            with torch.cuda.stream(side):
                junk = torch.tanh(junk) * 1.0001
        vol.buf.zero_()
        ops.corr_build_blocked(fm.data_ptr(), fm.data_ptr() + 4 * D * N, (pairs + 1) * D * N, D * N, vol, B, pairs, D)
        ko = ops.new_shadow(Planes.of(torch.empty(n, 324, N, device=dev)), dev)
        ops.corr_lookup_blocked(vol, Planes.of(coords), None, ko, B, pairs)
        torch.cuda.synchronize()
        # (the 3.3-GB volume is summed on the device: two integer checksums over different word sizes; the features are hashed whole)
        raw = vol.buf.view(torch.uint8)
        raw = raw[: raw.numel() // 4 * 4]
        sig = (int(torch.sum(raw.view(torch.int32), dtype=torch.int64)), int(torch.sum(raw.view(torch.int16), dtype=torch.int64)),
               hashlib.sha256(ko.base.cpu().numpy().tobytes()).hexdigest())
        if first is None:
            first = sig
        assert sig == first, rep
