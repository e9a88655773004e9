"""f16x3 GEMM-to-GEMM hand-over as split k-octet planes (SF_LAYOUT_SPLIT_KOCT / c_f16 = 4; reference: the 1x1 convolutions of
PCBlock4_Deep_nopool_res, core/update.py:14-16,30-36): a chain of two sf_gemm calls with the hidden tensor stored already split is
BIT-IDENTICAL to the same chain through fp32 planes (the consumer computes the same (hi, lo) pair itself), and matches float64."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _split_planes(n, rows, P, dev):
    from streamflow_amd.ops import Planes
    r8 = (rows + 7) // 8 * 8
    base = torch.full((n * r8 * P,), float("nan"), device=dev)
    return Planes(base, 0, 2 * r8 * P, n, rows, P, f16=True, koct=True, split=True)


# (C, hidden, M2, P, n): the SK blocks' ffn shapes (hidden not a multiple of 32 / 8), a ragged pixel count, one tiny case
@pytest.mark.parametrize("C,H,M2,P,n", [(256, 384, 256, 7040, 2), (324, 486, 324, 1000, 1), (640, 960, 128, 704, 3), (128, 192, 128, 36, 1)])
def test_split_koct_chain_is_bit_identical_to_fp32_planes(dev, C, H, M2, P, n):
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes, PackedLinear
    g = torch.Generator().manual_seed(C + H)
    W1, b1 = torch.randn(H, C, generator=g) / C ** 0.5, torch.randn(H, generator=g) * 0.1
    W2, b2 = torch.randn(M2, H, generator=g) / H ** 0.5, torch.randn(M2, generator=g) * 0.1
    X = torch.randn(n, C, P, generator=g)
    R = torch.randn(n, H, P, generator=g)
    prev = ops.set_precision("f16x3")
    try:
        A1, A2 = PackedLinear(W1.reshape(H, C, 1, 1), b1, dev), PackedLinear(W2.reshape(M2, H, 1, 1), b2, dev)
        Xp, Rp = Planes.of(X.to(dev)), Planes.of(R.to(dev))
        outs = []
        for epi, res in ((ops.EPI_GELU, None), (ops.EPI_RES_GELU, Rp), (ops.EPI_NONE, None)):
            hid32 = Planes.of(torch.full((n, H, P), float("nan"), device=dev))
            hid_s = _split_planes(n, H, P, dev)
            Y32 = Planes.of(torch.full((n, M2, P), float("nan"), device=dev))
            Ys = Planes.of(torch.full((n, M2, P), float("nan"), device=dev))
            ops.gemm(A1, Xp, hid32, epi, R=res)
            ops.gemm(A2, hid32, Y32, ops.EPI_NONE)
            ops.gemm(A1, Xp, hid_s, epi, R=res)
            ops.gemm(A2, hid_s, Ys, ops.EPI_NONE)
            torch.cuda.synchronize()
            h32, hs = hid32.tensor().cpu(), hid_s.tensor().cpu()
            # the stored pair is the truncating split of the fp32 value: hi + lo within 2^-20 of it, never above it in magnitude
            assert torch.isfinite(hs).all()
            assert ((hs - h32).abs() <= 2.0 ** -19 * h32.abs() + 1e-7).all()
            y32, ys = Y32.tensor().cpu(), Ys.tensor().cpu()
            assert torch.equal(y32, ys), (epi, (y32 - ys).abs().max().item())
            outs.append((epi, res is not None, y32))
    finally:
        ops.set_precision(prev)
    for epi, has_r, y in outs:
        t = torch.einsum("hc,ncp->nhp", W1.double(), X.double()) + b1.double()[None, :, None]
        if has_r:
            t = t + R.double()
        if epi != ops.EPI_NONE:
            t = F.gelu(t)
        ref = torch.einsum("mh,nhp->nmp", W2.double(), t) + b2.double()[None, :, None]
        assert (y.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_split_koct_needs_f16x3_and_the_128_row_tile(dev):
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes, PackedLinear
    g = torch.Generator().manual_seed(1)
    W = torch.randn(64, 128, generator=g)
    X = _split_planes(1, 128, 64, dev)
    X.base.zero_()
    Y = Planes.of(torch.zeros(1, 64, 64, device=dev))
    prev = ops.set_precision("f16x3")
    try:
        with pytest.raises(RuntimeError, match="SPLIT_KOCT"):
            ops.gemm(PackedLinear(W.reshape(64, 128, 1, 1), None, dev), X, Y)        # M = 64: no 128-row tile
    finally:
        ops.set_precision(prev)
    prev = ops.set_precision("f16x2")
    try:
        with pytest.raises(RuntimeError, match="f16x3"):
            ops.gemm(PackedLinear(torch.randn(128, 128, generator=g).reshape(128, 128, 1, 1), None, dev), X,
                     Planes.of(torch.zeros(1, 128, 64, device=dev)))
    finally:
        ops.set_precision(prev)


def test_engine_with_the_split_handover_is_bit_identical(dev):
    """EngineOptions.split_handover (off by default: measured slower) changes no bit of an fp32_class forward: the hidden tensors of
    every SK block and of the temporal MLP leave their producers as the (hi, lo) pair the consumers would have computed."""
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import EngineOptions, HotPathEngine
    B, T, h, w = 2, 4, 16, 24
    P = syn.make_params(5, T)
    fmaps, cnets = syn.make_features(5, B, T, h, w)
    outs = []
    for on in (False, True):
        eng = HotPathEngine(P, device=dev, T=T, use_graph=False, **dict(presets.engine_kwargs("fp32_class"), options=EngineOptions(split_handover=on)))
        ups, low = eng.forward(fmaps.to(dev), cnets.to(dev), iters=3)
        outs.append([u.cpu() for u in ups])
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and torch.equal(a, b)
