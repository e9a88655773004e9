"""sf_mask_upsample (csrc/mask_upsample.hip): the mask head's second layer (core/update.py:758,777) + convex upsampling
(core/models/streamflow.py:82-93) as ONE launch, through the C ABI (-m gpu): against float64 on the same fp16-rounded operand, against
the two launches it replaces (sf_gemm + sf_upsample_flow), ragged grids, one and two products; and the engine with and without it."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _weff(A, single):
    hi = A.hi.float().permute(1, 0, 2).reshape(A.lda_h, -1)[: A.M, : A.K].double().cpu()
    lo = A.lo.float().permute(1, 0, 2).reshape(A.lda_h, -1)[: A.M, : A.K].double().cpu()
    return (hi if single else hi + lo) / A.split_scale


def _ref_upsample(flow, mask):
    """streamflow.py:82-93 in float64: flow [n, 2, h, w], mask [n, 576, h, w]."""
    n, _, h, w = flow.shape
    m = torch.softmax(mask.view(n, 1, 9, 8, 8, h, w), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, w)
    return torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(n, 2, 8 * h, 8 * w)


@pytest.mark.parametrize("pm", [1, 2])
@pytest.mark.parametrize("hw", [(55, 128), (47, 156), (9, 13), (32, 32)])
def test_mask_upsample_vs_float64_and_two_launches(dev, hw, pm):
    from dataclasses import replace
    from streamflow_amd import ops
    from streamflow_amd.ops import PackedLinear, PackedMask, Planes
    h, w = hw
    n, P = 3, h * w
    g = torch.Generator().manual_seed(h * 7 + w + pm)
    W2, b2 = torch.randn(576, 256, generator=g) / 16 * 3, torch.randn(576, generator=g)
    A = PackedLinear(W2.view(576, 256, 1, 1), b2, dev)
    A.single = pm == 1
    pack = PackedMask(A)
    x = torch.relu(torch.randn(n, 256, P, generator=g))
    flow = torch.randn(n, 2, h, w, generator=g) * 5
    X = Planes.of(x.to(dev).contiguous())
    sh = ops.new_shadow(X, dev)
    ops.pack_koct(X, sh)
    X = replace(X, shadow=sh)
    cx = ops.Ctx(precision=ops.PRECISION_F16X2)
    assert ops.mask_upsample_ok(pack, X, cx)
    out = torch.full((n, 2, 8 * h, 8 * w), float("nan"), device=dev)
    ops.mask_upsample(pack, X, flow.to(dev).contiguous(), out, h, w, cx=cx)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all()), "a cell of the upsampled flow was not written"
    mask64 = 0.25 * (torch.einsum("mk,nkp->nmp", _weff(A, pm == 1), x.half().double()) + b2.double()[None, :, None])
    ref = _ref_upsample(flow.double(), mask64.view(n, 576, h, w))
    err = (out.double().cpu() - ref).abs().max().item()
    assert err <= 2e-4 * max(1.0, ref.abs().max().item()), (hw, pm, err)
    # the two launches: sf_gemm (mask.2, alpha 0.25) + sf_upsample_flow on the same operand
    mk = Planes.of(torch.empty(n, 576, P, device=dev))
    ops.gemm(A, X, mk, ops.EPI_NONE, alpha=0.25, cx=cx)
    up2 = ops.upsample_flow(flow.to(dev).contiguous(), mk.tensor().view(n, 576, h, w))
    torch.cuda.synchronize()
    assert (up2 - out).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


def test_engine_with_and_without_the_fused_mask_head(dev):
    """Same flows from the engine with EngineOptions.mask_upsample on and off (graph replay), to the softmax's rounding."""
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import EngineOptions, HotPathEngine
    T, B, h, w = 4, 2, 32, 48
    params = syn.make_params(5, T)
    fm, cn = (t.to(dev) for t in syn.make_features(78, B, T, h, w))
    kw = presets.engine_kwargs("config2_mixed")
    outs = []
    for on in (True, False):
        eng = HotPathEngine(params, device=dev, T=T, use_graph=True, options=EngineOptions(mask_upsample=on), **kw)
        eng.forward(fm, cn, iters=3)
        outs.append([f.clone() for f in eng.forward(fm, cn, iters=3)[0]])
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())


def test_mask_upsample_rejects_what_it_was_not_built_for(dev):
    import ctypes as Ct
    from streamflow_amd import _lib
    lib = _lib.load()
    g = _lib.SfMaskUpsample()
    assert lib.sf_mask_upsample(Ct.byref(g), None) != 0
    g.X16, g.strideX, g.ldx, g.wstream, g.wstream_bytes = 0x1000, 256 * 64, 64, 0x2000, 288 * 1024
    g.flow, g.out, g.n_img, g.h, g.w, g.K, g.M, g.pm, g.alpha = 0x3000, 0x4000, 1, 8, 8, 256, 512, 1, 0.25
    assert lib.sf_mask_upsample(Ct.byref(g), None) != 0 and b"built for" in lib.sf_last_error()
    g.M, g.wstream_bytes = 576, 1024
    assert lib.sf_mask_upsample(Ct.byref(g), None) != 0 and b"weight stream size" in lib.sf_last_error()
    assert lib.sf_mask_upsample_frags(1) == 288 and lib.sf_mask_upsample_frags(2) == 576 and lib.sf_mask_upsample_frags(0) == 0
