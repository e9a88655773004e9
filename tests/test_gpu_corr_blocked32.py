"""Blocked fp32 correlation volumes (csrc/corr_blocked32.hip; reference core/corr.py:7-54 in its own fp32) through the C ABI:
build vs the oracle pyramid, lookup vs the oracle lookup on the stored cells and end to end, per-pair addressing, robustness
against garbage in the padding cells, equality with the pitched row-major fp32 path, the reference's golden volumes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


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
    c[:, :, 6, 6] = torch.tensor([5.25, 11.75])                      # ys % 4 == 3: the one-row fourth piece
    return c


# block-aligned, ragged in both dims, minimal (a 1-pixel coarsest level), odd pooled sizes (KITTI 47 x 156), D not a multiple of 32
SHAPES = [(1, 32, 16, 24), (2, 16, 17, 19), (1, 40, 16, 16), (1, 256, 47, 156), (3, 16, 16, 40), (1, 64, 33, 65), (2, 256, 24, 41)]


@pytest.mark.parametrize("B,D,h,w", SHAPES)
def test_blocked32_build_and_lookup_vs_oracle(dev, B, D, h, w):
    from oracle import streamflow_oracle as orc
    import streamflow_amd as sfa
    g = torch.Generator().manual_seed(B * 1000 + h * 10 + w)
    f1, f2 = torch.randn(B, D, h, w, generator=g), torch.randn(B, D, h, w, generator=g)
    coords = _coords(orc, B, h, w, g)
    blk = sfa.CorrBlock(f1.to(dev), f2.to(dev), num_levels=4, radius=4, layout="blocked")
    assert blk.vol.f32
    out = blk(coords.to(dev)).cpu()
    lv = [t.cpu() for t in blk.corr_pyramid]
    pyr = orc.corr_pyramid(f1.double(), f2.double(), 4)
    for l, (a, b_) in enumerate(zip(lv, pyr)):
        assert a.dtype == torch.float32 and a.shape == b_.shape, (l, a.shape, b_.shape)
        # split fp16 operands, three products: ~2^-20 of sum |a_k b_k| / sqrt(D) ~ 2^-20 * 0.8 sqrt(D) / sqrt(D)
        assert (a.double() - b_).abs().max().item() < 1e-5, (l, (a.double() - b_).abs().max().item())
    cc = coords.clone()
    cc[torch.isnan(cc)] = -1.0e6
    ref = orc.corr_lookup(lv, cc, 4)                                  # the lookup alone, on the stored cells
    assert torch.isfinite(out).all()
    # (the oracle follows the reference through grid_sample's normalised coordinates, utils.py:65-79: a round trip worth
    # ~2^-24 w in the sampling position, times the cell-to-cell differences of ~3)
    tol = 2e-5 * max(1.0, w / 32)
    assert (out - ref).abs().max().item() < tol
    ref64 = orc.corr_lookup([t.float() for t in pyr], cc, 4)          # end to end
    assert (out - ref64).abs().max().item() < tol + 1e-5


@pytest.mark.parametrize("B,pairs,D,h,w", [(2, 3, 32, 17, 28), (1, 1, 16, 47, 156), (1, 3, 256, 55, 128)])
def test_blocked32_padding_garbage_and_pairs(dev, B, pairs, D, h, w):
    """Neither the levels nor the looked-up features change when the volume buffer was full of NaN bit patterns before the build
    (padding cells of partial blocks are never read as data); image (clip b, pair t) is the volume of frames (t, t + 1)."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    n, N = B * pairs, h * w
    g = torch.Generator().manual_seed(7 + h)
    fm = torch.randn(B, pairs + 1, D, h, w, generator=g).to(dev)
    coords = _coords(orc, n, h, w, g, 3.0).to(dev).contiguous()
    outs = []
    for fill in (0x00, 0xFF):
        vol = ops.new_blocked_volume(n, h, w, dev, f32=True)
        vol.buf.fill_(fill)
        ops.corr_build_blocked(fm.data_ptr(), fm.data_ptr() + 4 * D * N, (pairs + 1) * D * N, D * N, vol, B, pairs, D)
        out = torch.full((n, 324, N), float("nan"), device=dev)
        ops.corr_lookup_blocked(vol, Planes.of(coords), Planes.of(out), None, B, pairs)
        torch.cuda.synchronize()
        outs.append((out.cpu(), [t.cpu() for t in vol.levels()]))
    a, b = outs
    assert torch.isfinite(a[0]).all() and torch.equal(a[0], b[0])
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y)
    fmc = fm.cpu()
    for (b_, t) in ((0, 0), (B - 1, pairs - 1)):
        pyr = orc.corr_pyramid(fmc[b_:b_ + 1, t].double(), fmc[b_:b_ + 1, t + 1].double(), 4)
        for l in range(4):
            got = a[1][l][b_ * pairs + t].double().reshape(pyr[l].shape)
            assert (got - pyr[l]).abs().max().item() < 1e-5, (b_, t, l)


def test_blocked32_matches_pitched_row_major_path(dev):
    """Same arithmetic (split fp16 operands, three products in the same order per k-step), two layouts: cells and features agree
    to fp32 rounding of the accumulation order."""
    import streamflow_amd as sfa
    from oracle import streamflow_oracle as orc
    g = torch.Generator().manual_seed(3)
    f1, f2 = torch.randn(2, 256, 23, 37, generator=g).to(dev), torch.randn(2, 256, 23, 37, generator=g).to(dev)
    a = sfa.CorrBlock(f1, f2)
    b = sfa.CorrBlock(f1, f2, layout="blocked")
    for x, y in zip(a.corr_pyramid, b.corr_pyramid):
        assert (x - y).abs().max().item() < 2e-6
    c = _coords(orc, 2, 23, 37, g).to(dev)
    assert (a(c) - b(c)).abs().max().item() < 2e-5


@pytest.mark.parametrize("tag", ["corr_odd", "corr_b2"])
def test_blocked32_vs_reference_golden_volume(dev, tag):
    """The reference's own CorrBlock outputs (tests/golden/corr_*.npz, generated by tests/golden/make_golden.py)."""
    from tests import cases
    from streamflow_amd.corr import CorrBlock
    g = np.load(os.path.join(GOLDEN, tag + ".npz"))
    f1, f2, coords, ident = cases.corr_inputs(tag)
    blk = CorrBlock(f1.to(dev), f2.to(dev), num_levels=4, radius=4, layout="blocked")
    for i, lvl in enumerate(blk.corr_pyramid):
        ref = torch.from_numpy(g[f"level{i}"])
        assert lvl.shape == ref.shape and (lvl.cpu() - ref).abs().max().item() < 2e-5, (tag, i)
    assert (blk(coords.to(dev)).cpu() - torch.from_numpy(g["lookup"])).abs().max().item() < 2e-5
    assert (blk(ident.to(dev)).cpu() - torch.from_numpy(g["lookup_identity"])).abs().max().item() < 2e-5


def test_blocked32_is_deterministic_at_the_kitti_shape(dev):
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    B, pairs, D, h, w = 1, 1, 256, 47, 156
    N = h * w
    g = torch.Generator().manual_seed(5)
    fm = torch.randn(B, 2, D, h, w, generator=g).to(dev)
    coords = (torch.rand(1, 2, h, w, generator=g) * torch.tensor([w, h]).view(1, 2, 1, 1)).to(dev).contiguous()
    vol = ops.new_blocked_volume(1, h, w, dev, f32=True)
    res = []
    for _ in range(3):
        vol.buf.zero_()
        ops.corr_build_blocked(fm.data_ptr(), fm.data_ptr() + 4 * D * N, 2 * D * N, D * N, vol, B, pairs, D)
        out = torch.empty(1, 324, N, device=dev)
        ops.corr_lookup_blocked(vol, Planes.of(coords), Planes.of(out), None, B, pairs)
        torch.cuda.synchronize()
        res.append((vol.buf.clone(), out))
    for v, o in res[1:]:
        assert torch.equal(v, res[0][0]) and torch.equal(o, res[0][1])


def test_blocked32_at_the_spring_grid_last_pair(dev):
    """1080p feature grid (136 x 240, N = 32,640): 5.6 GB of blocked fp32 pyramids per pair -- 64-bit image / record addressing.
    Records of source pixels of the LAST of two pairs (highest addresses) against a direct float64 contraction and their own
    2 x 2 pooling; lookups of the same pixels against the oracle lookup on those records."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    B, pairs, D, h, w = 1, 2, 256, 136, 240
    N, n = h * w, 2
    g = torch.Generator().manual_seed(11)
    fm = torch.randn(B, pairs + 1, D, h, w, generator=g)
    fmd = fm.to(dev)
    vol = ops.new_blocked_volume(n, h, w, dev, f32=True)
    assert vol.img_stride > (1 << 32)
    ops.corr_build_blocked(fmd.data_ptr(), fmd.data_ptr() + 4 * D * N, (pairs + 1) * D * N, D * N, vol, B, pairs, D)
    coords = (orc.coords_grid(n, h, w) + torch.randn(n, 2, h, w, generator=g) * 6.0).contiguous()
    out = torch.empty(n, 324, N, device=dev)
    ops.corr_lookup_blocked(vol, Planes.of(coords.to(dev)), Planes.of(out), None, B, pairs)
    torch.cuda.synchronize()
    img = 1
    f2 = fm[0, 2].reshape(D, N).double()
    for i in (0, 12345, N - 1):
        rec = torch.as_strided(vol.buf, (vol.rec,), (1,), img * vol.img_stride + i * vol.rec).clone().cpu()
        lv = []
        for l in range(4):
            nby, nbx = vol.nby[l], vol.nbx[l]
            blk = rec[vol.off[l]: vol.off[l] + nby * nbx * 128].view(torch.float32).view(nby, nbx, 8, 4)    # [by][bx][tx % 8][ty % 4]
            lv.append(blk.permute(0, 3, 1, 2).reshape(nby * 4, nbx * 8)[: h >> l, : w >> l])
        ref = (fm[0, 1].reshape(D, N)[:, i].double() @ f2) / 16.0
        assert (lv[0].reshape(-1).double() - ref).abs().max().item() < 1e-5
        for l in range(3):
            hl, wl = h >> (l + 1), w >> (l + 1)
            pooled = lv[l][: 2 * hl, : 2 * wl].reshape(hl, 2, wl, 2).mean(dim=(1, 3))
            assert (pooled - lv[l + 1]).abs().max().item() < 1e-6, (i, l)
        c = coords[img, :, i // w, i % w].reshape(1, 2, 1, 1)
        want = orc.corr_lookup([t.reshape(1, 1, *t.shape) for t in lv], c, 4).reshape(324)
        assert (out[img, :, i].cpu() - want).abs().max().item() < 2e-4, i
    del vol, out
    torch.cuda.empty_cache()
