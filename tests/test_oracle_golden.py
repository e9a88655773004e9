"""CPU: pin the oracle (oracle/streamflow_oracle.py) against golden vectors produced by the
reference's own modules (tests/golden/make_golden.py).  No GPU, no /root/reference needed."""
import numpy as np
import pytest
import torch

from oracle import streamflow_oracle as orc
from tests import cases


def close(a, b, atol, rtol=0.0):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f"max err {err.max():.3e} (tol {atol:g}+{rtol:g}*|ref|), mean {err.mean():.3e}"


def test_coords_grid_bit_exact(golden):
    g = golden("coords_grid")
    out = orc.coords_grid(int(g["batch"]), int(g["ht"]), int(g["wd"]))
    assert np.array_equal(out.numpy(), g["out"])           # bit-exact requirement


def test_bilinear_sampler(golden):
    img, crd = cases.bilinear_inputs()
    close(orc.bilinear_sampler(img, crd), golden("bilinear_sampler")["out"], 2e-6)
    out, mask = orc.bilinear_sampler(img, crd, mask=True)
    g = golden("bilinear_sampler")["mask"]
    assert 0 < g.sum() < g.size                                  # the fixture has both inside and outside points
    assert np.array_equal(mask.numpy(), g)


@pytest.mark.parametrize("tag", list(cases.CORR_CASES))
def test_corr_pyramid_and_lookup(golden, tag):
    g = golden(tag)
    f1, f2, coords, ident = cases.corr_inputs(tag)
    pyr = orc.corr_pyramid(f1, f2)
    for i, lvl in enumerate(pyr):
        close(lvl, g[f"level{i}"], 2e-6, 1e-6)
    # level 0 is the same matmul -> expect bit equality with the reference here
    assert np.array_equal(pyr[0].numpy(), g["level0"])
    close(orc.corr_lookup(pyr, coords), g["lookup"], 5e-6)
    close(orc.corr_lookup(pyr, ident), g["lookup_identity"], 5e-6)


def test_lookup_channel_order_is_x_major(golden):
    """Channel l*81 + a*9 + b must read offset (dx=a-4, dy=b-4)."""
    f1, f2, _, ident = cases.corr_inputs("corr_odd")
    pyr = orc.corr_pyramid(f1, f2)
    out = orc.corr_lookup(pyr, ident)
    B, D, h, w, _ = cases.CORR_CASES["corr_odd"]
    vol = pyr[0].reshape(h, w, h, w)
    y, x, a, b = 8, 9, 6, 1                                   # dx=+2, dy=-3
    assert out[0, a * 9 + b, y, x].item() == pytest.approx(vol[y, x, y + (b - 4), x + (a - 4)].item(), abs=1e-6)


def test_gma(golden):
    g = golden("gma")
    P, inp, mf = cases.gma_inputs()
    attn = orc.gma_attention(inp, P["att.to_qk.weight"])
    close(attn, g["attn"], 1e-6, 1e-5)
    close(orc.gma_aggregate(torch.from_numpy(g["attn"]), mf, P["update_block.aggregator.to_v.weight"],
                            P["update_block.aggregator.gamma"]), g["aggregate"], 1e-5)


def test_skblocks(golden):
    from streamflow_amd import synthetic as syn
    g = golden("skblock")
    P = syn.make_params(cases.SKBLOCK_SEED, 4)
    for name, cin, cout, kc in cases.SKBLOCK_CASES:
        out = orc.skblock(cases.skblock_inputs(name, cin), P, "update_block." + name, kc)
        close(out, g[name.replace(".", "_")], 2e-5, 1e-5)


@pytest.mark.parametrize("tag", list(cases.UPDATE_CASES))
def test_update_block(golden, tag):
    g = golden(tag)
    B, T, h, w, _ = cases.UPDATE_CASES[tag]
    P, nets, inps, corrs, flows, attn = cases.update_inputs(tag)
    mf = orc.motion_encoder(flows, corrs, P, "update_block.encoder", (1, 15))
    close(mf, g["motion"], 3e-5, 1e-5)
    Pn = T - 1
    tok = mf.reshape(B, Pn, 128, h * w).permute(0, 3, 1, 2).reshape(B * h * w, Pn, 128)
    tok = orc.temporal_block(tok, P, "update_block.transformer_block.transformer_block")
    mft = tok.reshape(B, h * w, Pn, 128).permute(0, 2, 3, 1).reshape(B * Pn, 128, h, w)
    close(mft, g["temporal"], 3e-5, 1e-5)
    n2, masks, dflow = orc.update_block(nets, inps, corrs, flows, attn, Pn, P)
    close(n2, g["nets"], 5e-5, 1e-5)
    close(masks, g["masks"], 5e-5, 1e-5)
    close(dflow, g["dflow"], 5e-5, 1e-5)


def test_upsample(golden):
    flow, mask = cases.upsample_inputs()
    close(orc.upsample_flow(flow, mask), golden("upsample")["out"], 5e-6)


@pytest.mark.parametrize("tag", list(cases.FORWARD_CASES))
def test_full_forward(golden, tag):
    g = golden(tag)
    B, T, H, W, iters, seed, use_init = cases.FORWARD_CASES[tag]
    P, fmaps, cnets, finit, iters = cases.forward_inputs(tag)
    ups, low = orc.hotpath_forward(fmaps, cnets, P, iters, flow_init=finit)
    for i in range(T - 1):
        ref = torch.from_numpy(g[f"up{i}"])
        assert ups[i].shape == ref.shape == (B, 2, H, W)
        e = orc.epe(ups[i], ref)
        assert e < 1e-4, f"pair {i}: EPE vs reference forward {e:.3e}"
        if use_init:
            close(low[i], g[f"low{i}"], 1e-4)
    if "first0" in g:
        preds, _ = orc.hotpath_forward(fmaps, cnets, P, 1, all_iters=True)
        for i in range(T - 1):
            assert orc.epe(preds[i][0], torch.from_numpy(g[f"first{i}"])) < 1e-4


@pytest.mark.parametrize("tag", list(cases.INTERP_CASES))
def test_forward_interpolate(golden, tag):
    """f4: bit-exact against the reference's scipy griddata result (values are copies of input flow entries)."""
    out = orc.forward_interpolate(cases.interp_inputs(tag))
    assert torch.equal(out, torch.from_numpy(golden(tag)["out"]))


@pytest.mark.parametrize("tag", list(cases.TWINS_CASES))
def test_twins_csc_encoder(golden, tag):
    """f1: the restated Twins_CSC encoder against the reference's Twins_CSC.forward executed over the timm stand-in
    (token grids 64x24, 30x18 / 15x9: none a multiple of the 7x7 window, so the zero-padded window tokens matter)."""
    from oracle import twins_oracle as two
    P, x = cases.twins_inputs(tag)
    out = two.twins_csc_forward(x, P)
    g = golden(tag)["out"]
    B, T, H, W, _ = cases.TWINS_CASES[tag]
    assert tuple(out.shape) == g.shape == (B, T, 256, H // 8, W // 8)
    assert np.abs(g).mean() > 0.1
    close(out, g, 2e-4, 1e-4)
