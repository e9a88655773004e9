"""GPU parity tests proper: every HIP kernel, called through the C ABI, against
(i) the committed golden vectors produced by the reference and (ii) the CPU oracle on the same
seeded inputs.  Run on the MI355X box with `pytest -m gpu`.

Tolerances (fp32 path, exact-fp32 MFMA; differences come from summation order and from the
reference's normalise/unnormalise round trip of sampling coordinates):
  coords_grid: bit-exact;  corr volume/pyramid: 2e-5 abs + 1e-5 rel;  lookup: 2e-5;
  GMA / SKBlock / update block tensors: 1e-4 abs + 1e-4 rel;  final flows: EPE <= 1e-3 px
  (the north-star bound; observed values are printed).
"""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def close(a, b, atol, rtol=0.0, what=""):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), f"{what}: non-finite values in HIP output"
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} elements off; max err {err.max():.3e} at "
                           f"{np.unravel_index(err.argmax(), err.shape)} (tol {atol:g}+{rtol:g}*|ref|), mean {err.mean():.3e}")


@pytest.fixture(scope="module")
def dev():
    return _gpu()


@pytest.fixture(params=["fp32", "f16x3"])
def precision(request):
    """Every GEMM-shaped kernel is checked in both arithmetic modes (exact fp32 MFMA / split fp16x3)."""
    import streamflow_amd as sfa
    prev = sfa.set_precision(request.param)
    yield request.param
    sfa.set_precision(prev)


def test_library_loaded_is_in_tree():
    import os
    from streamflow_amd import _lib
    _lib.load()
    assert os.path.samefile(os.path.dirname(_lib.LIB_PATH), os.path.join(os.path.dirname(__file__), "..", "streamflow_amd"))
    with open("/proc/self/maps") as f:
        assert "libstreamflow_hip.so" in f.read()


def test_coords_grid_bit_exact(golden, dev):
    from streamflow_amd import ops
    g = golden("coords_grid")
    out = ops.coords_grid(int(g["batch"]), int(g["ht"]), int(g["wd"]), dev)
    assert np.array_equal(out.cpu().numpy(), g["out"])
    big = ops.coords_grid(3, 55, 128, dev).cpu()
    from oracle import streamflow_oracle as orc
    assert torch.equal(big, orc.coords_grid(3, 55, 128))


def test_bilinear_sampler(golden, dev):
    from streamflow_amd import ops
    img, crd = cases.bilinear_inputs()
    out, mask = ops.bilinear_sampler(img.to(dev), crd.to(dev).contiguous(), want_mask=True)
    close(out, golden("bilinear_sampler")["out"], 5e-6, what="bilinear_sampler")
    assert mask.shape == (5, 4, 9, 1)
    assert np.array_equal(mask.cpu().numpy(), golden("bilinear_sampler")["mask"])      # utils.py:75-77, strict bounds


@pytest.mark.parametrize("tag", list(cases.CORR_CASES))
def test_corr_build_and_lookup_vs_golden(golden, dev, tag, precision):
    from streamflow_amd.corr import CorrBlock
    g = golden(tag)
    f1, f2, coords, ident = cases.corr_inputs(tag)
    blk = CorrBlock(f1.to(dev), f2.to(dev), num_levels=4, radius=4)
    for i, lvl in enumerate(blk.corr_pyramid):
        close(lvl, g[f"level{i}"], 2e-5, 1e-5, what=f"{tag} level{i}")
    close(blk(coords.to(dev)), g["lookup"], 2e-5, what=f"{tag} lookup")
    close(blk(ident.to(dev)), g["lookup_identity"], 2e-5, what=f"{tag} lookup identity")
    vol = CorrBlock.corr(f1.to(dev), f2.to(dev))
    B, D, h, w, _ = cases.CORR_CASES[tag]
    assert vol.shape == (B, h, w, 1, h, w)
    close(vol.reshape(B * h * w, 1, h, w), g["level0"], 2e-5, 1e-5, what=f"{tag} CorrBlock.corr")


def test_gma_vs_golden(golden, dev, precision):
    from streamflow_amd.gma import Attention, Aggregate
    g = golden("gma")
    P, inp, mf = cases.gma_inputs()
    att = Attention(args=None, dim=128, heads=1, max_pos_size=160, dim_head=128).to(dev)
    att.load_state_dict({"to_qk.weight": P["att.to_qk.weight"]}, strict=True)
    agg = Aggregate(args=None, dim=128, dim_head=128, heads=1).to(dev)
    agg.load_state_dict({"to_v.weight": P["update_block.aggregator.to_v.weight"],
                         "gamma": P["update_block.aggregator.gamma"]}, strict=True)
    attn = att(inp.to(dev))
    close(attn, g["attn"], 1e-6, 1e-4, what="attention")
    close(agg(torch.from_numpy(g["attn"]).to(dev), mf.to(dev)), g["aggregate"], 1e-4, 1e-4, what="aggregate")


def _sub(params, prefix):
    return {k[len(prefix) + 1:]: v for k, v in params.items() if k.startswith(prefix + ".")}


def test_skblocks_vs_golden(golden, dev, precision):
    from streamflow_amd import synthetic as syn
    from streamflow_amd.update import PCBlock4_Deep_nopool_res
    g = golden("skblock")
    P = syn.make_params(cases.SKBLOCK_SEED, 4)
    for name, cin, cout, kc in cases.SKBLOCK_CASES:
        m = PCBlock4_Deep_nopool_res(cin, cout, list(kc)).to(dev)
        m.load_state_dict(_sub(P, "update_block." + name), strict=True)
        out = m(cases.skblock_inputs(name, cin).to(dev))
        close(out, g[name.replace(".", "_")], 1e-4, 1e-4, what="skblock " + name)


@pytest.mark.parametrize("tag", list(cases.UPDATE_CASES))
def test_update_block_vs_golden(golden, dev, tag, precision):
    from argparse import Namespace
    from streamflow_amd.update import SKUpdateBlock_TAM_v3
    g = golden(tag)
    B, T, h, w, _ = cases.UPDATE_CASES[tag]
    P, nets, inps, corrs, flows, attn = cases.update_inputs(tag)
    args = Namespace(decoder_dim=256, corr_levels=4, corr_radius=4, k_conv=[1, 15], PCUpdater_conv=[1, 7], T=T,
                     use_gma=True, num_heads=1, Encoder="InjectEncoder")
    ub = SKUpdateBlock_TAM_v3(args).to(dev)
    ub.load_state_dict(_sub(P, "update_block"), strict=True)
    mf = ub.encoder(flows.to(dev), corrs.to(dev))
    close(mf, g["motion"], 1e-4, 1e-4, what=tag + " motion encoder")
    n2, masks, dflow = ub(nets.to(dev), inps.to(dev), corrs.to(dev), flows.to(dev), attn.to(dev), T=T - 1)
    close(n2, g["nets"], 2e-4, 1e-4, what=tag + " nets")
    close(masks, g["masks"], 2e-4, 1e-4, what=tag + " masks")
    close(dflow, g["dflow"], 2e-4, 1e-4, what=tag + " delta flow")


def test_upsample_vs_golden(golden, dev):
    from streamflow_amd import ops
    flow, mask = cases.upsample_inputs()
    close(ops.upsample_flow(flow.to(dev), mask.to(dev)), golden("upsample")["out"], 2e-5, what="upsample")


@pytest.mark.parametrize("tag", list(cases.FORWARD_CASES))
@pytest.mark.parametrize("graph", [False, True])
def test_engine_forward_vs_golden(golden, dev, tag, graph, precision):
    """The fused engine against the reference's SKFlow_MF8.forward outputs (golden)."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd.engine import HotPathEngine
    g = golden(tag)
    B, T, H, W, iters, seed, use_init = cases.FORWARD_CASES[tag]
    P, fmaps, cnets, finit, iters = cases.forward_inputs(tag)
    eng = HotPathEngine(P, device=dev, T=T, use_graph=graph, precision=precision)
    finit_d = None if finit is None else [f.to(dev) for f in finit]
    for rep in range(2):                                    # second call exercises graph replay / buffer reuse
        ups, low = eng.forward(fmaps.to(dev), cnets.to(dev), iters=iters, flow_init=finit_d)
        for i in range(T - 1):
            e = orc.epe(ups[i].cpu(), torch.from_numpy(g[f"up{i}"]))
            print(f"{tag} {precision} graph={graph} rep={rep} pair {i}: EPE vs reference = {e:.3e}")
            assert e <= 1e-3, f"{tag} pair {i}: EPE {e}"
            if use_init:
                close(low[i], g[f"low{i}"], 1e-3, what=f"{tag} lowres {i}")


@pytest.mark.parametrize("tag", list(cases.FORWARD_CASES))
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("preset", ["config2_fp16", "config2_mixed"])
def test_engine_forward_config2_presets_vs_golden(golden, dev, tag, graph, preset):
    """the kernels the HEADLINE presets run (activation-stationary GEMMs, FFN pairs, the one-launch temporal
    block, blocked fp16 volumes, fp16-input depthwise, mask head -> upsampling) against the reference's own SKFlow_MF8.forward
    outputs (golden fixtures), not only against the oracle.  Bound: the class's 1e-3 px (flows here are 1-3 px)."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import presets
    from streamflow_amd.engine import HotPathEngine
    g = golden(tag)
    B, T, H, W, iters, seed, use_init = cases.FORWARD_CASES[tag]
    P, fmaps, cnets, finit, iters = cases.forward_inputs(tag)
    eng = HotPathEngine(P, device=dev, T=T, use_graph=graph, **presets.engine_kwargs(preset))
    finit_d = None if finit is None else [f.to(dev) for f in finit]
    for rep in range(2):
        ups, low = eng.forward(fmaps.to(dev), cnets.to(dev), iters=iters, flow_init=finit_d)
        for i in range(T - 1):
            e = orc.epe(ups[i].cpu(), torch.from_numpy(g[f"up{i}"]))
            print(f"{tag} {preset} graph={graph} rep={rep} pair {i}: EPE vs reference = {e:.3e}")
            assert e <= 1e-3, f"{tag} {preset} pair {i}: EPE {e}"
            if use_init:
                close(low[i], g[f"low{i}"], 2e-3, what=f"{tag} {preset} lowres {i}")


@pytest.mark.parametrize("tag", list(cases.UPDATE_CASES))
def test_update_block_f16x2_vs_golden(golden, dev, tag):
    """The update block's API class in the config-2 arithmetic (f16x2: fp16 activations into every product, weights hi + lo) against
    the reference's SKUpdateBlock_TAM_v3.forward golden: tensor tolerances of the class (a chain of ~25 contractions whose operands are
    rounded to 2^-11: 5e-3 abs + 5e-3 rel on O(1) tensors; observed values are printed)."""
    from argparse import Namespace
    import streamflow_amd as sfa
    from streamflow_amd.update import SKUpdateBlock_TAM_v3
    g = golden(tag)
    B, T, h, w, _ = cases.UPDATE_CASES[tag]
    P, nets, inps, corrs, flows, attn = cases.update_inputs(tag)
    args = Namespace(decoder_dim=256, corr_levels=4, corr_radius=4, k_conv=[1, 15], PCUpdater_conv=[1, 7], T=T,
                     use_gma=True, num_heads=1, Encoder="InjectEncoder")
    prev = sfa.set_precision("f16x2")
    try:
        ub = SKUpdateBlock_TAM_v3(args).to(dev)
        ub.load_state_dict(_sub(P, "update_block"), strict=True)
        mf = ub.encoder(flows.to(dev), corrs.to(dev))
        n2, masks, dflow = ub(nets.to(dev), inps.to(dev), corrs.to(dev), flows.to(dev), attn.to(dev), T=T - 1)
    finally:
        sfa.set_precision(prev)
    for name, out, ref in (("motion", mf, g["motion"]), ("nets", n2, g["nets"]), ("masks", masks, g["masks"]), ("dflow", dflow, g["dflow"])):
        err = float(np.abs(out.detach().float().cpu().numpy() - ref).max())
        print(f"{tag} f16x2 {name}: max |err| vs reference = {err:.3e} (|ref| max {float(np.abs(ref).max()):.2f})")
        close(out, ref, 5e-3, 5e-3, what=f"{tag} f16x2 {name}")


def test_model_api_forward_vs_golden(golden, dev, precision):
    """SKFlow_MF8 with the reference's signature (list of frames in 0..255, test_mode), stand-in encoder."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd.model import SKFlow_MF8, default_args
    tag = "forward_T4"
    g = golden(tag)
    B, T, H, W, iters, seed, use_init = cases.FORWARD_CASES[tag]
    P, fmaps, cnets, finit, iters = cases.forward_inputs(tag)
    model = SKFlow_MF8(default_args(T=T, Encoder="InjectEncoder")).to(dev)
    model.load_state_dict(dict(P), strict=True)
    model.fnet.features, model.cnet.features = fmaps.to(dev), cnets.to(dev)
    images = [torch.zeros(B, 3, H, W, device=dev) for _ in range(T)]
    ups = model(images, iters=iters, test_mode=True)
    assert len(ups) == T - 1 and ups[0].shape == (B, 2, H, W)
    for i in range(T - 1):
        assert orc.epe(ups[i].cpu(), torch.from_numpy(g[f"up{i}"])) <= 1e-3
    allp = model(images, iters=2, test_mode=False)          # training-mode return: per pair, per iteration
    assert len(allp) == T - 1 and len(allp[0]) == 2
    for i in range(T - 1):
        assert orc.epe(allp[i][0].cpu(), torch.from_numpy(g[f"first{i}"])) <= 1e-3


def test_f16x2_mode_within_parity_budget(dev):
    """f16x2 (activations rounded once to fp16, weights split) is looser than f16x3 but must stay inside the 1e-3 px EPE
    budget of the north star; 15 iterations at a small shape, against the CPU oracle.  The single-product `f16` mode
    (weights rounded to fp16 too: the arithmetic of an fp16-autocast deployment) is NOT parity-grade -- 3e-3 px here,
    2.5e-3 px at the headline shape -- and is only checked to stay in that class: it exists to document why the weights
    keep their lo part (245 vs 215 flow-fields/s would be the price of the budget)."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    B, T, h, w = 1, 4, 16, 24
    P = syn.make_params(5, T)
    fmaps, cnets = syn.make_features(5, B, T, h, w)
    ups_o, _ = orc.hotpath_forward(fmaps, cnets, P, 15)
    for prec, bound in (("f16", 1e-2), ("f16x2", 1e-3), ("f16x3", 5e-5)):
        eng = HotPathEngine(P, device=dev, T=T, precision=prec)
        ups, _ = eng.forward(fmaps.to(dev), cnets.to(dev), iters=15)
        e = max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))
        print(f"{prec}: 15-iteration EPE vs oracle = {e:.3e}")
        assert e <= bound, (prec, e)


def test_sintel_shape_vs_oracle(dev, precision):
    """Headline shape (440x1024 -> 55x128 grid, T=4), 2 iterations, against the CPU oracle; also the
    size-independent properties: level-1 == mean of level-0 2x2 blocks, attention rows sum to 1."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    B, T, h, w = 1, 4, 55, 128
    P = syn.make_params(7, T)
    fmaps, cnets = syn.make_features(7, B, T, h, w)
    eng = HotPathEngine(P, device=dev, T=T)
    ups, low = eng.forward(fmaps.to(dev), cnets.to(dev), iters=2)
    ups = [u.cpu() for u in ups]
    pl = eng.plan(B, h, w, 256)
    N = h * w
    l0 = pl.level_maps(0)[0]
    l1 = pl.level_maps(1)[0]
    pooled = l0[:, : 2 * (h // 2), :].reshape(N, h // 2, 2, w // 2, 2).mean(dim=(2, 4))
    assert (pooled - l1).abs().max().item() < 1e-5
    if pl.attn16 is not None:            # split precisions keep the attention weights in fp16 (engine._Plan)
        rows = pl.attn16[0].float().sum(dim=-1)
        assert (rows - 1).abs().max().item() < 2e-3
    else:
        rows = pl.attn[0].sum(dim=-1)
        assert (rows - 1).abs().max().item() < 1e-4
    torch.set_num_threads(max(1, torch.get_num_threads()))
    ups_o, low_o = orc.hotpath_forward(fmaps, cnets, P, 2)
    for i in range(T - 1):
        e = orc.epe(ups[i], ups_o[i])
        print(f"sintel-shape {precision} pair {i}: EPE vs oracle = {e:.3e}")
        assert e <= 1e-3


_HEADLINE_ORACLE = {}


@pytest.mark.parametrize("preset", ["config2_mixed", "config2_fp16", "fp32_class"])
def test_headline_config_batched_vs_oracle(dev, preset):
    """BASELINE.json's headline configuration exactly as bench.py times it: 440x1024 (55x128 grid), T=4, ALL 15
    iterations, 8 clips batched through every launch (the per-GPU share of config 4's batch 64), HIP-graph replay, in
    both named arithmetic configurations (streamflow_amd/presets.py: the bench default `config2_fp16` and the library
    default `fp32_class`).  The first and the LAST clip of the batch (image indices 0-2 and 21-23: the highest buffer
    offsets of the multi-GB volume / attention allocations) are compared with the CPU oracle run on those clips alone."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    B, T, h, w, iters = 8, 4, 55, 128, 15
    P = syn.make_params(0, T)
    fmaps, cnets = syn.make_features(1000, B, T, h, w)
    eng = HotPathEngine(P, device=dev, T=T, use_graph=True, **presets.engine_kwargs(preset))
    fd, cd = fmaps.to(dev), cnets.to(dev)
    eng.forward(fd, cd, iters=iters)                       # capture
    ups, low = eng.forward(fd, cd, iters=iters)            # replay
    ups = [u.cpu() for u in ups]
    assert all(torch.isfinite(u).all() for u in ups)
    for clip in (0, B - 1):
        if clip not in _HEADLINE_ORACLE:                          # ~20 s of CPU per clip: shared by the two presets
            _HEADLINE_ORACLE[clip] = orc.hotpath_forward(fmaps[clip:clip + 1], cnets[clip:clip + 1], P, iters)[0]
        ups_o = _HEADLINE_ORACLE[clip]
        for i in range(T - 1):
            e = orc.epe(ups[i][clip:clip + 1], ups_o[i])
            mag = ups_o[i].norm(dim=1).mean().item()
            print(f"headline config [{preset}], clip {clip} of {B}, pair {i}: EPE vs oracle after {iters} iterations = {e:.3e} px "
                  f"(mean |flow| {mag:.2f} px)")
            # north star: 1e-3 px; the mixed preset was chosen by ablation to stay below half of that
            assert e <= (5e-4 if preset == "config2_mixed" else 1e-3), (clip, i, e)
    # clips are independent: the same clip at another batch position must give the same flows
    fd2 = fd.clone()
    fd2[3], cd2 = fd[B - 1], cd.clone()
    cd2[3] = cd[B - 1]
    ups2, _ = eng.forward(fd2, cd2, iters=iters)
    for i in range(T - 1):
        assert orc.epe(ups2[i][3:4].cpu(), ups[i][B - 1:B]) <= 1e-4


def test_preset_selected_through_reference_args(dev):
    """The arithmetic preset is chosen the way the reference chooses its own (args.mixed_precision, evaluate_mf.py:1106)
    or by name (args.preset); the engine the model builds carries exactly that preset's settings."""
    import streamflow_amd as sfa
    from streamflow_amd import ops, presets, synthetic as syn
    P = syn.make_params(2, 4)
    for kw, want in ((dict(), "fp32_class"), (dict(mixed_precision=True), presets.MODEL_MIXED_PRESET),
                     (dict(preset="config2_fp16"), "config2_fp16"), (dict(preset="config2_mixed"), "config2_mixed")):
        m = sfa.SKFlow_MF8(sfa.default_args(T=4, Encoder="InjectEncoder", **kw))
        m.load_state_dict(dict(P), strict=True)
        assert m.preset_name() == want
        eng = m.engine(dev)
        cfg = presets.engine_kwargs(want)
        assert eng.precision == ops._PRECISION_NAMES[cfg["precision"]] and eng.corr_f16 == (cfg["corr_dtype"] == "f16")
        assert eng.single_layers == tuple(cfg.get("single_layers", ()))
        n_dw = sum(getattr(eng.W, b).dw_single for b in eng.W.SK_BLOCKS)
        assert sum(pl.single for pl in eng.W.layers().values()) + n_dw == len(cfg.get("single_layers", ()))
    assert sfa.StreamFlowT4(None, Encoder="InjectEncoder").preset_name() == presets.MODEL_MIXED_PRESET == "config2_fp16"
    assert sfa.StreamFlowT4(None, Encoder="InjectEncoder", preset=presets.BENCH_PRESET).preset_name() == "config2_mixed"
    with pytest.raises(RuntimeError, match="unknown preset"):
        sfa.SKFlow_MF8(sfa.default_args(T=4, Encoder="InjectEncoder", preset="nope")).engine(dev)


def test_single_product_layers_are_what_they_say(dev):
    """A layer marked single-product multiplies by the round-to-nearest fp16 image of its (scaled) weights alone: equal to
    the two-product GEMM run on weights that ARE fp16-representable, and different from the split-weight result."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes, PackedLinear
    g = torch.Generator().manual_seed(5)
    M, K, P_, n = 384, 256, 512, 2
    Wt = torch.randn(M, K, generator=g) / K ** 0.5
    X = torch.randn(n, K, P_, generator=g).to(dev)
    prev = ops.set_precision("f16x2")
    try:
        A = PackedLinear(Wt.reshape(M, K, 1, 1), None, dev)
        Y2, Y1, Yr = (torch.empty(n, M, P_, device=dev) for _ in range(3))
        ops.gemm(A, Planes.of(X), Planes.of(Y2))
        A.single = True
        ops.gemm(A, Planes.of(X), Planes.of(Y1))
        Ar = PackedLinear((A.hi.float().permute(1, 0, 2).reshape(A.lda_h, -1)[:M, :K] / A.split_scale).reshape(M, K, 1, 1), None, dev)
        ops.gemm(Ar, Planes.of(X), Planes.of(Yr))
        torch.cuda.synchronize()
    finally:
        ops.set_precision(prev)
    assert (Y1 - Yr).abs().max().item() < 1e-5
    assert (Y1 - Y2).abs().max().item() > 1e-5


def test_fp16_volume_mode_within_parity_budget(dev):
    """corr_dtype='f16' (fp16 correlation pyramids built with single f16 products; BASELINE configs 2 and 5): the
    final flows must stay inside the 1e-3 px budget against the fp32 oracle -- 15 iterations at a small shape, and the
    full Sintel shape (55x128 grid, 15 iterations, one clip)."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    for (B, T, h, w, seed) in ((1, 4, 16, 24, 5), (1, 4, 55, 128, 7)):
        P = syn.make_params(seed, T)
        fmaps, cnets = syn.make_features(seed, B, T, h, w)
        ups_o, _ = orc.hotpath_forward(fmaps, cnets, P, 15)
        eng = HotPathEngine(P, device=dev, T=T, corr_dtype="f16")
        ups, _ = eng.forward(fmaps.to(dev), cnets.to(dev), iters=15)
        assert eng.plan(B, h, w, 256).lvls[0].dtype == torch.float16
        e = max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))
        print(f"fp16 volumes, {h}x{w} grid: 15-iteration EPE vs fp32 oracle = {e:.3e}")
        assert e <= 1e-3, (h, w, e)


def test_engine_on_non_current_device():
    """An engine built for cuda:1 while cuda:0 is the current device must launch on cuda:1's streams."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from oracle import streamflow_oracle as orc
    from streamflow_amd import synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    torch.cuda.set_device(0)
    d1 = torch.device("cuda:1")
    B, T, h, w = 1, 4, 16, 24
    P = syn.make_params(3, T)
    fmaps, cnets = syn.make_features(3, B, T, h, w)
    for graph in (False, True):
        eng = HotPathEngine(P, device=d1, T=T, use_graph=graph)
        ups, _ = eng.forward(fmaps.to(d1), cnets.to(d1), iters=3)
        ups_o, _ = orc.hotpath_forward(fmaps, cnets, P, 3)
        assert torch.cuda.current_device() == 0
        assert max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o)) <= 1e-3


@pytest.mark.parametrize("preset", ["fp32_class", "config2_fp16", "config2_mixed"])
def test_kitti_shape_T2_vs_oracle(dev, preset):
    """BASELINE config 3: KITTI shape 376x1248 -> 47x156 grid (odd height, width not a multiple of 32, pooled levels of
    odd width 39 / 19), T=2 (one pair, single-token temporal block, 128->2 flow head), corr build + lookup + 2 iterations
    vs the oracle, in both presets (fp32 volumes: tight; fp16 volumes: one fp16 rounding of each stored cell)."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    B, T, h, w = 1, 2, 47, 156
    P = syn.make_params(13, T)
    fmaps, cnets = syn.make_features(13, B, T, h, w)
    eng = HotPathEngine(P, device=dev, T=T, **presets.engine_kwargs(preset))
    ups, low = eng.forward(fmaps.to(dev), cnets.to(dev), iters=2)
    pl = eng.plan(B, h, w, 256)
    if preset == "fp32_class":
        pyr = orc.corr_pyramid(fmaps[:, 0], fmaps[:, 1])
        for l in range(4):
            got = pl.level_maps(l)[0].reshape(pyr[l].shape).cpu()        # (strided view of the row-pitched maps: 156 -> 160 cells)
            assert (got - pyr[l]).abs().max().item() < 5e-5, f"level {l}"
    else:
        pyr = orc.corr_pyramid(fmaps[:, 0].half().float(), fmaps[:, 1].half().float())
        assert pl.corr_blocked                                   # the preset's fp16 volumes live in the blocked layout
        lv = pl.vol.levels()
        for l in range(4):
            got = lv[l].reshape(pyr[l].shape).float().cpu()
            assert lv[l].dtype == torch.float16
            assert ((got - pyr[l]).abs() <= 2.0 ** -11 * pyr[l].abs() + 3e-5).all(), f"level {l}"
    ups_o, _ = orc.hotpath_forward(fmaps, cnets, P, 2)
    e = orc.epe(ups[0].cpu(), ups_o[0])
    print(f"kitti-shape T=2 [{preset}]: EPE vs oracle = {e:.3e}")
    assert ups[0].shape == (1, 2, 376, 1248) and e <= 1e-3


def test_chunked_attention_path_vs_golden(golden, dev):
    """The high-resolution attention path (rows recomputed chunk by chunk every iteration, never an N x N tensor;
    SURVEY K6') forced at a small shape: must agree with the reference forward like the materialised path."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd.engine import HotPathEngine
    tag = "forward_T4"
    g = golden(tag)
    B, T, H, W, iters, seed, use_init = cases.FORWARD_CASES[tag]
    P, fmaps, cnets, finit, iters = cases.forward_inputs(tag)
    eng = HotPathEngine(P, device=dev, T=T)
    eng.attn_chunk_rows = 100                      # 384 pixels -> chunks of 100, 100, 100, 84 rows
    ups, _ = eng.forward(fmaps.to(dev), cnets.to(dev), iters=iters)
    assert eng.plan(B, H // 8, W // 8, 256).attn_rows == 100
    for i in range(T - 1):
        e = orc.epe(ups[i].cpu(), torch.from_numpy(g[f"up{i}"]))
        assert e <= 1e-3, e


@pytest.mark.parametrize("P", [64, 323, 1000, 7040])
@pytest.mark.parametrize("qkp", [1, 2, 3])
@pytest.mark.parametrize("stats", [False, True])
def test_gma_flash_kernel_vs_float64(dev, P, qkp, stats):
    """sf_gma_flash_*: out = mf + gamma * softmax(scale q k^T) v (demo.py:235-258 == gma.py:53-65,91-104) against a
    float64 evaluation on the same q, k, v.  P = 323 / 1000 exercise the padded key tail and the partial query tile.
    qk_products = 3 is the split-precision (fp32-class) logit path; 1 and 2 round k (and q) to fp16 once.
    stats: the softmax statistics are computed once by pack_qk (what the engine does: q, k are constant over the loop)
    instead of online in every aggregate call; logits with a large spread check that the stored maximum is the right one."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    gen = torch.Generator().manual_seed(P * 10 + qkp)
    n = 2
    qk = torch.randn(n, 256, P, generator=gen)
    qk[:, :128] *= 1.5                                        # logits with a spread of a few units after scaling
    v = torch.randn(n, 128, P, generator=gen)
    mf = torch.randn(n, 128, P, generator=gen)
    gamma = torch.tensor([0.61])
    ws = torch.empty(ops.gma_flash_ws_bytes(n, P), dtype=torch.uint8, device=dev)
    out = torch.full((n, 128, P), float("nan"), device=dev)
    scale = 128 ** -0.5
    sharp = P >= 323
    if sharp:
        qk[:, :128, : P // 2] *= 6.0                          # half of the queries with logits of +-40: exp2 must not overflow
    ops.gma_flash_pack_qk(Planes.of(qk.to(dev)), ws, scale, stats_qk_products=qkp if stats else 0)
    ops.gma_flash_aggregate(ws, Planes.of(v.to(dev)), Planes.of(mf.to(dev)), gamma.to(dev), Planes.of(out), qkp, use_stats=stats)
    torch.cuda.synchronize()
    if stats:       # same logits; the fp16 weights are rounded relative to the final maximum instead of the running one (2^-11 each)
        online = torch.full((n, 128, P), float("nan"), device=dev)
        ops.gma_flash_aggregate(ws, Planes.of(v.to(dev)), Planes.of(mf.to(dev)), gamma.to(dev), Planes.of(online), qkp, use_stats=False)
        d = (online - out).abs().max().item()
        print(f"flash P={P} qk_products={qkp}: stored statistics vs online softmax: {d:.2e}")
        assert d < 1.5e-3, d
    q64, k64 = qk[:, :128].double(), qk[:, 128:].double()
    attn = torch.softmax(scale * torch.einsum("ndi,ndj->nij", q64, k64), dim=-1)
    ref = mf.double() + 0.61 * torch.einsum("nij,ndj->ndi", attn, v.double())
    err = (out.double().cpu() - ref).abs().max().item()
    tol = {3: 4e-4, 2: 1.5e-3, 1: 3e-3}[qkp]                  # fp16 softmax weights / v: ~2^-11 relative on O(1) values
    if sharp:                                                 # peaked rows: one weight of ~1 carries its fp16 rounding (2^-11 |v|) alone,
        tol = {3: 2e-3, 2: 1.2e-2, 1: 2.5e-2}[qkp]            # and fp16-rounded q, k move logits of +-40 by up to 1e-2
    print(f"flash P={P} qk_products={qkp}: max abs err vs float64 = {err:.2e}")
    assert err < tol, (P, qkp, err)
    # optional k-octet fp16 copy of the result (the next GEMM's operand format): same fp32 output, copy = its rounding
    from dataclasses import replace
    out2 = torch.full((n, 128, P), float("nan"), device=dev)
    sh = ops.new_shadow(Planes.of(out2), dev)
    ops.gma_flash_aggregate(ws, Planes.of(v.to(dev)), Planes.of(mf.to(dev)), gamma.to(dev),
                            replace(Planes.of(out2), shadow=sh), qkp, use_stats=stats)
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    assert torch.equal(sh.tensor().float(), out.half().float())
    # v as fp16 rows (sf_gma_flash_aggregate_f16v, the to_v GEMM's c_f16 = 1 hand-over): the pack rounds v to fp16 anyway, so
    # feeding the rounded values through either entry point gives bit-identical results (P even: fp16 rows in a float buffer)
    if P % 2 == 0:
        v16 = v.half().to(dev).contiguous()
        V16 = Planes(v16.view(-1).view(torch.float32), 0, 128 * P, n, 128, P, f16=True)
        o_h = torch.full((n, 128, P), float("nan"), device=dev)
        o_f = torch.full((n, 128, P), float("nan"), device=dev)
        ops.gma_flash_aggregate(ws, V16, Planes.of(mf.to(dev)), gamma.to(dev), Planes.of(o_h), qkp, use_stats=stats)
        ops.gma_flash_aggregate(ws, Planes.of(v16.float()), Planes.of(mf.to(dev)), gamma.to(dev), Planes.of(o_f), qkp, use_stats=stats)
        torch.cuda.synchronize()
        assert torch.equal(o_h, o_f) and torch.equal(o_h, out)


@pytest.mark.parametrize("P", [64, 323, 1000, 7040])
@pytest.mark.parametrize("single", [False, True])
def test_gma_flash_project_v_vs_gemm_and_float64(dev, P, single):
    """sf_gma_flash_project_v (to_v + the v pack in one launch, gma.py:93) against (a) float64 on the same fp16-rounded features and
    (b) the two-launch path it replaces (sf_gemm to fp16 rows -> pack inside sf_gma_flash_aggregate): the packed v planes must give
    the same aggregation result up to the summation order of one 128-term dot product and one fp16 rounding of v.  P = 323 / 1000:
    padded keys must be exact zeros (the workspace is poisoned first)."""
    from dataclasses import replace
    from streamflow_amd import ops
    from streamflow_amd.ops import PackedLinear, Planes
    gen = torch.Generator().manual_seed(P + 7 * int(single))
    n = 3
    x = torch.randn(n, 128, P, generator=gen)
    Wv = torch.randn(128, 128, generator=gen) / 128 ** 0.5
    qk = torch.randn(n, 256, P, generator=gen)
    mf = torch.randn(n, 128, P, generator=gen)
    gamma = torch.tensor([0.61]).to(dev)
    A = PackedLinear(Wv.view(128, 128, 1, 1), None, dev)
    A.single = single
    X = Planes.of(x.to(dev).contiguous())
    sh = ops.new_shadow(X, dev)
    ops.pack_koct(X, sh)
    X = replace(X, shadow=sh)
    cx = ops.Ctx(precision=ops.PRECISION_F16X2)
    assert ops.gma_flash_project_ok(A, X, cx)
    ws = torch.empty(ops.gma_flash_ws_bytes(n, P), dtype=torch.uint8, device=dev)
    ws.fill_(0x7e)                                            # (0x7e7e = a large finite fp16 / NaN-free poison: padded keys must be rewritten)
    ops.gma_flash_pack_qk(Planes.of(qk.to(dev)), ws, 128 ** -0.5, stats_qk_products=1, cx=cx)
    out_f = torch.full((n, 128, P), float("nan"), device=dev)
    ops.gma_flash_project_v(ws, A, X, cx=cx)
    ops.gma_flash_aggregate(ws, None, Planes.of(mf.to(dev)), gamma, Planes.of(out_f), 1, use_stats=True, cx=cx)
    # the two-launch path: GEMM in the same arithmetic (fp16 activations, hi + lo or hi weights) to fp16 rows, then the pack
    Pe = P + (P % 2)
    v16 = torch.zeros(n * 128 * Pe // 2 + 8, device=dev)
    if P % 2 == 0:
        V16 = Planes(v16, 0, 128 * P, n, 128, P, f16=True)
        ops.gemm(A, X, V16, ops.EPI_NONE, cx=cx)
        out_g = torch.full((n, 128, P), float("nan"), device=dev)
        ops.gma_flash_aggregate(ws, V16, Planes.of(mf.to(dev)), gamma, Planes.of(out_g), 1, use_stats=True, cx=cx)
        torch.cuda.synchronize()
        d = (out_f - out_g).abs().max().item()
        print(f"project_v P={P} single={single}: fused vs gemm + pack: {d:.2e}")
        assert d < 2e-3, d                                     # (one fp16 ulp of v values of ~3 = 2e-3, weighted by softmax rows)
    torch.cuda.synchronize()
    hi = A.hi.float().permute(1, 0, 2).reshape(128, 128).double().cpu()
    lo = A.lo.float().permute(1, 0, 2).reshape(128, 128).double().cpu()
    W_eff = (hi if single else hi + lo) / A.split_scale
    v64 = torch.einsum("dc,ncp->ndp", W_eff, x.half().double()).half().double()           # v enters P V as fp16
    attn = torch.softmax(128 ** -0.5 * torch.einsum("ndi,ndj->nij", qk[:, :128].half().double(), qk[:, 128:].half().double()), dim=-1)
    ref = mf.double() + 0.61 * torch.einsum("nij,ndj->ndi", attn, v64)
    err = (out_f.double().cpu() - ref).abs().max().item()
    print(f"project_v P={P} single={single}: max abs err vs float64 = {err:.2e}")
    assert err < 3e-3, (P, single, err)


@pytest.mark.parametrize("qkp", [1, 2, 3])
def test_engine_flash_mode_vs_golden(golden, dev, qkp):
    """The whole loop with the fused GMA aggregation (gma_mode='flash': no attention matrix at all) against the
    reference forward, for every logit precision; the 15-iteration small-shape run bounds the accumulated effect."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    tag = "forward_T4"
    g = golden(tag)
    B, T, H, W, iters, seed, use_init = cases.FORWARD_CASES[tag]
    P, fmaps, cnets, finit, iters = cases.forward_inputs(tag)
    for graph in (False, True):
        eng = HotPathEngine(P, device=dev, T=T, use_graph=graph, gma_mode="flash", flash_qk_products=qkp)
        ups, _ = eng.forward(fmaps.to(dev), cnets.to(dev), iters=iters)
        pl = eng.plan(B, H // 8, W // 8, 256)
        assert pl.flash and pl.attn.numel() <= pl.n * pl.P
        for i in range(T - 1):
            e = orc.epe(ups[i].cpu(), torch.from_numpy(g[f"up{i}"]))
            assert e <= 1e-3, (qkp, graph, i, e)
    B, T, h, w = 1, 4, 16, 24
    P = syn.make_params(5, T)
    fmaps, cnets = syn.make_features(5, B, T, h, w)
    ups_o, _ = orc.hotpath_forward(fmaps, cnets, P, 15)
    eng = HotPathEngine(P, device=dev, T=T, gma_mode="flash", flash_qk_products=qkp)
    ups, _ = eng.forward(fmaps.to(dev), cnets.to(dev), iters=15)
    e = max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))
    print(f"flash GMA, qk_products={qkp}: 15-iteration EPE vs oracle = {e:.3e}")
    assert e <= 1e-3, (qkp, e)


def test_spring_shape_smoke(dev):
    """BASELINE config 5 shape (1080x1920 -> 1088x1920 padded -> 136x240 grid, N = 32640): the 4-GB-per-pair
    volumes and the chunked attention path run; outputs finite; pyramid pooling property holds on a sample."""
    from streamflow_amd import synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    B, T, h, w = 1, 4, 136, 240
    P = syn.make_params(17, T)
    fmaps, cnets = syn.make_features(17, B, T, h, w)
    from streamflow_amd import presets
    eng2 = HotPathEngine(P, device=dev, T=T, **presets.engine_kwargs(presets.BENCH_PRESET))      # the bench preset at T = 4
    ups2, _ = eng2.forward(fmaps.to(dev), cnets.to(dev), iters=1)
    assert eng2.plan(B, h, w, 256).corr_blocked and all(torch.isfinite(u).all() for u in ups2)
    del eng2
    torch.cuda.empty_cache()
    eng = HotPathEngine(P, device=dev, T=T)
    ups, low = eng.forward(fmaps.to(dev), cnets.to(dev), iters=1)
    assert max((a - b).abs().max().item() for a, b in zip(ups, ups2)) < 0.05      # two arithmetic classes, one iteration
    pl = eng.plan(B, h, w, 256)
    assert pl.attn_rows < h * w
    assert ups[0].shape == (1, 2, 1088, 1920)
    for u in ups:
        assert torch.isfinite(u).all()
    N = h * w
    i = 12345                                           # one source pixel of pair 0
    l0 = pl.level_maps(0)[0][i]
    l1 = pl.level_maps(1)[0][i]
    assert (l0.reshape(h // 2, 2, w // 2, 2).mean(dim=(1, 3)) - l1).abs().max().item() < 1e-5
    ref = (fmaps[0, 0].reshape(256, N)[:, i].to(dev) @ fmaps[0, 1].reshape(256, N).to(dev)) / 16.0
    assert (l0.reshape(-1) - ref).abs().max().item() < 1e-3


def test_stress_shape_1440x2560(dev):
    """The reference's own memory-smoke shape (test_memory.py:478-483: 1 x 4 x 3 x 1440 x 2560 -> 180 x 320 grid, N = 57,600):
    9.1 GB of blocked fp16 pyramids per pair (64-bit image / record addressing: a row-major fp16 level 0 alone is 6.6 GB,
    past any 32-bit offset), fused GMA, one iteration in the bench preset; outputs finite, and the volume of one source
    pixel agrees with a direct contraction and its own 2 x 2 pooling."""
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    B, T, h, w = 1, 4, 180, 320
    P = syn.make_params(23, T)
    fmaps, cnets = syn.make_features(23, B, T, h, w)
    eng = HotPathEngine(P, device=dev, T=T, **presets.engine_kwargs(presets.BENCH_PRESET))
    ups, low = eng.forward(fmaps.to(dev), cnets.to(dev), iters=1)
    assert ups[0].shape == (1, 2, 1440, 2560)
    for u in ups:
        assert torch.isfinite(u).all()
    pl = eng.plan(B, h, w, 256)
    assert pl.corr_blocked and pl.flash
    N, i, img = h * w, 43210, 2                                 # one source pixel of the LAST pair (highest addresses)
    rec = torch.as_strided(pl.vol.buf, (pl.vol.rec,), (1,), img * pl.vol.img_stride + i * pl.vol.rec).clone()
    lv = []
    for l in range(2):
        nby, nbx = pl.vol.nby[l], pl.vol.nbx[l]
        blk = rec[pl.vol.off[l]: pl.vol.off[l] + nby * nbx * 128].view(torch.float16).view(nby, nbx, 8, 8)
        lv.append(blk.permute(0, 3, 1, 2).reshape(nby * 8, nbx * 8)[: h >> l, : w >> l].float())
    ref = (fmaps[0, 2].reshape(256, N)[:, i].half().float().to(dev) @ fmaps[0, 3].reshape(256, N).half().float().to(dev)) / 16.0
    assert (lv[0].reshape(-1) - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())
    assert (lv[0].view(h // 2, 2, w // 2, 2).mean(dim=(1, 3)) - lv[1]).abs().max().item() < 2e-3
    del eng, pl
    torch.cuda.empty_cache()


def test_debug_range_check_catches_fp16_overflow(dev, monkeypatch):
    """SF_DEBUG_RANGE: an activation beyond the fp16 range (the split modes would saturate its `hi` half silently) raises
    before the contraction is launched; in-range data and the exact fp32 mode pass."""
    from streamflow_amd import ops
    from streamflow_amd.ops import PackedLinear, Planes
    monkeypatch.setattr(ops, "DEBUG_RANGE", True)
    W = PackedLinear(torch.randn(64, 32, 1, 1) * 0.1, None, dev)
    x = torch.randn(1, 32, 256, device=dev)
    y = torch.empty(1, 64, 256, device=dev)
    prev = ops.set_precision("f16x2")
    try:
        ops.gemm(W, Planes.of(x), Planes.of(y))
        x[0, 3, 7] = 1.0e5
        with pytest.raises(RuntimeError, match="SF_DEBUG_RANGE"):
            ops.gemm(W, Planes.of(x), Planes.of(y))
        with pytest.raises(RuntimeError, match="SF_DEBUG_RANGE"):
            ops.dwconv_res_gelu(Planes.of(x.view(1, 32, 256)), torch.zeros(32, 225, device=dev), torch.zeros(32, device=dev),
                                Planes.of(torch.empty(1, 32, 256, device=dev)), 16, 16, 15)
        ops.set_precision("fp32")
        ops.gemm(W, Planes.of(x), Planes.of(y))
        torch.cuda.synchronize()
        assert torch.isfinite(y).all()
    finally:
        ops.set_precision(prev)


def test_spring_shape_one_pair_vs_oracle(dev):
    """BASELINE config 5 at full resolution against the oracle: 1088x1920 (136x240 grid, N = 32,640), ONE frame pair
    (T=2), 2 iterations, in both presets -- the fused GMA kernel (the N x N matrix is never stored), the 4.3 GB (fp32) /
    2.1 GB (fp16) correlation volume, lookups and the update block at that size.  The oracle needs ~15 GB of host RAM."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    B, T, h, w, iters = 1, 2, 136, 240, 2
    P = syn.make_params(19, T)
    fmaps, cnets = syn.make_features(19, B, T, h, w)
    ups_o, _ = orc.hotpath_forward(fmaps, cnets, P, iters)
    for name in ("fp32_class", "config2_fp16", "config2_mixed"):
        eng = HotPathEngine(P, device=dev, T=T, **presets.engine_kwargs(name))
        ups, _ = eng.forward(fmaps.to(dev), cnets.to(dev), iters=iters)
        pl = eng.plan(B, h, w, 256)
        assert pl.flash and pl.attn.numel() <= pl.n * pl.P          # no attention matrix was materialised
        e = orc.epe(ups[0].cpu(), ups_o[0])
        print(f"spring shape, one pair, {iters} iterations [{name}]: EPE vs oracle = {e:.3e}")
        assert ups[0].shape == (1, 2, 1088, 1920) and e <= 1e-3, (name, e)
        del eng, pl
        torch.cuda.empty_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("K", [200, 1544])          # with and without a partial last k-tile
def test_gemm_stored_fp16_operand(dev, K):
    """attn @ v with the attention matrix stored in fp16 (SF_LAYOUT_F16_K_MINOR): C = R + gamma * A B^T with
    A = v [M][K] fp32 (split hi+lo in the kernel), B [N][K] fp16 used as stored.  Reference in float64 on the SAME
    fp16 values, so only the split-product error (~2^-22) and the fp32 accumulation remain."""
    from streamflow_amd import ops
    from streamflow_amd._lib import LAYOUT_F16_K_MINOR, LAYOUT_K_MINOR
    prev = ops.PRECISION
    ops.set_precision("f16x3")
    try:
        g = torch.Generator().manual_seed(K)
        M, N, n = 128, 300, 2
        A = torch.randn(n, M, K, generator=g).to(dev)
        B16 = torch.softmax(torch.randn(n, N, K, generator=g) * 3, dim=-1).half().to(dev)
        R = torch.randn(n, M, N, generator=g).to(dev)
        gamma = torch.tensor([0.37], device=dev)
        C = torch.empty(n, M, N, device=dev)
        ops.gemm_raw(A=A.data_ptr(), B=B16.data_ptr(), C=C.data_ptr(), R=R.data_ptr(), gamma=gamma.data_ptr(), M=M, N=N,
                     K=K, batch=n, lda=K, ldb=K, ldc=N, ldr=N, strideA=M * K, strideB=N * K, strideC=M * N,
                     strideR=M * N, a_layout=LAYOUT_K_MINOR, b_layout=LAYOUT_F16_K_MINOR, alpha=1.0,
                     epilogue=ops.EPI_AXPY)
        torch.cuda.synchronize()
        ref = R.double() + 0.37 * torch.einsum("zmk,znk->zmn", A.double(), B16.double())
        err = (C.double() - ref).abs().max().item()
        assert err < 2e-6, err
    finally:
        ops.PRECISION = prev


@pytest.mark.gpu
@pytest.mark.parametrize("k", [15, 7])
@pytest.mark.parametrize("hw", [(55, 128), (23, 37), (16, 24), (70, 52)])
def test_dwconv_vs_torch(dev, precision, k, hw):
    """y = gelu(x + dwconv_kxk(x) + b) (update.py:33-34) against torch in float64: the fp32 stencil and, in the split
    precisions, the Toeplitz-on-MFMA kernel (ragged widths / heights exercise the zero padding and the tile tails)."""
    import torch.nn.functional as F
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    h, w = hw
    g = torch.Generator().manual_seed(k * 1000 + h)
    n_img, C = 3, 5
    x = torch.randn(n_img, C, h * w, generator=g)
    wgt = torch.randn(C, k, k, generator=g) / k
    b = torch.randn(C, generator=g) * 0.1
    X, Y = Planes.of(x.to(dev)), Planes.of(torch.empty(n_img, C, h * w, device=dev))
    ops.dwconv_res_gelu(X, wgt.to(dev).contiguous(), b.to(dev), Y, h, w, k)
    torch.cuda.synchronize()
    xd = x.double().view(n_img, C, h, w)
    ref = F.gelu(xd + F.conv2d(xd, wgt.double().view(C, 1, k, k), b.double(), padding=k // 2, groups=C))
    got = Y.tensor().view(n_img, C, h, w).double().cpu()
    err = (got - ref).abs().max().item()
    assert err < (2e-5 if precision != "fp32" else 2e-5), (precision, err)
    # fp16 row output (the hand-over to the folded pw GEMM): the same values, rounded ONCE to fp16 (the compiler may fold
    # the last multiply of the GELU into the conversion, so this is within half an fp16 ulp of the fp32 result, not
    # necessarily the rounding of the rounded fp32 value; and the two instantiations contract the erf polynomial
    # differently: ~1e-7 absolute in the far negative tail where gelu is ~1e-6)
    store = torch.zeros(n_img, C, h * w, device=dev)
    Y16 = Planes(store.view(-1), 0, C * h * w, n_img, C, h * w, f16=True)
    ops.dwconv_res_gelu(X, wgt.to(dev).contiguous(), b.to(dev), Y16, h, w, k)
    torch.cuda.synchronize()
    y32 = Y.tensor()
    assert bool(((Y16.tensor().float() - y32).abs() <= 2.0 ** -11 * y32.abs() * (1 + 1e-3) + 3e-7).all())


@pytest.mark.gpu
@pytest.mark.parametrize("k", [15, 7])
@pytest.mark.parametrize("hw", [(55, 128), (23, 37)])
def test_dwconv_two_product_mode(dev, hw, k):
    """f16x2 / f16 modes (K = 15 and K = 7 on the matrix cores): the activation enters the products rounded to fp16 (weights hi + lo, residual exact).
    Against float64 with the activation rounded the same way the error is the split's (1e-5); against the exact
    convolution it is bounded by 2^-11 * sum|w||x|."""
    import torch.nn.functional as F
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    h, w = hw
    n_img, C = 2, 6
    g = torch.Generator().manual_seed(h)
    x = torch.randn(n_img, C, h * w, generator=g)
    wgt = torch.randn(C, k, k, generator=g) / k
    b = torch.randn(C, generator=g) * 0.1
    X, Y = Planes.of(x.to(dev)), Planes.of(torch.empty(n_img, C, h * w, device=dev))
    prev = ops.set_precision("f16x2")
    try:
        ops.dwconv_res_gelu(X, wgt.to(dev).contiguous(), b.to(dev), Y, h, w, k)
        torch.cuda.synchronize()
    finally:
        ops.set_precision(prev)
    xd = x.double().view(n_img, C, h, w)
    xr = x.half().double().view(n_img, C, h, w)
    conv = lambda t, ww: F.conv2d(t, ww.double().view(C, 1, k, k), None, padding=k // 2, groups=C)
    got = Y.tensor().view(n_img, C, h, w).double().cpu()
    ref_same = F.gelu(xd + conv(xr, wgt) + b.double().view(1, C, 1, 1))
    assert (got - ref_same).abs().max().item() < 2e-5
    ref = F.gelu(xd + conv(xd, wgt) + b.double().view(1, C, 1, 1))
    bound = 2.0 ** -11 * conv(xd.abs(), wgt.abs()) * 1.13 + 2e-5          # |gelu'| <= 1.13
    assert bool(((got - ref).abs() <= bound).all())


@pytest.mark.gpu
@pytest.mark.parametrize("single", [False, True])
@pytest.mark.parametrize("k", [15, 7])
@pytest.mark.parametrize("hw,n_img,C", [((55, 128), 24, 40), ((55, 128), 5, 3), ((23, 37), 3, 6), ((16, 24), 7, 4), ((70, 56), 9, 2),
                                        ((136, 240), 3, 2)])
def test_dwconv_f16_input(dev, hw, n_img, C, k, single):
    """sf_dwconv_res_gelu_f16in: x2 handed over as fp16 ROWS (config-2 presets).  Widths of whole octets take the DMA-staged,
    double-buffered form (several images per workgroup: 24 x 40 planes; strips: the 136-row grid); other widths the register
    form.  Against float64 on the SAME fp16 input values the error is the weight split's and the fp16 rounding of the
    output; and the result equals (to that rounding) what the fp32-input kernel gives for the same values.  The output buffer
    is NaN-filled: every cell of every image must be written."""
    import torch.nn.functional as F
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    h, w = hw
    g = torch.Generator().manual_seed(k * 100 + h + n_img)
    x16 = (torch.randn(n_img, C, h * w, generator=g) * 2).half()
    wgt = torch.randn(C, k, k, generator=g) / k
    b = torch.randn(C, generator=g) * 0.1
    xs = x16.to(dev).contiguous()
    X16 = Planes(xs.view(-1).view(torch.float32), 0, C * h * w, n_img, C, h * w, f16=True)   # (even number of halves per image)
    ys = torch.full((n_img, C, h * w), float("nan"), dtype=torch.float16, device=dev)
    Y16 = Planes(ys.view(-1).view(torch.float32), 0, C * h * w, n_img, C, h * w, f16=True)
    cx = ops.Ctx(precision=ops.PRECISION_F16X2)
    ops.dwconv_res_gelu(X16, wgt.to(dev).contiguous(), b.to(dev), Y16, h, w, k, single=single, cx=cx)
    torch.cuda.synchronize()
    got = ys.double().cpu().view(n_img, C, h, w)
    assert bool(torch.isfinite(got).all())
    xd = x16.double().view(n_img, C, h, w)
    wd = (wgt.half() if single else wgt).double()
    pre = xd + F.conv2d(xd, wd.view(C, 1, k, k), b.double(), padding=k // 2, groups=C)
    ref = F.gelu(pre)
    # fp16 result: half an ulp of the value + the polynomial GELU (5.2e-5 absolute, 1.1e-5 relative) and the split error of the
    # fp32 result before rounding
    tol = 2.0 ** -11 * ref.abs() * 1.03 + 8e-5
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), (hw, n_img, C, k, single, float((got - ref).abs().max()), int(bad.sum()))
    # the fp32-input kernel on the same values
    Xf = Planes.of(x16.float().to(dev).contiguous())
    yf = torch.full((n_img, C, h * w), float("nan"), dtype=torch.float16, device=dev)
    Yf = Planes(yf.view(-1).view(torch.float32), 0, C * h * w, n_img, C, h * w, f16=True)
    ops.dwconv_res_gelu(Xf, wgt.to(dev).contiguous(), b.to(dev), Yf, h, w, k, single=single, cx=cx)
    torch.cuda.synchronize()
    d = (ys.float() - yf.float()).abs()
    # (each may sit one fp16 spacing from the float64 value near the top of a binade -- the bound above is 2^-11 RELATIVE --, so the
    # pair may differ by two: the DMA form folds the residual into the centre tap, the register form adds x in the epilogue)
    # absolute slack: in the tail of the polynomial GELU (pre-activation near -4, results ~1e-4, |error| <= 5.2e-5 by construction) a
    # 2e-6 difference of the pre-activation moves the result by up to ~2e-5 (at the clamp of the polynomial)
    assert bool((d <= 2.0 ** -9 * torch.maximum(ys.float().abs(), yf.float().abs()) + 2.5e-5).all()), float(d.max())


@pytest.mark.gpu
@pytest.mark.parametrize("k,C,hw", [(15, 256, (55, 128)), (7, 640, (55, 128)), (15, 324, (22, 36))])
def test_skblock_pw_fold_vs_float64(dev, k, C, hw):
    """f16x2 mode: x4 = gelu(x3 + pw(x3)) is computed as gelu((W + I) x3) with x3 handed over in fp16 rows by the
    depthwise kernel (engine.run_skblock).  The whole block against torch in float64, with the fold and without
    (Ctx.pw_fold = False): both within the mode's error, and the fold no worse than 2x the plain path."""
    import torch.nn.functional as F
    from streamflow_amd import ops
    from streamflow_amd.engine import SKBlockWeights, run_skblock
    from streamflow_amd.ops import Planes
    h, w = hw
    P, n, Cm, Co = h * w, 2, C * 3 // 2, 128
    g = torch.Generator().manual_seed(C + k)
    r = lambda *s: torch.randn(*s, generator=g)
    sd = {"b.ffn1.0.weight": r(Cm, C, 1, 1) / C ** 0.5, "b.ffn1.0.bias": r(Cm) * 0.1,
          "b.ffn1.2.weight": r(C, Cm, 1, 1) / Cm ** 0.5, "b.ffn1.2.bias": r(C) * 0.1,
          "b.conv_list.0.weight": r(C, 1, 1, 1) * 0.3, "b.conv_list.0.bias": r(C) * 0.1,
          "b.conv_list.1.weight": r(C, 1, k, k) / k, "b.conv_list.1.bias": r(C) * 0.1,
          "b.pw.weight": r(C, C, 1, 1) / C ** 0.5, "b.pw.bias": r(C) * 0.1,
          "b.ffn2.0.weight": r(Cm, C, 1, 1) / C ** 0.5, "b.ffn2.0.bias": r(Cm) * 0.1,
          "b.ffn2.2.weight": r(Co, Cm, 1, 1) / Cm ** 0.5, "b.ffn2.2.bias": r(Co) * 0.1}
    x = r(n, C, h, w)
    d = {kk: v.double() for kk, v in sd.items()}
    xd = x.double()
    c1 = lambda t, nm: F.conv2d(t, d["b." + nm + ".weight"], d["b." + nm + ".bias"])
    t = F.gelu(xd + c1(F.gelu(c1(xd, "ffn1.0")), "ffn1.2"))
    t = F.gelu(t + F.conv2d(t, d["b.conv_list.0.weight"], d["b.conv_list.0.bias"], groups=C))
    t = F.gelu(t + F.conv2d(t, d["b.conv_list.1.weight"], d["b.conv_list.1.bias"], padding=k // 2, groups=C))
    t = F.gelu(t + c1(t, "pw"))
    ref = c1(F.gelu(c1(t, "ffn2.0")), "ffn2.2").view(n, Co, P)
    W = SKBlockWeights(sd, "b", dev)
    X = Planes.of(x.view(n, C, P).to(dev))
    mk = lambda rows: Planes.of(torch.zeros(n, rows + 8, P, device=dev))
    errs = {}
    for fold in ("1", "0"):
        y = torch.full((n, Co, P), float("nan"), device=dev)
        run_skblock(W, X, Planes.of(y), mk(Cm), mk(C), mk(C), h, w, cx=ops.Ctx(ops.PRECISION_F16X2, pw_fold=(fold == "1")))
        torch.cuda.synchronize()
        errs[fold] = (y.double().cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert errs["0"] < 4e-3 * scale and errs["1"] < 4e-3 * scale, (errs, scale)
    assert errs["1"] < 2.0 * errs["0"] + 1e-6, errs


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(cases.INTERP_CASES))
def test_forward_interpolate_vs_reference(golden, dev, tag):
    """f4: the GPU nearest-neighbour kernel returns exactly the reference's scipy griddata result."""
    import streamflow_amd as sfa
    out = sfa.forward_interpolate(cases.interp_inputs(tag).to(dev))
    assert out.is_cuda and torch.equal(out.cpu(), torch.from_numpy(golden(tag)["out"]))


@pytest.mark.gpu
def test_forward_interpolate_full_size_vs_oracle(dev):
    """Sintel-size low-resolution flows (55x128, batch of 3) against the CPU restatement, plus the properties that hold
    at any size: every output vector is one of the input vectors whose warped position is valid, and a zero flow maps
    to itself except on the excluded border row / column (x1 > 0, y1 > 0 are strict in the reference)."""
    from oracle import streamflow_oracle as orc
    import streamflow_amd as sfa
    from streamflow_amd import synthetic as syn
    flow = syn.randn(91, "interp.full", (3, 2, 55, 128), 6.0)
    out = sfa.forward_interpolate(flow.to(dev)).cpu()
    for i in range(3):
        assert torch.equal(out[i], orc.forward_interpolate(flow[i]))
    pairs_in = {tuple(v) for v in flow[0].reshape(2, -1).t().tolist()}
    assert all(tuple(v) in pairs_in for v in out[0].reshape(2, -1).t().tolist())
    z = sfa.forward_interpolate(torch.zeros(1, 2, 16, 24, device=dev))
    assert torch.count_nonzero(z).item() == 0


@pytest.mark.gpu
def test_warm_start_clip_loop_vs_oracle(dev):
    """evaluate_mf.py:286-304: two consecutive clips of a scene, the second started from the forward-interpolated
    low-resolution flows of the first.  The whole loop (model forward + GPU forward_interpolate) against the oracle's."""
    from oracle import streamflow_oracle as orc
    from streamflow_amd import demo, synthetic as syn
    from streamflow_amd.model import SKFlow_MF8, default_args
    B, T, h, w, iters = 1, 4, 16, 24, 4
    P = syn.make_params(11, T)
    feats = [syn.make_features(100 + c, B, T, h, w) for c in range(2)]
    model = SKFlow_MF8(default_args(T=T, Encoder="InjectEncoder")).to(dev)
    model.load_state_dict(dict(P), strict=True)
    clips = [[torch.zeros(B, 3, 8 * h, 8 * w, device=dev) for _ in range(T)] for _ in range(2)]
    calls = {"n": 0}
    inner = model._features

    def features(imgs):                               # the stand-in encoders inject the clip's seeded features
        f, c = feats[calls["n"]]
        calls["n"] += 1
        model.fnet.features, model.cnet.features = f.to(dev), c.to(dev)
        return inner(imgs)
    model._features = features
    got = demo.predict_clips_warm_start(model, clips, iters=iters)
    # oracle loop
    init = None
    for c in range(2):
        ups, low = orc.hotpath_forward(feats[c][0], feats[c][1], P, iters, flow_init=init)
        for i in range(T - 1):
            assert orc.epe(got[c][i].cpu(), ups[i]) <= 1e-3, (c, i)
        init = [orc.forward_interpolate(l[0])[None] for l in low]


@pytest.mark.gpu
def test_bench_contract_line(dev):
    """bench.py prints ONE JSON line with the driver's contract fields plus `roofline` and `cpu_baseline`
    (a reduced run: demo256 workload, 2 clips, 2 steps)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--workload", "demo256", "--clips", "2"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "flow_fields_per_sec" and d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert d["epe_vs_oracle"]["value"] <= 1e-3
    # the clock under the step, measured by a probe that is a branch of the step's own graph: a plausible shader clock
    c = d["roofline"]["clock"]
    assert 500.0 < c["sustained_mhz"] <= 2600.0 and 500.0 < c["idle_probe_mhz"] <= 2600.0 and len(c["reads_mhz"]) == 4


@pytest.mark.gpu
def test_clock_probe_branch_leaves_the_results_alone(dev):
    """EngineOptions.clock_probe_us forks sf_clock_probe beside the forward (eager and as a branch of the captured graph): same flows
    bit for bit, a plausible clock in engine.clock_counts, and the forward lasts at least the probe's spin."""
    import time
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import EngineOptions, HotPathEngine
    T, B, h, w = 4, 2, 32, 48
    params = syn.make_params(3, T)
    fm, cn = (t.to(dev) for t in syn.make_features(77, B, T, h, w))
    kw = presets.engine_kwargs("config2_mixed")
    ref = [f.clone() for f in HotPathEngine(params, device=dev, T=T, use_graph=True, **kw).forward(fm, cn, iters=3)[0]]
    for graph in (False, True):
        eng = HotPathEngine(params, device=dev, T=T, use_graph=graph, options=EngineOptions(clock_probe_us=20000), **kw)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = eng.forward(fm, cn, iters=3)[0]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        assert all(torch.equal(a, b) for a, b in zip(ref, out))
        cyc, ticks = (int(v) for v in eng.clock_counts.tolist())
        assert ticks >= 20000 * 100 and 500.0 < 100.0 * cyc / ticks <= 2600.0
        assert dt >= 0.02


@pytest.mark.gpu
def test_bench_two_ranks_self_launched(dev):
    """`python bench.py --gpus 2` as typed (no external launcher): the process spawns its two ranks before touching
    the GPU, each rank runs main()'s N > 1 branch (process group, per-rank device, barrier, max-over-ranks all-reduce);
    both ranks share the one GPU of this box and rendezvous over gloo.  One JSON line, whole-job aggregate."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-device", "--dist-backend",
                        "gloo", "--workload", "demo256", "--clips", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["clips_per_step_all_gpus"] == 4 and d["config"]["parallelism"] == "replicas2"
    # value = flow fields of ALL ranks / max-over-ranks time
    assert abs(d["value"] - 2 * 2 * 3 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d                       # rank 0 at N = 1 only


@pytest.mark.gpu
def test_bench_strong_scaling_three_ranks(dev):
    """BASELINE config 4's mode at toy size, end to end: `bench.py --gpus 3 --total-clips 5 --clips 2` -- the SAME 5 clips at any
    N, round-robined over the ranks (2 / 2 / 1 clips, the uneven case), each rank's share run in launches of at most 2 clips;
    `scaling` = "strong", per-rank times in the JSON line, value = total flow fields / max-over-ranks time.  The three ranks
    share this box's one GPU and rendezvous over gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--share-device", "--dist-backend", "gloo",
                        "--workload", "demo256", "--total-clips", "5", "--clips", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["scaling"] == "strong"
    assert d["config"]["clips_per_step_all_gpus"] == 5 and d["config"]["clips_per_rank"] == [2, 2, 1]
    assert len(d["per_rank_ms_per_step"]) == 3 and max(d["per_rank_ms_per_step"]) <= d["ms_per_step"] * 1.001
    assert abs(d["value"] - 5 * 3 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6      # 5 clips x 3 pairs x 2 steps


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(cases.TWINS_CASES))
def test_twins_csc_encoder_vs_reference(golden, dev, tag, precision):
    """f1: the Twins_CSC encoder on the HIP kernels (sf_gemm for every Linear / strided conv, sf_layernorm_cm,
    sf_window_attn, sf_subsample_attn, sf_dwconv3x3_res) against the reference's Twins_CSC.forward output (golden:
    the reference's PatchEmbed-over-(T H) and stage loop executed over the timm stand-in) and against the CPU oracle."""
    from oracle import twins_oracle as two
    from streamflow_amd.encoders import Twins_CSC
    P, x = cases.twins_inputs(tag)
    enc = Twins_CSC().to(dev)
    enc.load_state_dict(dict(P), strict=True)
    out = enc(x.to(dev))
    B, T, H, W, _ = cases.TWINS_CASES[tag]
    assert out.shape == (B, T, 256, H // 8, W // 8) and out.is_cuda
    close(out, golden(tag)["out"], 5e-4, 2e-4, what=tag + " vs reference")
    close(out, two.twins_csc_forward(x, P), 5e-4, 2e-4, what=tag + " vs oracle")


@pytest.mark.gpu
def test_real_frames_end_to_end_with_twins_encoder(dev):
    """No stand-in anywhere: frames in 0..255 -> Twins_CSC fnet / cnet -> refinement loop -> flows, through the
    reference's SKFlow_MF8 signature; against the CPU oracles chained the same way (encoder oracle -> hot-path oracle), in the
    three presets (encoders AND loop in the preset's arithmetic, fp16 hand-over everywhere in the config-2 presets).
    The random-weight network over these frames is ill-conditioned (features up to |f| = 19, flows of 20-60 px).  The fp32 class
    stays inside the ABSOLUTE 1e-3 px budget even so.  The deviation of the fp16-activation class is proportional to the flow:
    its bound is 1e-3 of the mean flow magnitude, and the mixed preset may be at most 8x the all-split preset's deviation
    (measured 4.4x) -- a regression of the single-product layer set on ill-conditioned inputs stays visible."""
    from oracle import streamflow_oracle as orc, twins_oracle as two
    from streamflow_amd import synthetic as syn
    from streamflow_amd.model import SKFlow_MF8, default_args
    B, T, H, W, iters = 1, 4, 128, 192, 4
    hot, ef, ec = syn.make_params(21, T), syn.make_twins_params(22), syn.make_twins_params(23)
    sd = dict(hot)
    sd.update({"fnet." + k: v for k, v in ef.items()})
    sd.update({"cnet." + k: v for k, v in ec.items()})
    frames = [(syn.randn(24, f"frame{t}", (B, 3, H, W)).sigmoid() * 255.0) for t in range(T)]
    imgs = 2 * (torch.stack(frames, dim=1) / 255.0) - 1.0
    fmaps = two.twins_csc_forward(imgs, ef)
    cnets = two.twins_csc_forward(imgs[:, :-1], ec)
    ups_o, _ = orc.hotpath_forward(fmaps, cnets, hot, iters)
    mag = float(torch.stack([o.norm(dim=1).mean() for o in ups_o]).mean())
    worst = {}
    for preset in ("fp32_class", "config2_fp16", "config2_mixed"):
        model = SKFlow_MF8(default_args(T=T, preset=preset)).to(dev)
        model.load_state_dict(sd, strict=True)
        ups = model([f.to(dev) for f in frames], iters=iters, test_mode=True)
        assert len(ups) == T - 1 and ups[0].shape == (B, 2, H, W)
        worst[preset] = max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))
        print(f"real frames -> Twins_CSC -> loop [{preset}]: max EPE vs chained oracles = {worst[preset]:.3e} (mean |flow| {mag:.1f} px)")
        del model
    assert worst["fp32_class"] <= 1e-3, worst
    assert worst["config2_fp16"] <= 1e-3 * max(1.0, mag) and worst["config2_mixed"] <= 1e-3 * max(1.0, mag), (worst, mag)
    assert worst["config2_mixed"] <= 8.0 * worst["config2_fp16"], worst


@pytest.mark.gpu
@pytest.mark.parametrize("iters", cases.HARD_ITERS)
@pytest.mark.parametrize("seed", cases.HARD_SEEDS)
def test_hard_case_sweep_vs_oracle(dev, seed, iters):
    """the ill-conditioned input class (frames -> exact Twins_CSC features of a random-weight encoder -> loop at
    128 x 192, 4 iterations, flows of 4-40 px) on SIX weight / frame seeds, every one against the CPU ORACLE (not against another
    engine).  Bounds: the fp32 class inside the absolute 1e-3 px budget; the config-2 class (fp16 activations entering every
    product) inside 1e-3 of the mean flow magnitude -- its deviation is relative, DESIGN.md section 6 carries the per-seed table
    (profiles/r05_hard_case_ablation.jsonl: the oracle itself moves by <= 1.1e-3 px under a 2^-11 relative perturbation of its
    inputs, i.e. the cases are NOT chaotic; the deviation is the activations' fp16 rounding, 5-10x above everything else).
    Round 6: swept at 4 iterations (the form every earlier round reported) AND at the 15 the reference deploys
    (scripts/infer.sh:17, demo.py:419)."""
    from oracle import streamflow_oracle as orc, twins_oracle as two
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    B, T, H, W, _ = cases.HARD_SHAPE
    ps, _, a, b = cases.hard_case_seeds(seed)
    hot = syn.make_params(ps, T)
    imgs = 2 * (torch.stack(cases.hard_case_frames(seed), dim=1) / 255.0) - 1.0
    fm = two.twins_csc_forward(imgs, syn.make_twins_params(a))
    cn = two.twins_csc_forward(imgs[:, :-1], syn.make_twins_params(b))
    ups_o, _ = orc.hotpath_forward(fm, cn, hot, iters)
    mag = float(torch.stack([o.norm(dim=1).mean() for o in ups_o]).mean())
    worst = {}
    for preset in ("fp32_class", "config2_fp16", "config2_mixed"):
        eng = HotPathEngine(hot, device=dev, T=T, **presets.engine_kwargs(preset))
        ups, _ = eng.forward(fm.to(dev).contiguous(), cn.to(dev).contiguous(), iters=iters)
        worst[preset] = max(orc.epe(u.cpu(), o) for u, o in zip(ups, ups_o))
    print(f"hard case seed {seed}, {iters} iterations: mean |flow| {mag:.2f} px, EPE vs oracle {worst}")
    assert worst["fp32_class"] <= 1e-3, (seed, worst)
    assert worst["config2_fp16"] <= 1e-3 * max(1.0, mag), (seed, worst, mag)
    assert worst["config2_mixed"] <= 1e-3 * max(1.0, mag), (seed, worst, mag)
