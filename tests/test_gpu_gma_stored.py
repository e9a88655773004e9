"""GPU: the stored-weights GMA path (sf_gma_flash_store_p + sf_gma_stored_aggregate; core/gma.py:53-65 keeps `attn` for the loop,
gma.py:99-102 multiplies it every iteration) against the fused recompute kernel it must equal BIT FOR BIT (same statistics, same
fp16 weights, same MFMA sequence), against float64, and at the engine level."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _inputs(P, n, seed, sharp):
    gen = torch.Generator().manual_seed(seed)
    qk = torch.randn(n, 256, P, generator=gen)
    v = torch.randn(n, 128, P, generator=gen)
    mf = torch.randn(n, 128, P, generator=gen)
    if sharp:
        qk[:, :128, : P // 2] *= 6.0                         # half of the queries with logits of +-40
    return qk, v, mf


@pytest.mark.parametrize("P", [64, 323, 1000, 7040])
@pytest.mark.parametrize("qkp", [1, 3])
@pytest.mark.parametrize("n", [3, 24])
def test_stored_aggregate_equals_flash_bit_for_bit_and_float64(dev, P, qkp, n):
    from dataclasses import replace
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    if n == 24 and P != 7040:
        pytest.skip("the many-image case runs at the headline grid only")
    qk, v, mf = _inputs(P, n, 100 + P + qkp, sharp=P >= 323)
    gamma = torch.tensor([0.61]).to(dev)
    scale = 128 ** -0.5
    ws = torch.empty(ops.gma_flash_ws_bytes(n, P), dtype=torch.uint8, device=dev)
    ws.fill_(0x7F)                                           # poison: padded keys must come out as exact zeros
    pbuf = torch.empty(ops.gma_stored_p_bytes(n, P), dtype=torch.uint8, device=dev)
    pbuf.fill_(0x7E)                                         # fp16 0x7E7E = NaN: every fragment the kernel reads must have been written
    QK, V, MF = Planes.of(qk.to(dev)), Planes.of(v.to(dev)), Planes.of(mf.to(dev))
    ops.gma_flash_pack_qk(QK, ws, scale, stats_qk_products=qkp)
    ref = torch.full((n, 128, P), float("nan"), device=dev)
    ops.gma_flash_aggregate(ws, V, MF, gamma, Planes.of(ref), qkp, use_stats=True)
    ops.gma_flash_store_p(ws, pbuf, n, P, qkp)
    out = torch.full((n, 128, P), float("nan"), device=dev)
    sh = ops.new_shadow(Planes.of(out), dev)
    ops.gma_stored_aggregate(ws, pbuf, V, MF, gamma, replace(Planes.of(out), shadow=sh))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert torch.equal(out, ref), f"P={P} qkp={qkp}: max diff {(out - ref).abs().max().item():.3e}"
    assert torch.equal(sh.tensor().float(), out.half().float())
    if n == 3:
        q64, k64 = qk[:, :128].double(), qk[:, 128:].double()
        attn = torch.softmax(scale * torch.einsum("ndi,ndj->nij", q64, k64), dim=-1)
        r64 = mf.double() + 0.61 * torch.einsum("nij,ndj->ndi", attn, v.double())
        err = (out.double().cpu() - r64).abs().max().item()
        tol = ({3: 2e-3, 1: 2.5e-2} if P >= 323 else {3: 4e-4, 1: 3e-3})[qkp]
        print(f"stored P={P} qk_products={qkp}: max abs err vs float64 = {err:.2e}")
        assert err < tol
    # twenty launches beside a competing stream: bit-identical (the V ring is refilled right behind a barrier)
    if P == 7040 and n == 3:
        side = torch.cuda.Stream(device=dev)
        junk = torch.empty(64 << 20, device=dev)
        for _ in range(20):
            with torch.cuda.stream(side):
                junk.normal_()
            o2 = torch.full((n, 128, P), float("nan"), device=dev)
            ops.gma_stored_aggregate(ws, pbuf, V, MF, gamma, Planes.of(o2))
            torch.cuda.synchronize()
            assert torch.equal(o2, ref)


def test_stale_weights_poison_the_result(dev):
    """Weights stored for one pack call must not be used with the statistics of another (the header check): NaN, not silence."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    P, n = 256, 2
    qk, v, mf = _inputs(P, n, 7, False)
    gamma = torch.tensor([0.5]).to(dev)
    ws = torch.zeros(ops.gma_flash_ws_bytes(n, P), dtype=torch.uint8, device=dev)
    pbuf = torch.zeros(ops.gma_stored_p_bytes(n, P), dtype=torch.uint8, device=dev)
    QK, V, MF = Planes.of(qk.to(dev)), Planes.of(v.to(dev)), Planes.of(mf.to(dev))
    ops.gma_flash_pack_qk(QK, ws, 128 ** -0.5, stats_qk_products=1)
    ops.gma_flash_store_p(ws, pbuf, n, P, 1)
    ops.gma_flash_pack_qk(QK, ws, 128 ** -0.5, stats_qk_products=1)         # a new pack invalidates the stored weights
    out = torch.zeros(n, 128, P, device=dev)
    ops.gma_stored_aggregate(ws, pbuf, V, MF, gamma, Planes.of(out))
    torch.cuda.synchronize()
    assert torch.isnan(out).all()


@pytest.mark.parametrize("preset", ["config2_mixed", "fp32_class"])
def test_engine_stored_mode_equals_flash_mode(dev, preset):
    """HotPathEngine(gma_mode='stored') == gma_mode='flash', bit for bit, eager and graph replay (three clips: no key split; one
    clip at a small grid: the key-split form with its combine pass)."""
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    T = 4
    params = syn.make_params(5, T)
    for B, h, w in ((3, 32, 48), (1, 40, 64)):
        fm, cn = (t.to(dev) for t in syn.make_features(81, B, T, h, w))
        outs = {}
        from streamflow_amd.engine import EngineOptions
        for mode in ("flash", "stored"):
            # (stored_auto_px = 0: 'flash' means the recompute kernel -- by default split-precision logits keep their weights, i.e. ARE 'stored')
            kw = dict(presets.engine_kwargs(preset), gma_mode=mode, options=EngineOptions(stored_auto_px=0))
            for graph in (False, True):
                eng = HotPathEngine(params, device=dev, T=T, use_graph=graph, **kw)
                eng.forward(fm, cn, iters=3)
                outs[(mode, graph)] = [f.clone() for f in eng.forward(fm, cn, iters=3)[0]]
                assert (mode == "stored") == (next(iter(eng._plans.values())).pbuf is not None)
        for key, val in outs.items():
            for a, b in zip(val, outs[("flash", False)]):
                assert torch.equal(a, b), key


@pytest.mark.parametrize("P", [64, 323, 1000, 7040])
@pytest.mark.parametrize("qkp", [1, 2, 3])
def test_pipelined_recompute_kernel_equals_the_round5_kernel(dev, P, qkp, monkeypatch):
    """gma_flash_pipe_kernel (round 6: the wave overlaps the softmax of tile t with the logits of tile t + 1) computes the same
    logits, the same fp16 weights and the same P V sequence as gma_flash_kernel<QKP, 1>: bit-identical outputs (SF_FLASH_PIPE=0
    selects the old kernel), incl. the key-split form (n = 1) and twenty launches beside a competing stream."""
    from streamflow_amd import ops
    from streamflow_amd.ops import Planes
    for n in (3, 1):
        qk, v, mf = _inputs(P, n, 300 + P + qkp, sharp=P >= 323)
        gamma = torch.tensor([0.61]).to(dev)
        ws = torch.empty(ops.gma_flash_ws_bytes(n, P), dtype=torch.uint8, device=dev)
        ws.fill_(0x7F)
        QK, V, MF = Planes.of(qk.to(dev)), Planes.of(v.to(dev)), Planes.of(mf.to(dev))
        ops.gma_flash_pack_qk(QK, ws, 128 ** -0.5, stats_qk_products=qkp)
        outs = {}
        for pipe in ("0", "2"):                              # 0: the round-5 kernel, 2: the pipelined form for every product count
            monkeypatch.setenv("SF_FLASH_PIPE", pipe)
            o = torch.full((n, 128, P), float("nan"), device=dev)
            ops.gma_flash_aggregate(ws, V, MF, gamma, Planes.of(o), qkp, use_stats=True)
            torch.cuda.synchronize()
            outs[pipe] = o
        assert torch.isfinite(outs["2"]).all()
        assert torch.equal(outs["0"], outs["2"]), f"P={P} qkp={qkp} n={n}: max diff {(outs['0'] - outs['2']).abs().max().item():.3e}"
        if P == 7040 and n == 3:
            side = torch.cuda.Stream(device=dev)
            junk = torch.empty(64 << 20, device=dev)
            for _ in range(20):
                with torch.cuda.stream(side):
                    junk.normal_()
                o2 = torch.full((n, 128, P), float("nan"), device=dev)
                ops.gma_flash_aggregate(ws, V, MF, gamma, Planes.of(o2), qkp, use_stats=True)
                torch.cuda.synchronize()
                assert torch.equal(o2, outs["2"])


def test_engine_hybrid_mode_equals_flash_mode_at_the_headline_grid(dev):
    """gma_mode='hybrid' (round 6): the first half of the images through the stored weights on a chain stream, the second half through the
    recompute kernel beside it -- same results as 'flash', bit for bit (6 clips at the 55 x 128 grid: both halves are large enough to run
    without the key-split form, the condition of the hybrid schedule), eager and graph replay."""
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    T, B, h, w = 4, 6, 55, 128
    params = syn.make_params(5, T)
    fm, cn = (t.to(dev) for t in syn.make_features(83, B, T, h, w))
    outs = {}
    from streamflow_amd.engine import EngineOptions
    for mode in ("flash", "flash_one_launch", "hybrid"):
        # 'flash': one fused launch per half-batch chain (EngineOptions.gma_per_chain, the default); 'flash_one_launch': round 5's form
        kw = dict(presets.engine_kwargs("config2_mixed"), gma_mode=mode.split("_")[0],
                  options=EngineOptions(gma_per_chain=(mode != "flash_one_launch")))
        for graph in (False, True):
            eng = HotPathEngine(params, device=dev, T=T, use_graph=graph, **kw)
            eng.forward(fm, cn, iters=2)
            outs[(mode, graph)] = [f.clone() for f in eng.forward(fm, cn, iters=2)[0]]
            pl = next(iter(eng._plans.values()))
            assert pl.n_store == (9 if mode == "hybrid" else 0) and (pl.pbuf is not None) == (mode == "hybrid")
            del eng
    for key, val in outs.items():
        for a, b in zip(val, outs[("flash", False)]):
            assert torch.equal(a, b), key


def test_split_precision_logits_keep_their_weights_by_default(dev):
    """EngineOptions.stored_auto_px: with flash_qk_products >= 2 (fp32_class) the fused path stores its softmax weights once per clip
    (same results as the recompute: the engine-level equality test above); with one product at a small grid it does not."""
    from streamflow_amd import presets, synthetic as syn
    from streamflow_amd.engine import HotPathEngine
    T, B, h, w = 4, 1, 16, 24
    params = syn.make_params(5, T)
    fm, cn = (t.to(dev) for t in syn.make_features(84, B, T, h, w))
    for preset, stored in (("fp32_class", True), ("config2_mixed", False)):
        eng = HotPathEngine(params, device=dev, T=T, **presets.engine_kwargs(preset))
        eng.forward(fm, cn, iters=2)
        pl = next(iter(eng._plans.values()))
        assert (pl.pbuf is not None) == stored and pl.auto_stored == stored
