"""CPU: the C-ABI library loads and exports every symbol include/streamflow_hip.h declares; host-side
logic (weight packing, state-dict contract, loud failure without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from streamflow_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from streamflow_amd import _lib
    hdr = open(os.path.join(REPO, "include", "streamflow_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(sf_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    # ... and nothing else: the dynamic symbol table (`nm -D`) holds exactly the declared sf_* entry points
    import shutil
    import subprocess
    nm = shutil.which("nm") or shutil.which("llvm-nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    if os.path.exists(nm):
        syms = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
        exported = {ln.split()[-1] for ln in syms.splitlines() if ln.split() and ln.split()[-1].startswith("sf_")}
        assert exported == declared, exported ^ declared
    version = int(re.search(r"#define SF_VERSION (\d+)", hdr).group(1))
    assert lib.sf_version() == version >= 101


def test_struct_layout_matches_header():
    from streamflow_amd._lib import SfGemm
    # 8 pointers + 4 i32 + 8 i64 + 2 i32 + (i32,pad,i64)*2 + 3 i32 + f32 + 2 i32  (natural alignment)
    assert ctypes.sizeof(SfGemm) == 304                         # ... + split-K fields + c_f16 (offset 264) + C16, strideC16, r_f16, a_k_pad, algo
    assert SfGemm.lda.offset == 80 and SfGemm.b_group_stride.offset == 160 and SfGemm.alpha.offset == 196
    assert SfGemm.c_f16.offset == 264 and SfGemm.C16.offset == 272 and SfGemm.strideC16.offset == 280
    assert SfGemm.r_f16.offset == 288 and SfGemm.a_k_pad.offset == 292 and SfGemm.algo.offset == 296


def test_blocked_volume_geometry(lib):
    """sf_corr_blocked_geometry (host-only): records of 8 x 8-cell blocks, levels floor-halved (corr.py:19-21)."""
    from streamflow_amd import ops
    rec, off, nby, nbx, src = ops.blocked_geometry(55, 128)                  # Sintel 440 x 1024
    assert (nby, nbx) == ((7, 4, 2, 1), (16, 8, 4, 2)) and rec == 128 * (112 + 32 + 8 + 2) == 19712
    assert off == (0, 112 * 128, 144 * 128, 152 * 128) and src == 7040
    rec, off, nby, nbx, src = ops.blocked_geometry(47, 156)                  # KITTI 376 x 1248: 23x78, 11x39, 5x19
    assert (nby, nbx) == ((6, 3, 2, 1), (20, 10, 5, 3)) and rec == 128 * (120 + 30 + 10 + 3) and src == 7424
    assert lib.sf_corr_blocked_bytes(3, 55, 128) == 3 * 7040 * 19712
    assert lib.sf_corr_blocked_geometry(7, 64, None, None, None, None, None) == -1
    assert b"too small" in lib.sf_last_error()
    assert lib.sf_corr_build_blocked_ws_bytes(2, 256, 55, 128) == 2 * 2 * 32 * 7048 * 16
    # fp32 cells (csrc/corr_blocked32.hip): blocks of 4 rows x 8 columns
    rec, off, nby, nbx, src = ops.blocked_geometry(55, 128, f32=True)
    assert (nby, nbx) == ((14, 7, 4, 2), (16, 8, 4, 2)) and rec == 128 * (224 + 56 + 16 + 4) == 38400 and src == 7040
    assert off == (0, 224 * 128, 280 * 128, 296 * 128)
    rec, off, nby, nbx, src = ops.blocked_geometry(47, 156, f32=True)
    assert (nby, nbx) == ((12, 6, 3, 2), (20, 10, 5, 3)) and rec == 128 * (240 + 60 + 15 + 6) and src == 7424
    assert lib.sf_corr_blocked32_bytes(3, 55, 128) == 3 * 7040 * 38400
    assert lib.sf_corr_blocked32_geometry(7, 64, None, None, None, None, None) == -1
    assert lib.sf_corr_build_blocked32_ws_bytes(2, 256, 55, 128) == 2 * 2 * 2 * 32 * 7040 * 16


def test_gemm_output_format_rules_hold_for_every_b_layout(lib):
    """the c_f16 rules were skipped for fp16 B operands (the common hand-over case): a K-major fp16 B with
    c_f16 = 1 and N = 6 would have run the scalar fp32 epilogue into a buffer sized for halves.  Validation happens
    before any launch, so this runs without a GPU (dummy, never dereferenced pointers)."""
    from streamflow_amd import _lib
    from streamflow_amd._lib import SfGemm

    def desc(**kw):
        g = SfGemm()
        g.A, g.B, g.C, g.A_hi, g.A_lo = 4096, 4096, 4096, 4096, 8192
        g.M, g.N, g.K, g.batch = 128, 6, 64, 1
        g.lda, g.ldb, g.ldc, g.lda_h = 128, 6, 6, 128
        g.a_layout, g.alpha, g.precision = _lib.LAYOUT_SPLIT_F16, 1.0, _lib.PRECISION_F16X2
        for k, v in kw.items():
            setattr(g, k, v)
        return g

    for blay in (_lib.LAYOUT_F16_K_MAJOR, _lib.LAYOUT_F16_KOCT):
        for cf in (1, 3):
            g = desc(b_layout=blay, c_f16=cf, C16=4096)
            assert lib.sf_gemm(ctypes.byref(g), None) == -2, (blay, cf)
            assert b"c_f16" in lib.sf_last_error()
    g = desc(b_layout=_lib.LAYOUT_F16_KOCT, c_f16=2, N=8, ldb=8, ldc=8, epilogue=_lib.EPI_RELU)
    assert lib.sf_gemm(ctypes.byref(g), None) == -2 and b"c_f16 = 2" in lib.sf_last_error()
    g = desc(b_layout=_lib.LAYOUT_F16_KOCT, N=8, ldb=8, ldc=8, k_splits=2, c_f16=1)
    assert lib.sf_gemm(ctypes.byref(g), None) == -2
    # the split k-octet hand-over (SF_LAYOUT_SPLIT_KOCT / c_f16 = 4) belongs to F16X3 and to the 128-row tile
    for kw in (dict(b_layout=_lib.LAYOUT_SPLIT_KOCT), dict(c_f16=4)):
        g = desc(N=8, ldb=8, ldc=8, **kw)
        assert lib.sf_gemm(ctypes.byref(g), None) != 0 and b"F16X3" in lib.sf_last_error(), kw
    g = desc(b_layout=_lib.LAYOUT_SPLIT_KOCT, precision=_lib.PRECISION_F16X3, M=64, N=8, ldb=8, ldc=8)
    assert lib.sf_gemm(ctypes.byref(g), None) == -2 and b"SPLIT_KOCT" in lib.sf_last_error()
    g = desc(c_f16=4, precision=_lib.PRECISION_F16X3, N=8, ldb=8, ldc=8, epilogue=_lib.EPI_RELU)
    assert lib.sf_gemm(ctypes.byref(g), None) == -2 and b"c_f16 = 2 / 4" in lib.sf_last_error()
    # k-octet residual: only with the RES_GELU_DW1 vector epilogue
    g = desc(b_layout=_lib.LAYOUT_F16_KOCT, N=8, ldb=8, ldc=8, ldr=8, R=4096, r_f16=2, epilogue=_lib.EPI_RES_GELU)
    assert lib.sf_gemm(ctypes.byref(g), None) == -2 and b"r_f16" in lib.sf_last_error()
    g = desc(b_layout=_lib.LAYOUT_F16_KOCT, N=6, ldb=6, ldc=6, ldr=6, R=4096, r_f16=2, epilogue=_lib.EPI_RES_GELU_DW1,
             dw_w=4096, dw_b=4096)
    assert lib.sf_gemm(ctypes.byref(g), None) == -2 and b"r_f16" in lib.sf_last_error()


def test_bad_arguments_return_error_codes(lib):
    assert lib.sf_coords_grid(None, 1, 4, 4, None) == -1
    assert b"sf_coords_grid" in lib.sf_last_error()
    assert lib.sf_gemm(None, None) == -1


def test_cpu_tensors_fail_loudly():
    import streamflow_amd as sfa
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sfa.coords_grid(1, 4, 4, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU"):
        sfa.CorrBlock(torch.zeros(1, 8, 16, 16), torch.zeros(1, 8, 16, 16))
    with pytest.raises(RuntimeError):
        sfa.HotPathEngine({}, device="cpu")


def test_state_dict_contract_strict_load():
    """Keys/shapes of the reference checkpoint contract load with strict=True, with and
    without the DataParallel 'module.' prefix handled by the checkpoint loader."""
    import streamflow_amd as sfa
    from streamflow_amd import synthetic as syn
    for T in (2, 3, 4):
        m = sfa.SKFlow_MF8(sfa.default_args(T=T, Encoder="InjectEncoder"))
        P = syn.make_params(0, T)
        m.load_state_dict(dict(P), strict=True)
        sd = m.state_dict()
        assert set(sd) == set(P)
        for k in P:
            assert tuple(sd[k].shape) == tuple(P[k].shape), k
    assert sum(v.numel() for v in syn.make_params(0, 4).values()) == 5780737 + 32768


def test_checkpoint_loader_accepts_module_prefix(tmp_path):
    import streamflow_amd as sfa
    from streamflow_amd import synthetic as syn
    P = syn.make_params(1, 4)
    ck = tmp_path / "ck.pth"
    torch.save({"model": {"module." + k: v for k, v in P.items()}}, ck)
    m = sfa.StreamFlowT4(str(ck), Encoder="InjectEncoder")
    assert torch.equal(m.state_dict()["update_block.gru.pw.weight"], P["update_block.gru.pw.weight"])
    bad = dict(P)
    bad.pop("update_block.aggregator.gamma")
    torch.save(bad, ck)
    with pytest.raises(RuntimeError, match="mismatch"):
        sfa.StreamFlowT4(str(ck), Encoder="InjectEncoder")


def test_packed_linear_layout():
    from streamflow_amd.ops import PackedLinear
    w = torch.arange(6 * 5, dtype=torch.float32).view(6, 5, 1, 1)
    p = PackedLinear(w, None, "cpu")
    assert (p.K, p.M, p.lda) == (5, 6, 128) and tuple(p.wt.shape) == (32, 128)
    assert torch.equal(p.wt[:5, :6], w.view(6, 5).t()) and p.wt[:, 6:].abs().sum() == 0 and p.wt[5:].abs().sum() == 0
    # split image: k-octet planes [K up to 128 / 8][M up to 128][8]; element (m, k) at [k // 8, m, k % 8]
    assert tuple(p.hi.shape) == (16, 128, 8) and p.lda_h == 128 and p.k_pad == 128     # (K padded to 128: whole weight stages)
    # the image holds split_scale * W (power of two, max|w| scaled into [1, 2)); undone by alpha / bias in ops.gemm
    assert p.split_scale == 2.0 ** -4 and p.split_error == 0.0
    full = (p.hi.float() + p.lo.float()).permute(1, 0, 2).reshape(128, 128) / p.split_scale
    assert torch.equal(full[:6, :5], w.view(6, 5)) and full[6:].abs().sum() == 0 and full[:, 5:].abs().sum() == 0
    # a layer of tiny weights keeps ~21 bits after scaling (unscaled, its lo parts would be fp16 subnormals: ~15 bits)
    tiny = PackedLinear(torch.randn(64, 96, 1, 1, generator=torch.Generator().manual_seed(0)) * 1e-3,
                        torch.ones(64), "cpu")
    assert tiny.split_scale >= 128 and tiny.split_error < 2.0 ** -20
    assert torch.equal(tiny.bias_split, tiny.bias * tiny.split_scale)
    w3 = torch.randn(4, 3, 3, 3)
    p3 = PackedLinear(w3, None, "cpu", conv3x3=True)
    assert p3.K == 27 and p3.wt[(1 * 3 + 2) * 3 + 1, 2] == w3[2, 1, 1, 2]


def test_input_padder_matches_reference_formula():
    from streamflow_amd import InputPadder
    p = InputPadder((1, 3, 436, 1024))
    x = torch.zeros(1, 3, 436, 1024)
    (y,) = p.pad(x)
    assert y.shape[-2:] == (440, 1024) and p._pad == [0, 0, 2, 2]
    assert p.unpad(y).shape == x.shape
    # [left, right, top, bottom] as the reference's class computes them (utils.py:10-16; values generated from it in the build container)
    table = [("sintel", 436, 1024, [0, 0, 2, 2]), ("sintel", 375, 1242, [3, 3, 0, 1]), ("sintel", 376, 1248, [0, 0, 0, 0]),
             ("sintel", 1080, 1920, [0, 0, 0, 0]), ("sintel", 124, 188, [2, 2, 2, 2]), ("sintel", 17, 23, [0, 1, 3, 4]),
             ("kitti", 436, 1024, [0, 0, 0, 4]), ("kitti", 375, 1242, [3, 3, 0, 1]), ("kitti", 124, 188, [2, 2, 0, 4]),
             ("kitti", 17, 23, [0, 1, 0, 7]), ("kitti", 64, 64, [0, 0, 0, 0])]
    for mode, h, w, pad in table:
        q = InputPadder((2, 3, h, w), mode)
        assert q._pad == pad, (mode, h, w, q._pad)
        z = torch.randn(2, 3, h, w)
        (zp,) = q.pad(z)
        assert zp.shape[-2] % 8 == 0 and zp.shape[-1] % 8 == 0 and torch.equal(q.unpad(zp), z)
        assert torch.equal(q.pad_list([z])[0], zp)


def test_full_checkpoint_contract_with_twins_encoder(tmp_path):
    """f1: with the Twins_CSC encoder the model exposes the reference's COMPLETE state dict -- fnet.svt.* and cnet.svt.*
    (74 keys each: first two twins_svt_large stages + the surviving final norm, twins_csc.py:40-57) next to att.* and
    update_block.* -- and StreamFlowT4(ckpt) loads a DataParallel-style checkpoint file strictly (demo.py:388-389)."""
    import streamflow_amd as sfa
    from streamflow_amd import synthetic as syn
    hot = syn.make_params(0, 4)
    enc = syn.make_twins_params(1)
    assert len(enc) == 74 and enc["svt.blocks.0.1.attn.sr.weight"].shape == (128, 128, 8, 8)
    assert enc["svt.blocks.1.0.attn.qkv.bias"].shape == (768,) and enc["svt.norm.weight"].shape == (1024,)
    sd = dict(hot)
    sd.update({"fnet." + k: v for k, v in enc.items()})
    sd.update({"cnet." + k: v + 0.0 for k, v in enc.items()})
    m = sfa.SKFlow_MF8(sfa.default_args(T=4))                      # default encoder = Twins_CSC
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd, strict=True)
    path = tmp_path / "streamflow.pth"
    torch.save({"model": {"module." + k: v for k, v in sd.items()}}, path)
    t4 = sfa.StreamFlowT4(str(path))
    assert torch.equal(t4.state_dict()["cnet.svt.pos_block.1.proj.0.weight"], sd["cnet.svt.pos_block.1.proj.0.weight"])
    bad = dict(sd)
    bad.pop("fnet.svt.blocks.0.0.attn.qkv.bias")
    torch.save(bad, path)
    with pytest.raises(RuntimeError, match="fnet.svt.blocks.0.0.attn.qkv.bias"):
        sfa.StreamFlowT4(str(path))


def test_mixed_preset_layer_sets():
    """config2_mixed (profiles/r05_preset_select.jsonl): exactly the 17 layers of the oracle-referenced selection
    keep split weights, the other 22 contraction layers and two of the 15x15 depthwise layers are single-product; every name exists
    in the engine's layer inventory; the committed selection record names the same sets; the other presets name no single layer."""
    import json
    from streamflow_amd import presets
    from streamflow_amd.engine import HotPathWeights
    inventory = set(HotPathWeights.PLAIN_LAYERS) | {f"{b}.{l}" for b in HotPathWeights.SK_BLOCKS for l in HotPathWeights.SK_LAYERS}
    assert len(inventory) == 39
    kw = presets.engine_kwargs("config2_mixed")
    single = set(kw["single_layers"])
    gemm_single = single & inventory
    assert set(presets.MIXED_KEEP_SPLIT) <= inventory and len(presets.MIXED_KEEP_SPLIT) == 17
    assert gemm_single == inventory - set(presets.MIXED_KEEP_SPLIT) and len(gemm_single) == 22
    assert single - inventory == {"convf2.dw", "conv.dw"}
    assert {"flow_head.ffn2_2", "gru.pw", "qkv", "fc2", "conv.pw"} <= set(presets.MIXED_KEEP_SPLIT)
    with open(os.path.join(REPO, "profiles", "r05_preset_select.jsonl")) as f:
        rec = json.loads(f.read().strip().splitlines()[-1])
    assert set(rec["keep_split"]) == set(presets.MIXED_KEEP_SPLIT)
    assert set(rec["single_depthwise"]) == set(presets.MIXED_SINGLE_DEPTHWISE)
    assert max(e / max(1.0, m) for e, m in zip(rec["epe_selected"][3:], rec["mean_flow_px"][3:])) < 0.6e-3       # hard cases
    assert all(v["epe_selected"] <= 0.6e-3 * max(1.0, v["mean_flow_px"]) for v in rec["validation_hard"])      # held-out validation
    for name in ("fp32_class", "config2_fp16"):
        assert not presets.engine_kwargs(name).get("single_layers")
    assert presets.BENCH_PRESET == "config2_mixed"


def test_ffn_pair_weight_stream_layout(lib):
    """ops.PackedPair: the fragment stream of sf_ffn_pair (include/streamflow_hip.h SfFfnPair): per 32 hidden rows the layer-1
    fragments in (k-step, plane, 16-row tile) order, then the layer-2 fragments with their 32 columns in the order the layer-1
    accumulators hold the hidden rows, zero-padded to whole 16-fragment stages.  Host-only (packing + sf_ffn_pair_frags)."""
    import random
    from streamflow_amd import ops
    torch.manual_seed(0)
    K1, H, M2 = 324, 486, 256
    A1 = ops.PackedLinear(torch.randn(H, K1, 1, 1), torch.randn(H), "cpu")
    A2 = ops.PackedLinear(torch.randn(M2, H, 1, 1), torch.randn(M2), "cpu")
    pp = ops.PackedPair(A1, A2)
    rnd = random.Random(1)
    for pm in ((1, 1), (2, 1), (2, 2)):
        st = pp.stream(*pm).view(-1, 64, 8)
        nk1, nt2, hp = (K1 + 31) // 32, (M2 + 15) // 16, (H + 31) // 32
        fpad = lib.sf_ffn_pair_frags(K1, M2, pm[0], pm[1])
        assert fpad % 16 == 0 and fpad >= 2 * nk1 * pm[0] + nt2 * pm[1] and st.shape[0] == hp * fpad
        h1, l1 = pp._split(A1, hp * 32, nk1 * 32)
        h2, l2 = pp._split(A2, nt2 * 16, hp * 32)
        assert torch.equal(h1[:H, :K1].float() + l1[:H, :K1].float(), (A1.hi.float() + A1.lo.float()).permute(1, 0, 2).reshape(A1.lda_h, -1)[:H, :K1])
        for _ in range(300):
            m, u, s_, lane, i = rnd.randrange(hp), rnd.randrange(2), rnd.randrange(nk1), rnd.randrange(64), rnd.randrange(8)
            row, kq = lane & 15, lane >> 4
            for pl in range(pm[0]):
                W = l1 if (pm[0] == 2 and pl == 0) else h1
                f = (u * nk1 + s_) * pm[0] + pl
                assert st[m * fpad + f, lane, i] == W[32 * m + 16 * u + row, 32 * s_ + 8 * kq + i]
            t = rnd.randrange(nt2)
            hid = 32 * m + (4 * kq + i if i < 4 else 16 + 4 * kq + i - 4)
            for pl in range(pm[1]):
                W = l2 if (pm[1] == 2 and pl == 0) else h2
                assert st[m * fpad + 2 * nk1 * pm[0] + t * pm[1] + pl, lane, i] == W[16 * t + row, hid]
        assert not bool(st.view(hp, fpad, -1)[:, 2 * nk1 * pm[0] + nt2 * pm[1]:].any())          # padding fragments are zero
    assert lib.sf_ffn_pair_frags(256, 256, 3, 1) == 0


def test_ffn_pair_argument_validation(lib):
    """sf_ffn_pair rejects what it cannot run before any launch (dummy, never dereferenced pointers): unbuilt shapes, a short
    weight stream, misaligned operands, mode 1 without the depthwise parameters."""
    import ctypes
    from streamflow_amd import _lib
    g = _lib.SfFfnPair()
    g.X, g.strideX, g.ldx = 0x1000, 256 * 64, 64
    g.wstream, g.wstream_bytes = 0x2000, 1 << 24
    g.N, g.batch, g.K1, g.H, g.M2, g.pm1, g.pm2, g.mode = 64, 1, 256, 384, 192, 1, 1, 0
    g.C16, g.strideC16, g.ldc16 = 0x3000, 192 * 64, 64
    g.alpha1 = g.alpha2 = 1.0
    bad = dict(K1=640, H=960, M2=128)                                  # the GRU: not built
    for k, v in bad.items():
        setattr(g, k, v)
    assert lib.sf_ffn_pair(ctypes.byref(g), None) != 0 and b"not built" in lib.sf_last_error() or b"H <=" in lib.sf_last_error()
    g.K1, g.H, g.M2 = 256, 384, 192
    g.wstream_bytes = 1024
    assert lib.sf_ffn_pair(ctypes.byref(g), None) != 0 and b"weight stream too small" in lib.sf_last_error()
    g.wstream_bytes = 1 << 24
    g.X = 0x1004
    assert lib.sf_ffn_pair(ctypes.byref(g), None) != 0 and b"16-byte aligned" in lib.sf_last_error()
    g.X = 0x1000
    g.mode, g.M2 = 1, 256
    assert lib.sf_ffn_pair(ctypes.byref(g), None) != 0 and b"mode 1 needs" in lib.sf_last_error()
    g.mode, g.M2, g.pm1 = 0, 192, 3
    assert lib.sf_ffn_pair(ctypes.byref(g), None) != 0


def test_temporal_block_weight_stream_layout_and_validation(lib):
    """ops.PackedTemporal: the fragment stream of sf_temporal_block (include/streamflow_hip.h SfTemporalBlock) -- q / k rows of qkv,
    then per proj k-step the two v row tiles and proj's eight, then per fc2 k-step the two fc1 row tiles and fc2's eight; the columns
    of proj / fc1 / fc2 k-steps in accumulator order; `lo` before `hi`.  Host-only; plus the entry point's argument checks."""
    import ctypes
    import random
    from streamflow_amd import _lib, ops
    torch.manual_seed(3)
    Cc, Hh = 128, 256
    L = [ops.PackedLinear(torch.randn(3 * Cc, Cc, 1, 1), None, "cpu"), ops.PackedLinear(torch.randn(Cc, Cc, 1, 1), torch.randn(Cc), "cpu"),
         ops.PackedLinear(torch.randn(Hh, Cc, 1, 1), torch.randn(Hh), "cpu"), ops.PackedLinear(torch.randn(Cc, Hh, 1, 1), torch.randn(Cc), "cpu")]
    pk = ops.PackedTemporal(*L)
    assert pk.built() and pk.products(ops.Ctx(precision=ops.PRECISION_F16X2)) == 2
    L[1].single = True
    assert pk.products(ops.Ctx(precision=ops.PRECISION_F16X2)) is None and pk.products(ops.Ctx(precision=ops.PRECISION_F16)) == 1
    L[1].single = False
    planes = [ops.PackedPair._split(l, l.M, l.K) for l in L]                 # (hi, lo) per layer
    rnd = random.Random(2)
    perm = lambda kq, i: 4 * kq + i if i < 4 else 16 + 4 * kq + i - 4
    for pm in (1, 2):
        st = pk.stream(pm).view(-1, 64, 8)
        assert st.shape[0] == lib.sf_temporal_block_frags(pm) == 256 * pm
        pick = lambda layer, pl: planes[layer][1] if (pm == 2 and pl == 0) else planes[layer][0]
        for _ in range(400):
            lane, i, pl = rnd.randrange(64), rnd.randrange(8), rnd.randrange(pm)
            row, kq = lane & 15, lane >> 4
            m, s_ = rnd.randrange(16), rnd.randrange(4)                     # phase 1: q / k row tile m, k-step s
            assert st[(m * 4 + s_) * pm + pl, lane, i] == pick(0, pl)[16 * m + row, 32 * s_ + 8 * kq + i]
            p_, u, mo = rnd.randrange(4), rnd.randrange(2), rnd.randrange(8)   # phase 2, proj k-step p
            base = 64 * pm + p_ * 16 * pm
            assert st[base + (u * 4 + s_) * pm + pl, lane, i] == pick(0, pl)[16 * (16 + 2 * p_ + u) + row, 32 * s_ + 8 * kq + i]
            assert st[base + 8 * pm + mo * pm + pl, lane, i] == pick(1, pl)[16 * mo + row, 32 * p_ + perm(kq, i)]
            h = rnd.randrange(8)                                             # phase 3, fc2 k-step h
            base = 128 * pm + h * 16 * pm
            assert st[base + (u * 4 + s_) * pm + pl, lane, i] == pick(2, pl)[16 * (2 * h + u) + row, 32 * s_ + perm(kq, i)]
            assert st[base + 8 * pm + mo * pm + pl, lane, i] == pick(3, pl)[16 * mo + row, 32 * h + perm(kq, i)]
    g = _lib.SfTemporalBlock()
    g.X16, g.strideX, g.ldx, g.wstream, g.wstream_bytes = 0x1000, 128 * 64, 64, 0x2000, 512 * 1024
    g.ln1_w = g.ln1_b = g.ln2_w = g.ln2_b = 0x3000
    g.Y, g.strideY, g.ldy = 0x4000, 128 * 64, 64
    g.N, g.B, g.TT, g.C, g.H, g.pm = 64, 1, 3, 128, 256, 2
    g.TT = 4
    assert lib.sf_temporal_block(ctypes.byref(g), None) != 0 and b"built for" in lib.sf_last_error()
    g.TT, g.wstream_bytes = 3, 1024
    assert lib.sf_temporal_block(ctypes.byref(g), None) != 0 and b"weight stream size" in lib.sf_last_error()
    g.wstream_bytes, g.X16 = 512 * 1024, 0x1004
    assert lib.sf_temporal_block(ctypes.byref(g), None) != 0 and b"16-byte aligned" in lib.sf_last_error()
    g.X16, g.Y = 0x1000, None
    assert lib.sf_temporal_block(ctypes.byref(g), None) != 0 and b"NULL operand" in lib.sf_last_error()


def test_a_translation_unit_compiles_from_a_clean_directory(tmp_path):
    """the in-tree build is mtime-incremental and the library travels prebuilt -- prove that a source compiles
    from NOTHING with the real build's flags: the smallest kernel file into an empty directory (object + resource record), the
    unit `python -m streamflow_amd.build --clean` repeats for every source."""
    from streamflow_amd import build
    obj = build.compile_source("mask_upsample.hip", str(tmp_path))
    assert os.path.getsize(obj) > 10000
    res = open(obj[:-2] + ".res").read()
    assert "mask_upsample_kernel" in res and "ScratchSize" in res
    # the clean path removes what it says (on a copy of the directory names, not the real build)
    assert callable(build.clean)


def test_sk_tail_weight_stream_layout(lib):
    """ops.PackedTail: the stream has sf_sk_tail_frags fragments; a fragment is [k-half][row][8 halves]; the k-steps of ffn2.0 / ffn2.2
    carry their columns in accumulator-register order (csrc/sk_tail.hip); units are zero-padded to 16-fragment stages."""
    from streamflow_amd.ops import PackedLinear, PackedPair, PackedTail
    C, H, M2 = 128, 192, 64
    g = torch.Generator().manual_seed(3)
    mk = lambda m, k: PackedLinear(torch.randn(m, k, 1, 1, generator=g), torch.randn(m, generator=g), "cpu")
    Ap, A0, A2 = mk(C, C), mk(H, C), mk(M2, H)
    tail = PackedTail(Ap, A0, A2)
    for pm in (1, 2):
        frags = lib.sf_sk_tail_frags(C, H, M2, pm)
        st = tail.stream(pm).view(frags, 2, 32, 8)                              # [fragment][k-half][row][i]
        nc, nh, nm, ks = 4, 6, 2, 8
        u1, u2 = -(-ks * pm // 16) * 16, -(-(ks + 2 * nm) * pm // 16) * 16
        assert frags == nc * u1 + nh * u2
        h1 = PackedPair._split(Ap, C, C)[0]
        h2 = PackedPair._split(A0, H, C)[0]
        h3 = PackedPair._split(A2, M2, H)[0]
        hi = pm - 1                                                             # position of the hi plane inside a (lo, hi) pair
        # pw tile 1, k-step 3: natural columns
        f = st[1 * u1 + 3 * pm + hi]
        for kh in (0, 1):
            assert torch.equal(f[kh], h1[32:64, 48 + 8 * kh: 48 + 8 * kh + 8])
        # ffn2.0 hidden tile 2, k-step 5 = (x4 tile 2, half 1): accumulator order
        f = st[nc * u1 + 2 * u2 + 5 * pm + hi]
        for kh in (0, 1):
            cols = [32 * 2 + 16 * 1 + (i & 3) + 8 * (i >> 2) + 4 * kh for i in range(8)]
            assert torch.equal(f[kh], h2[64:96][:, cols])
        # ffn2.2 from hidden tile 2, half s = 1, row tile m = 1
        f = st[nc * u1 + 2 * u2 + (ks + 1 * nm + 1) * pm + hi]
        for kh in (0, 1):
            cols = [32 * 2 + 16 * 1 + (i & 3) + 8 * (i >> 2) + 4 * kh for i in range(8)]
            assert torch.equal(f[kh], h3[32:64][:, cols])
        # padding of the last stage of a unit is zero
        if (ks + 2 * nm) * pm % 16:
            assert bool((st[nc * u1 + u2 - 1] == 0).all())
    assert lib.sf_sk_tail_frags(324, 486, 256, 2) == 0 and lib.sf_sk_tail_frags(256, 384, 192, 1) == 0
