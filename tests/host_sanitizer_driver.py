"""Run by tests/test_host_sanitizer_cpu.py inside a python that has the AddressSanitizer runtime preloaded and SF_HIP_LIB pointing at
the host-instrumented build (streamflow_amd.build.build_asan): drives the HOST side of the C ABI -- argument checks, launch planning,
dispatch tables, error strings -- with dummy device pointers on a machine without a GPU.  A valid problem runs its whole planning
code and then fails at the launch ('no device'); an invalid one is rejected with SF_ERR_BAD_ARG / SF_ERR_UNSUPPORTED.  Any sanitizer
report aborts the process (the test checks the exit code and stderr)."""
import ctypes as C
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamflow_amd import _lib  # noqa: E402

lib = _lib.load()
assert "asan" in _lib.LIB_PATH, _lib.LIB_PATH
calls = rejected = 0
PTR = [0x10000 * (i + 1) for i in range(16)]          # 16-byte aligned, never dereferenced by host code


def call(fn, *args):
    global calls, rejected
    rc = fn(*args)
    calls += 1
    if rc != 0:
        rejected += 1
        assert lib.sf_last_error()                     # every failure leaves a message
    return rc


# geometry / workspace helpers
rec, src = C.c_int64(), C.c_int64()
off, nby, nbx = (C.c_int64 * 4)(), (C.c_int32 * 4)(), (C.c_int32 * 4)()
for h, w in ((55, 128), (47, 156), (136, 240), (7, 64), (8, 8), (1, 1)):
    call(lib.sf_corr_blocked_geometry, h, w, C.byref(rec), off, nby, nbx, C.byref(src))
    lib.sf_corr_build_ws_bytes(2, 3, 256, h, w)
    lib.sf_corr_build_blocked_ws_bytes(6, 256, h, w)
    lib.sf_corr_blocked_bytes(6, h, w)
    call(lib.sf_corr_blocked32_geometry, h, w, C.byref(rec), off, nby, nbx, C.byref(src))
    lib.sf_corr_build_blocked32_ws_bytes(6, 256, h, w)
    lib.sf_corr_blocked32_bytes(6, h, w)
    lib.sf_gma_flash_ws_bytes(6, h * w)
for K1, M2, a, b in itertools.product((128, 256, 324, 640, 0), (6, 64, 126, 256, 324), (1, 2, 3), (1, 2)):
    lib.sf_ffn_pair_frags(K1, M2, a, b)

# sf_gemm: every layout / epilogue / precision / algo combination over the update block's layer shapes
shapes = [(960, 640), (640, 960), (486, 324), (256, 384), (128, 128), (6, 576), (126, 384), (576, 256), (64, 192), (256, 1152)]
for (M, K), bl, epi, prec, algo, cf in itertools.product(shapes, (0, 1, 4, 5, 6), range(7), (0, 1, 2, 3), (0, 1, 2), (0, 1, 2, 3, 4)):
    g = _lib.SfGemm()
    g.A, g.B, g.C, g.bias, g.R, g.dw_w, g.dw_b, g.gamma = PTR[0], PTR[1], PTR[2], PTR[3], PTR[4], PTR[5], PTR[6], PTR[7]
    g.M, g.N, g.K, g.batch = M, 7040, K, 24
    g.lda, g.ldb, g.ldc, g.ldr = (M + 127) // 128 * 128, 7040, 7040, 7040
    g.strideA, g.strideB, g.strideC, g.strideR = 0, K * 7040, M * 7040, M * 7040
    g.a_layout, g.b_layout = (2 if prec else 0), bl
    g.alpha, g.epilogue, g.precision, g.algo = 1.0, epi, prec, algo
    g.A_hi, g.A_lo, g.lda_h, g.a_padded, g.a_k_pad = PTR[8], PTR[9], (M + 127) // 128 * 128, 1, 128
    g.c_f16, g.C16, g.strideC16 = cf, PTR[10], (M + 7) // 8 * 8 * 7040
    if K == 1152:
        g.conv3x3, g.h, g.w = 1, 55, 128
    call(lib.sf_gemm, C.byref(g), None)
    lib.sf_gemm_split_ws_floats(M, 7040, K, 24)

# sf_ffn_pair: built and unbuilt shapes, both modes, product combinations
for (K1, H, M2), mode, pm1, pm2, part in itertools.product(((128, 192, 64), (128, 192, 128), (256, 384, 192), (256, 384, 126), (256, 384, 256),
                                                             (324, 486, 256), (324, 486, 324), (640, 960, 128), (384, 576, 6)),
                                                            (0, 1), (1, 2, 3), (1, 2), (0, 1)):
    p = _lib.SfFfnPair()
    p.X, p.strideX, p.ldx = PTR[0], (K1 + 7) // 8 * 8 * 7040, 7040
    p.wstream, p.wstream_bytes = PTR[1], 1 << 26
    p.bias1, p.bias2, p.dw_w, p.dw_b = PTR[2], PTR[3], PTR[4], PTR[5]
    p.C, p.strideC, p.ldc = PTR[6], M2 * 7040, 7040
    p.C16, p.strideC16, p.ldc16 = PTR[7], (M2 + 7) // 8 * 8 * 7040, 7040
    p.N, p.batch, p.K1, p.H, p.M2, p.pm1, p.pm2, p.mode, p.c16_partial = 7040, 24, K1, H, M2, pm1, pm2, mode, part
    p.alpha1 = p.alpha2 = 1.0
    call(lib.sf_ffn_pair, C.byref(p), None)
    for xg, xs in ((128, 640 * 7040), (48, 640 * 7040), (128, 3), (-8, 0)):          # the flow head's grouped view, and what is not one
        p.x_group, p.x_group_stride = xg, xs
        call(lib.sf_ffn_pair, C.byref(p), None)
    p.x_group, p.x_group_stride = 0, 0
    p.R32, p.strideR32, p.ldr32 = PTR[8], M2 * 7040, 7040          # the fp32-residual form (mode 1 only)
    call(lib.sf_ffn_pair, C.byref(p), None)

# the SK block's back half in one launch: every shape x product count, outputs, bad arguments
for (C_, H_, M2_), pm in itertools.product(((256, 384, 192), (256, 384, 126), (384, 576, 6), (128, 192, 64), (324, 486, 256), (256, 100, 192)), (1, 2, 3)):
    t = _lib.SfSkTail()
    fr = lib.sf_sk_tail_frags(C_, H_, M2_, pm)
    t.X, t.strideX, t.ldx = PTR[0], C_ * 7040, 7040
    t.wstream, t.wstream_bytes = PTR[1], max(fr, 1) * 1024
    t.bias1, t.bias2, t.bias3 = PTR[2], PTR[3], None
    t.N, t.batch, t.C, t.H, t.M2, t.pm = 7040, 24, C_, H_, M2_, pm
    t.alpha1 = t.alpha2 = t.alpha3 = 1.0
    for y, y16, part in ((PTR[4], None, 0), (None, PTR[5], 0), (PTR[4], PTR[5], 1), (None, None, 0), (PTR[4], 0x1008, 0)):
        t.Y, t.strideY, t.ldy = y, M2_ * 7040, 7040
        t.Y16, t.strideY16, t.ldy16, t.y16_partial = y16, (M2_ + 7) // 8 * 8 * 7040, 7040, part
        call(lib.sf_sk_tail, C.byref(t), None)
    t.wstream_bytes = 1024                                          # stream too small
    call(lib.sf_sk_tail, C.byref(t), None)
call(lib.sf_sk_tail, None, None)

# correlation, GMA, depthwise, element-wise entry points: good and bad arguments
strides = (C.c_int64 * 4)(*[8 * 7040 * (55 >> l) * (128 >> l) for l in range(4)])
for pitch in (None, (C.c_int32 * 4)(128, 64, 32, 32), (C.c_int32 * 4)(100, 64, 32, 16)):
    for prec in (0, 1, 3, 7):
        call(lib.sf_corr_build_pyramid_pitched, PTR[0], PTR[1], 4 * 256 * 7040, 256 * 7040, PTR[2], PTR[3], PTR[4], PTR[5], strides,
             pitch, 8, 3, 256, 55, 128, 4, prec, PTR[6], 1 << 30, None)
    call(lib.sf_corr_lookup_pitched, PTR[2], PTR[3], PTR[4], PTR[5], strides, pitch, PTR[6], PTR[7], 324 * 7040, None, 0, 8, 3, 55, 128,
         4, 4, 0, None)
call(lib.sf_corr_build_blocked, PTR[0], PTR[1], 4 * 256 * 7040, 256 * 7040, PTR[2], 7040 * 19712, 8, 3, 256, 55, 128, PTR[3], 1 << 30, None)
call(lib.sf_corr_lookup_blocked, PTR[2], 7040 * 19712, PTR[3], None, 0, PTR[4], 328 * 7040, 8, 3, 55, 128, None)
for stride in (7040 * 38400, 7040 * 38400 - 128, 100):
    call(lib.sf_corr_build_blocked32, PTR[0], PTR[1], 4 * 256 * 7040, 256 * 7040, PTR[2], stride, 8, 3, 256, 55, 128, PTR[3], 1 << 30, None)
    call(lib.sf_corr_lookup_blocked32, PTR[2], stride, PTR[3], PTR[4], 324 * 7040, 8, 3, 55, 128, None)
call(lib.sf_corr_build_blocked32, PTR[0], PTR[1], 4 * 256 * 7040, 256 * 7040, PTR[2], 7040 * 38400, 8, 3, 256, 55, 128, PTR[3], 1000, None)
call(lib.sf_corr_lookup_blocked32, PTR[2], 7040 * 38400, None, PTR[4], 324 * 7040, 8, 3, 55, 128, None)
for n, P, qk in ((24, 7040, 1), (3, 32640, 3), (0, 7040, 1), (24, 7040, 5)):
    ws = lib.sf_gma_flash_ws_bytes(max(n, 1), P)
    call(lib.sf_gma_flash_pack_qk, PTR[0], 256 * P, PTR[1], ws, n, P, 0.088, qk if qk < 4 else 0, None)
    call(lib.sf_gma_flash_aggregate, PTR[1], ws, PTR[2], 128 * P, PTR[3], 128 * P, PTR[4], PTR[5], 128 * P, PTR[6], 128 * P, n, P, qk, 1, None)
    call(lib.sf_gma_flash_project_v, PTR[1], ws, PTR[2], 128 * P, P, PTR[3], PTR[4], 128, 1.0, min(qk, 3), n, P, None)
    pb = lib.sf_gma_stored_p_bytes(max(n, 1), P)
    call(lib.sf_gma_flash_store_p, PTR[1], ws, PTR[7], pb, n, P, qk, None)
    call(lib.sf_gma_flash_store_p, PTR[1], ws, PTR[7], pb - 1, n, P, min(qk, 3), None)          # weight buffer one byte short
    for v, vf in ((PTR[2], 0), (PTR[2], 1), (None, 0), (PTR[2], 2)):
        call(lib.sf_gma_stored_aggregate, PTR[1], ws, PTR[7], pb, v, vf, 128 * P, PTR[3], 128 * P, PTR[4], PTR[5], 128 * P, PTR[6],
             128 * P, n, P, None)
for ks, prec, C_ in itertools.product((15, 7, 9), (0, 1, 2, 3), (128, 324, 640)):
    call(lib.sf_dwconv_res_gelu, PTR[0], C_ * 7040, PTR[1], PTR[2], PTR[3], C_ * 7040, 0, 24, C_, 55, 128, ks, prec, None)
    call(lib.sf_dwconv_res_gelu_f16in, PTR[0], C_ * 7040, PTR[1], PTR[2], PTR[3], C_ * 7040, 24, C_, 55, 128, ks, prec, None)
call(lib.sf_coords_grid, PTR[0], 8, 55, 128, None)
call(lib.sf_coords_grid, None, 8, 55, 128, None)
for TT, pm, Cc, wb in itertools.product((1, 2, 3, 4, 0), (1, 2, 3), (128, 64), (None, 0)):
    t = _lib.SfTemporalBlock()
    t.X16, t.strideX, t.ldx, t.wstream = PTR[0], 128 * 7040, 7040, PTR[1]
    t.wstream_bytes = (lib.sf_temporal_block_frags(pm) * 1024) if wb is None else wb
    t.ln1_w, t.ln1_b, t.ln2_w, t.ln2_b, t.bias_proj, t.bias_fc1, t.bias_fc2 = PTR[2], PTR[3], PTR[4], PTR[5], PTR[6], PTR[7], PTR[8]
    t.Y, t.strideY, t.ldy, t.Y16, t.strideY16, t.ldy16 = PTR[9], 640 * 7040, 7040, PTR[10], 640 * 7040, 7040
    t.N, t.B, t.TT, t.C, t.H, t.pm = 7040, 8, TT, Cc, 256, pm
    call(lib.sf_temporal_block, C.byref(t), None)
call(lib.sf_temporal_block, None, None)
for pm, M, wb, hh in itertools.product((1, 2, 3), (576, 512), (None, 0), (55, 0)):
    m = _lib.SfMaskUpsample()
    m.X16, m.strideX, m.ldx, m.wstream = PTR[0], 256 * 7040, 7040, PTR[1]
    m.wstream_bytes = (lib.sf_mask_upsample_frags(pm) * 1024) if wb is None else wb
    m.bias, m.flow, m.out = PTR[2], PTR[3], PTR[4]
    m.n_img, m.h, m.w, m.K, m.M, m.pm, m.alpha = 24, hh, 128, 256, M, pm, 0.25
    call(lib.sf_mask_upsample, C.byref(m), None)
call(lib.sf_mask_upsample, None, None)
for out, us in ((PTR[0], 1000), (None, 1000), (PTR[0], 0), (PTR[0], 3000000)):
    call(lib.sf_clock_probe, out, us, None)
print(f"host sanitizer driver: {calls} calls, {rejected} rejected or failed at the launch, version {lib.sf_version()}")
