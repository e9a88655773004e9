"""ctypes binding of libstreamflow_hip.so (the C ABI declared in include/streamflow_hip.h).

There is NO fallback: if the shared library is missing, fails to load, or a tensor is not a
contiguous fp32 CUDA(HIP) tensor, the call raises.  The product path never routes through
PyTorch ops or the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  (imported first so the process-wide HIP runtime is torch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SF_HIP_LIB: load an alternative build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("SF_HIP_LIB") or os.path.join(_HERE, "libstreamflow_hip.so")

LAYOUT_K_MAJOR, LAYOUT_K_MINOR, LAYOUT_SPLIT_F16, LAYOUT_F16_K_MINOR, LAYOUT_F16_K_MAJOR, LAYOUT_F16_KOCT, LAYOUT_SPLIT_KOCT = 0, 1, 2, 3, 4, 5, 6
PRECISION_FP32, PRECISION_F16X3, PRECISION_F16X2, PRECISION_F16 = 0, 1, 2, 3
EPI_NONE, EPI_GELU, EPI_RELU, EPI_RES, EPI_RES_GELU, EPI_RES_GELU_DW1, EPI_AXPY = range(7)
ALGO_AUTO, ALGO_TILED, ALGO_BSTAT = 0, 1, 2

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float


class SfGemm(C.Structure):
    _fields_ = [
        ("A", _vp), ("B", _vp), ("C", _vp), ("bias", _vp), ("R", _vp), ("dw_w", _vp), ("dw_b", _vp), ("gamma", _vp),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("batch", C.c_int32),
        ("lda", _i64), ("ldb", _i64), ("ldc", _i64), ("ldr", _i64),
        ("strideA", _i64), ("strideB", _i64), ("strideC", _i64), ("strideR", _i64),
        ("a_layout", C.c_int32), ("b_layout", C.c_int32),
        ("b_group", C.c_int32), ("b_group_stride", _i64),
        ("r_group", C.c_int32), ("r_group_stride", _i64),
        ("conv3x3", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("alpha", _f), ("epilogue", C.c_int32), ("precision", C.c_int32),
        ("A_hi", _vp), ("A_lo", _vp), ("lda_h", _i64), ("a_padded", C.c_int32),
        ("k_splits", C.c_int32), ("split_stride", _i64),
        ("split_ws", _vp), ("split_ws_floats", _i64),
        ("c_f16", C.c_int32),
        ("C16", _vp), ("strideC16", _i64),
        ("r_f16", C.c_int32),
        ("a_k_pad", C.c_int32), ("algo", C.c_int32),
    ]


class SfFfnPair(C.Structure):
    _fields_ = [
        ("X", _vp), ("strideX", _i64), ("ldx", _i64),
        ("wstream", _vp), ("wstream_bytes", _i64),
        ("bias1", _vp), ("bias2", _vp), ("dw_w", _vp), ("dw_b", _vp),
        ("C", _vp), ("strideC", _i64), ("ldc", _i64),
        ("C16", _vp), ("strideC16", _i64), ("ldc16", _i64),
        ("N", C.c_int32), ("batch", C.c_int32), ("K1", C.c_int32), ("H", C.c_int32), ("M2", C.c_int32),
        ("pm1", C.c_int32), ("pm2", C.c_int32), ("mode", C.c_int32), ("gelu_out", C.c_int32), ("c16_partial", C.c_int32),
        ("alpha1", _f), ("alpha2", _f),
        ("x_group", C.c_int32), ("x_group_stride", _i64),
        ("R32", _vp), ("strideR32", _i64), ("ldr32", _i64), ("r32_group_stride", _i64),
    ]


class SfSkTail(C.Structure):
    _fields_ = [
        ("X", _vp), ("strideX", _i64), ("ldx", _i64),
        ("wstream", _vp), ("wstream_bytes", _i64),
        ("bias1", _vp), ("bias2", _vp), ("bias3", _vp),
        ("Y", _vp), ("strideY", _i64), ("ldy", _i64),
        ("Y16", _vp), ("strideY16", _i64), ("ldy16", _i64),
        ("N", C.c_int32), ("batch", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("M2", C.c_int32), ("pm", C.c_int32),
        ("gelu_out", C.c_int32), ("y16_partial", C.c_int32),
        ("alpha1", _f), ("alpha2", _f), ("alpha3", _f),
    ]


class SfTemporalBlock(C.Structure):
    _fields_ = [
        ("X16", _vp), ("strideX", _i64), ("ldx", _i64),
        ("wstream", _vp), ("wstream_bytes", _i64),
        ("ln1_w", _vp), ("ln1_b", _vp), ("ln2_w", _vp), ("ln2_b", _vp), ("bias_proj", _vp), ("bias_fc1", _vp), ("bias_fc2", _vp),
        ("Y", _vp), ("strideY", _i64), ("ldy", _i64),
        ("Y16", _vp), ("strideY16", _i64), ("ldy16", _i64),
        ("N", C.c_int32), ("B", C.c_int32), ("TT", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("pm", C.c_int32),
        ("alpha_qkv", _f), ("alpha_proj", _f), ("alpha_fc1", _f), ("alpha_fc2", _f), ("ss_proj", _f), ("ss_fc2", _f),
        ("eps", _f), ("scale", _f),
    ]


class SfMaskUpsample(C.Structure):
    _fields_ = [
        ("X16", _vp), ("strideX", _i64), ("ldx", _i64),
        ("wstream", _vp), ("wstream_bytes", _i64),
        ("bias", _vp), ("flow", _vp), ("out", _vp),
        ("n_img", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("K", C.c_int32), ("M", C.c_int32), ("pm", C.c_int32),
        ("alpha", _f),
    ]


# name -> (restype, argtypes); must list every symbol of include/streamflow_hip.h
SIGNATURES = {
    "sf_version": (_i, []),
    "sf_last_error": (C.c_char_p, []),
    "sf_coords_grid": (_i, [_vp, _i, _i, _i, _vp]),
    "sf_bilinear_sampler": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sf_corr_build_pyramid": (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, C.POINTER(_i64), _i, _i, _i, _i, _i, _i, _i,
                                   _vp, _i64, _vp]),
    "sf_corr_build_ws_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "sf_corr_lookup": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_i64), _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sf_corr_build_pyramid_pitched": (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, C.POINTER(_i64), C.POINTER(C.c_int32),
                                           _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "sf_corr_lookup_pitched": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_i64), C.POINTER(C.c_int32), _vp, _vp, _i64, _vp, _i64,
                                    _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sf_corr_blocked_geometry": (_i, [_i, _i, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.POINTER(_i64)]),
    "sf_corr_blocked_bytes": (_i64, [_i, _i, _i]),
    "sf_corr_build_blocked_ws_bytes": (_i64, [_i, _i, _i, _i]),
    "sf_corr_build_blocked": (_i, [_vp, _vp, _i64, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "sf_corr_lookup_blocked": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp]),
    "sf_corr_blocked32_geometry": (_i, [_i, _i, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(_i64)]),
    "sf_corr_blocked32_bytes": (_i64, [_i, _i, _i]),
    "sf_corr_build_blocked32_ws_bytes": (_i64, [_i, _i, _i, _i]),
    "sf_corr_build_blocked32": (_i, [_vp, _vp, _i64, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "sf_corr_lookup_blocked32": (_i, [_vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "sf_gemm": (_i, [C.POINTER(SfGemm), _vp]),
    "sf_gemm_split_ws_floats": (_i64, [_i, _i, _i, _i]),
    "sf_gma_flash_ws_bytes": (_i64, [_i, _i]),
    "sf_gma_flash_pack_qk": (_i, [_vp, _i64, _vp, _i64, _i, _i, _f, _i, _vp]),
    "sf_gma_flash_aggregate": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp]),
    "sf_gma_flash_aggregate_f16v": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp]),
    "sf_gma_flash_project_v": (_i, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _i, _f, _i, _i, _i, _vp]),
    "sf_gma_stored_p_bytes": (_i64, [_i, _i]),
    "sf_gma_flash_store_p": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _vp]),
    "sf_gma_stored_aggregate": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i, _i, _vp]),
    "sf_ffn_pair": (_i, [C.POINTER(SfFfnPair), _vp]),
    "sf_ffn_pair_frags": (_i, [_i, _i, _i, _i]),
    "sf_sk_tail": (_i, [C.POINTER(SfSkTail), _vp]),
    "sf_sk_tail_frags": (_i, [_i, _i, _i, _i]),
    "sf_sk_tail_layout": (_i, [_i, _i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sf_temporal_block": (_i, [C.POINTER(SfTemporalBlock), _vp]),
    "sf_temporal_block_frags": (_i, [_i]),
    "sf_mask_upsample": (_i, [C.POINTER(SfMaskUpsample), _vp]),
    "sf_mask_upsample_frags": (_i, [_i]),
    "sf_softmax_rows": (_i, [_vp, _i64, _i, _vp, _vp]),
    "sf_splitk_combine": (_i, [_vp, _i64, _i, _i64, _vp, _i64, _vp, _vp, _i64, _i, _i64, _vp]),
    "sf_dwconv_res_gelu": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sf_dwconv_res_gelu_f16in": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp]),
    "sf_layernorm_cm": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _f, _vp]),
    "sf_temporal_attn": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sf_temporal_attn_f16in": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sf_pack_koct": (_i, [_vp, _i64, _i, _i, _i, _vp, _i64, _vp]),
    "sf_clock_probe": (_i, [_vp, _i, _vp]),
    "sf_context_split": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _vp]),
    "sf_flow_update": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp]),
    "sf_window_attn": (_i, [_vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp]),
    "sf_window_attn_mfma": (_i, [_vp, _i64, _i, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sf_subsample_attn": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp]),
    "sf_subsample_attn_ws_bytes": (_i64, [_i, _i, _i]),
    "sf_subsample_attn_mfma": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp, _i64, _i, _vp]),
    "sf_dwconv3x3_res": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "sf_upsample_flow": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "sf_forward_interpolate": (_i, [_vp, _vp, _i, _i, _i, _vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (once).  Raises RuntimeError if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -m streamflow_amd.build` (needs hipcc); there is no CPU/PyTorch fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.sf_version() < 116:
        raise RuntimeError("libstreamflow_hip.so is too old; rebuild")
    _lib = lib
    return lib


CALLS = 0          # entry-point calls checked so far (a diagnostic counter: HotPathEngine reads the difference over a graph capture)


def check(status: int, what: str = "") -> None:
    global CALLS
    CALLS += 1
    if status != 0:
        msg = load().sf_last_error().decode(errors="replace")
        raise RuntimeError(f"libstreamflow_hip {what} failed ({status}): {msg}")


def ptr(t: Optional[torch.Tensor], offset_floats: int = 0) -> Optional[int]:
    """Raw device pointer of a contiguous fp32 HIP tensor (+ element offset)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("streamflow_amd kernels need tensors on the MI355X (cuda:N); got a CPU tensor "
                           "(there is no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("expected a contiguous tensor")
    return t.data_ptr() + 4 * offset_floats


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream
