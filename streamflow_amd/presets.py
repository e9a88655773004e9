"""Named arithmetic configurations of the hot path (one place, used by bench.py, the tests and the docs).

``fp32_class`` -- the library default.  Every contraction in split precision (x = hi + lo in fp16, three MFMA products,
    fp32 accumulation: ~2^-20 relative), correlation pyramids kept in fp32 as the reference keeps them (corr.py:13; rows
    padded to whole cache lines where the dense maps would straddle them), GMA aggregation by the fused recompute kernel
    with SPLIT q and k (three MFMA products per logit: the logits are fp32-class, 2^-20; only the softmax weights and v enter
    the second contraction in fp16, as the materialised matrix of rounds 1-4 stored them).  Round 5 shipped ONE fp16 product per
    logit here: that puts |logit| 2^-11 of absolute error on the logits, harmless for the random-weight networks of the test
    inputs (small logits, EPE unchanged) but 12x looser on peaked attention (logits of +-40: 2.5e-2 against 2e-3 in
    tests/test_gpu_parity.py::test_gma_flash_kernel_vs_float64) -- not fp32-class for trained GMA weights, so the
    default went back to three products (127.9 -> 140.7 ms per step) -- and the engine now computes them ONCE per clip: with
    split-precision logits the fused path keeps its fp16 softmax weights for the refinement loop and streams them every iteration
    (EngineOptions.stored_auto_px; bit-identical to the recompute, 142.8 -> 132.2 ms per step).  Flows agree with the fp32 reference
    to ~2e-5 px at the headline shape and stay inside 1e-3 px on every ill-conditioned input tried.
``config2_fp16`` -- BASELINE.json configuration 2 ("Sintel-shape 436x1024 T=4 iters=15 bf16"): the arithmetic class of
    the reference's own deployment, which runs the whole network under fp16 autocast (evaluate_mf.py:1106,
    demo.py:427-456), but with fp32 accumulation everywhere and split-precision WEIGHTS: activations are rounded once
    to fp16 when they enter a contraction (f16x2), correlation volumes are fp16 cells built with single f16 products,
    the GMA aggregation is the fused recompute kernel with fp16 q / k / v (what flash_attn_func computes in the
    reference's demo).  ~1.3e-4 px EPE against the fp32 oracle at the headline shape after 15 iterations: 8x inside
    the 1e-3 budget.

``config2_mixed`` -- ``config2_fp16`` with SINGLE-product weights (the round-to-nearest fp16 image alone: one MFMA per
    product, no `lo` plane) in the 22 of 39 contraction layers and 2 of 6 depthwise layers where that stays inside the
    selection caps below.  The layers that keep hi + lo are the OUTPUT halves of the blocks (pw, ffn2.0, ffn2.2 of conv, convc2,
    gru and the flow head, convc1.ffn2_0) and the whole temporal block; every ffn1 pair, the flow branch, the GMA projections and
    the mask head are single.

Selection of the single-product set (round 5, ``tests/analysis/preset_select_v2.py``, data in ``profiles/r05_preset_select.jsonl``):
every case is referenced to the CPU ORACLE; none of them is a bench or test input.  Headline-type cases (55 x 128, 15
iterations, seeds 11-13) may lose 25 % against ``config2_fp16``; ill-conditioned cases (frames -> random-weight Twins_CSC
features -> 128 x 192, 4 iterations; ten seeds) must stay under 0.6e-3 of the mean flow magnitude.  Layers are ranked by the
worst share of a cap's head-room they use alone and admitted greedily while every case holds.  Result: <= 2.3e-4 px on the
headline cases, <= 0.56e-3 of the flow on the ten selection seeds, 0.35 - 0.49e-3 on five further seeds that took no part in
the selection (the reference's own fp16-autocast arithmetic sits at 0.75 - 1.2e-3 of the flow on these inputs,
``profiles/r05_reference_autocast_deviation.jsonl``).
(Round 4's set -- 30 single GEMM layers, 4 single depthwise -- had been selected on data from a defective build and against
the fp32-class engine, ``profiles/r05_hard_case_at_88f9fb4.jsonl``; re-measured against the oracle it reached 1.4e-3 of the flow
on one hard seed.)

Not a preset: ``precision='f16'`` (weights rounded to fp16 as well, one MFMA per product -- plain fp16-autocast
arithmetic with fp32 accumulation) runs at 245 flow-fields/s but lands at 2.5e-3 px: outside the budget.  The systematic
rounding of the WEIGHTS is what costs the accuracy, not the rounding of activations; hence split weights everywhere.
"""
from __future__ import annotations

from typing import Dict

PRESETS: Dict[str, Dict[str, object]] = {
    # (GMA through the fused kernel -- no N x N matrix -- with split q / k: fp32-class logits whatever their magnitude; round 5's
    # single fp16 product per logit was 9 % faster and indistinguishable on the random-weight test networks, but is not fp32-class
    # on peaked attention: see the module docstring)
    "fp32_class": dict(precision="f16x3", corr_dtype="f32", gma_mode="flash", flash_qk_products=3),
    "config2_fp16": dict(precision="f16x2", corr_dtype="f16", gma_mode="flash", flash_qk_products=1),
    "config2_mixed": dict(precision="f16x2", corr_dtype="f16", gma_mode="flash", flash_qk_products=1,
                          single_layers="all_but_keep"),
}
# layers that keep split (hi + lo) weights in `config2_mixed` (profiles/r05_preset_select.jsonl, last line: "keep_split")
MIXED_KEEP_SPLIT = ("conv.ffn2_0", "conv.ffn2_2", "conv.pw", "convc1.ffn2_0", "convc2.ffn2_0", "convc2.ffn2_2", "convc2.pw",
                    "fc1", "fc2", "flow_head.ffn2_0", "flow_head.ffn2_2", "flow_head.pw", "gru.ffn2_0", "gru.ffn2_2", "gru.pw",
                    "proj", "qkv")
# K x K depthwise layers with single-product weights in `config2_mixed` (same selection run: "single_depthwise")
MIXED_SINGLE_DEPTHWISE = ("convf2.dw", "conv.dw")
BENCH_PRESET = "config2_mixed"
# what `args.mixed_precision = True` (the reference's autocast switch) selects through the model API: the all-split form of the
# class.  The mixed preset is opt-in (`args.preset = "config2_mixed"`, or bench.py's default): its single-product layer set
# is a measured trade (selection below) and an application with differently conditioned weights should re-run the selection.
MODEL_MIXED_PRESET = "config2_fp16"


def engine_kwargs(name: str) -> Dict[str, object]:
    if name not in PRESETS:
        raise RuntimeError(f"unknown preset {name!r} (have {list(PRESETS)})")
    kw = dict(PRESETS[name])
    if kw.get("single_layers") == "all_but_keep":
        from .engine import HotPathWeights
        names = list(HotPathWeights.PLAIN_LAYERS) + [f"{b}.{l}" for b in HotPathWeights.SK_BLOCKS for l in HotPathWeights.SK_LAYERS]
        kw["single_layers"] = tuple(n for n in names if n not in MIXED_KEEP_SPLIT) + MIXED_SINGLE_DEPTHWISE
    return kw
