"""Named arithmetic configurations of the hot path (one place, used by bench.py, the tests and the docs).

``fp32_class`` -- the library default.  Every contraction in split precision (x = hi + lo in fp16, three MFMA products,
    fp32 accumulation: ~2^-20 relative), correlation pyramids kept in fp32 as the reference keeps them (corr.py:13; rows
    padded to whole cache lines where the dense maps would straddle them), GMA aggregation by the fused recompute kernel
    with fp16 q / k / v (as precise as the materialised matrix, whose softmax weights were stored in fp16 too).  Flows agree
    with the fp32 reference to ~2e-5 px at the headline shape and stay inside 1e-3 px on every ill-conditioned input tried.
``config2_fp16`` -- BASELINE.json configuration 2 ("Sintel-shape 436x1024 T=4 iters=15 bf16"): the arithmetic class of
    the reference's own deployment, which runs the whole network under fp16 autocast (evaluate_mf.py:1106,
    demo.py:427-456), but with fp32 accumulation everywhere and split-precision WEIGHTS: activations are rounded once
    to fp16 when they enter a contraction (f16x2), correlation volumes are fp16 cells built with single f16 products,
    the GMA aggregation is the fused recompute kernel with fp16 q / k / v (what flash_attn_func computes in the
    reference's demo).  ~1.3e-4 px EPE against the fp32 oracle at the headline shape after 15 iterations: 8x inside
    the 1e-3 budget.

``config2_mixed`` -- ``config2_fp16`` with SINGLE-product weights (the round-to-nearest fp16 image alone: one MFMA per
    product, no `lo` plane) in the 30 of 39 contraction layers where that is invisible in the flows.  Measured one layer
    at a time and cumulatively (tools/layer_ablation.py, tools/preset_sets.py): the whole motion encoder,
    GMA projections, mask head, the GRU's first FFN and most of the flow head change the 15-iteration EPE by < 2e-5 px;
    the damage of plain fp16 weights (2.5e-3 px) comes from NINE layers, which keep hi + lo: the GRU's pw / ffn2 (its output
    path), the temporal block's qkv / proj / fc1 and the flow head's pw / ffn2 (1.99e-3 px from ``flow_head.ffn2_2`` alone).
    1.7e-4 .. 2.0e-4 px on four weight / feature seeds (config2_fp16: 1.4e-4 .. 1.5e-4).

Selection of the single-product set (round 4, ``tools/preset_select.py``, data in ``profiles/r04_preset_select.jsonl``): on
weight / feature seeds 11, 12, 13 -- which neither bench.py nor any test uses -- at two conditionings (the headline shape,
15 iterations; frames -> random-init Twins_CSC features at 128 x 192, 4 iterations), layers are ranked by their worst
relative damage over the six cases and admitted in that order while the cumulative EPE stays within +25 % of
``config2_fp16`` in EVERY case.  That procedure returns exactly the nine split layers and four single-product depthwise layers
below (the same set round 3 had picked on the headline inputs alone).

Not a preset: ``precision='f16'`` (weights rounded to fp16 as well, one MFMA per product -- plain fp16-autocast
arithmetic with fp32 accumulation) runs at 245 flow-fields/s but lands at 2.5e-3 px: outside the budget.  The systematic
rounding of the WEIGHTS is what costs the accuracy, not the rounding of activations; hence split weights everywhere.
"""
from __future__ import annotations

from typing import Dict

PRESETS: Dict[str, Dict[str, object]] = {
    # (round 5: GMA through the fused kernel with fp16 q / k instead of the materialised fp16 attention matrix -- the matrix path
    # stores its softmax weights in fp16 as well; EPE unchanged on the six hard seeds, profiles/r05_hard_case_ablation.jsonl row
    # "fp32_class+flash1", 134.8 -> 127.9 ms per step)
    "fp32_class": dict(precision="f16x3", corr_dtype="f32", gma_mode="flash", flash_qk_products=1),
    "config2_fp16": dict(precision="f16x2", corr_dtype="f16", gma_mode="flash", flash_qk_products=1),
    "config2_mixed": dict(precision="f16x2", corr_dtype="f16", gma_mode="flash", flash_qk_products=1,
                          single_layers="all_but_keep"),
}
# layers that keep split (hi + lo) weights in `config2_mixed`
MIXED_KEEP_SPLIT = ("gru.pw", "gru.ffn2_0", "gru.ffn2_2", "qkv", "proj", "fc1",
                    "flow_head.pw", "flow_head.ffn2_0", "flow_head.ffn2_2")
# K x K depthwise layers with single-product weights in `config2_mixed` (tools/dw_ablation.py: the four 15 x 15 layers of the
# motion encoder move the EPE by < 5e-6 px; the GRU's 7 x 7 and the flow head's 15 x 15 add 5e-5 .. 1.8e-4 and stay split)
MIXED_SINGLE_DEPTHWISE = ("convc1.dw", "convc2.dw", "convf2.dw", "conv.dw")
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
