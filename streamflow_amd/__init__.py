"""streamflow_amd: MI355X-native (gfx950) implementation of the StreamFlow multi-frame optical-flow hot path.

Drop-in surface (names, signatures and state-dict keys follow the reference):
    CorrBlock, bilinear_sampler, coords_grid, Attention, Aggregate, PCBlock4_Deep_nopool_res (SKBlock),
    SKMotionEncoder6_Deep_nopool_res, TransformerBlock, TemporalLayer2, SKUpdateBlock_TAM_v3,
    SKFlow_MF8, StreamFlowT4, InputPadder
All arithmetic runs in hand-written HIP kernels reached through the C ABI in include/streamflow_hip.h;
there is no CPU or PyTorch-op fallback.
"""
from .corr import CorrBlock  # noqa: F401
from .gma import Aggregate, Attention  # noqa: F401
from .model import SKFlow_MF8, StreamFlowT4, default_args  # noqa: F401
from .update import (PCBlock4_Deep_nopool_res, SKBlock, SKMotionEncoder6_Deep_nopool_res,  # noqa: F401
                     SKUpdateBlock_TAM_v3, TemporalLayer2, TransformerBlock)
from .utils import InputPadder, bilinear_sampler, coords_grid, forward_interpolate  # noqa: F401
from .engine import HotPathEngine  # noqa: F401
from .ops import set_precision, precision_name  # noqa: F401
from .demo import group_clips, predict_frames  # noqa: F401
