"""Model-forward API of the reference, kept as the drop-in surface:

* ``SKFlow_MF8(args).forward(images: list[T x [B,3,H,W] in 0..255], iters, flow_init, upsample, test_mode)``
  (reference core/models/streamflow.py:30-149)
* ``StreamFlowT4(ckpt).forward(images [B,T,3,H,W] in [-1,1], iters=15, flow_init, upsample, test_mode=True)``
  (reference demo.py:376-470)

State-dict keys of the hot path (``att.*``, ``update_block.*``) are the reference's, so published
checkpoints load.  The refinement loop runs in :class:`streamflow_amd.engine.HotPathEngine` (HIP kernels);
the encoder is the reference's ``Twins_CSC`` on the same kernels (encoders.py; stand-ins exist for tests).
"""
from __future__ import annotations

from argparse import Namespace
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from .encoders import ENCODERS
from .engine import HotPathEngine
from .gma import Attention
from .update import SKUpdateBlock_TAM_v3

UPDATE_BLOCKS = {"SKUpdateBlock_TAM_v3": SKUpdateBlock_TAM_v3}


def default_args(T: int = 4, Encoder: str = "Twins_CSC", **kw) -> Namespace:
    """The canonical StreamFlow flag set (reference scripts/infer.sh:12-26, train_mf.py:375,396,462)."""
    a = Namespace(model_name="SKFlow_MF8", Encoder=Encoder, UpdateBlock="SKUpdateBlock_TAM_v3",
                  MotionEncoder="SKMotionEncoder6_Deep_nopool_res", use_gma=True, k_conv=[1, 15],
                  PCUpdater_conv=[1, 7], T=T, num_heads=1, decoder_dim=256, mixed_precision=False, dropout=0,
                  corr_levels=4, corr_radius=4, use_graph=False, preset=None)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


class SKFlow_MF8(nn.Module):
    def __init__(self, args: Namespace):
        super().__init__()
        self.args = args
        if getattr(args, "decoder_dim", None) is None:
            args.decoder_dim = 256
        self.context_dim = cdim = args.decoder_dim // 2
        self.hidden_dim = args.decoder_dim // 2
        args.corr_levels = 4                                   # reference streamflow.py:38-39
        args.corr_radius = 4
        if args.Encoder not in ENCODERS:
            raise RuntimeError(f"Encoder '{args.Encoder}' is not available in this build (have {list(ENCODERS)}); "
                               "the other encoders of the reference are ablations no script selects (SURVEY.md 2a)")
        if args.UpdateBlock not in UPDATE_BLOCKS:
            raise RuntimeError(f"UpdateBlock '{args.UpdateBlock}' is not built (have {list(UPDATE_BLOCKS)})")
        self.fnet = ENCODERS[args.Encoder](args, norm_fn="instance")
        self.cnet = ENCODERS[args.Encoder](args, norm_fn="batch")
        self.update_block = UPDATE_BLOCKS[args.UpdateBlock](args)
        if not args.use_gma:
            raise RuntimeError("use_gma=False is not built")
        self.att = Attention(args=args, dim=cdim, heads=args.num_heads, max_pos_size=160, dim_head=cdim)
        self.ratio = 8
        self._engine: Optional[HotPathEngine] = None
        self._engine_key = None

    # -- engine management -----------------------------------------------------------------------
    def _hot_state(self):
        return {k: v for k, v in self.state_dict().items() if k.startswith(("att.", "update_block."))}

    def preset_name(self) -> str:
        """Arithmetic configuration of the refinement loop, chosen the way the reference chooses its own: `args.preset`
        names one of streamflow_amd.presets explicitly; otherwise `args.mixed_precision` (the reference's autocast switch,
        evaluate_mf.py:1106, streamflow.py:106,118,135) selects the reduced-precision class (`config2_fp16`: split weights in every
        layer; the mixed preset of the benchmark is opt-in by name) and its absence
        the fp32-class arithmetic (`fp32_class`)."""
        from . import presets
        name = getattr(self.args, "preset", None)
        if name is None:
            name = presets.MODEL_MIXED_PRESET if getattr(self.args, "mixed_precision", False) else "fp32_class"
        if name not in presets.PRESETS:
            raise RuntimeError(f"unknown preset {name!r} (have {list(presets.PRESETS)})")
        return name

    def engine(self, device) -> HotPathEngine:
        from . import presets
        params = [p for n, p in self.named_parameters() if n.startswith(("att.", "update_block."))]
        name = self.preset_name()
        key = (str(device), name) + tuple((p.data_ptr(), p._version) for p in params)
        if self._engine is None or self._engine_key != key:
            self._engine = HotPathEngine(self._hot_state(), device=device, T=self.args.T,
                                         use_graph=bool(getattr(self.args, "use_graph", False)),
                                         **presets.engine_kwargs(name))
            self._engine_key = key
        return self._engine

    # -- forward ------------------------------------------------------------------------------------
    def _features(self, images: torch.Tensor):
        """fnet over the T frames, cnet over the first T-1 (streamflow.py:112-117).  The encoders run in the arithmetic class
        of the selected preset (the reference's autocast region covers them too, streamflow.py:106-108): split precision
        for `fp32_class`, fp16 activations with the fp16 k-octet hand-over for the config-2 presets."""
        from . import presets
        from .encoders import Twins_CSC
        prec = presets.PRESETS[self.preset_name()]["precision"]
        run = lambda enc, x: enc(x, precision=prec) if isinstance(enc, Twins_CSC) else enc(x)     # (stand-in encoders: no arithmetic)
        fmaps = run(self.fnet, images).float().contiguous()
        cnets = run(self.cnet, images[:, :-1]).float().contiguous()
        return fmaps, cnets

    @torch.no_grad()
    def forward(self, images: Sequence[torch.Tensor], iters: int = 12, flow_init=None, upsample: bool = True,
                test_mode: bool = False):
        """images: list of T tensors [B,3,H,W] with values in 0..255 (reference streamflow.py:95-100).
        `upsample` is part of the reference signature but is never read in its body (streamflow.py:95-149 always
        upsamples); it is accepted and ignored here for the same reason."""
        imgs = torch.stack(list(images), dim=1)
        imgs = 2 * (imgs / 255.0) - 1.0
        return self._forward_normalised(imgs, iters, flow_init, test_mode)

    def _forward_normalised(self, imgs: torch.Tensor, iters: int, flow_init, test_mode: bool):
        B, T, C, H, W = imgs.shape
        if H % 8 or W % 8:
            raise RuntimeError("H and W must be multiples of 8 (pad with InputPadder first)")
        fmaps, cnets = self._features(imgs)
        eng = self.engine(imgs.device)
        if test_mode:
            ups, low = eng.forward(fmaps, cnets, iters=iters, flow_init=flow_init)
            ups = [u.clone() for u in ups]
            if flow_init is None:
                return ups
            return ups, [l.clone() for l in low]
        # training-mode return (reference streamflow.py:149): every iteration's upsampled prediction per pair
        preds: List[List[torch.Tensor]] = eng.forward_all_iterations(fmaps, cnets, iters=iters, flow_init=flow_init)
        return preds


class StreamFlowT4(SKFlow_MF8):
    """Self-contained T=4 model of the reference's demo (demo.py:376-470): images [B,T,3,H,W] already in [-1,1],
    iters=15, test_mode=True by default.  `ckpt` may be a path or an already loaded object ({'model': state_dict} or a
    bare dict, keys optionally prefixed 'module.').  With the default Twins_CSC encoder the whole checkpoint is loaded
    strictly (demo.py:388-389); with a stand-in encoder only the hot-path keys are (and must match exactly)."""

    def __init__(self, ckpt=None, Encoder: str = "Twins_CSC", use_graph: bool = True, preset: Optional[str] = None):
        # the reference's demo hard-wires autocast(enabled=True) (demo.py:427,439,456): reduced-precision class by default
        super().__init__(default_args(T=4, Encoder=Encoder, use_graph=use_graph, mixed_precision=True, preset=preset))
        if ckpt is not None:
            obj = torch.load(ckpt, map_location="cpu") if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "read") else ckpt
            sd = obj["model"] if isinstance(obj, dict) and "model" in obj else obj
            sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
            if Encoder == "Twins_CSC":
                self.load_state_dict(sd, strict=True)
            else:
                hot = {k: v for k, v in sd.items() if k.startswith(("att.", "update_block."))}
                mine = {k for k in self.state_dict() if k.startswith(("att.", "update_block."))}
                if set(hot) != mine:
                    raise RuntimeError(f"checkpoint hot-path keys mismatch: missing {sorted(mine - set(hot))[:5]}, "
                                       f"unexpected {sorted(set(hot) - mine)[:5]}")
                self.load_state_dict(hot, strict=False)
        for p in self.parameters():
            p.requires_grad = False

    @torch.no_grad()
    def forward(self, images: torch.Tensor, iters: int = 15, flow_init=None, upsample: bool = True,
                test_mode: bool = True):
        return self._forward_normalised(images, iters, flow_init, test_mode)
