#!/usr/bin/env python3
"""Generate golden vectors by EXECUTING THE REFERENCE's own hot-path modules on CPU.

Run in the build container only (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden.py

It imports ``core/corr.py``, ``core/gma.py``, ``core/utils/utils.py`` directly and
``core/update.py`` / ``core/models/streamflow.py`` through two in-memory stand-ins that are this
repo's own code (nothing from the reference is copied into the repo):

* a ``timm`` stub exposing the four symbols those files import (``Attention``, ``Mlp``,
  ``DropPath``, ``to_2tuple``), written from timm's published definition (the package is not
  installed in this image and is unpinned upstream, so that boundary is "parity unpinned");
* an ``encoders`` stub whose ``InjectEncoder`` returns pre-generated feature maps, because the
  hot path starts at encoder outputs (reference streamflow.py:106-108).

Inputs and weights are rebuilt from seeds by ``streamflow_amd.synthetic`` (also used by the
tests), so the ``.npz`` files hold only the reference OUTPUTS plus the seeds/shapes.
"""
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"

from streamflow_amd import synthetic as syn  # noqa: E402
from tests import cases  # noqa: E402


# ------------------------------------------------------------------------------------------
# stand-ins (own code)
# ------------------------------------------------------------------------------------------
def _module(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


class _TimmAttention(nn.Module):
    """timm.models.vision_transformer.Attention (published semantics, qk_norm=False, no dropout)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_norm=False, attn_drop=0.0, proj_drop=0.0,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        assert not qk_norm and attn_drop == 0.0 and proj_drop == 0.0
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class _TimmMlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _DropPath(nn.Identity):
    def __init__(self, drop_prob=0.0):
        super().__init__()


# ---- timm.models.twins stand-in (published semantics of timm/models/twins.py; own code) ------------------------------------
class _LocallyGroupedAttn(nn.Module):
    def __init__(self, dim, num_heads=8, attn_drop=0.0, proj_drop=0.0, ws=1):
        super().__init__()
        assert ws != 1 and dim % num_heads == 0
        self.dim, self.num_heads, self.ws = dim, num_heads, ws
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, size):
        import torch.nn.functional as F
        B, N, C = x.shape
        H, W = size
        x = x.view(B, H, W, C)
        pad_r = (self.ws - W % self.ws) % self.ws
        pad_b = (self.ws - H % self.ws) % self.ws
        x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        _, Hp, Wp, _ = x.shape
        _h, _w = Hp // self.ws, Wp // self.ws
        x = x.reshape(B, _h, self.ws, _w, self.ws, C).transpose(2, 3)
        qkv = self.qkv(x).reshape(B, _h * _w, self.ws * self.ws, 3, self.num_heads, C // self.num_heads).permute(3, 0, 1, 4, 2, 5)
        q, k, v = qkv.unbind(0)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        x = (attn @ v).transpose(2, 3).reshape(B, _h, _w, self.ws, self.ws, C)
        x = x.transpose(2, 3).reshape(B, _h * self.ws, _w * self.ws, C)
        if pad_r > 0 or pad_b > 0:
            x = x[:, :H, :W, :].contiguous()
        return self.proj(x.reshape(B, N, C))


class _GlobalSubSampleAttn(nn.Module):
    def __init__(self, dim, num_heads=8, attn_drop=0.0, proj_drop=0.0, sr_ratio=1):
        super().__init__()
        self.dim, self.num_heads, self.sr_ratio = dim, num_heads, sr_ratio
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=True)
        self.kv = nn.Linear(dim, dim * 2, bias=True)
        self.proj = nn.Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)
        else:
            self.sr = self.norm = None

    def forward(self, x, size):
        B, N, C = x.shape
        q = self.q(x).reshape(B, N, self.num_heads, C // self.num_heads).permute(0, 2, 1, 3)
        if self.sr is not None:
            x = x.permute(0, 2, 1).reshape(B, C, *size)
            x = self.sr(x).reshape(B, C, -1).permute(0, 2, 1)
            x = self.norm(x)
        kv = self.kv(x).reshape(B, -1, 2, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        k, v = kv.unbind(0)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class _TwinsBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, sr_ratio, ws):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = (_GlobalSubSampleAttn(dim, num_heads, 0.0, 0.0, sr_ratio) if ws == 1
                     else _LocallyGroupedAttn(dim, num_heads, 0.0, 0.0, ws))
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _TimmMlp(dim, int(dim * mlp_ratio))

    def forward(self, x, size):
        x = x + self.attn(self.norm1(x), size)
        return x + self.mlp(self.norm2(x))


class _PosConv(nn.Module):
    def __init__(self, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Sequential(nn.Conv2d(in_chans, embed_dim, 3, 1, 1, bias=True, groups=embed_dim))

    def forward(self, x, size):
        B, N, C = x.shape
        feat = x.transpose(1, 2).view(B, C, *size)
        return (self.proj(feat) + feat).flatten(2).transpose(1, 2)


class _TwinsSvtLarge(nn.Module):
    """What `timm.create_model('twins_svt_large')` returns, as far as twins_csc.py touches it: four stages of
    patch_embeds / pos_drops / blocks / pos_block, a final norm and a head (stages 3-4 and the head are deleted by the
    reference, twins_csc.py:52-57, so they are placeholders here)."""

    def __init__(self):
        super().__init__()
        dims, heads, depths, srs = [128, 256, 512, 1024], [4, 8, 16, 32], [2, 2, 18, 2], [8, 4, 2, 1]
        self.patch_embeds = nn.ModuleList([nn.Identity() for _ in dims])          # replaced by the reference (:42-47)
        self.pos_drops = nn.ModuleList([nn.Dropout(p=0.0) for _ in dims])
        self.blocks = nn.ModuleList([
            nn.ModuleList([_TwinsBlock(dims[k], heads[k], 4, srs[k], 1 if i % 2 == 1 else 7) for i in range(depths[k])])
            if k < 2 else nn.ModuleList() for k in range(4)])
        self.pos_block = nn.ModuleList([_PosConv(d, d) if k < 2 else nn.Identity() for k, d in enumerate(dims)])
        self.norm = nn.LayerNorm(1024, eps=1e-6)
        self.head = nn.Linear(1024, 1000)


class InjectEncoder(nn.Module):
    """Stand-in encoder: ignores pixel values, returns the tensor stored in ``.features``
    (sliced to the number of frames it is called with)."""

    def __init__(self, args=None, norm_fn=None):
        super().__init__()
        self.features = None

    def forward(self, x):
        return self.features[:, : x.shape[1]]


def install_stubs():
    timm_mod = _module("timm")

    def create_model(name, pretrained=False, **kw):
        assert name == "twins_svt_large" and not pretrained
        return _TwinsSvtLarge()
    timm_mod.create_model = create_model
    _module("timm.models")
    tw = _module("timm.models.twins")
    tw.GlobalSubSampleAttn, tw.LocallyGroupedAttn = _GlobalSubSampleAttn, _LocallyGroupedAttn
    vt = _module("timm.models.vision_transformer")
    vt.Attention = _TimmAttention
    for name in ("timm.layers", "timm.models.layers"):
        m = _module(name)
        m.Mlp, m.DropPath, m.to_2tuple = _TimmMlp, _DropPath, (lambda x: (x, x))
    enc = _module("encoders")
    enc.InjectEncoder = InjectEncoder
    enc.__all__ = ["InjectEncoder"]
    sys.path.insert(0, os.path.join(REF, "core"))


def ref_args(T):
    return Namespace(decoder_dim=256, corr_levels=4, corr_radius=4, k_conv=list(syn.K_CONV),
                     PCUpdater_conv=list(syn.GRU_CONV), T=T, use_gma=True, num_heads=1,
                     Encoder="InjectEncoder", UpdateBlock="SKUpdateBlock_TAM_v3",
                     mixed_precision=False, dropout=0)


ONLY = set(sys.argv[1:])          # optional: fixture names to (re)write; default all


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def sub(params, prefix):
    """state-dict slice with `prefix.` stripped."""
    return {k[len(prefix) + 1:]: v for k, v in params.items() if k.startswith(prefix + ".")}


def main():
    torch.set_num_threads(4)
    install_stubs()
    import corr as ref_corr
    import gma as ref_gma
    import update as ref_update
    from utils import utils as ref_utils
    import models.streamflow as ref_model
    from einops import rearrange

    with torch.no_grad():
        # ---- a5 coords_grid -------------------------------------------------------------
        save("coords_grid", out=ref_utils.coords_grid(2, 7, 9).contiguous(), batch=2, ht=7, wd=9)

        # ---- a4 bilinear_sampler (incl. out-of-range and exactly-integer coordinates) ----
        img, crd = cases.bilinear_inputs()
        out_m, in_mask = ref_utils.bilinear_sampler(img, crd, mask=True)          # utils.py:75-77: strict in-bounds mask
        assert torch.equal(out_m, ref_utils.bilinear_sampler(img, crd))
        save("bilinear_sampler", out=out_m, mask=in_mask)

        # ---- a1-a3 corr build, pyramid, lookup (odd grid 17x19 -> levels 17x19, 8x9, 4x4, 2x2) ----
        for tag, (B, D, h, w, seed) in cases.CORR_CASES.items():
            f1, f2, coords, ident = cases.corr_inputs(tag)
            assert torch.equal(ident, ref_utils.coords_grid(B, h, w))
            blk = ref_corr.CorrBlock(f1, f2, num_levels=4, radius=4)
            save(tag, lookup=blk(coords), lookup_identity=blk(ident),
                 **{f"level{i}": v for i, v in enumerate(blk.corr_pyramid)})

        # ---- a6/a7 GMA attention + aggregate ------------------------------------------------
        P, inp, mf = cases.gma_inputs()
        args = ref_args(4)
        att = ref_gma.Attention(args=args, dim=128, heads=1, max_pos_size=160, dim_head=128)
        att.load_state_dict(sub(P, "att"), strict=True)
        agg = ref_gma.Aggregate(args=args, dim=128, dim_head=128, heads=1)
        agg.load_state_dict(sub(P, "update_block.aggregator"), strict=True)
        assert agg.project is None
        attn = att(inp)
        save("gma", attn=attn, aggregate=agg(attn, mf))

        # ---- a8 SKBlock instances --------------------------------------------------------
        P = syn.make_params(cases.SKBLOCK_SEED, 4)
        outs = {}
        for name, cin, cout, kc in cases.SKBLOCK_CASES:
            m = ref_update.PCBlock4_Deep_nopool_res(cin, cout, list(kc))
            m.load_state_dict(sub(P, "update_block." + name), strict=True)
            outs[name.replace(".", "_")] = m(cases.skblock_inputs(name, cin))
        save("skblock", **outs)

        # ---- a9 motion encoder, a10 temporal block, a11 update block -------------------------
        for tag, (B, T, h, w, seed) in cases.UPDATE_CASES.items():
            Pn = T - 1
            P, nets, inps, corrs, flows, attn = cases.update_inputs(tag)
            ub = ref_update.SKUpdateBlock_TAM_v3(ref_args(T))
            ub.load_state_dict(sub(P, "update_block"), strict=True)
            mf = ub.encoder(flows, corrs)
            tok = rearrange(mf, "(B T) C H W -> (B H W) T C", T=Pn)
            mft = ub.transformer_block(tok, HW=(h, w))
            n2, masks, dflow = ub(nets, inps, corrs, flows, attn, T=Pn)
            save(tag, motion=mf, temporal=mft, nets=n2, masks=masks, dflow=dflow)

        # ---- K12 convex upsample ---------------------------------------------------------
        flow, mask = cases.upsample_inputs()
        model = ref_model.SKFlow_MF8(ref_args(4))
        save("upsample", out=model.upsample_flow(flow, mask))

        # ---- f4 forward_interpolate (scipy griddata nearest; warm start of the next clip) -------------
        for tag in cases.INTERP_CASES:
            save(tag, out=ref_utils.forward_interpolate(cases.interp_inputs(tag)))

        # ---- f1 Twins_CSC encoder: the reference's PatchEmbed-over-(T H) and stage loop over the timm stand-in ----
        import importlib.util
        spec = importlib.util.spec_from_file_location("ref_twins_csc", os.path.join(REF, "core", "encoders", "twins_csc.py"))
        ref_twins = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref_twins)
        for tag in cases.TWINS_CASES:
            P, x = cases.twins_inputs(tag)
            enc = ref_twins.Twins_CSC(pretrained=False)
            enc.load_state_dict(dict(P), strict=True)            # key names / shapes of the checkpoint contract
            enc.eval()
            save(tag, out=enc(x))

        # ---- f3 flow file formats: the reference's own core/utils/frame_utils.py, imported over in-memory cv2 / h5py stubs
        # (readFlow / writeFlow / readPFM touch neither; the KITTI functions hand their uint16 arithmetic to cv2.imread /
        # cv2.imwrite, which the stub captures: the PNG codec itself is cv2's, not the reference's) ----
        if not ONLY or "frame_utils" in ONLY:
            import tempfile
            cap = {}
            cv2 = _module("cv2")
            cv2.setNumThreads = lambda n: None
            cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
            cv2.IMREAD_ANYDEPTH, cv2.IMREAD_COLOR = 2, 1
            cv2.imwrite = lambda fn, arr: cap.__setitem__("written", np.array(arr))
            cv2.imread = lambda fn, flags: cap["to_read"].copy()
            _module("h5py")
            from utils import frame_utils as ref_fu
            flow, kitti, pfm3, pfm1 = cases.flow_io_inputs()
            with tempfile.TemporaryDirectory() as td:
                fn = os.path.join(td, "a.flo")
                ref_fu.writeFlow(fn, flow)
                flo_bytes = np.fromfile(fn, np.uint8)
                flo_back = ref_fu.readFlow(fn)
                ref_fu.writeFlowKITTI(os.path.join(td, "k.png"), flow)
                cap["to_read"] = kitti
                kflow, kvalid = ref_fu.readFlowKITTI(os.path.join(td, "k.png"))
                pf = {}
                for tag, arr, hdr, end in (("pfm3_le", pfm3, b"PF", "<"), ("pfm1_be", pfm1, b"Pf", ">")):
                    fnp = os.path.join(td, tag + ".pfm")
                    with open(fnp, "wb") as f:
                        f.write(hdr + b"\n" + f"{arr.shape[1]} {arr.shape[0]}\n".encode() + (b"-1.0\n" if end == "<" else b"1.0\n"))
                        np.flipud(arr).astype(end + "f4").tofile(f)
                    pf[tag + "_file"] = np.fromfile(fnp, np.uint8)
                    pf[tag] = np.ascontiguousarray(ref_fu.readPFM(fnp)).astype(np.float32)
            save("frame_utils", flo_bytes=flo_bytes, flo_back=flo_back, kitti_written_bgr=cap["written"], kitti_flow=kflow,
                 kitti_valid=kvalid, **pf)

        # ---- a12 full forward through SKFlow_MF8.forward (config-1 style plumbing) -----------
        for tag, (B, T, H, W, iters, seed, use_init) in cases.FORWARD_CASES.items():
            P, fmaps, cnets, finit, iters = cases.forward_inputs(tag)
            model = ref_model.SKFlow_MF8(ref_args(T))
            model.load_state_dict(dict(P), strict=True)       # encoders stand-ins hold no parameters
            model.fnet.features = fmaps
            model.cnet.features = cnets
            model.eval()
            images = [torch.zeros(B, 3, H, W) for _ in range(T)]
            if use_init:
                ups, low = model(images, iters=iters, flow_init=finit, test_mode=True)
                save(tag, **{f"up{i}": v for i, v in enumerate(ups)}, **{f"low{i}": v for i, v in enumerate(low)})
            else:
                ups = model(images, iters=iters, test_mode=True)
                extra = {}
                if H * W <= 128 * 192:
                    # training-mode return (per pair, per iteration): keep iteration 0 as an extra pin
                    allp = model(images, iters=iters, test_mode=False)
                    extra = {f"first{i}": allp[i][0] for i in range(T - 1)}
                save(tag, **{f"up{i}": v for i, v in enumerate(ups)}, **extra)


if __name__ == "__main__":
    main()
