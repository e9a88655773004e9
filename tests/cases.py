"""Seeded input builders shared by the golden generator (tests/golden/make_golden.py), the oracle
tests and the GPU parity tests, so all three see byte-identical inputs."""
import torch

from streamflow_amd import synthetic as syn

CORR_CASES = {"corr_odd": (1, 32, 17, 19, 21), "corr_b2": (2, 16, 16, 18, 22)}       # B, D, h, w, seed
UPDATE_CASES = {"update_T4": (1, 4, 9, 12, 51), "update_T3_b2": (2, 3, 8, 8, 52),
                "update_T2": (1, 2, 7, 10, 53)}                                      # B, T, h, w, seed
SKBLOCK_CASES = [("encoder.convc1", 324, 256, syn.K_CONV), ("encoder.convf2", 128, 64, syn.K_CONV),
                 ("encoder.conv", 256, 126, syn.K_CONV), ("gru", 640, 128, syn.GRU_CONV),
                 ("flow_head", 384, 6, syn.K_CONV)]
SKBLOCK_SEED, SKBLOCK_HW = 41, (9, 10)
FORWARD_CASES = {"forward_T4": (1, 4, 128, 192, 4, 71, False),
                 "forward_T3_b2_init": (2, 3, 128, 128, 3, 72, True),
                 "forward_demo256": (1, 4, 256, 256, 4, 73, False)}                   # B,T,H,W,iters,seed,init
GMA_CASE = (31, 2, 12, 16)                                                           # seed, BT, h, w
UPSAMPLE_SEED = 61
INTERP_CASES = {"interp_a": (81, 23, 31, 4.0), "interp_b": (82, 16, 40, 12.0), "interp_c": (83, 9, 7, 1.0)}   # seed, h, w, flow scale


def _grid(B, h, w):
    xs = torch.arange(w, dtype=torch.float32).view(1, w).expand(h, w)
    ys = torch.arange(h, dtype=torch.float32).view(h, 1).expand(h, w)
    return torch.stack([xs, ys], 0)[None].repeat(B, 1, 1, 1)


def bilinear_inputs():
    img = syn.randn(11, "bs.img", (5, 3, 6, 7))
    crd = syn.randn(11, "bs.coords", (5, 4, 9, 2), 3.0) + 3.0
    crd[0, 0, :, :] = torch.tensor([[0.0, 0.0], [6.0, 5.0], [-1.0, 2.0], [7.0, 2.0], [3.0, -1.0],
                                    [3.0, 6.0], [2.5, 2.5], [6.0, 0.0], [0.0, 5.0]])
    return img, crd


def corr_inputs(tag):
    B, D, h, w, seed = CORR_CASES[tag]
    f1 = syn.randn(seed, "corr.f1", (B, D, h, w))
    f2 = syn.randn(seed, "corr.f2", (B, D, h, w))
    coords = _grid(B, h, w) + syn.randn(seed, "corr.flow", (B, 2, h, w), 3.0)
    coords[:, :, 0, 0] = torch.tensor([-6.0, 2.0])                       # window partly outside
    coords[:, :, 1, 1] = torch.tensor([float(w + 9), float(h + 9)])      # fully outside
    coords[:, :, 2, 2] = torch.tensor([3.0, 4.0])                        # exactly integer
    coords[:, :, 3, 3] = torch.tensor([float(w - 1), float(h - 1)])      # last cell
    return f1, f2, coords, _grid(B, h, w)


def gma_inputs():
    seed, BT, h, w = GMA_CASE
    P = syn.make_params(seed, 4)
    inp = torch.relu(syn.randn(seed, "gma.inp", (BT, 128, h, w)))
    mf = syn.randn(seed, "gma.mf", (BT, 128, h, w))
    return P, inp, mf


def skblock_inputs(name, cin):
    h, w = SKBLOCK_HW
    return syn.randn(SKBLOCK_SEED, "sk.x." + name, (2, cin, h, w))


def update_inputs(tag):
    B, T, h, w, seed = UPDATE_CASES[tag]
    Pn = T - 1
    N = h * w
    P = syn.make_params(seed, T)
    nets = torch.tanh(syn.randn(seed, "ub.nets", (B * Pn, 128, h, w)))
    inps = torch.relu(syn.randn(seed, "ub.inps", (B * Pn, 128, h, w)))
    corrs = syn.randn(seed, "ub.corrs", (B * Pn, 324, h, w))
    flows = syn.randn(seed, "ub.flows", (B * Pn, 2, h, w), 2.0)
    attn = torch.softmax(syn.randn(seed, "ub.attn", (B * Pn, 1, N, N), 2.0), dim=-1)
    return P, nets, inps, corrs, flows, attn


def upsample_inputs():
    flow = syn.randn(UPSAMPLE_SEED, "up.flow", (2, 2, 9, 11), 3.0)
    mask = syn.randn(UPSAMPLE_SEED, "up.mask", (2, 576, 9, 11), 2.0)
    return flow, mask


def forward_inputs(tag):
    B, T, H, W, iters, seed, use_init = FORWARD_CASES[tag]
    h, w = H // 8, W // 8
    P = syn.make_params(seed, T)
    fmaps, cnets = syn.make_features(seed, B, T, h, w)
    finit = [syn.randn(seed, f"flow_init{i}", (B, 2, h, w), 1.5) for i in range(T - 1)] if use_init else None
    return P, fmaps, cnets, finit, iters


def interp_inputs(tag):
    """Low-resolution flow [2, h, w] for forward_interpolate (utils.py:34-62); the larger scales push many source
    points outside the image (dropped by the reference's `valid` mask)."""
    seed, h, w, scale = INTERP_CASES[tag]
    return syn.randn(seed, "interp.flow", (2, h, w), scale)


TWINS_CASES = {"twins_a": (1, 4, 64, 96, 91), "twins_b": (2, 3, 40, 72, 92)}      # B, T, H, W, seed (token grids not multiples of 7)


def twins_inputs(tag):
    """Normalised frames [B,T,3,H,W] in [-1,1] and the encoder's synthetic parameters."""
    B, T, H, W, seed = TWINS_CASES[tag]
    x = torch.tanh(syn.randn(seed, "twins.x", (B, T, 3, H, W)))
    return syn.make_twins_params(seed), x


def flow_io_inputs():
    """Seeded inputs of the frame_utils fixture (f3): a flow with fractional parts that sit on the uint16 truncation edge of
    the KITTI code in float32 vs float64 arithmetic, a 16-bit KITTI image, PFM payloads."""
    import numpy as np
    rng = np.random.default_rng(91)
    flow = (rng.standard_normal((5, 7, 2)) * 20.0).astype(np.float32)
    flow[0, 0] = [0.0, -512.0]
    flow[0, 1] = [1.0 - 2.0 ** -7, 511.984375]          # 64 u + 2^15 lands 0.5 below an integer: float32 keeps it, truncation differs if mis-rounded
    flow[0, 2] = [np.float32(3.99999976), np.float32(-0.0078125)]
    kitti = rng.integers(0, 65536, size=(4, 6, 3)).astype(np.uint16)          # as cv2.imread returns it: B, G, R = valid, v, u
    kitti[:, :, 0] = rng.integers(0, 2, size=(4, 6))
    pfm3 = rng.standard_normal((3, 4, 3)).astype(np.float32)
    pfm1 = rng.standard_normal((3, 4)).astype(np.float32)
    return flow, kitti, pfm3, pfm1


# ---- hard cases: frames -> exact Twins_CSC features -> loop at 128 x 192 (ill-conditioned random-weight network, flows of 4-40 px) ----
# seed 21 = the case bench.py / the tests have carried since round 3 (hot-path params 21, frames 24, Twins params 22 / 23); the
# others are the held-out construction of round 4's tools/preset_select.py (params s, frames 100 + s, Twins 200 + s / 300 + s).
# every consumer sweeps ALL of them and reports the maximum.
HARD_SEEDS = (21, 11, 12, 13, 31, 32)
HARD_SHAPE = (1, 4, 128, 192, 4)                      # B, T, H, W, iterations (the round-3 ... 5 form of the sweep)
HARD_ITERS = (4, 15)                                  # ... and the count the reference deploys (scripts/infer.sh:17, demo.py:419): VERDICT r5 #6


def hard_case_seeds(seed):
    """(hot-path params seed, frame seed, fnet Twins seed, cnet Twins seed)"""
    return (21, 24, 22, 23) if seed == 21 else (seed, 100 + seed, 200 + seed, 300 + seed)


def hard_case_frames(seed):
    B, T, H, W, _ = HARD_SHAPE
    fs = hard_case_seeds(seed)[1]
    return [(syn.randn(fs, f"frame{t}", (B, 3, H, W)).sigmoid() * 255.0) for t in range(T)]
