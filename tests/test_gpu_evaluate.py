"""Row f3 on the GPU: validate_sintel_mf / validate_kitti_mf (streamflow_amd/evaluate.py, the reference's evaluate_mf.py:468-503
and :106-142) over synthetic dataset trees written with this package's codecs, driven by the HIP model (frames -> Twins_CSC ->
hot path) and, for comparison, by the CPU oracles chained the same way and scored by the same loops."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _smooth_frames(rng, n, H, W):
    """A textured image translated by a few pixels per frame (so that consecutive frames are related)."""
    base = rng.integers(0, 256, size=(H + 64, W + 64, 3)).astype(np.float32)
    for _ in range(2):                                                   # cheap blur: correlated texture
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) / 4.0
    base = (base - base.min()) / (base.max() - base.min()) * 255.0
    return [base[32 + 2 * i: 32 + 2 * i + H, 32 - 3 * i + 16: 32 - 3 * i + 16 + W].round().astype(np.uint8) for i in range(n)]


class _OracleModel:
    """The CPU oracles behind the reference's test-mode call signature."""
    def __init__(self, hot, ef, ec, T):
        self.hot, self.ef, self.ec, self.T = hot, ef, ec, T

    def __call__(self, images, iters=6, test_mode=True):
        from oracle import streamflow_oracle as orc, twins_oracle as two
        imgs = 2 * (torch.stack([i.cpu().float() for i in images], dim=1) / 255.0) - 1.0
        fmaps = two.twins_csc_forward(imgs, self.ef)
        cnets = two.twins_csc_forward(imgs[:, :-1], self.ec)
        return orc.hotpath_forward(fmaps, cnets, self.hot, iters)[0]


def _models(dev, T, preset):
    from streamflow_amd import synthetic as syn
    from streamflow_amd.model import SKFlow_MF8, default_args
    hot, ef, ec = syn.make_params(31, T), syn.make_twins_params(32), syn.make_twins_params(33)
    sd = dict(hot)
    sd.update({"fnet." + k: v for k, v in ef.items()})
    sd.update({"cnet." + k: v for k, v in ec.items()})
    model = SKFlow_MF8(default_args(T=T, preset=preset)).to(dev)
    model.load_state_dict(sd, strict=True)
    return model, _OracleModel(hot, ef, ec, T)


@pytest.mark.parametrize("preset,tol", [("fp32_class", 1e-3), ("config2_mixed", 2e-2)])
def test_validate_sintel_mf_hip_vs_oracle(tmp_path, dev, preset, tol):
    """Two scenes (5 and 4 frames, T = 3: the second one needs the end-aligned tail clip with a -1 frame id), clean and final
    passes, 124 x 188 frames (padded to 128 x 192 by InputPadder).  The HIP model's dataset scores must equal the oracle model's:
    both are scored against the same (random) ground truth by the same loop, so the difference is the flow deviation alone."""
    from streamflow_amd import evaluate, flow_io
    rng = np.random.default_rng(3)
    H, W, T, iters = 124, 188, 3, 3
    for scene, n in (("ambush_9", 5), ("cave_9", 4)):
        for dstype in ("clean", "final"):
            os.makedirs(tmp_path / "training" / dstype / scene)
            for i, img in enumerate(_smooth_frames(rng, n, H, W)):
                flow_io.write_png(str(tmp_path / "training" / dstype / scene / f"frame_{i + 1:04d}.png"), img)
        os.makedirs(tmp_path / "training" / "flow" / scene)
        for i in range(n - 1):
            flow_io.write_flo(str(tmp_path / "training" / "flow" / scene / f"frame_{i + 1:04d}.flo"),
                              rng.normal(0, 3, size=(H, W, 2)).astype(np.float32))
    model, oracle = _models(dev, T, preset)
    got = evaluate.sintel_report(model, iters=iters, root=str(tmp_path), nframes=T, device=dev)
    ref = evaluate.sintel_report(oracle, iters=iters, root=str(tmp_path), nframes=T, device=torch.device("cpu"))
    assert set(got) == {"clean", "final"}
    for k in ("clean", "final"):
        assert got[k]["pairs"] == ref[k]["pairs"] == 4 + 3                       # every pair of both scenes exactly once
        assert abs(got[k]["epe"] - ref[k]["epe"]) <= tol, (preset, k, got[k], ref[k])
        for r in ("1px", "3px", "5px"):
            assert abs(got[k][r] - ref[k][r]) <= 5e-3, (preset, k, r, got[k], ref[k])
    res = evaluate.validate_sintel_mf(model, iters=iters, root=str(tmp_path), nframes=T, device=dev)
    assert set(res) == {"clean", "final"} and res["clean"] == got["clean"]["epe"]


def test_validate_kitti_mf_hip_vs_oracle(tmp_path, dev):
    """Two sequences of the multi-frame KITTI layout (frames 09..11 for T = 3, ground truth of pair 10 -> 11 as a sparse 16-bit
    PNG), 'kitti' padding; EPE and F1-all from the HIP model against the oracle model's."""
    from streamflow_amd import evaluate, flow_io
    rng = np.random.default_rng(4)
    H, W, T, iters = 122, 180, 3, 3
    os.makedirs(tmp_path / "training" / "image_2")
    os.makedirs(tmp_path / "training" / "flow_occ")
    for s in range(2):
        for i, img in zip(range(12 - T, 12), _smooth_frames(rng, T, H, W)):
            flow_io.write_png(str(tmp_path / "training" / "image_2" / ("%06d_%02d.png" % (s, i))), img)
        gt = rng.normal(0, 6, size=(H, W, 2)).astype(np.float32)
        valid = rng.random((H, W)) < 0.4
        enc = flow_io.kitti_encode(gt)
        enc[..., 2] = valid
        flow_io.write_png(str(tmp_path / "training" / "flow_occ" / ("%06d_10.png" % s)), enc)
    model, oracle = _models(dev, T, "fp32_class")
    got = evaluate.validate_kitti_mf(model, iters=iters, multi_root=str(tmp_path), nframes=T, device=dev)
    ref = evaluate.validate_kitti_mf(oracle, iters=iters, multi_root=str(tmp_path), nframes=T, device=torch.device("cpu"))
    assert set(got) == {"kitti_epe", "kitti_f1"}
    assert abs(got["kitti_epe"] - ref["kitti_epe"]) <= 1e-3, (got, ref)
    assert abs(got["kitti_f1"] - ref["kitti_f1"]) <= 0.05, (got, ref)              # F1 is a count over ~9000 valid pixels
