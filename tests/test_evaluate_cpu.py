"""Row f3 scoring loops (streamflow_amd/evaluate.py) on the CPU: the clip schedule against a literal restatement of the
reference's dataset loop (core/mf_datasets.py:1125-1149), and validate_sintel_mf / validate_kitti_mf over synthetic dataset
trees written with this package's own codecs, with a stand-in model whose error per pair is known in closed form."""
import os

import numpy as np
import pytest
import torch

from streamflow_amd import evaluate, flow_io


def _reference_loop(n_images, nframes):
    """core/mf_datasets.py:1125-1149, restated line by line (window starts and frame ids only)."""
    out, i = [], 0
    while True:
        if i + nframes <= n_images:
            out.append((i, [j for j in range(i, i + nframes)]))
        else:
            out.append((n_images - nframes, [-1 if j < i else j for j in range(n_images - nframes, n_images)]))
        if i + nframes >= n_images:
            break
        i += nframes - 1
    return out


def test_sintel_clip_schedule_matches_the_reference_loop():
    for T in range(2, 7):
        for n in range(T, 60):
            sched = evaluate.sintel_clip_schedule(n, T)
            assert sched == _reference_loop(n, T), (n, T)
            scored = sorted(ids[i] for _, ids in sched for i in range(T - 1) if ids[i] != -1)
            assert scored == list(range(n - 1)), (n, T)                 # every pair of the scene exactly once
    with pytest.raises(ValueError):
        evaluate.sintel_clip_schedule(3, 4)


def _tag_frame(rng, H, W, scene, idx):
    img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    img[0, 0] = (scene, idx, 77)                                         # the stand-in model reads (scene, frame) back from here
    return img


def _noise(scene, idx, H, W):
    g = np.random.default_rng(1000 * scene + idx)
    return g.normal(0.0, 2.0, size=(2, H, W)).astype(np.float32)


def test_validate_sintel_mf_on_a_synthetic_tree(tmp_path, capsys):
    rng = np.random.default_rng(0)
    H, W, T = 44, 60, 4                                                  # not multiples of 8: the padder is exercised
    lengths = {"alley_1": 5, "market_2": 9}
    gts = {}
    for s, (scene, n) in enumerate(lengths.items()):
        for dstype in ("clean", "final"):
            os.makedirs(tmp_path / "training" / dstype / scene)
            for i in range(n):
                flow_io.write_png(str(tmp_path / "training" / dstype / scene / f"frame_{i + 1:04d}.png"), _tag_frame(rng, H, W, s, i))
        os.makedirs(tmp_path / "training" / "flow" / scene)
        for i in range(n - 1):
            gt = rng.normal(0.0, 5.0, size=(H, W, 2)).astype(np.float32)
            gts[(s, i)] = gt
            flow_io.write_flo(str(tmp_path / "training" / "flow" / scene / f"frame_{i + 1:04d}.flo"), gt)
    calls = []

    def model(images, iters=0, test_mode=False):
        assert test_mode and len(images) == T and all(im.shape == (1, 3, 48, 64) for im in images)
        pad_t, pad_l = (48 - H) // 2, (64 - W) // 2                      # 'sintel' padding: split on both sides
        flows = []
        for im in images[:-1]:
            s, i = int(im[0, 0, pad_t, pad_l]), int(im[0, 1, pad_t, pad_l])
            f = torch.zeros(1, 2, 48, 64)
            f[0, :, pad_t:pad_t + H, pad_l:pad_l + W] = torch.from_numpy(gts[(s, i)]).permute(2, 0, 1) + torch.from_numpy(_noise(s, i, H, W))
            flows.append(f)
        calls.append(int(images[0][0, 1, pad_t, pad_l]))
        return flows

    res = evaluate.validate_sintel_mf(model, iters=3, root=str(tmp_path), nframes=T)
    want = np.concatenate([np.sqrt((_noise(s, i, H, W) ** 2).sum(0)).reshape(-1)
                           for s, n in enumerate(lengths.values()) for i in range(n - 1)])
    assert set(res) == {"clean", "final"}
    assert abs(res["clean"] - want.mean()) < 1e-5 and abs(res["final"] - want.mean()) < 1e-5
    assert calls == [0, 1, 0, 3, 5] * 2                                 # 5 frames: clips at 0 and (tail) 1; 9 frames: 0, 3, tail 5
    rep = evaluate.sintel_report(model, iters=3, root=str(tmp_path), nframes=T, dstypes=("clean",))["clean"]
    assert rep["pairs"] == 4 + 8 and abs(rep["3px"] - (want < 3).mean()) < 1e-9 and abs(rep["1px"] - (want < 1).mean()) < 1e-9
    assert "Validation (clean) EPE:" in capsys.readouterr().out


def test_validate_kitti_mf_on_a_synthetic_tree(tmp_path):
    rng = np.random.default_rng(1)
    H, W, T = 37, 124, 3                                                 # 'kitti' padding: bottom / both sides
    os.makedirs(tmp_path / "training" / "image_2")
    os.makedirs(tmp_path / "training" / "flow_occ")
    epes, outs, gts = [], [], {}
    for s in range(3):
        for fr in range(12 - T, 12):
            flow_io.write_png(str(tmp_path / "training" / "image_2" / ("%06d_%02d.png" % (s, fr))), _tag_frame(rng, H, W, s, fr))
        gt = (rng.normal(0.0, 20.0, size=(H, W, 2)) * 64).round() / 64      # representable in the 16-bit code
        valid = rng.random((H, W)) < 0.7
        enc = flow_io.kitti_encode(gt.astype(np.float64))
        enc[:, :, 2] = valid
        flow_io.write_png(str(tmp_path / "training" / "flow_occ" / ("%06d_10.png" % s)), enc)
        gts[s] = gt.astype(np.float32)
        noise = _noise(s, 10, H, W) * 2.0
        epe = np.sqrt((noise ** 2).sum(0)).reshape(-1)
        mag = np.sqrt((gt.astype(np.float32) ** 2).sum(-1)).reshape(-1)
        v = valid.reshape(-1)
        epes.append(epe[v].mean())
        outs.append(((epe > 3.0) & (epe / mag > 0.05))[v])

    def model(images, iters=0, test_mode=False):
        Hp, Wp = images[0].shape[-2:]
        assert (Hp, Wp) == (40, 128) and len(images) == T
        pad_l = (Wp - W) // 2
        s = int(images[-2][0, 0, 0, pad_l])
        f = torch.zeros(1, 2, Hp, Wp)
        f[0, :, :H, pad_l:pad_l + W] = torch.from_numpy(gts[s]).permute(2, 0, 1) + torch.from_numpy(_noise(s, 10, H, W) * 2.0)
        return [torch.full((1, 2, Hp, Wp), 1e6)] * (T - 2) + [f]           # pairs without ground truth must not be scored

    res = evaluate.validate_kitti_mf(model, iters=2, multi_root=str(tmp_path), nframes=T)
    assert set(res) == {"kitti_epe", "kitti_f1"}
    assert abs(res["kitti_epe"] - np.mean(epes)) < 1e-4
    assert abs(res["kitti_f1"] - 100 * np.concatenate(outs).mean()) < 1e-4            # (float32 mean, as the reference takes it)
    with pytest.raises(ValueError):
        evaluate.validate_kitti_mf(model)
