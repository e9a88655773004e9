"""CPU: the demo's sliding-window / tail-flag schedule (reference demo.py:517-532): every consecutive frame pair
gets exactly one flow field, in order, for any video length >= T."""
import pytest
import torch

from streamflow_amd.demo import group_clips, predict_frames


def _reference_schedule(n, T):
    """Line-by-line restatement of the reference loop, recording (window start, flags) -- the test oracle."""
    out, i = [], 0
    while True:
        if i + T <= n:
            start, flags = i, [j for j in range(i, i + T)]
        else:
            start, flags = n - T, [-1 if j < i else j for j in range(n - T, n)]
        out.append((start, flags))
        if i + T >= n:
            break
        i = i + T - 1
    return out


@pytest.mark.parametrize("T", [2, 3, 4])
@pytest.mark.parametrize("n", [4, 5, 6, 7, 8, 9, 10, 13, 23])
def test_schedule_matches_reference_loop_and_covers_every_pair_once(n, T):
    if n < T:
        pytest.skip("shorter than one window")
    sched = group_clips(n, T)
    ref = _reference_schedule(n, T)
    assert [(s, [f != -1 for f in fl[: T - 1]]) for s, fl in ref] == sched
    pairs = [s + k for s, keep in sched for k in range(T - 1) if keep[k]]
    assert pairs == list(range(n - 1))


def test_predict_frames_pads_groups_and_unpads():
    T, H, W = 4, 20, 30                       # not multiples of 8 -> padded to 24 x 32 inside
    frames = [torch.full((3, H, W), float(j)) for j in range(9)]
    seen = []

    def fake_model(imgs):                      # imgs [1,T,3,H',W']; "flow" of pair k encodes its first frame id
        assert imgs.shape == (1, T, 3, 24, 32)
        ids = imgs[0, :, 0, 12, 16]
        seen.append(ids.tolist())
        return [torch.full((1, 2, 24, 32), float(ids[k])) for k in range(T - 1)]

    flows = predict_frames(fake_model, frames, T=T)
    assert len(flows) == 8 and all(f.shape == (2, H, W) for f in flows)
    assert [int(f[0, 0, 0]) for f in flows] == list(range(8))
    assert seen == [[0, 1, 2, 3], [3, 4, 5, 6], [5, 6, 7, 8]]


def test_warm_start_loop_plumbing(monkeypatch):
    """predict_clips_warm_start hands every clip the forward-interpolated low-resolution flows of the previous one
    (evaluate_mf.py:296-304); the first clip starts from zeros.  Model and interpolation are fakes here (CPU)."""
    import torch
    from streamflow_amd import demo, utils
    seen = []

    def fake_interp(f):
        return f + 1.0
    monkeypatch.setattr(utils, "forward_interpolate", fake_interp)

    def model(images, iters, flow_init, test_mode):
        assert test_mode and iters == 3 and len(flow_init) == len(images) - 1
        seen.append([f.clone() for f in flow_init])
        low = [f + 10.0 for f in flow_init]
        return [torch.zeros(1, 2, 16, 24) for _ in low], low
    clips = [[torch.zeros(1, 3, 16, 24) for _ in range(3)] for _ in range(3)]
    out = demo.predict_clips_warm_start(model, clips, iters=3)
    assert len(out) == 3 and len(out[0]) == 2
    assert all(float(f.abs().max()) == 0.0 for f in seen[0]) and seen[0][0].shape == (1, 2, 2, 3)
    assert all(torch.all(f == 11.0) for f in seen[1]) and all(torch.all(f == 22.0) for f in seen[2])
