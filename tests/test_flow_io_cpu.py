"""CPU: .flo codec, KITTI uint16 arithmetic and the evaluation metrics (reference frame_utils.py / evaluate_mf.py)."""
import numpy as np
import pytest

from streamflow_amd import flow_io


def test_flo_round_trip_and_header_bytes(tmp_path):
    rng = np.random.default_rng(0)
    flow = rng.standard_normal((5, 7, 2)).astype(np.float32) * 10
    p = tmp_path / "a.flo"
    flow_io.write_flo(str(p), flow)
    raw = p.read_bytes()
    assert raw[:4] == b"PIEH"                                  # 202021.25f, the Middlebury tag
    assert np.frombuffer(raw[4:12], np.int32).tolist() == [7, 5]   # width first, then height
    assert len(raw) == 12 + 5 * 7 * 2 * 4
    assert np.frombuffer(raw[12:20], np.float32).tolist() == flow[0, 0].tolist()   # (u, v) interleaved
    assert np.array_equal(flow_io.read_flo(str(p)), flow)
    p.write_bytes(b"XXXX" + raw[4:])
    with pytest.raises(IOError):
        flow_io.read_flo(str(p))


def test_kitti_codec_arithmetic():
    flow = np.array([[[1.5, -2.25], [0.0, 100.0]]], np.float32)          # multiples of 1/64 are exact
    enc = flow_io.kitti_encode(flow)
    assert enc.dtype == np.uint16 and enc[0, 0].tolist() == [32768 + 96, 32768 - 144, 1]
    dec, valid = flow_io.kitti_decode(enc)
    assert np.array_equal(dec, flow) and valid.tolist() == [[1.0, 1.0]]


def test_metrics_formulas():
    gt = np.zeros((2, 2, 3))
    gt[0] = 10.0                                            # |gt| = 10
    flow = gt.copy()
    flow[0, 0, 0] += 3.0                                    # epe 3   -> not > 3: inlier
    flow[0, 0, 1] += 4.0                                    # epe 4, 40 % -> outlier
    flow[1, 1, 2] += 0.5                                    # epe 0.5
    e = flow_io.epe_map(flow, gt)
    assert np.allclose(e, [[3, 4, 0], [0, 0, 0.5]])
    m = flow_io.sintel_metrics(e)
    assert m["epe"] == pytest.approx(7.5 / 6) and m["1px"] == pytest.approx(4 / 6) and m["5px"] == 1.0
    valid = np.ones((2, 3))
    valid[0, 0] = 0
    k = flow_io.kitti_f1(flow, gt, valid)
    assert k["f1"] == pytest.approx(100 * 1 / 5) and k["epe"] == pytest.approx(4.5 / 5)
    # HWC layout gives the same numbers
    assert np.allclose(flow_io.epe_map(flow.transpose(1, 2, 0), gt.transpose(1, 2, 0)), e)
