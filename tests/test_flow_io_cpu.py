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


def test_kitti_png_round_trip_and_filters(tmp_path):
    """16-bit KITTI flow PNG (frame_utils.py:117-122,137-141): write -> read is exact on the 1/64 px grid, the file is a
    16-bit RGB PNG with u in the RED channel (cv2's BGR flip in the reference), and the reader decodes all five PNG row
    filters (a hand-filtered file: devkit PNGs are written by libpng with adaptive filtering)."""
    import struct
    import zlib
    from streamflow_amd import flow_io
    rng = np.random.default_rng(5)
    flow = np.round(rng.normal(0, 30, (13, 17, 2)) * 64) / 64
    path = str(tmp_path / "f.png")
    flow_io.write_flow_kitti(path, flow)
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">IIBB", raw[16:26]) == (17, 13, 16, 2)
    got, valid = flow_io.read_flow_kitti(path)
    assert got.dtype == np.float32 and np.array_equal(got, flow.astype(np.float32)) and (valid == 1).all()
    img = flow_io.read_png(path)
    assert np.array_equal(img[:, :, 0], (64 * flow[:, :, 0] + 2 ** 15).astype(np.uint16))       # R = u
    # re-encode the same pixels with filter types 0..4 cycling over the rows
    rows = img.astype(">u2").reshape(13, -1).view(np.uint8).reshape(13, -1).astype(np.int32)
    bpp, enc, prev = 6, [], np.zeros(rows.shape[1], np.int32)
    for y, line in enumerate(rows):
        ft = y % 5
        a = np.concatenate([np.zeros(bpp, np.int32), line[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        p = a + prev - c
        pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
        paeth = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        pred = [0, a, prev, (a + prev) >> 1, paeth][ft]
        enc.append(bytes([ft]) + ((line - pred) & 255).astype(np.uint8).tobytes())
        prev = line

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    p2 = str(tmp_path / "g.png")
    open(p2, "wb").write(raw[:8] + chunk(b"IHDR", struct.pack(">IIBBBBB", 17, 13, 16, 2, 0, 0, 0)) +
                         chunk(b"IDAT", zlib.compress(b"".join(enc))[:40]) + chunk(b"IDAT", zlib.compress(b"".join(enc))[40:]) +
                         chunk(b"IEND", b""))
    assert np.array_equal(flow_io.read_png(p2), img)
    got2, _ = flow_io.read_flow_kitti(p2)
    assert np.array_equal(got2, got)
    with pytest.raises(IOError):
        flow_io.read_png(str(tmp_path / "f.png") + "x") if False else flow_io.read_flow_kitti(_gray_png(tmp_path, flow_io))


def _gray_png(tmp_path, flow_io):
    p = str(tmp_path / "gray.png")
    flow_io.write_png(p, np.zeros((4, 4), np.uint8))
    return p


def test_pfm_round_trip(tmp_path):
    from streamflow_amd import flow_io
    rng = np.random.default_rng(6)
    for shape in ((5, 7), (5, 7, 3)):
        a = rng.normal(size=shape).astype(np.float32)
        p = str(tmp_path / "a.pfm")
        flow_io.write_pfm(p, a)
        assert np.array_equal(flow_io.read_pfm(p), a)
    # big-endian file with a positive scale, rows bottom-up (frame_utils.py:70-81)
    b = rng.normal(size=(3, 4)).astype(np.float32)
    p = str(tmp_path / "b.pfm")
    with open(p, "wb") as f:
        f.write(b"Pf\n4 3\n1.0\n")
        np.flipud(b).astype(">f4").tofile(f)
    assert np.array_equal(flow_io.read_pfm(p), b)
